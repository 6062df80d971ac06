# Round 5, call F (the measurement record): (1) SQ counters of the GEMM main loops on both MFMA shapes beside the vendor kernel,
# four shapes at M = 32768 (tools/pmc_gemm_compare.sh); (2) HBM-traffic passes on the SHIPPED kernels only (tools/pmc_traffic.sh:
# time / FETCH_SIZE / WRITE_SIZE / hit counters, one rocprofv3 pass each); (3) the N = 2 code paths on this tree (two ranks
# sharing the one GPU over gloo: weak, strong, sharded train step).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r05f_gemm_pmc.txt
for shape in "32768 3072 12288" "32768 12288 3072" "32768 9216 3072" "32768 3072 15360"; do
  echo "## shape $shape (M N K)" >> gpurun_out/r05f_gemm_pmc.txt
  SHAPE="$shape" bash tools/pmc_gemm_compare.sh "gemm8_m16:FK_GEMM_MFMA=16" "gemm8_m32:FK_GEMM_MFMA=32" "hipBLASLt:FK_PROF_VENDOR=1" >> gpurun_out/r05f_gemm_pmc.txt 2>&1
done
tail -28 gpurun_out/r05f_gemm_pmc.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
( PMC_PASSES="time fetch write hit" timeout 1200 bash tools/pmc_traffic.sh > gpurun_out/r05f_traffic.log 2>&1; echo "traffic rc=$?" ); tail -5 gpurun_out/r05f_traffic.log
( python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/r05f_traffic > gpurun_out/r05f_traffic_summary.log 2>&1; python tools/traffic_json.py gpurun_out/r05f_traffic >> gpurun_out/r05f_traffic_summary.log 2>&1; echo "summary rc=$?" ); head -40 gpurun_out/r05f_traffic.md
export FK_BENCH_BACKEND=gloo
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r05f_n2_weak.json 2> gpurun_out/r05f_n2_weak.err; echo "n2 weak rc=$?" )
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --scaling strong --global-batch 4 --no-extra --cpu-baseline none > gpurun_out/r05f_n2_strong.json 2> gpurun_out/r05f_n2_strong.err; echo "n2 strong rc=$?" )
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/smoke_train_2rank.py > gpurun_out/r05f_n2_train.json 2> gpurun_out/r05f_n2_train.err; echo "n2 train rc=$?" )
for f in n2_weak n2_strong n2_train; do tail -c 400 gpurun_out/r05f_$f.json; echo; done
