# Round 4, call E: stream-K test shapes, 2-rank smokes (one GPU, gloo), SQ counters of the attention grids, HBM-traffic PMC
# passes on the current kernels (conv3x3_halo in the decoder), the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_cfg3.py tests/test_hip_backward.py -m gpu -x -q -s -k "stream_k" > gpurun_out/r04e_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04e_tests.log ); tail -3 gpurun_out/r04e_tests.log
export FK_BENCH_BACKEND=gloo
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r04e_n2_weak.json 2> gpurun_out/r04e_n2_weak.err; echo "n2 weak rc=$?" )
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --scaling strong --global-batch 4 --no-extra --cpu-baseline none > gpurun_out/r04e_n2_strong.json 2> gpurun_out/r04e_n2_strong.err; echo "n2 strong rc=$?" )
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/smoke_train_2rank.py > gpurun_out/r04e_n2_train.json 2> gpurun_out/r04e_n2_train.err; echo "n2 train rc=$?" )
unset FK_BENCH_BACKEND
for f in n2_weak n2_strong n2_train; do tail -c 600 gpurun_out/r04e_$f.json; echo; done
( SHAPE="1 8704" KIND=attention bash tools/pmc_attention.sh "plain:FK_ATTN_SPLIT=0" "streamk:FK_ATTN_SPLIT=1" > gpurun_out/r04e_attention_pmc.txt 2>&1; SHAPE="1 8704" KIND=attention_bwd bash tools/pmc_attention.sh "plain:FK_ATTN_SPLIT=0" "streamk:FK_ATTN_SPLIT=1" >> gpurun_out/r04e_attention_pmc.txt 2>&1; SHAPE="1 2560" KIND=attention bash tools/pmc_attention.sh "s2560:FK_ATTN_SPLIT=1" >> gpurun_out/r04e_attention_pmc.txt 2>&1 ); cat gpurun_out/r04e_attention_pmc.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
( TRAFFIC_SKIP="B32,278528,B8" PMC_PASSES="time fetch write hit" bash tools/pmc_traffic.sh > gpurun_out/r04e_traffic_passes.log 2>&1 ); cat gpurun_out/traffic/passes.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/r04e_traffic > gpurun_out/r04e_traffic_summary.log 2>&1; tail -3 gpurun_out/r04e_traffic_summary.log; head -30 gpurun_out/r04e_traffic.md
( timeout 900 python bench.py > gpurun_out/r04e_bench_default.json 2> gpurun_out/r04e_bench_default.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e_bench_default.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention'])
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'))
t=e.get('cfg5_train_step_1024x1024_bs1',{}); print('cfg5', {k: v for k, v in t.items() if k in ('value','ms_per_step','error','peak_memory_gb','host_enqueue_ms_per_step','host_work_ms_per_step','T_step_e2e')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'))
PY
