# Round 6, call Q (the measurement record on the final library): (1) SQ counters of gemm8, gemm10 (one workgroup per CU) and hipBLASLt,
# four shapes at M = 32768 (tools/pmc_gemm_compare.sh); (2) beyond-L2 traffic passes on the shipped kernels (tools/pmc_traffic.sh).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r06q_gemm_pmc.txt
for shape in "32768 3072 12288" "32768 12288 3072" "32768 9216 3072" "32768 3072 15360"; do
  echo "## shape $shape (M N K)" >> gpurun_out/r06q_gemm_pmc.txt
  SHAPE="$shape" bash tools/pmc_gemm_compare.sh "gemm8_m16:FK_GEMM_BN=256" "gemm10:FK_GEMM_BN=1024" "hipBLASLt:FK_PROF_VENDOR=1" >> gpurun_out/r06q_gemm_pmc.txt 2>&1
done
cat gpurun_out/r06q_gemm_pmc.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
( PMC_PASSES="time fetch write hit" timeout 1500 bash tools/pmc_traffic.sh > gpurun_out/r06q_traffic.log 2>&1; echo "traffic rc=$?" ); tail -5 gpurun_out/r06q_traffic.log
( python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/r06q_traffic > gpurun_out/r06q_traffic_summary.log 2>&1; python tools/traffic_json.py gpurun_out/r06q_traffic >> gpurun_out/r06q_traffic_summary.log 2>&1; echo "summary rc=$?" ); head -45 gpurun_out/r06q_traffic.md
