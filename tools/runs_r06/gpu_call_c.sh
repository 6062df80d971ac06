# Round 6, call C: SQ counters of gemm10 (first schedule) beside gemm8 (m16) and hipBLASLt on two M = 32768 shapes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for SH in "32768 3072 12288" "32768 12288 3072"; do
  echo "## shape $SH (M N K)"
  SHAPE="$SH" bash tools/pmc_gemm_compare.sh "gemm8_m16:FK_GEMM_BN=256" "gemm10:FK_GEMM_BN=1024" "hipBLASLt:FK_PROF_VENDOR=1"
done > gpurun_out/r06c_gemm10_pmc.txt 2>&1
cat gpurun_out/r06c_gemm10_pmc.txt
