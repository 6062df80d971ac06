# Round 6, call E: gemm10's K loop under s_memtime -- cycles per K-tile and wave of the shipped schedule and of its measurement
# forms (gemm10_gen.py EXPERIMENTS; library built with -DFK_G10_EXPERIMENTS).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for X in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  FK_G10_X=$X timeout 120 python tools/g10_cycles.py 32768 3072 12288 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06e_gemm10_cycles.txt 2>&1
cat gpurun_out/r06e_gemm10_cycles.txt
