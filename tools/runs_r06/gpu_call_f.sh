# Round 6, call F: where a gemm10 tile's time goes outside the K loop (entry -> loop, loop -> exit, gaps between workgroups on a CU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for SH in "32768 3072 12288" "32768 12288 3072" "2560 12288 3072"; do
for X in 1 10; do
  FK_G10_X=$X timeout 120 python tools/g10_cycles.py $SH 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r06f_gemm10_timeline.txt 2>&1
cat gpurun_out/r06f_gemm10_timeline.txt
