# Round 6, call H: gemm10 -- time before / inside the asm prologue, and the persistent grid (one workgroup per CU walking the tile list)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for SH in "32768 3072 12288" "32768 12288 3072" "2560 12288 3072"; do
  FK_G10_X=1 timeout 120 python tools/g10_cycles.py $SH 2>&1 | grep -v amdgpu.ids
  FK_G10_X=1 FK_G10_PERSIST=1 timeout 120 python tools/g10_cycles.py $SH 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06h_gemm10_timeline.txt 2>&1
cat gpurun_out/r06h_gemm10_timeline.txt
for P in 0 1; do echo "== FK_G10_PERSIST=$P"; ( FK_G10_PERSIST=$P AB_SHAPES="32768x3072x12288,32768x12288x3072,32768x9216x3072,2560x12288x3072,8704x12288x3072" AB_VARIANTS="256m16,1024m16,vendor" timeout 600 python tools/ab_gemm_variants.py 5 0 2>&1 | grep -v amdgpu.ids ); done > gpurun_out/r06h_gemm10_ab.txt 2>&1
cat gpurun_out/r06h_gemm10_ab.txt
