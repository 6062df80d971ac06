# Round 6, call D: what each ingredient of gemm10's K loop costs -- measurement forms of the loop (gemm10_gen.py EXPERIMENTS,
# library built with -DFK_G10_EXPERIMENTS), each in its own process against gemm8 (m16) in the same process.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "gemm10" 2>&1 | tail -2 )
for X in 0 1 2 3 4 5; do
  echo "== FK_G10_X=$X"
  AB_NOCHECK=1 FK_G10_X=$X AB_SHAPES="32768x3072x12288,32768x12288x3072,2560x12288x3072" AB_VARIANTS="256m16,1024m16" timeout 300 python tools/ab_gemm_variants.py 3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06d_gemm10_ingredients.txt 2>&1
cat gpurun_out/r06d_gemm10_ingredients.txt
