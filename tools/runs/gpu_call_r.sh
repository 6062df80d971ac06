cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_cfg3.py -x -q -m gpu -k "attention" > gpurun_out/r02r_tests.log 2>&1; echo "pytest rc=$?" )
tail -2 gpurun_out/r02r_tests.log
for i in 1 2; do
  FK_LIB_PATH=$PWD/gpt_image_edit_amd/libfk_base_gfx950.so timeout 300 python tools/ab_attention.py base 2>&1 | tee -a gpurun_out/r02r_ab.txt
  timeout 300 python tools/ab_attention.py new 2>&1 | tee -a gpurun_out/r02r_ab.txt
done
