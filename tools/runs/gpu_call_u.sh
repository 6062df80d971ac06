cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
D=$PWD/gpt_image_edit_amd
for i in 1 2; do
  timeout 300 python tools/ab_attention.py fd2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02u_ab.txt
  FK_LIB_PATH=$D/libfk_fd3_gfx950.so timeout 300 python tools/ab_attention.py fd3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02u_ab.txt
  FK_LIB_PATH=$D/libfk_fd4_gfx950.so timeout 300 python tools/ab_attention.py fd4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02u_ab.txt
done
