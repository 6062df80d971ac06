cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
N=$PWD/gpt_image_edit_amd/libfk_nobar_gfx950.so
for i in 1 2; do
  timeout 300 python tools/ab_attention.py lockstep 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02t_ab.txt
  FK_LIB_PATH=$N timeout 300 python tools/ab_attention.py nobarrier 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02t_ab.txt
done
SHAPE="4 8704" bash tools/pmc_attention.sh "nobar:FK_LIB_PATH=$N" 2>&1 | tee gpurun_out/r02t_pmc.txt
