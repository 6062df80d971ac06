# Round 4, call L: tile-order depth of the large-tile GEMM kernels (fk_gemm_set_group_m): whole-edit A/B at 512^2 and 1024^2.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( AB_ARMS="gm=8;gm=16;gm=4;gm=2" timeout 400 python tools/ab_edit_plans.py cfg2_single_512x512_28step 3 2 > gpurun_out/r04l_gm_512.txt 2>&1; echo "rc=$?" ); grep -v amdgpu gpurun_out/r04l_gm_512.txt | tail -10
( AB_ARMS="gm=8;gm=40;gm=16;gm=4" timeout 600 python tools/ab_edit_plans.py single_1024x1024_28step 2 2 > gpurun_out/r04l_gm_1024.txt 2>&1; echo "rc=$?" ); grep -v amdgpu gpurun_out/r04l_gm_1024.txt | tail -10
