cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
B=$PWD/gpt_image_edit_amd/libfk_base_gfx950.so
( SHAPE="4 8704" bash tools/pmc_attention.sh "base:FK_LIB_PATH=$B" "new:FK_X=1"; SHAPE="1 2560" bash tools/pmc_attention.sh "base:FK_LIB_PATH=$B" "new:FK_X=1" ) 2>&1 | tee gpurun_out/r02s_pmc.txt
