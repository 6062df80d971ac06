# Round 5, call J: attention_fwd4 -- where a tile's LDS-DMA requests sit (0 before / 1 behind the first K-fragment reads / 2 one
# per slot of the first group) and FK_A4_EARLY (V^T reads one group earlier); control = call I's best build (a4_m0).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05j_attention_ab.txt
L=build_ab
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
: > $O
export FK_ATTN_KERNEL=4
run FK_LIB_PATH=$L/a4_m0/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py callI_best
run timeout 200 python tools/ab_attention.py d1
run FK_LIB_PATH=$L/a4_d0/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d0
run FK_LIB_PATH=$L/a4_d2/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d2
run FK_LIB_PATH=$L/a4_d1e1/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d1e1
run FK_LIB_PATH=$L/a4_d2e1/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d2e1
run timeout 200 python tools/ab_attention.py d1
run FK_ATTN_KERNEL=8 timeout 200 python tools/ab_attention.py k8
cat $O
for v in d2 d2e1 d1e1; do ( FK_LIB_PATH=$PWD/$L/a4_$v/gpt_image_edit_amd/libfk_gfx950.so timeout 600 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" > gpurun_out/r05j_tests_$v.log 2>&1; echo "pytest $v rc=$?" | tee -a gpurun_out/r05j_tests_$v.log ); tail -2 gpurun_out/r05j_tests_$v.log; done
