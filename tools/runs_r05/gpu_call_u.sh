# Round 5, call U: attention_fwd4 -- where the tile's 8 LDS-DMA requests sit, continued: form 2 (all in the first group), 3 (four
# in each block's third group), 4 (two in groups 1 and 3 of each block); all with s_nop 2 in front of the first softmax step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05u_attention_ab.txt
: > $O
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
for r in 1 2; do
run FK_LIB_PATH=build_ab/a4_ew2/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d2
run FK_LIB_PATH=build_ab/a4_d3/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d3
run FK_LIB_PATH=build_ab/a4_d4/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention.py d4
done
cat $O
for v in d3 d4; do ( FK_LIB_PATH=$PWD/build_ab/a4_$v/gpt_image_edit_amd/libfk_gfx950.so timeout 600 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" > gpurun_out/r05u_tests_$v.log 2>&1; echo "pytest $v rc=$?" | tee -a gpurun_out/r05u_tests_$v.log ); tail -2 gpurun_out/r05u_tests_$v.log; done
