cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_ATTN_VARIANT=2 timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_cfg3.py -x -q -k "attention" > gpurun_out/r02g_attn_pp_tests.log 2>&1; echo "pytest(attn pp) rc=$?" )
tail -4 gpurun_out/r02g_attn_pp_tests.log
( timeout 600 python tools/ab_attn_variants.py 4 > gpurun_out/r02g_ab_attn.log 2>&1; echo "ab rc=$?" )
tail -7 gpurun_out/r02g_ab_attn.log
