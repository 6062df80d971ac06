cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( AB_VARIANTS=1,2,3,4 timeout 600 python tools/ab_attn_variants.py 4 > gpurun_out/r02g_ab_attn2.log 2>&1; echo "ab rc=$?" )
tail -7 gpurun_out/r02g_ab_attn2.log
