# call T: SQ counters of the attention backward kernels (two-pass form and the three-pass form) and of the forward, S = 8704
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( SHAPE="1 8704" KIND=attention_bwd bash tools/pmc_attention.sh "two_pass:FK_ATTN_BWD=1" "three_pass:FK_ATTN_BWD=0"; SHAPE="1 8704" KIND=attention bash tools/pmc_attention.sh "fwd:FK_X=0" ) > gpurun_out/r03t_attention_pmc.txt 2>&1
cat gpurun_out/r03t_attention_pmc.txt | cut -c1-260
