# Round 5, call L: the two attention forwards bit for bit (row sums per tile in the 4-wave kernel as in the 8-wave one), the
# attention / training tests, rates, and the SQ counters of both kernels at B1 S8704 and B1 S2560.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" tests/test_hip_training.py -k attention tests/test_hip_train_step.py > gpurun_out/r05l_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05l_tests.log ); tail -4 gpurun_out/r05l_tests.log
( timeout 200 python tools/ab_attention.py k4; FK_ATTN_KERNEL=8 timeout 200 python tools/ab_attention.py k8 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l_attention_rates.txt
( SHAPE="1 8704" bash tools/pmc_attention.sh "k4:FK_ATTN_KERNEL=4" "k8:FK_ATTN_KERNEL=8"; SHAPE="1 2560" bash tools/pmc_attention.sh "k4:FK_ATTN_KERNEL=4" "k8:FK_ATTN_KERNEL=8" ) 2>&1 | tee gpurun_out/r05l_attention_pmc.txt
