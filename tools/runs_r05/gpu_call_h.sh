# Round 5, call H: the 4-wave x 64-row attention forward, third form (hand-placed MFMA slots with a scheduling fence each, S^T
# chains as inline-asm MFMAs on VGPR accumulators) against the 8-wave kernel: rates + checksums; its parity tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_ATTN_KERNEL=4 timeout 200 python tools/ab_attention.py k4 ; FK_ATTN_KERNEL=8 timeout 200 python tools/ab_attention.py k8 ; FK_ATTN_KERNEL=4 timeout 200 python tools/ab_attention.py k4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05h_attention_4wave_ab.txt; cat gpurun_out/r05h_attention_4wave_ab.txt
( FK_ATTN_KERNEL=4 timeout 600 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" > gpurun_out/r05h_tests_attn4.log 2>&1; echo "pytest attn4 rc=$?" | tee -a gpurun_out/r05h_tests_attn4.log ); tail -3 gpurun_out/r05h_tests_attn4.log
