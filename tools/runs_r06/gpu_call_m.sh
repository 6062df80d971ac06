# Round 6, call M: host time per block backward, per-launch route vs block-level C entry points, from an idle queue
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python tools/bwd_block_host_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06m_bwd_block_host_time.txt; cat gpurun_out/r06m_bwd_block_host_time.txt
