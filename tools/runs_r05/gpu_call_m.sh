# Round 5, call M: where the two attention forwards differ at a stream-K shape (B1 H24 S5632): outputs and log-sum-exps.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for k in 4 8; do echo "kernel $k S5632 default grid"; FK_ATTN_KERNEL=$k timeout 100 python tools/attn_dump.py 1 24 5632 /tmp/k$k.pt; done
python tools/attn_diff.py /tmp/k4.pt /tmp/k8.pt 24
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05m_attn_diff.txt
