cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for n in 12 2 3 4; do echo "== NRING=$n"; NRING=$n timeout 200 python tools/cold_weights_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/   vendor.*//' | head -4; done | tee gpurun_out/r02ac_ring.txt
