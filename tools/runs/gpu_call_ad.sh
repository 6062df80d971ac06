cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/cold_weights_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/   vendor.*//' | tee gpurun_out/r02ad_cold.txt
