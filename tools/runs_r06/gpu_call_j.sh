# Round 6, call J: price of a fabric byte -- gemm8 under tile orders of different L2 reuse (tools/pmc_gemm_tile_order.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
SHAPE="8704 9216 3072" bash tools/pmc_gemm_tile_order.sh 1 2 4 8 16 64
SHAPE="32768 3072 12288" bash tools/pmc_gemm_tile_order.sh 1 2 4 8 16 128
SHAPE="2560 12288 3072" bash tools/pmc_gemm_tile_order.sh 1 2 4 8 16
} > gpurun_out/r06j_tile_order_traffic.txt 2>&1
cat gpurun_out/r06j_tile_order_traffic.txt
