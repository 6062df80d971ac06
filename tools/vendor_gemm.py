"""Context measurement: the vendor library (hipBLASLt through torch.nn.functional.linear) on the GEMM shapes of
the path, same timing method as tools/ab_gemm.py.  Not used by the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_gemm import SHAPES  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

for (M, N, K, _) in SHAPES + [(8192, 8192, 8192, 0)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).bfloat16()
    b = (torch.rand(N, device="cuda") * 2 - 1).bfloat16()
    t = timeit(lambda: torch.nn.functional.linear(a, w, b))
    print(f"[hipBLASLt] {M}x{N}x{K}: {2.0 * M * N * K / t / 1e12:.0f} TF/s", flush=True)
