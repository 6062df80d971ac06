"""Context measurement: sustained (seconds-long) GEMM throughput, this repo's kernels vs the vendor library, to see
what the power/clock management leaves of the short-burst numbers of tools/ab_gemm.py.  Not used by the product."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
shapes = [(2560, 3072, 15360), (2560, 9216, 3072), (32768, 3072, 12288)]
for (M, N, K) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(BF)
    b = (torch.rand(N, device="cuda") * 2 - 1).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    for name, fn in (("fk   ", lambda: ops.gemm(a, w, b, out=out)), ("vendor", lambda: torch.nn.functional.linear(a, w, b))):
        fl = 2.0 * M * N * K
        n_per = max(1, int(0.25 / (fl / 1.0e15)))
        rates = []
        torch.cuda.synchronize()
        t_end = time.time() + 3.0
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_per):
                fn()
            e1.record()
            e1.synchronize()
            rates.append(fl * n_per / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        print(f"{name} {M}x{N}x{K}: " + " ".join(f"{r:.0f}" for r in rates), flush=True)
