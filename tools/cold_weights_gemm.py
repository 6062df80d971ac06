"""Does it matter that, inside an edit, every GEMM streams weights nobody has touched since the previous step (24 GB of
parameters per MMDiT forward, far beyond the 256 MiB Infinity Cache), while a micro-benchmark re-reads one warm weight?
Times the path's GEMM shapes with ONE weight buffer and with a ring of NRING distinct ones (same activations)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
NRING = int(os.environ.get("NRING", "12"))   # 2-3 weights still fit the 256 MiB Infinity Cache together


def rate(fn, fl, n_per):
    rs = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n_per):
            fn(i)
        e1.record()
        e1.synchronize()
        rs.append(fl * n_per / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return statistics.median(rs)


for (M, N, K, epi) in [(2560, 9216, 3072, 0), (2560, 12288, 3072, ops.FK_EPI_GELU_TANH), (2560, 3072, 15360, 0),
                       (2560, 3072, 12288, 0), (8704, 9216, 3072, 0), (8704, 3072, 15360, 0)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
    ws = [((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(BF) for _ in range(NRING)]
    b = (torch.rand(N, device="cuda") * 2 - 1).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    fl = 2.0 * M * N * K
    n_per = max(NRING, int(0.12 / (fl / 1.1e15)) // NRING * NRING)
    warm = rate(lambda i: ops.gemm(a, ws[0], b, out=out, epilogue=epi), fl, n_per)
    cold = rate(lambda i: ops.gemm(a, ws[i % NRING], b, out=out, epilogue=epi), fl, n_per)
    vw = rate(lambda i: torch.nn.functional.linear(a, ws[0], b), fl, n_per)
    vc = rate(lambda i: torch.nn.functional.linear(a, ws[i % NRING], b), fl, n_per)
    print(f"{M}x{N}x{K} epi{epi}: fk warm {warm:.0f} cold {cold:.0f} ({(cold / warm - 1) * 100:+.1f} %)   "
          f"vendor warm {vw:.0f} cold {vc:.0f} ({(vc / vw - 1) * 100:+.1f} %)", flush=True)
