"""K-major GEMM operands (fk_gemm_args.layout 1 / 2) against the transposed-copy path they replace, on the backward pass's
shapes at S = 8704: ms of [transposes + layout-0 GEMM] vs ms of the layout-1 / 2 GEMM alone (and the layout-0 GEMM alone).
`python tools/ab_gemm_layouts.py`"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
rnd = lambda *sh, sc=1.0: ((torch.rand(*sh, device=dev, generator=g) * 2 - 1) * sc).to(BF)  # noqa: E731


def timed(fn, n=10, rounds=3):
    fn()
    ms = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / n)
    return statistics.median(ms)


S = 8704
for (Nout, Kin) in [(3072, 3072), (12288, 3072), (3072, 12288), (9216, 3072), (3072, 15360)]:
    dy, x, W = rnd(1, S, Nout), rnd(1, S, Kin), rnd(Nout, Kin, sc=0.05)
    WT = torch.empty(Kin, Nout, device=dev, dtype=BF)
    dx = torch.empty(1, S, Kin, device=dev, dtype=BF)
    fl_d = 2.0 * S * Nout * Kin
    t_tr = timed(lambda: ops.transpose(W.view(1, Nout, Kin), WT.view(1, Kin, Nout)))
    t_g0 = timed(lambda: ops.gemm(dy, WT, out=dx))
    t_g1 = timed(lambda: ops.gemm(dy, W, out=dx, layout=1))
    print(f"dgrad {S}x{Kin}x{Nout}: transpose {t_tr * 1e3:.0f} us + layout 0 {t_g0 * 1e3:.0f} us ({fl_d / t_g0 / 1e9:.0f} TF/s) | layout 1 "
          f"{t_g1 * 1e3:.0f} us ({fl_d / t_g1 / 1e9:.0f} TF/s)", flush=True)
    dyT, xT = torch.empty(Nout, S, device=dev, dtype=BF), torch.empty(Kin, S, device=dev, dtype=BF)
    dW = torch.empty(Nout, Kin, device=dev, dtype=BF)

    def old():
        ops.transpose(dy, dyT.view(1, Nout, S))
        ops.transpose(x, xT.view(1, Kin, S))
        ops.gemm(dyT, xT, out=dW)
    t_old = timed(old)
    t_g0 = timed(lambda: ops.gemm(dyT, xT, out=dW))
    t_g2 = timed(lambda: ops.gemm(dy, x, out=dW, layout=2))
    print(f"wgrad {Nout}x{Kin}x{S}: transposes + layout 0 {t_old * 1e3:.0f} us (GEMM alone {t_g0 * 1e3:.0f} us, {fl_d / t_g0 / 1e9:.0f} TF/s) | "
          f"layout 2 {t_g2 * 1e3:.0f} us ({fl_d / t_g2 / 1e9:.0f} TF/s)", flush=True)
