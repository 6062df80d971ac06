"""Within-process interleaved A/B of the large-tile GEMM kernels (fk_gemm_set_variant) and the vendor library on the
path's shapes: N rounds, every variant once per round, ~0.12 s of back-to-back launches per measurement; prints the
median and the best TF/s per (shape, variant).  `python tools/ab_gemm_variants.py [rounds] [epi]`."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import libfk, ops  # noqa: E402

BF = torch.bfloat16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
epi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = libfk.load()
shapes = [tuple(int(x) for x in t.split("x")) for t in os.environ["AB_SHAPES"].split(",")] if os.environ.get("AB_SHAPES") else [(2560, 9216, 3072), (2560, 12288, 3072), (2560, 3072, 12288), (2560, 3072, 15360), (2560, 3072, 3072),
          (8704, 9216, 3072), (8704, 12288, 3072), (8704, 3072, 12288), (8704, 3072, 15360), (8704, 3072, 3072),
          (32768, 12288, 3072), (32768, 3072, 12288), (32768, 9216, 3072)]
# a variant is a launch form (fk_gemm_set_variant) with an optional MFMA shape suffix: "256m16" = 256 x 256 tiles on v_mfma_f32_16x16x32_bf16
variants = [v for v in os.environ.get("AB_VARIANTS", "128,256,384,512,0,vendor").split(",")]


def form_of(v):
    return int(v.split("m")[0]), (int(v.split("m")[1]) if "m" in v else 32)
for (M, N, K) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(BF)
    b = (torch.rand(N, device="cuda") * 2 - 1).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    fl = 2.0 * M * N * K
    n_per = max(3, int(0.12 / (fl / 1.1e15)))
    res = {v: [] for v in variants}
    ref, used = None, {}
    for r in range(rounds + 1):
        for v in variants:
            if v == "vendor":
                fn = lambda: torch.nn.functional.linear(a, w, b)  # noqa: E731
            else:
                ops.gemm_set_variant(form_of(v)[0])
                ops.gemm_set_mfma(form_of(v)[1])
                fn = lambda: ops.gemm(a, w, b, out=out, epilogue=epi)  # noqa: E731
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_per):
                fn()
            e1.record()
            e1.synchronize()
            if r:
                res[v].append(fl * n_per / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            elif v != "vendor":   # first round: the variants must agree bit for bit (split-K pairs, other MFMA shape: to the last bits)
                used[v] = ops.gemm_last_variant()
                if ref is None:
                    ref, ref_m, ref_v = out.clone(), form_of(v)[1], used[v]
                elif used[v] in (512, 640) or ref_v in (512, 640) or form_of(v)[1] != ref_m:
                    d = (ref.float() - out.float()).abs().max().item()
                    assert d <= 2 ** -7 * ref.float().abs().max().item(), f"split-K differs by {d} on {M}x{N}x{K}"
                elif os.environ.get("AB_NOCHECK") != "1":   # AB_NOCHECK=1: measurement forms whose results are wrong by construction
                    assert torch.equal(ref, out), f"variant {v} differs from variant 128 on {M}x{N}x{K}"
    ops.gemm_set_variant(0)
    ops.gemm_set_mfma(0)
    print(f"{M}x{N}x{K} epi{epi}: " + "  ".join(f"{v}{'' if str(used.get(v, v)) == v.split('m')[0] else '->' + str(used[v])}: med "
                                                 f"{statistics.median(x):.0f} best {max(x):.0f}" for v, x in res.items()), flush=True)
