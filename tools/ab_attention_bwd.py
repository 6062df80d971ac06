"""Attention backward at the path's training shapes for ONE build of the library (FK_LIB_PATH selects it); run it
alternately on two builds for an A/B.  Interleaves the two forms the build offers (fk_attention_bwd_set_mode: 1 = dQ pass +
paired dK / dV pass, 0 = three passes) and prints ms per call, the TF/s-equivalent at 8 tile products (what the three
passes execute) and a checksum of the three gradients (bit-identical builds print identical checksums).

    FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so python tools/ab_attention_bwd.py [tag]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import libfk, ops  # noqa: E402

BF = torch.bfloat16
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
H, D = 24, 3072
for B, S in [(1, int(x)) for x in os.environ.get("AB_S", "8704,2560").split(",")]:
    g = torch.Generator(device="cuda").manual_seed(S)
    q = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    k = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    qkv = torch.randn(B, S, 3 * D, device="cuda", generator=g).to(BF)
    do = torch.randn(B, S, D, device="cuda", generator=g).to(BF)
    o = torch.empty(B, S, D, device="cuda", dtype=BF)
    lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
    ops.attention_lse(q, k, qkv[:, :, 2 * D:], o, lse)
    dsum = ops.rowdot(do, o, H)
    dq, dk, dqkv = torch.empty_like(q), torch.empty_like(k), torch.zeros_like(qkv)
    fn = lambda: ops.attention_bwd(q, k, qkv[:, :, 2 * D:], do, lse, dsum, dq, dk, dqkv[:, :, 2 * D:])  # noqa: E731
    lib = libfk.load()
    modes = [int(m) for m in os.environ.get("AB_MODES", "1,0").split(",")] if os.environ.get("AB_MODES", "") != "none" else [None]
    ms = {m: [] for m in modes}
    cs = {}
    for rnd in range(5):
        for m in modes:
            if m is not None:
                ops.attention_bwd_set_mode(m)
            fn()
            torch.cuda.synchronize()
            if rnd == 0:
                cs[m] = dq.float().abs().sum().item() + dk.float().abs().sum().item() + dqkv.float().abs().sum().item()
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            e1.synchronize()
            ms[m].append(e0.elapsed_time(e1) / 8)
    if modes[0] is not None:
        ops.attention_bwd_set_mode(1)
    fl = 8 * 2.0 * B * H * S * S * 128
    for m in modes:
        t = statistics.median(ms[m])
        print(f"{tag} mode {m} attention_bwd B{B} S{S}: {t:.3f} ms  {fl / (t * 1e-3) / 1e12:.0f} TF/s at 8 products  checksum {cs[m]:.8e}", flush=True)
