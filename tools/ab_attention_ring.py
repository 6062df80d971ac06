"""Attention forward, K / V ring of 3 stages (a barrier per tile) vs 4 (a barrier per two tiles), interleaved in one process."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import libfk, ops  # noqa: E402

BF = torch.bfloat16
lib = libfk.load()
H, D = 24, 3072
for B, S in [(1, 2560), (1, 5632), (1, 8704), (4, 8704)]:
    g = torch.Generator(device="cuda").manual_seed(S + B)
    q = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    k = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    qkv = torch.randn(B, S, 3 * D, device="cuda", generator=g).to(BF)
    o = torch.empty(B, S, D, device="cuda", dtype=BF)
    fl = 4.0 * B * H * S * S * 128
    res = {3: [], 4: []}
    for r in range(5):
        for ring in (3, 4):
            lib.fk_attention_set_ring(ring)
            ops.attention(q, k, qkv[:, :, 2 * D:], o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = max(3, int(0.12 / (fl / 1.0e15)))
            e0.record()
            for _ in range(iters):
                ops.attention(q, k, qkv[:, :, 2 * D:], o)
            e1.record()
            e1.synchronize()
            if r:
                res[ring].append(fl * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    lib.fk_attention_set_ring(3)
    print(f"attention B{B} S{S}: " + "  ".join(f"ring {k_}: med {statistics.median(v):.0f} best {max(v):.0f} TF/s" for k_, v in res.items()), flush=True)
