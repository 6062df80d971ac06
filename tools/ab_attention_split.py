"""Within-process interleaved A/B of the attention forward's grid forms (fk_attention_set_split): 0 = one workgroup per
(b, h, 256-row block), 1 = stream-K persistent grid where the plain grid wastes part of a round.  Random operands,
~0.15 s of back-to-back launches per measurement, arms alternating inside every round; median / best of 5 rounds.

    python tools/ab_attention_split.py
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
H, D = 24, 3072
SHAPES = [(1, 2560), (1, 5632), (1, 8704), (2, 8704), (4, 8704), (1, 4608)]
if os.environ.get("AB_SHAPES"):          # e.g. AB_SHAPES=1x2560,4x8704
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["AB_SHAPES"].split(",")]
for B, S in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(S + B)
    q = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    k = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    qkv = torch.randn(B, S, 3 * D, device="cuda", generator=g).to(BF)
    o = torch.empty(B, S, D, device="cuda", dtype=BF)
    fl = 4.0 * B * H * S * S * 128
    rates = {0: [], 1: []}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mode in (0, 1):
        ops.attention_set_split(mode)
        ops.attention(q, k, qkv[:, :, 2 * D:], o)
    torch.cuda.synchronize()
    e0.record(); ops.attention(q, k, qkv[:, :, 2 * D:], o); e1.record(); e1.synchronize()
    iters = max(3, int(0.15 / (e0.elapsed_time(e1) * 1e-3)))
    for _ in range(5):
        for mode in (0, 1):
            ops.attention_set_split(mode)
            e0.record()
            for _ in range(iters):
                ops.attention(q, k, qkv[:, :, 2 * D:], o)
            e1.record(); e1.synchronize()
            rates[mode].append(fl * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    n_items = B * H * ((S + 255) // 256)
    print(f"attention B{B} S{S} ({n_items} items = {n_items / 256:.2f} rounds): plain grid med {statistics.median(rates[0]):.0f} best "
          f"{max(rates[0]):.0f} TF/s | stream-K med {statistics.median(rates[1]):.0f} best {max(rates[1]):.0f} TF/s", flush=True)
ops.attention_set_split(1)
