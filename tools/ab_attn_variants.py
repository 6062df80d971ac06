"""Within-process interleaved A/B of the attention kernels (fk_attention_set_variant): lockstep (1) vs two-group (2).
Prints median / best TF/s per shape; the two must agree bit for bit (same arithmetic, same order)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import libfk, ops  # noqa: E402

BF = torch.bfloat16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = libfk.load()
H = 24
for (B, S) in [(1, 2560), (1, 5632), (1, 8704), (4, 8704), (1, 1000), (2, 4096)]:
    q = torch.randn(B, H, S, 128, device="cuda").to(BF)
    k = torch.randn(B, H, S, 128, device="cuda").to(BF)
    qkv = torch.randn(B, S, 3 * H * 128, device="cuda").to(BF)
    outs = {v: torch.zeros(B, S, H * 128, device="cuda", dtype=BF) for v in (1, 2)}
    fl = 4.0 * B * H * S * S * 128
    n_per = max(3, int(0.12 / (fl / 0.9e15)))
    res = {1: [], 2: []}
    for r in range(rounds + 1):
        for v in (1, 2):
            lib.fk_attention_set_variant(v)
            fn = lambda: ops.attention(q, k, qkv[:, :, 2 * H * 128:], outs[v])  # noqa: E731
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_per):
                fn()
            e1.record()
            e1.synchronize()
            if r:
                res[v].append(fl * n_per / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    lib.fk_attention_set_variant(0)
    same = torch.equal(outs[1], outs[2])
    print(f"attention B{B} S{S}: " + "  ".join(f"v{v}: med {statistics.median(x):.0f} best {max(x):.0f}" for v, x in res.items())
          + f"  bit-identical={same}  max|d|={(outs[1].float() - outs[2].float()).abs().max().item():.3e}", flush=True)
