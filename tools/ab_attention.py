"""Attention forward rate at the path's shapes for ONE build of the library (FK_LIB_PATH selects it); run it
alternately on two builds for an A/B (tools/runs/*.sh).  Random operands, ~0.15 s of back-to-back launches per
measurement, median / best of 4 rounds.

    FK_LIB_PATH=.../libfk_base_gfx950.so python tools/ab_attention.py [tag]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
    H, D = 24, 3072
    for B, S in [(1, 2560), (1, 5632), (1, 8704), (4, 8704), (2, 4096)]:
        g = torch.Generator(device="cuda").manual_seed(S + B)
        q = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
        k = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
        qkv = torch.randn(B, S, 3 * D, device="cuda", generator=g).to(BF)
        o = torch.empty(B, S, D, device="cuda", dtype=BF)
        fl = 4.0 * B * H * S * S * 128
        ops.attention(q, k, qkv[:, :, 2 * D:], o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.attention(q, k, qkv[:, :, 2 * D:], o); e1.record(); e1.synchronize()
        iters = max(3, int(0.15 / (e0.elapsed_time(e1) * 1e-3)))
        rates = []
        for _ in range(4):
            e0.record()
            for _ in range(iters):
                ops.attention(q, k, qkv[:, :, 2 * D:], o)
            e1.record(); e1.synchronize()
            rates.append(fl * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        print(f"{tag} attention B{B} S{S}: med {statistics.median(rates):.0f} best {max(rates):.0f} TF/s  "
              f"checksum {o.float().abs().sum().item():.6e}", flush=True)


if __name__ == "__main__":
    main()
