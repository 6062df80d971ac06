"""Run ONE kernel shape a few times (target for `rocprofv3 --pmc ...`).

    python tools/prof_one.py gemm 2560 9216 3072 [epi]      |  attention B S      |  attention_bwd B S
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16


def rnd(*s):
    return (torch.rand(*s, device="cuda") * 2 - 1).to(BF)


kind = sys.argv[1]
if kind == "gemm":
    M, N, K = (int(v) for v in sys.argv[2:5])
    epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    a, w, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    if os.environ.get("FK_PROF_VENDOR") == "1":   # context: hipBLASLt on the same operands
        for _ in range(5):
            torch.nn.functional.linear(a, w, b)
    else:
        for _ in range(5):
            ops.gemm(a, w, b, out=out, epilogue=epi)
else:
    B, S = int(sys.argv[2]), int(sys.argv[3])
    H = 24
    q, k, qkv = rnd(B, H, S, 128), rnd(B, H, S, 128), rnd(B, S, 3 * H * 128)
    o = torch.empty(B, S, H * 128, device="cuda", dtype=BF)
    if kind == "attention_bwd":    # both forms of the backward (FK_ATTN_BWD picks the default one)
        do, lse = rnd(B, S, H * 128), torch.empty(B, H, S, device="cuda", dtype=torch.float32)
        ops.attention_lse(q, k, qkv[:, :, 2 * H * 128:], o, lse)
        dsum = ops.rowdot(do, o, H)
        dq, dk, dqkv = torch.empty_like(q), torch.empty_like(k), torch.zeros_like(qkv)
        for _ in range(5):
            ops.attention_bwd(q, k, qkv[:, :, 2 * H * 128:], do, lse, dsum, dq, dk, dqkv[:, :, 2 * H * 128:])
    else:
        for _ in range(5):
            ops.attention(q, k, qkv[:, :, 2 * H * 128:], o)
torch.cuda.synchronize()
