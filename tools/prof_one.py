"""Run ONE kernel shape a few times (target for `rocprofv3 --pmc ...`).

    python tools/prof_one.py gemm 2560 9216 3072 [epi]      |  attention B S
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
    for _ in range(5):
        ops.attention(q, k, qkv[:, :, 2 * H * 128:], o)
torch.cuda.synchronize()
