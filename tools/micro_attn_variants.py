"""Time fk_attention_fwd_bf16 at the edit's shapes (FK_ATTN_VARIANT selects an experimental instantiation)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
torch.manual_seed(0)
for (B, H, S) in [(1, 24, 2560), (1, 24, 5632), (1, 24, 8704), (4, 24, 8704)]:
    q = torch.randn(B, H, S, 128, device="cuda").to(BF)
    k = torch.randn(B, H, S, 128, device="cuda").to(BF)
    v = torch.randn(B, S, H * 128, device="cuda").to(BF)
    o = torch.empty(B, S, H * 128, device="cuda", dtype=BF)
    for _ in range(5):
        ops.attention(q, k, v, o)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        ops.attention(q, k, v, o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"variant {os.environ.get('FK_ATTN_VARIANT', '0')} B{B} H{H} S{S}: {dt * 1e6:.1f} us "
          f"{4 * B * H * S * S * 128 / dt / 1e12:.0f} TF/s", flush=True)
