"""Write the attention forward's output for one shape (the kernel FK_ATTN_KERNEL selects) to a file, twice in a row, and
report whether the two launches agree; tools/attn_diff.py compares two such files row by row.
    FK_ATTN_KERNEL=4 python tools/attn_dump.py B H S out.pt [grid]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

B, H, S = (int(v) for v in sys.argv[1:4])
if len(sys.argv) > 5:
    ops.attention_set_split(int(sys.argv[5]))
g = torch.Generator(device="cuda").manual_seed(1000 * S + B)
q = torch.randn(B, H, S, 128, device="cuda", generator=g).to(torch.bfloat16)
k = torch.randn(B, H, S, 128, device="cuda", generator=g).to(torch.bfloat16)
qkv = torch.randn(B, S, 3 * H * 128, device="cuda", generator=g).to(torch.bfloat16)
outs = []
for _ in range(3):
    o = torch.zeros(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, H, S, device="cuda", dtype=torch.float32)
    ops.attention_lse(q, k, qkv[:, :, 2 * H * 128:], o, lse)
    torch.cuda.synchronize()
    outs.append(o.cpu())
print("launches agree:", torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]))
torch.save(outs[0], sys.argv[4])
torch.save(lse.cpu(), sys.argv[4] + '.lse')
