"""Rows / heads in which two attention outputs (tools/attn_dump.py) differ.  python tools/attn_diff.py a.pt b.pt H"""
import sys

import torch

a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
H = int(sys.argv[3])
B, S, D = a.shape
d = (a.float() - b.float()).view(B, S, H, 128)
bad = (d != 0).any(-1)            # [B, S, H]
print("differing (row, head) pairs:", int(bad.sum()), "of", bad.numel(), " max |d|", d.abs().max().item())
for bb in range(B):
    for h in range(H):
        rows = bad[bb, :, h].nonzero().flatten()
        if len(rows):
            blocks = sorted(set((rows // 256).tolist()))
            print(f"  b{bb} h{h}: {len(rows)} rows, 256-row blocks {blocks[:12]}{'...' if len(blocks) > 12 else ''}  elements/row {float((d[bb, rows, h] != 0).sum(-1).float().mean()):.1f}")

import os
if os.path.exists(sys.argv[1] + ".lse"):
    la, lb = torch.load(sys.argv[1] + ".lse"), torch.load(sys.argv[2] + ".lse")      # [B, H, S]
    dl = (la != lb)
    print("rows whose lse differs:", int(dl.sum()), "of", dl.numel(), " max |d|", (la - lb).abs().max().item())
    both = bad.permute(0, 2, 1) & dl
    print("rows with an output difference that also differ in lse:", int(both.sum()), "of", int(bad.sum()))
    for bb in range(B):
        for h in range(min(H, 4)):
            rows = dl[bb, h].nonzero().flatten()
            if len(rows):
                blocks = sorted(set((rows // 256).tolist()))
                print(f"  lse b{bb} h{h}: {len(rows)} rows in 256-row blocks {blocks}; rows mod 64 histogram (32-row halves): {[int(((rows % 64) // 32 == x).sum()) for x in (0, 1)]}")
