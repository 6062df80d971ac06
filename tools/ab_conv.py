"""3 x 3 convolutions of the VAE decoder at the 1024^2 sizes: implicit-GEMM kernel (+ separate GroupNorm-apply pass) vs the LDS
halo-tiled kernel with the GroupNorm + SiLU prologue (csrc/conv_halo.hip), interleaved in one process."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402
from gpt_image_edit_amd.vae import _pack_conv  # noqa: E402

BF = torch.bfloat16
for (B, H, W, cin, cout, up) in [(1, 128, 128, 512, 512, False), (1, 128, 128, 512, 512, True), (1, 256, 256, 512, 512, False),
                                 (1, 512, 512, 256, 256, False), (1, 1024, 1024, 128, 128, False), (1, 512, 512, 512, 256, False)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, W, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.03).to(BF)
    bias = torch.randn(cout, device="cuda", generator=g).to(BF)
    gamma = (1 + 0.1 * torch.randn(cin, device="cuda", generator=g)).to(BF)
    beta = (0.1 * torch.randn(cin, device="cuda", generator=g)).to(BF)
    wp = _pack_conv(wt)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    fl = 2.0 * B * Ho * Wo * cout * 9 * cin

    def old():
        xn = ops.group_norm_nhwc(x, gamma, beta, True) if not up else x
        return ops.conv2d_nhwc(xn, wp, bias, cout, ksize=3, stride=1, pad=1, upsample2x=up)

    def new():
        gn = (ops.group_norm_stats(x), gamma, beta, True) if not up else None
        return ops.conv3x3_halo(x, wp, bias, cout, upsample2x=up, gn=gn)
    res = {"implicit-gemm (+GN pass)": [], "halo (+GN prologue)": []}
    for r in range(4):
        for name, fn in (("implicit-gemm (+GN pass)", old), ("halo (+GN prologue)", new)):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            if r:
                res[name].append(e0.elapsed_time(e1) / 5)
    print(f"conv3x3 B{B} {H}x{W}{' up2x' if up else ''} {cin}->{cout}: " + "  ".join(
        f"{k}: {statistics.median(v) * 1e3:.0f} us = {fl / (statistics.median(v) * 1e-3) / 1e12:.0f} TF/s" for k, v in res.items()), flush=True)
