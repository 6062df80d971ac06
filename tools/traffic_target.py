"""Target of the HBM-traffic PMC passes (tools/pmc_traffic.sh): a fixed sequence of kernels at the path's real shapes,
each launched REPS times back to back and followed by a one-wave separator kernel (fk_silu on 8 elements), so that the
rocprofv3 per-dispatch counter rows can be grouped per item by position.  Writes the item list with each item's
ALGORITHMIC bytes / FLOPs per launch to gpurun_out/traffic_items.json.

Items whose operands exceed the 256 MiB Infinity Cache are the ones whose counters can be read as HBM traffic
(MI355X_MICROARCH.md: "scale past L3 before reading FETCH_SIZE as over-fetch evidence"); the small ones are listed
with that caveat.  A plain device copy of 1 GiB calibrates the counters' units on a known byte count.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402
from gpt_image_edit_amd.vae import HipAutoencoderKL  # noqa: E402

BF = torch.bfloat16
REPS = 3
items = []


def rnd(*s, scale=1.0):
    return ((torch.rand(*s, device="cuda") * 2 - 1) * scale).to(BF)


def sep():
    ops.silu(torch.zeros(8, device="cuda", dtype=BF))


SKIP = [t for t in os.environ.get("TRAFFIC_SKIP", "").split(",") if t]    # substrings of item names to leave out (saves box time)


def skipped(name):
    return any(t in name for t in SKIP)


def item(name, kernel_substr, fn, alg_read, alg_write, flops=0.0, note=""):
    if skipped(name):
        return
    for _ in range(REPS):
        fn()
    sep()
    items.append(dict(name=name, kernel=kernel_substr, reps=REPS, alg_read_bytes=alg_read, alg_write_bytes=alg_write,
                      flops=flops, note=note))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic_items.json"
    sep()
    # ---- calibration: 1 GiB device-to-device copy through torch's vectorised copy kernel and through ln_modulate ----
    n = 1 << 29
    src, dst = torch.empty(n, device="cuda", dtype=BF).normal_(), torch.empty(n, device="cuda", dtype=BF)
    item("calib_copy_1GiB", "AUnaryFunctor<c10::BFloat16", lambda: dst.copy_(src * 1), 2 * n, 2 * n,
         note="torch's vectorised `src * 1` kernel: 1 GiB read + 1 GiB written per launch (the copy_ that follows is a blit)")
    del src, dst
    D = 3072
    for B, S in ((1, 8704), (32, 8704)):
        if skipped(f"ln_modulate B{B} S{S}"):
            continue
        x, o = rnd(B, S, D), torch.empty(B, S, D, device="cuda", dtype=BF)
        mod = rnd(B, 6 * D, scale=0.3)
        item(f"ln_modulate B{B} S{S}", "ln_modulate", lambda: ops.ln_modulate(x, mod[:, :D], mod[:, D:2 * D], out=o),
             B * S * D * 2, B * S * D * 2, note="one read + one write of the [B,S,3072] stream")
        del x, o
    # ---- GEMMs (bf16 in / out; algorithmic = A + W read once, C written once) -------------------------------------
    # one item per launch class of the 512^2 and the 1024^2 edit (the launcher picks the form: mixed grid, split-K pair ...)
    for M, N, K, epi, tag in ((2560, 9216, 3072, 0, "qkv 512^2"), (2560, 12288, 3072, 1, "mlp-up 512^2 (GELU)"),
                              (2560, 3072, 15360, 0, "proj_out 512^2 (K-long)"), (2560, 3072, 3072, 0, "out-proj 512^2"),
                              (8704, 9216, 3072, 0, "qkv 1024^2"), (8704, 12288, 3072, 1, "mlp-up 1024^2 (GELU)"),
                              (8704, 3072, 15360, 0, "proj_out 1024^2 (K-long)"),
                              (278528, 3072, 3072, 0, "out-proj cfg3 (B=32)"),
                              (278528, 3072, 12288, 0, "ff.net.2 cfg3 (B=32, K-long: gemm10_kernel)")):
        if skipped(f"gemm {M}x{N}x{K} {tag}"):
            continue
        a, w, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N)
        c = torch.empty(M, N, device="cuda", dtype=BF)
        item(f"gemm {M}x{N}x{K} {tag}", "gemm", lambda: ops.gemm(a, w, b, out=c, epilogue=epi),
             (M * K + N * K) * 2, M * N * 2, flops=2.0 * M * N * K)
        del a, w, c
    # ---- attention (algorithmic: Q, K, V read once, O written once = 4 * S * 128 * 2 B per head) ------------------
    H = 24
    for B, S in ((1, 2560), (1, 8704), (8, 8704)):
        if skipped(f"attention B{B} S{S}"):
            continue
        q, k, qkv = rnd(B, H, S, 128), rnd(B, H, S, 128), rnd(B, S, 3 * H * 128)
        o = torch.empty(B, S, H * 128, device="cuda", dtype=BF)
        item(f"attention B{B} S{S}", "attention_fwd", lambda: ops.attention(q, k, qkv[:, :, 2 * H * 128:], o),
             3 * B * H * S * 128 * 2, B * H * S * 128 * 2, flops=4.0 * B * H * S * S * 128,
             note="K/V of one head are re-read by every query block of that head (from L2 / MALL when they fit)")
        del q, k, qkv, o
    # ---- VAE decode at the 1024^2 size: conv (implicit GEMM) and GroupNorm kernels ---------------------------------
    if not skipped("vae.decode B4 128x128 latent"):
        vae = HipAutoencoderKL(device="cuda", init="synthetic", seed=1)
        z = rnd(4, 16, 128, 128)
        vae.decode(z, return_dict=False)     # packs weights
        sep()
    item("vae.decode B4 128x128 latent", "*", lambda: vae.decode(z, return_dict=False), 4 * 13.4e9 / 2, 4 * 13.4e9 / 2,
         flops=4 * 10.47e12, note="whole decoder (all its kernels summed): BASELINE.md quotes ~13.4 GB fused-ideal activation traffic per 1024^2 image")
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(items, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
