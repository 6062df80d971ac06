"""Per-kernel micro-benchmarks on one MI355X (development aid; bench.py is the contract benchmark).

    python tools/bench_kernels.py [--quick]

Times the hot kernels at the shapes of BASELINE.json's configs with HIP events on the launch stream and
prints achieved TFLOP/s (MFMA kernels) or GB/s (HBM-bound kernels) next to the gfx950 peaks.
Inputs are uniform random in [-1, 1) (never zero-filled: clocks and softmax work depend on the data).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*shape):
    return (torch.rand(*shape, device="cuda") * 2 - 1).to(BF)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    rows = []
    D = 3072
    Ms = [2560] if args.quick else [2560, 8704, 32768]
    for M in Ms:
        for (N, K, epi, name) in [(3 * D, D, ops.FK_EPI_NONE, "qkv"), (D, D, ops.FK_EPI_NONE, "out"),
                                  (4 * D, D, ops.FK_EPI_GELU_TANH, "mlp_up"), (D, 4 * D, ops.FK_EPI_NONE, "mlp_down"),
                                  (D, 5 * D, ops.FK_EPI_NONE, "single_out")]:
            a, w, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
            out = torch.empty(M, N, device="cuda", dtype=BF)
            t = timeit(lambda: ops.gemm(a, w, b, out=out, epilogue=epi))
            tf = 2.0 * M * N * K / t / 1e12
            rows.append(dict(kernel=f"gemm_{name}", M=M, N=N, K=K, ms=t * 1e3, tflops=tf, frac=tf / PEAK_TF))
            print(rows[-1], flush=True)
    for (B, S) in ([(1, 2560)] if args.quick else [(1, 2560), (1, 8704), (4, 8704)]):
        H = 24
        q, k = rnd(B, H, S, 128), rnd(B, H, S, 128)
        qkv = rnd(B, S, 3 * D)
        o = torch.empty(B, S, H * 128, device="cuda", dtype=BF)
        t = timeit(lambda: ops.attention(q, k, qkv[:, :, 2 * D:], o))
        tf = 4.0 * B * H * S * S * 128 / t / 1e12
        rows.append(dict(kernel="attention", B=B, S=S, ms=t * 1e3, tflops=tf, frac=tf / PEAK_TF))
        print(rows[-1], flush=True)
        wn = rnd(128)
        cos, sin = torch.rand(S, 128, device="cuda"), torch.rand(S, 128, device="cuda")
        t = timeit(lambda: ops.qkv_post(qkv, q, k, wn, wn, wn, wn, cos, sin, 512))
        gbs = (2 * B * S * 2 * D * 2) / t / 1e9
        rows.append(dict(kernel="qkv_post", B=B, S=S, ms=t * 1e3, gbs=gbs, frac=gbs / PEAK_GBS))
        print(rows[-1], flush=True)
        x, mod = rnd(B, S, D), rnd(B, 6 * D)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.ln_modulate(x, mod[:, :D], mod[:, D:2 * D], out=y))
        gbs = (2 * B * S * D * 2) / t / 1e9
        rows.append(dict(kernel="ln_modulate", B=B, S=S, ms=t * 1e3, gbs=gbs, frac=gbs / PEAK_GBS))
        print(rows[-1], flush=True)
    # modulation GEMM: weight streaming, M = batch
    for B in (1, 32):
        Ntot = 19 * 12 * D + 38 * 3 * D + 2 * D
        a, w, b = rnd(B, D), rnd(Ntot, D) * 0.05, rnd(Ntot)
        out = torch.empty(B, Ntot, device="cuda", dtype=BF)
        t = timeit(lambda: ops.gemm(a, w, b, out=out), iters=5)
        gbs = Ntot * D * 2 / t / 1e9
        rows.append(dict(kernel="modulation_gemm", B=B, ms=t * 1e3, gbs=gbs, frac=gbs / PEAK_GBS))
        print(rows[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_kernels.json", "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
