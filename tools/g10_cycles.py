"""gemm10 measurement forms (library built with -DFK_G10_EXPERIMENTS, FK_G10_X=<n>): loop duration in shader cycles per K-tile
and wave (s_memtime either side of the asm statement, written over the first bytes of C) and the launch's TF/s, beside gemm8.
    FK_G10_X=3 python tools/g10_cycles.py [M N K]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import ops  # noqa: E402

BF = torch.bfloat16
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 3072, 12288)
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF)
w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(BF)
out_full = torch.zeros(M + 256, N, device="cuda", dtype=BF)      # the measurement forms write their records behind C's last row
out = out_full[:M]
fl = 2.0 * M * N * K
tiles = (M // 256) * (N // 256)


def rate(variant, n=6):
    ops.gemm_set_variant(variant)
    ops.gemm_set_mfma(16)
    for _ in range(2):
        ops.gemm(a, w, None, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, None, out=out)
    e1.record()
    e1.synchronize()
    return fl * n / (e0.elapsed_time(e1) * 1e-3) / 1e12


r8 = rate(256)
r10 = rate(1024)
line = f"X={os.environ.get('FK_G10_X', '0'):>2} {M}x{N}x{K}: gemm8 {r8:6.0f} TF/s  gemm10 form {r10:6.0f} TF/s"
if os.environ.get("FK_G10_X", "0") != "0":
    n_rec = min(tiles, 256) if os.environ.get("FK_G10_PERSIST", "0") != "0" else tiles
    rec = out_full[M:].reshape(-1).view(torch.int64)[: n_rec * 8].reshape(n_rec, 8).cpu().tolist()
    nkt = K // 64
    loop = sorted((r[5] - r[4]) / nkt for r in rec)
    pre = sorted(r[4] - r[3] for r in rec)
    pre_c = sorted(r[7] - r[3] for r in rec)
    epi = sorted(r[6] - r[5] for r in rec)
    med = statistics.median
    line += (f"  loop cycles per K-tile: median {med(loop):7.1f} p10 {loop[len(loop) // 10]:7.1f} p90 {loop[9 * len(loop) // 10]:7.1f} (MFMAs alone 2048)"
             f"  |  per tile: entry->loop {med(pre):6.0f} (of which before the asm statement {med(pre_c):6.0f})  loop {med(loop) * nkt:8.0f}  loop->exit {med(epi):6.0f} cycles")
    # per CU: wall-clock gaps between consecutive workgroups (s_memrealtime ticks: 100 MHz) and the share of time inside loops
    if os.environ.get("FK_G10_PERSIST", "0") != "0":
        print(line + "\n      (persistent grid: one record slot per workgroup, the last tile's)", flush=True)
        sys.exit(0)
    by_cu = {}
    for r in rec:
        by_cu.setdefault((r[0] >> 32, r[0] & 0xffffffc0), []).append(r)       # (XCC_ID, HW_ID without wave / SIMD bits)
    gaps, spans, busy = [], [], []
    for lst in by_cu.values():
        lst.sort(key=lambda r: r[1])
        gaps += [b[1] - a[2] for a, b in zip(lst, lst[1:])]
        spans.append(lst[-1][2] - lst[0][1])
        busy.append(sum(r[2] - r[1] for r in lst))
    if gaps:
        gaps.sort()
        line += (f"\n      {len(by_cu)} CUs seen, {len(rec) / len(by_cu):.2f} workgroups each; gap between consecutive workgroups on a CU: median "
                 f"{med(gaps) / 100:.2f} us p90 {gaps[9 * len(gaps) // 10] / 100:.2f} us; workgroup wall time median {med([r[2] - r[1] for r in rec]) / 100:.1f} us;"
                 f" CU span median {med(spans) / 100:.1f} us (max {max(spans) / 100:.1f}), of which inside workgroups {100 * sum(busy) / sum(spans):.1f} %")
print(line, flush=True)
