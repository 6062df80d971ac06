"""Development: run the FK_TRACE build of the 4-wave GEMM and print per-iteration s_memtime deltas
(events: 0 loop top, 1 before vmcnt wait, 2 after it, 3 after lgkmcnt, 4 after the barrier)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import libfk, ops  # noqa: E402

M, N, K = 32768, 3072, 12288
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
trace = torch.zeros(64 * 4 * 128, device="cuda", dtype=torch.int32)
args, _ = ops._gemm_args(a, w, None, out, 0, None, None, False, 1.0)
args.rope_cos = trace.data_ptr()
lib = libfk.load()
for _ in range(3):
    libfk.check(lib.fk_gemm_bf16(ctypes.byref(args), torch.cuda.current_stream().cuda_stream), "gemm")
torch.cuda.synchronize()
t = trace.cpu().view(64, 4, 16, 8).long()
names = os.environ.get("FK_TRACE_EVENTS", "k0,k1,k2,k3,pre_lgkm,pre_barrier").split(",")
order = [int(x) for x in os.environ.get("FK_TRACE_ORDER", "0,1,2,4,5,3").split(",")]   # program order of the event ids
for blk in (0, 1):
    for wv in range(4):
        ev = t[blk, wv]
        top = ev[:, order[0]]
        per_iter = (top[1:] - top[:-1]) & 0xffffffff
        segs = []
        for a, b in zip(order[:-1], order[1:]):
            segs.append(f"{names[a]}->{names[b]} {(((ev[:, b] - ev[:, a]) & 0xffffffff).float().mean().item()):.0f}")
        last = ((ev[1:, order[0]] - ev[:-1, order[-1]]) & 0xffffffff).float().mean().item()
        print(f"block {blk * 64} wave {wv}: iter {per_iter.float().mean():.0f} cyc (min {per_iter.min()} max {per_iter.max()}); " + ", ".join(segs) + f", {names[order[-1]]}->next {last:.0f}")
