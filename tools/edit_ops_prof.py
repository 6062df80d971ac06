"""Which torch operators still launch kernels inside one cfg 2 edit (VAE encode + N x MMDiT + VAE decode): one warm edit,
then one edit of EDIT_STEPS (default 4) denoise steps under torch.profiler; operators by device time with input shapes and
the Python frames that issued them.  `python tools/edit_ops_prof.py [rows]`"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
steps = int(os.environ.get("EDIT_STEPS", "4"))
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
pipe = bench.build_pipeline(device)
inp = bench.make_inputs(os.environ.get("EDIT_WORKLOAD", "cfg2_single_512x512_28step"), device, 0)
bench.run_edit(pipe, inp, steps)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.run_edit(pipe, inp, steps)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=5)
evs = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
n = 0
for e in evs:
    t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if t <= 0 or not e.key.startswith(("aten::", "Memcpy", "Memset")):
        continue
    stack = " <- ".join(s.split("/")[-1] for s in (e.stack or [])[:5])
    print(f"{t / 1e3:9.3f} ms  x{e.count:5d}  {e.key:28s} {str(e.input_shapes)[:70]:70s} {stack[:230]}")
    n += 1
    if n >= rows:
        break
