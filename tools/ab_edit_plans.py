"""Within-process interleaved A/B of whole edits under different GEMM launch plans (fk_gemm_set_plan) and other
process-level switches: the pipeline is built once, every arm runs once per round (1 warm-up + `edits` timed edits),
medians over the rounds are printed, then one instrumented edit per arm (per-family HIP-event sums, as bench.py).

    python tools/ab_edit_plans.py [workload] [rounds] [edits]         AB_PLANS="0,1,3" (default)
"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gpt_image_edit_amd import ops  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2_single_512x512_28step"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
edits = int(sys.argv[3]) if len(sys.argv) > 3 else 2
plans = [int(v) for v in os.environ.get("AB_PLANS", "0,1,3").split(",")]
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
inp = bench.make_inputs(workload, dev, seed=42)
res = {p: [] for p in plans}
for r in range(rounds):
    for p in plans:
        ops.gemm_set_plan(p)
        bench.run_edit(pipe, inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(edits):
            bench.run_edit(pipe, inp)
        torch.cuda.synchronize()
        res[p].append((time.perf_counter() - t0) / edits * 1e3)
for p in plans:
    print(f"{workload} plan {p}: ms/edit median {statistics.median(res[p]):.1f}  all {[round(x, 1) for x in res[p]]}", flush=True)
for p in plans:
    ops.gemm_set_plan(p)
    fam = bench.instrumented_edit(pipe, inp)
    print(f"{workload} plan {p}: " + "  ".join(f"{k} {v['ms']:.1f} ms {v['tflops']:.0f} TF/s ({v['launches']})" for k, v in fam.items()),
          flush=True)
ops.gemm_set_plan(3)
