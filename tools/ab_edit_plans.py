"""Within-process interleaved A/B of whole edits under process-level switches: the pipeline is built once, every arm runs
once per round (1 warm-up + `edits` timed edits), medians over the rounds are printed, then one instrumented edit per arm
(per-family HIP-event sums, as bench.py).  An arm = comma-separated switches:
    plan=<0..3>   fk_gemm_set_plan (bit 0 mixed grids, bit 1 split-K pairs)
    split=<0|1>   fk_attention_set_split (stream-K attention grids where the plain grid wastes a round)
    side=<0|1|auto>  transformer.OVERLAP_MLP (single blocks' MLP-up GEMM on a second stream)
    gm=<depth>    fk_gemm_set_group_m (row tiles per group of the tile order; 0 = default 8)

    AB_ARMS="plan=0;plan=1;plan=3" python tools/ab_edit_plans.py [workload] [rounds] [edits]
"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gpt_image_edit_amd import libfk, ops, transformer  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2_single_512x512_28step"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
edits = int(sys.argv[3]) if len(sys.argv) > 3 else 2
arms = [a.strip() for a in os.environ.get("AB_ARMS", "plan=0;plan=1;plan=3").split(";") if a.strip()]
lib = libfk.load()
DEFAULT_SIDE = transformer.OVERLAP_MLP


def apply(arm):
    ops.gemm_set_plan(3)
    ops.attention_set_split(1)
    transformer.OVERLAP_MLP = DEFAULT_SIDE
    ops.gemm_set_group_m(0)
    for kv in arm.split(","):
        k, v = kv.split("=")
        if k == "plan":
            ops.gemm_set_plan(int(v))
        elif k == "split":
            ops.attention_set_split(int(v))
        elif k == "gm":
            ops.gemm_set_group_m(int(v))
        elif k == "side":
            transformer.OVERLAP_MLP = {"0": False, "1": True}.get(v, "auto")
        else:
            raise SystemExit(f"unknown switch {k}")


dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
inp = bench.make_inputs(workload, dev, seed=42)
res = {a: [] for a in arms}
for r in range(rounds):
    for a in arms:
        apply(a)
        bench.run_edit(pipe, inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(edits):
            bench.run_edit(pipe, inp)
        torch.cuda.synchronize()
        res[a].append((time.perf_counter() - t0) / edits * 1e3)
for a in arms:
    print(f"{workload} [{a}]: ms/edit median {statistics.median(res[a]):.1f}  all {[round(x, 1) for x in res[a]]}", flush=True)
for a in arms:
    apply(a)
    fam = bench.instrumented_edit(pipe, inp)
    print(f"{workload} [{a}]: " + "  ".join(f"{k} {v['ms']:.1f} ms {v['tflops']:.0f} TF/s ({v['launches']})" for k, v in fam.items()),
          flush=True)
apply("plan=3")
