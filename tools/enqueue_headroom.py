"""How far ahead of the GPU the host runs: wall time to ENQUEUE one edit (python + ctypes launches, no sync)
against the time the GPU needs to execute it.  `python tools/enqueue_headroom.py [workload]` on a GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2_single_512x512_28step"
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev)
    inp = bench.make_inputs(workload, dev, 42)
    bench.run_edit(pipe, inp)
    torch.cuda.synchronize()
    # host-only cost: the same number of launches on a tiny problem (GPU time per kernel ~ launch latency)
    bench.WORKLOADS["tiny"] = (1, 64, 64, 64, 64, 64)
    tiny = bench.make_inputs("tiny", dev, 42)
    bench.run_edit(pipe, tiny)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter()
        bench.run_edit(pipe, tiny)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"tiny (64x64, same launch count): enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms",
              flush=True)
    for _ in range(3):
        t0 = time.perf_counter()
        bench.run_edit(pipe, inp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{workload}: enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms "
              f"(host busy {100 * (t1 - t0) / (t2 - t0):.0f} % of the edit)", flush=True)


if __name__ == "__main__":
    main()
