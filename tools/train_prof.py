"""Target of the train-step kernel trace (rocprofv3 --kernel-trace --stats): bench.train_step_bench with 2 warm-up + 3 timed
steps (BASELINE.json configs[4] on one GPU); prints the bench dict."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
r = bench.train_step_bench(torch.device("cuda", 0), steps=int(os.environ.get("TRAIN_STEPS", "3")), warmup=int(os.environ.get("TRAIN_WARMUP", "2")),
                           e2e=os.environ.get("TRAIN_E2E", "1") != "0")
print(json.dumps(r))
