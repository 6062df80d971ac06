"""Builds and runs the gfx950 hardware-semantics probe on the GPU box and keeps its output.

Not a parity test: it records measured instruction semantics (LDS transpose read, LDS-DMA) under
gpurun_out/ for the next kernel round.  Skipped when hipcc or a GPU is unavailable.
"""
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lds_probe(tmp_path):
    if not torch.cuda.is_available() or shutil.which("hipcc") is None:
        pytest.skip("needs GPU + hipcc")
    exe = tmp_path / "probe_lds"
    src = os.path.join(ROOT, "tests", "probes", "probe_lds.hip")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", src, "-o", str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "probe_lds.txt"), "w") as f:
        f.write(r.stdout + "\n--- stderr ---\n" + r.stderr)
    assert r.returncode == 0 and "status: no error" in r.stdout
