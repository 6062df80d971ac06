import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A GPU test that hangs the device (a deadlocked inter-workgroup hand-off) must end the run, not sit in it until the
    # box's own limit: pytest-timeout's thread method (a blocked hipDeviceSynchronize never returns to the interpreter).
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
                item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def bf16_ulp_diff(a, b):
    """Element-wise distance in bf16 ulps between two bf16 tensors (on CPU)."""
    import torch

    ai = a.contiguous().view(torch.int16).to(torch.int32)
    bi = b.contiguous().view(torch.int16).to(torch.int32)
    # map sign-magnitude to a monotonic integer line
    ai = torch.where(ai < 0, -(ai & 0x7FFF), ai)
    bi = torch.where(bi < 0, -(bi & 0x7FFF), bi)
    return (ai - bi).abs()


def report(name, got, ref):
    """Print error statistics (kept in the pytest -s log that comes back from the GPU box)."""
    import torch

    g, r = got.float().cpu(), ref.float().cpu()
    d = (g - r).abs()
    rel = d / (r.abs() + 1e-6)
    print(f"[parity] {name}: max_abs={d.max().item():.3e} mean_abs={d.mean().item():.3e} "
          f"max_rel={rel.max().item():.3e} ref_absmax={r.abs().max().item():.3e} "
          f"nan={int(torch.isnan(g).sum())}", flush=True)
    return d
