"""Oracle: ``FlowMatchEulerDiscreteScheduler`` (diffusers 0.32.2) as the pipeline uses it.

TEST INFRASTRUCTURE.  PARITY UNPINNED against the third-party source; restated from SURVEY.md
Appendix A.2.  Reference call sites: ``univa/utils/flux_pipeline.py:991-1008`` (sigmas + mu),
``:1052`` (set_begin_index), ``:1099`` (step).
"""
import math

import numpy as np
import torch


def shifted_sigmas(num_inference_steps, mu, sigmas=None):
    """set_timesteps(sigmas=linspace(1, 1/N, N), mu) -> (timesteps [N] fp32, sigmas [N+1] fp32)."""
    if sigmas is None:
        sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
    s = np.array(sigmas).astype(np.float32)
    s = math.exp(mu) / (math.exp(mu) + (1 / s - 1) ** 1.0)  # dynamic shifting; stays float32
    s = torch.from_numpy(np.asarray(s, dtype=np.float32))
    timesteps = s * 1000.0
    return timesteps, torch.cat([s, torch.zeros(1)])


def euler_step(model_output, sigma, sigma_next, sample):
    """scheduler.step: x.float() + (sigma_next - sigma) * v, cast back to v.dtype.

    ``sigma``/``sigma_next`` are 0-dim fp32 tensors, so torch's promotion rules apply exactly as in
    the reference (with a bf16 ``model_output`` the product is formed in bf16).
    """
    x = sample.to(torch.float32)
    prev = x + (sigma_next - sigma) * model_output
    return prev.to(model_output.dtype)
