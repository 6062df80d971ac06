"""FlowMatchEulerDiscreteScheduler as the reference pipeline drives it (host logic + one HIP kernel).

Interface used by ``univa/utils/flux_pipeline.py``: ``config.get(...)`` (:994-999),
``set_timesteps(sigmas=, mu=, device=)`` via ``retrieve_timesteps`` (:1000-1006), ``timesteps``,
``order``, ``set_begin_index(0)`` (:1052), ``step(noise_pred, t, latents, return_dict=False)[0]`` (:1099).
The sigma schedule is host arithmetic in float32 (28 numbers); the update itself is
``fk_euler_step_bf16`` with the reference's rounding (bf16(dsigma) * v in bf16, sum in fp32 -> bf16).
"""
import math

import numpy as np
import torch

from . import flux_spec, ops


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, **config):
        self.config = dict(flux_spec.SCHEDULER_CONFIG)
        self.config.update(config)
        self.timesteps = None
        self.sigmas = None
        self._sigmas_host = None
        self._step_index = None
        self._begin_index = None

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index=0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        cfg = self.config
        if cfg["use_dynamic_shifting"] and mu is None:
            raise ValueError(" you have a pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            ts = np.linspace(cfg["num_train_timesteps"], 1, num_inference_steps)
            sigmas = ts / cfg["num_train_timesteps"]
        s = np.array(sigmas).astype(np.float32)
        if cfg["use_dynamic_shifting"]:
            s = math.exp(mu) / (math.exp(mu) + (1 / s - 1) ** 1.0)
        else:
            s = cfg["shift"] * s / (1 + (cfg["shift"] - 1) * s)
        s = np.asarray(s, dtype=np.float32)
        self.num_inference_steps = len(s)
        self._sigmas_host = np.concatenate([s, np.zeros(1, dtype=np.float32)])
        self.timesteps = torch.from_numpy(s * np.float32(cfg["num_train_timesteps"])).to(device=device)
        self.sigmas = torch.from_numpy(self._sigmas_host.copy()).to(device=device)
        self._step_index = None
        self._begin_index = None

    def dsigma(self, i):
        """sigma[i+1] - sigma[i] in float32, as the reference forms it on the device."""
        return float(np.float32(self._sigmas_host[i + 1]) - np.float32(self._sigmas_host[i]))

    def _init_step_index(self, timestep):
        if self._begin_index is None:
            t = float(timestep)
            idx = (self.timesteps.cpu() == t).nonzero()
            self._step_index = int(idx[1 if len(idx) > 1 else 0])
        else:
            self._step_index = self._begin_index

    def step(self, model_output, timestep, sample, return_dict=True, **unused):
        """x <- x.float() + (sigma_next - sigma) * v, cast to v.dtype (out of place, like the reference)."""
        if self._step_index is None:
            self._init_step_index(timestep)
        if model_output.dtype != torch.bfloat16 or sample.dtype != torch.bfloat16:
            raise TypeError("the HIP scheduler step works on bf16 latents")
        prev = sample.contiguous().clone()
        B, S, _ = prev.shape
        ops.euler_step(prev, model_output.contiguous(), S, self.dsigma(self._step_index))
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return {"prev_sample": prev}
