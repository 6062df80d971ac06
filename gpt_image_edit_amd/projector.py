"""``denoise_projector`` of ``UnivaDenoiseTower`` on HIP.

Reference: ``univa/models/modeling_univa_denoise_tower.py:31-47`` builds
``nn.Sequential(Linear(3584, 12288), SiLU(), Linear(12288, 4096))`` and
``univa/models/qwen2p5vl/modeling_univa_qwen2p5vl.py:521-523`` applies it to the VLM's last hidden states to make
the first part of ``prompt_embeds``.  Here: two ``fk_gemm_bf16`` calls, the SiLU fused into the first one's epilogue.
Parameter names are the Sequential's (``0.weight, 0.bias, 2.weight, 2.bias``), i.e. what
``checkpoint.read_projector`` returns.
"""
import torch
from torch import nn

from . import flux_spec, ops

BF16 = torch.bfloat16


class HipDenoiseProjector(nn.Module):
    def __init__(self, input_hidden_size=3584, output_hidden_size=4096, device="cuda", init="empty", seed=0):
        super().__init__()
        shapes = {k[len("denoise_projector."):]: v
                  for k, v in flux_spec.projector_param_shapes(input_hidden_size, output_hidden_size).items()}
        if init == "synthetic":
            state = flux_spec.synthetic_state(shapes, seed=seed, device=device, dtype=BF16)
        else:
            state = {k: torch.empty(s, device=device, dtype=BF16) for k, s in shapes.items()}
        for k, v in state.items():
            self.register_parameter(k.replace(".", "__"), nn.Parameter(v, requires_grad=False))

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        return type(sd)((k.replace("__", "."), v) for k, v in sd.items())

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        return super().load_state_dict({k.replace(".", "__"): v.to(BF16) for k, v in state_dict.items()}, strict=strict, **kwargs)

    @torch.no_grad()
    def forward(self, hidden_states):
        """[B, L, 3584] (or [L, 3584]) bf16 -> [B, L, 4096] bf16."""
        if not hidden_states.is_cuda:
            raise RuntimeError("HipDenoiseProjector needs GPU tensors: there is no CPU fallback")
        x = hidden_states.to(BF16).contiguous()
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        h = ops.gemm(x, getattr(self, "0__weight"), getattr(self, "0__bias"), epilogue=ops.FK_EPI_SILU)
        y = ops.gemm(h, getattr(self, "2__weight"), getattr(self, "2__bias"))
        return y[0] if squeeze else y
