"""Host logic of the denoiser's optimisation step (section 8(f) rank 3) -- scalars and names only.

What is here is the per-step bookkeeping of the reference's ``train_denoiser.py`` that does not touch activations:
which parameters train (:70-122), how a step's noise levels are drawn and shifted (:935-993, :779-788) and how the
loss is weighted.  The tensor work of the step runs in libfk: noisy-input mix fused with the token packing, the
flow-matching loss fused with its gradient, the global gradient norm and AdamW (``csrc/train_kernels.hip``,
``ops.flow_noisy_tokens / flow_loss / sumsq / adamw_step``), the MMDiT forward (``transformer.py``) and its backward
(``backward.py``); ``oracle/train.py`` is the CPU restatement they are checked against.  Nothing here substitutes
torch arithmetic for any of it.
"""
import math

import torch

from . import helpers

__all__ = ["get_trainable_params", "check_param_is_in_components", "trainable_names", "apply_flux_schedule_shift",
           "sample_sigmas", "get_sigmas", "loss_weighting"]

DOUBLE_COMPONENTS = ("attn.norm_q", "attn.norm_k", "attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out", "norm1.linear")
SINGLE_COMPONENTS = ("attn.norm_q", "attn.norm_k", "attn.to_q", "attn.to_k", "attn.to_v", "norm.linear")
DOUBLE_TEXT_BRANCH = ("norm1_context.linear", "attn.norm_added_q", "attn.norm_added_k", "ff.net", "ff_context.net")
SINGLE_TEXT_BRANCH = ("proj_mlp", "proj_out")


def get_trainable_params(layers_to_train=tuple(range(57)), num_transformer_blocks=19, only_img_branch=True):
    """``train_denoiser.py:70-118``: name fragments (under ``denoise_tower.denoiser.``) of what is unfrozen; layer
    l < 19 is ``transformer_blocks.l``, otherwise ``single_transformer_blocks.(l - 19)``."""
    double = list(DOUBLE_COMPONENTS) + ([] if only_img_branch else list(DOUBLE_TEXT_BRANCH))
    single = list(SINGLE_COMPONENTS) + ([] if only_img_branch else list(SINGLE_TEXT_BRANCH))
    components = []
    for layer in layers_to_train:
        if layer < num_transformer_blocks:
            prefix, comps = f"denoise_tower.denoiser.transformer_blocks.{layer}", double
        else:
            prefix, comps = f"denoise_tower.denoiser.single_transformer_blocks.{layer - num_transformer_blocks}", single
        components.extend(f"{prefix}.{c}" for c in comps)
    return components


def check_param_is_in_components(name, components):
    """``train_denoiser.py:121-122``: plain substring match (so ``...blocks.1.`` fragments never match ``...blocks.12.``
    only because every fragment ends in a component name)."""
    return any(component in name for component in components)


def trainable_names(state_dict_keys, layers_to_train=tuple(range(57)), only_img_branch=True,
                    prefix="denoise_tower.denoiser."):
    """The keys of a denoiser state dict (without the UniWorld prefix) that the reference would leave trainable."""
    comps = get_trainable_params(layers_to_train, only_img_branch=only_img_branch)
    return [k for k in state_dict_keys if check_param_is_in_components(prefix + k, comps)]


def apply_flux_schedule_shift(sigmas, latent_h, latent_w, base_image_seq_len=256, max_image_seq_len=4096,
                              base_shift=0.5, max_shift=1.15):
    """``train_denoiser.py:972-986``: sigma * e^mu / (1 + (e^mu - 1) * sigma), mu linear in the packed length h*w/4."""
    mu = helpers.calculate_shift((latent_h * latent_w) // 4, base_image_seq_len, max_image_seq_len, base_shift, max_shift)
    shift = math.exp(mu)
    return (sigmas * shift) / (1 + (shift - 1) * sigmas)


def sample_sigmas(bsz, latent_h, latent_w, generator=None, device="cpu", **scheduler_config):
    """``discrete_timestep: false`` (:988-993): (sigmas [B] fp32, timesteps = 1000 * sigmas)."""
    sigmas = torch.sigmoid(1.0 * torch.randn((bsz,), generator=generator, device=device, dtype=torch.float32))
    sigmas = apply_flux_schedule_shift(sigmas, latent_h, latent_w, **scheduler_config)
    return sigmas, sigmas * 1000.0


def get_sigmas(timesteps, schedule_timesteps, schedule_sigmas, n_dim=4, dtype=torch.float32):
    """``discrete_timestep: true`` (:779-788): the schedule's sigma at each drawn timestep, shaped to broadcast."""
    step_indices = [(schedule_timesteps == t).nonzero().item() for t in timesteps]
    sigma = schedule_sigmas.to(dtype)[step_indices].flatten()
    while len(sigma.shape) < n_dim:
        sigma = sigma.unsqueeze(-1)
    return sigma


def loss_weighting(weighting_scheme, sigmas, sigmas_as_weight=False):
    """:1107-1110 -- ``compute_loss_weighting_for_sd3`` of diffusers (sigma_sqrt, cosmap, else ones) or the sigmas."""
    if sigmas_as_weight:
        return sigmas
    if weighting_scheme == "sigma_sqrt":
        return (sigmas ** -2.0).float()
    if weighting_scheme == "cosmap":
        return 2 / (math.pi * (1 - 2 * sigmas + 2 * sigmas ** 2))
    return torch.ones_like(sigmas)
