"""Oracle: one optimisation step of the denoiser.  TEST INFRASTRUCTURE.

Restates the arithmetic of ``train_denoiser.py:829-1181`` for the configuration the reference ships for FLUX-Kontext
(``scripts/denoiser/flux_qwen2p5vl_7b_vlm_stage2_1024.yaml``: continuous timesteps, ``weighting_scheme`` logit_normal
(= unit weights), guidance 1.0, AdamW(lr 1e-6, betas (0.9, 0.99), eps 1e-8, weight decay 0), gradient clipping at 1.0,
``only_tune_image_branch``), with the MMDiT of ``oracle/mmdit.py`` differentiated by torch autograd.

Pinned against the reference's own text (``oracle/make_golden.py::g_train`` lifts the functions out of
``train_denoiser.py`` with ``ast`` and runs them): ``get_trainable_params`` :70-118, ``check_param_is_in_components``
:121-122, the nested ``calculate_shift`` / ``apply_flux_schedule_shift`` :960-986 and ``get_sigmas`` :779-788.
PARITY UNPINNED for what lives in diffusers' ``training_utils`` (not installed): ``compute_density_for_timestep_sampling``
and ``compute_loss_weighting_for_sd3`` are restated from the SD3 paper (section 3.1) as that module publishes them.
No HIP counterpart of the backward pass exists yet; this file is the checker it will be built against.
"""
import math

import torch

from . import helpers, mmdit


# ---- which parameters train (train_denoiser.py:70-122) ------------------------------------------------------
def get_trainable_params(layers_to_train=tuple(range(57)), num_transformer_blocks=19, only_img_branch=True):
    """Name fragments of the trainable denoiser parameters; layer l < 19 is double block l, else single block l - 19."""
    double = ["attn.norm_q", "attn.norm_k", "attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out", "norm1.linear"]
    single = ["attn.norm_q", "attn.norm_k", "attn.to_q", "attn.to_k", "attn.to_v", "norm.linear"]
    if not only_img_branch:
        double = double + ["norm1_context.linear", "attn.norm_added_q", "attn.norm_added_k", "ff.net", "ff_context.net"]
        single = single + ["proj_mlp", "proj_out"]
    out = []
    for layer in layers_to_train:
        if layer < num_transformer_blocks:
            prefix, comps = f"denoise_tower.denoiser.transformer_blocks.{layer}", double
        else:
            prefix, comps = f"denoise_tower.denoiser.single_transformer_blocks.{layer - num_transformer_blocks}", single
        out.extend(f"{prefix}.{c}" for c in comps)
    return out


def check_param_is_in_components(name, components):
    return any(c in name for c in components)   # substring match: "transformer_blocks.1" also matches "...blocks.12"


# ---- timesteps / sigmas ----------------------------------------------------------------------------------------
def apply_flux_schedule_shift(sigmas, latent_h, latent_w, base_image_seq_len=256, max_image_seq_len=4096,
                              base_shift=0.5, max_shift=1.15):
    """Resolution-dependent shift of the sampled sigmas (:960-986): mu from the packed sequence length h*w/4."""
    mu = helpers.calculate_shift((latent_h * latent_w) // 4, base_image_seq_len, max_image_seq_len, base_shift, max_shift)
    shift = math.exp(mu)
    return (sigmas * shift) / (1 + (shift - 1) * sigmas)


def sample_sigmas_continuous(bsz, latent_h, latent_w, generator=None, **sched):
    """``discrete_timestep: false`` (:988-993): sigma = shift(sigmoid(N(0,1))), timestep = 1000 * sigma."""
    sigmas = torch.sigmoid(1.0 * torch.randn((bsz,), generator=generator, dtype=torch.float32))
    sigmas = apply_flux_schedule_shift(sigmas, latent_h, latent_w, **sched)
    return sigmas, sigmas * 1000.0


def compute_density_for_timestep_sampling(weighting_scheme, batch_size, logit_mean=0.0, logit_std=1.0, mode_scale=1.29,
                                          generator=None):
    """u in [0, 1) per sample (diffusers training_utils; SD3 section 3.1).  PARITY UNPINNED."""
    if weighting_scheme == "logit_normal":
        u = torch.normal(mean=logit_mean, std=logit_std, size=(batch_size,), generator=generator)
        return torch.sigmoid(u)
    u = torch.rand(size=(batch_size,), generator=generator)
    if weighting_scheme == "mode":
        u = 1 - u - mode_scale * (torch.cos(math.pi * u / 2) ** 2 - 1 + u)
    return u


def get_sigmas(timesteps, schedule_timesteps, schedule_sigmas, n_dim=4, dtype=torch.float32):
    """``discrete_timestep: true`` lookup (:779-788): the scheduler's sigma at each sampled timestep, broadcastable."""
    idx = [(schedule_timesteps == t).nonzero().item() for t in timesteps]
    sigma = schedule_sigmas.to(dtype)[idx].flatten()
    while sigma.dim() < n_dim:
        sigma = sigma.unsqueeze(-1)
    return sigma


def compute_loss_weighting_for_sd3(weighting_scheme, sigmas):
    """diffusers training_utils (SD3 section 3.1).  PARITY UNPINNED."""
    if weighting_scheme == "sigma_sqrt":
        return (sigmas ** -2.0).float()
    if weighting_scheme == "cosmap":
        return 2 / (math.pi * (1 - 2 * sigmas + 2 * sigmas ** 2))
    return torch.ones_like(sigmas)


# ---- the step --------------------------------------------------------------------------------------------------
def flow_matching_loss(model_pred, model_input, noise, weighting, weight_mask=None):
    """:1106-1166: target = noise - x; mean over everything of w * (pred - target)^2 in fp32."""
    target = noise - model_input
    loss = (weighting.float() * (model_pred.float() - target.float()) ** 2).reshape(target.shape[0], -1)
    if weight_mask is not None:
        return loss.sum() / weight_mask.sum() / model_pred.shape[1]
    return loss.mean()


def denoiser_loss(sd, model_input, cond_latents, noise, sigmas, prompt_embeds, pooled, guidance_scale=1.0,
                  weighting_scheme="logit_normal", flux_config=None):
    """Forward half of the step for one batch of unpadded, equally sized samples.

    model_input [B,16,h,w]: VAE latents of the target, already (z - shift) * scale; cond_latents likewise or None;
    sigmas [B] fp32.  Mirrors :994-1104: noisy input, 2x2 packing, [target | condition] tokens with condition ids
    carrying 1 in the first slot, transformer(timestep = sigma), slice, unpack."""
    dtype = prompt_embeds.dtype
    B, _, h, w = model_input.shape
    sig = sigmas.view(B, 1, 1, 1).to(model_input.dtype)
    noisy = (1.0 - sig) * model_input + sig * noise
    tokens = helpers.pack_latents(noisy.to(dtype))
    ids = helpers.prepare_latent_image_ids(h // 2, w // 2, dtype)
    S_tgt = tokens.shape[1]
    if cond_latents is not None:
        tokens = torch.cat([tokens, helpers.pack_latents(cond_latents.to(dtype))], dim=1)
        ids = torch.cat([ids, helpers.prepare_latent_image_ids(cond_latents.shape[2] // 2, cond_latents.shape[3] // 2,
                                                               dtype, first=1.0)], dim=0)
    txt_ids = torch.zeros(prompt_embeds.shape[1], 3, dtype=dtype)
    guidance = torch.full([B], guidance_scale, dtype=torch.float32)
    timestep = (sigmas * 1000.0).to(dtype) / 1000          # `timesteps / 1000` handed to the transformer
    pred = mmdit.flux_forward(sd, tokens, prompt_embeds, pooled, timestep, ids, txt_ids, guidance, config=flux_config)
    pred = helpers.unpack_latents(pred[:, :S_tgt], h * 8, w * 8)
    weighting = compute_loss_weighting_for_sd3(weighting_scheme, sig)
    return flow_matching_loss(pred, model_input, noise, weighting)


def adamw_step(params, grads, state, lr=1e-6, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0):
    """Global-norm clipping (accelerator.clip_grad_norm_, :1171-1177) then torch.optim.AdamW's update on fp32 masters.

    params / grads / state: dicts keyed by parameter name; state[name] = dict(step, exp_avg, exp_avg_sq).
    Returns (new_params, new_state, total_norm)."""
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads.values()))
    coef = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
    new_p, new_s = {}, {}
    for k, p in params.items():
        g = grads[k].float() * coef
        st = state.get(k) or dict(step=0, exp_avg=torch.zeros_like(p, dtype=torch.float32),
                                  exp_avg_sq=torch.zeros_like(p, dtype=torch.float32))
        step = st["step"] + 1
        m = st["exp_avg"] * betas[0] + (1 - betas[0]) * g
        v = st["exp_avg_sq"] * betas[1] + (1 - betas[1]) * g * g
        w = p.float() * (1 - lr * weight_decay)
        denom = (v.sqrt() / math.sqrt(1 - betas[1] ** step)) + eps
        w = w - (lr / (1 - betas[0] ** step)) * m / denom
        new_p[k], new_s[k] = w, dict(step=step, exp_avg=m, exp_avg_sq=v)
    return new_p, new_s, total


def train_step(sd, trainable, batch, state, flux_config=None, **opt):
    """loss, gradients (autograd through the oracle MMDiT) and the AdamW update of the ``trainable`` keys of ``sd``."""
    work = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v.detach()) for k, v in sd.items()}
    loss = denoiser_loss(work, flux_config=flux_config, **batch)
    grads = dict(zip(trainable, torch.autograd.grad(loss, [work[k] for k in trainable])))
    new_p, new_s, norm = adamw_step({k: sd[k] for k in trainable}, grads, state, **opt)
    return dict(loss=loss.detach(), grads=grads, params=new_p, state=new_s, grad_norm=norm)
