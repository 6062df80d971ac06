"""Oracle: the whole FLUX-Kontext edit on the CPU (TEST INFRASTRUCTURE; parity unpinned, see __init__).

Restates ``FluxKontextPipeline.__call__`` for the embeds-driven path
(reference ``univa/utils/flux_pipeline.py:874-1130``): condition-image VAE encode + (z - shift) * scale,
2x2 packing, position ids (condition ids carry 1 in the first slot), dynamic-shift sigma schedule,
N x {cat(target, condition) -> transformer(t/1000) -> slice -> Euler step}, unpack, un-scale, VAE decode.
Runs in the dtype of the given weights (bf16 = the reference's rounding points, fp32 = exact).
"""
import torch

from . import helpers, mmdit, scheduler, vae


def kontext_edit(sd_flux, sd_vae, cond_image, prompt_embeds, pooled, noise, height, width,
                 num_inference_steps=28, guidance_scale=3.5, flux_config=None, decode=True,
                 negative_prompt_embeds=None, negative_pooled=None, true_cfg_scale=1.0):
    """cond_image [B,3,Hc,Wc] in [-1,1] (already at its final size); noise [B,16,height/8,width/8].

    Returns dict(latents=[B,S_tgt,64], image=[B,3,height,width] or None, per_step=[...]).
    """
    dtype = prompt_embeds.dtype
    B = prompt_embeds.shape[0]
    # flux_pipeline.py:600-613, 679-698
    z_c = vae.encode_for_pipeline(sd_vae, cond_image.to(dtype))
    image_latents = helpers.pack_latents(z_c)
    hc, wc = z_c.shape[2] // 2, z_c.shape[3] // 2
    image_ids = helpers.prepare_latent_image_ids(hc, wc, dtype, first=1.0)
    ht, wt = noise.shape[2] // 2, noise.shape[3] // 2
    latent_ids = helpers.prepare_latent_image_ids(ht, wt, dtype)
    latents = helpers.pack_latents(noise.to(dtype))
    ids = torch.cat([latent_ids, image_ids], dim=0)
    txt_ids = torch.zeros(prompt_embeds.shape[1], 3, dtype=dtype)
    # :991-1008
    S_tgt = latents.shape[1]
    mu = helpers.calculate_shift(S_tgt)
    timesteps, sigmas = scheduler.shifted_sigmas(num_inference_steps, mu)
    guidance = torch.full([B], guidance_scale, dtype=torch.float32)
    per_step = []
    for i, t in enumerate(timesteps):  # :1054-1120
        model_in = torch.cat([latents, image_latents], dim=1)
        timestep = t.expand(B).to(dtype)
        v = mmdit.flux_forward(sd_flux, model_in, prompt_embeds, pooled, timestep / 1000, ids, txt_ids, guidance,
                               config=flux_config)
        v = v[:, :S_tgt]
        if true_cfg_scale > 1 and negative_prompt_embeds is not None and negative_pooled is not None:  # :928, :1080-1095
            neg_txt_ids = torch.zeros(negative_prompt_embeds.shape[1], 3, dtype=dtype)
            vn = mmdit.flux_forward(sd_flux, model_in, negative_prompt_embeds, negative_pooled, timestep / 1000, ids,
                                    neg_txt_ids, guidance, config=flux_config)[:, :S_tgt]
            v = vn + vae._scalar_op(v - vn, "mul", true_cfg_scale)
        latents = scheduler.euler_step(v, sigmas[i], sigmas[i + 1], latents)
        per_step.append(latents)
    image = None
    if decode:  # :1127-1129
        z = helpers.unpack_latents(latents, height, width)
        image = vae.decode_for_pipeline(sd_vae, z)
    return dict(latents=latents, image=image, per_step=per_step)
