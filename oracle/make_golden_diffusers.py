"""Pin the oracle's THIRD-PARTY restatements to the real diffusers.  TEST INFRASTRUCTURE.  One command:

    pip install diffusers==0.32.2          # the reference's pin: /root/reference/requirements.txt:22
    python oracle/make_golden_diffusers.py # writes tests/golden/diffusers_{mmdit,vae,sched}.npz
    python -m pytest tests/test_oracle_golden.py -k diffusers

It cannot run in the build container or on the GPU box (diffusers is not installed and there is no network): until a
maintainer runs it, ``oracle/mmdit.py`` / ``oracle/vae.py`` / ``oracle/scheduler.py`` stay "parity unpinned" and
``tests/test_oracle_golden.py::test_oracle_matches_diffusers_fixture`` SKIPS with exactly that reason.  Once the three
files exist, the same test turns every claim of the HIP-vs-oracle suite into a claim about diffusers.

What is pinned, i.e. which tensors decide that the restated WIRING is the library's:

* ``diffusers_mmdit.npz`` -- ``FluxTransformer2DModel`` (the class the reference instantiates at
  ``univa/models/modeling_univa_denoise_tower.py:21,103-110`` and calls at ``univa/utils/flux_pipeline.py:1067-1077``) built
  from a config, loaded ``strict=True`` with ``flux_spec.synthetic_state`` (so the KEY NAMES and SHAPES of
  ``flux_spec.flux_param_shapes`` are pinned too), on two configs:
    "tiny"  2 + 2 blocks, 4 heads x 16, batch 2, ragged text / image lengths, distinct timestep / guidance per sample;
            output + the (encoder_hidden_states, hidden_states) pair after EVERY double block + the stream after EVERY
            single block (forward hooks) -> chunk orders, gate / residual order, cat([txt, img]) order, cat([attn, mlp])
            order, RoPE pairing, norm_out scale-then-shift;
    "width" 1 + 1 blocks at the real width (24 heads x 128, D = 3072, axes (16, 56, 56)), batch 1, fp32 AND bf16
            execution -> the per-head RMSNorm cast order and the bf16 rounding points the HIP kernels reproduce.
* ``diffusers_vae.npz`` -- ``AutoencoderKL`` at the FLUX topology with narrow channels: ``encode().latent_dist.parameters``
  (the (0, 1, 0, 1) stride-2 pad, mid-block attention, mean | logvar order), ``.mode()``, ``decode()``.
* ``diffusers_sched.npz`` -- ``FlowMatchEulerDiscreteScheduler`` with the pipeline's arguments
  (``flux_pipeline.py:991-1008``): timesteps and sigmas for several (N, mu), and ``step()`` on fp32 and bf16 tensors.

Only data is written (seeds, inputs, outputs); every input is regenerated from its seed by the consuming test.
"""
import inspect
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from gpt_image_edit_amd import flux_spec  # noqa: E402
from oracle.helpers import prepare_latent_image_ids  # noqa: E402

# ---- the cases (shared with tests/test_oracle_golden.py, which regenerates the inputs from these seeds) -----------------
MMDIT_CASES = {
    "tiny": dict(cfg=dict(num_layers=2, num_single_layers=2, attention_head_dim=16, num_attention_heads=4,
                          joint_attention_dim=32, pooled_projection_dim=24, in_channels=16, out_channels=16,
                          axes_dims_rope=(4, 6, 6)),
                 weight_seed=3, input_seed=5, batch=2, s_txt=5, grid=(2, 3), timestep=(0.75, 0.31), guidance=(3.5, 1.0),
                 dtypes=("float32",)),
    "width": dict(cfg=dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1),
                  weight_seed=11, input_seed=7, batch=1, s_txt=8, grid=(4, 4), timestep=(0.5,), guidance=(3.5,),
                  dtypes=("float32", "bfloat16")),
}
VAE_CASE = dict(cfg=dict(block_out_channels=(32, 32, 64, 64), latent_channels=4), weight_seed=4, input_seed=9,
                latent_hw=(4, 6), image_hw=(32, 48))
SCHED_CASES = [(28, 1.15), (28, 0.5), (4, 0.63), (50, 0.8958)]


def mmdit_inputs(case, dtype=torch.float32):
    """Seeded inputs of one MMDiT case: target ids (first = 0) followed by condition ids (first = 1), as prepare_latents does."""
    g = torch.Generator().manual_seed(case["input_seed"])
    B, (gh, gw) = case["batch"], case["grid"]
    cfg = case["cfg"]
    s_img = 2 * gh * gw
    hs = torch.randn(B, s_img, cfg["in_channels"], generator=g)
    enc = torch.randn(B, case["s_txt"], cfg["joint_attention_dim"], generator=g)
    pooled = torch.randn(B, cfg["pooled_projection_dim"], generator=g)
    img_ids = torch.cat([prepare_latent_image_ids(gh, gw), prepare_latent_image_ids(gh, gw, first=1.0)])
    txt_ids = torch.zeros(case["s_txt"], 3)
    return dict(hidden_states=hs.to(dtype), encoder_hidden_states=enc.to(dtype), pooled_projections=pooled.to(dtype),
                timestep=torch.tensor(case["timestep"]).to(dtype), guidance=torch.tensor(case["guidance"]).to(dtype),
                img_ids=img_ids.to(dtype), txt_ids=txt_ids.to(dtype))


def vae_inputs(case=VAE_CASE):
    g = torch.Generator().manual_seed(case["input_seed"])
    z = torch.randn(1, case["cfg"]["latent_channels"], *case["latent_hw"], generator=g)
    im = torch.rand(1, 3, *case["image_hw"], generator=g) * 2 - 1
    return z, im


def sched_step_inputs():
    g = torch.Generator().manual_seed(21)
    return torch.randn(2, 6, 8, generator=g), torch.randn(2, 6, 8, generator=g)     # (model_output, sample)


def _np(t):
    return t.detach().float().cpu().numpy()


def golden_mmdit(diffusers):
    out = {"diffusers_version": np.array(diffusers.__version__)}
    for name, case in MMDIT_CASES.items():
        cfg = case["cfg"]
        kw = dict(patch_size=1, in_channels=cfg["in_channels"], num_layers=cfg["num_layers"],
                  num_single_layers=cfg["num_single_layers"], attention_head_dim=cfg["attention_head_dim"],
                  num_attention_heads=cfg["num_attention_heads"], joint_attention_dim=cfg["joint_attention_dim"],
                  pooled_projection_dim=cfg["pooled_projection_dim"], guidance_embeds=True,
                  axes_dims_rope=tuple(cfg["axes_dims_rope"]))
        if "out_channels" in inspect.signature(diffusers.FluxTransformer2DModel.__init__).parameters:
            kw["out_channels"] = cfg.get("out_channels")
        sd = flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=case["weight_seed"])
        for dt in case["dtypes"]:
            dtype = getattr(torch, dt)
            model = diffusers.FluxTransformer2DModel(**kw).eval()
            model.load_state_dict(sd, strict=True)         # key names + shapes of flux_spec are the library's
            model = model.to(dtype)
            taps = {}

            def tap(key):
                def hook(_m, _i, o):
                    taps[key] = o
                return hook
            hooks = [blk.register_forward_hook(tap(f"double{i}")) for i, blk in enumerate(model.transformer_blocks)]
            hooks += [blk.register_forward_hook(tap(f"single{i}")) for i, blk in enumerate(model.single_transformer_blocks)]
            with torch.no_grad():
                y = model(**mmdit_inputs(case, dtype), return_dict=False)[0]
            for h in hooks:
                h.remove()
            out[f"{name}.{dt}.out"] = _np(y)
            s_txt = case["s_txt"]
            for key, o in taps.items():
                if key.startswith("double"):               # FluxTransformerBlock returns (encoder_hidden_states, hidden_states)
                    out[f"{name}.{dt}.{key}.c"], out[f"{name}.{dt}.{key}.h"] = _np(o[0]), _np(o[1])
                else:                                      # single block: the joint stream [txt | img] (0.32.x); later
                    o = torch.cat(o, dim=1) if isinstance(o, tuple) else o   # versions return the pair (txt, img)
                    assert o.shape[1] == s_txt + 2 * case["grid"][0] * case["grid"][1]
                    out[f"{name}.{dt}.{key}.s"] = _np(o)
    np.savez(os.path.join(OUT, "diffusers_mmdit.npz"), **out)


def golden_vae(diffusers):
    c = dict(flux_spec.FLUX_VAE_CONFIG)
    c.update(VAE_CASE["cfg"])
    vae = diffusers.AutoencoderKL(
        in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
        block_out_channels=tuple(c["block_out_channels"]), layers_per_block=c["layers_per_block"], act_fn="silu",
        latent_channels=c["latent_channels"], norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159,
        use_quant_conv=False, use_post_quant_conv=False, mid_block_add_attention=True).eval()
    sd = flux_spec.synthetic_state(flux_spec.vae_param_shapes(VAE_CASE["cfg"]), seed=VAE_CASE["weight_seed"])
    vae.load_state_dict(sd, strict=True)
    z, im = vae_inputs()
    with torch.no_grad():
        dist = vae.encode(im).latent_dist
        out = {"diffusers_version": np.array(diffusers.__version__), "moments": _np(dist.parameters), "mode": _np(dist.mode()),
               "decode": _np(vae.decode(z, return_dict=False)[0])}
    np.savez(os.path.join(OUT, "diffusers_vae.npz"), **out)


def golden_sched(diffusers):
    out = {"diffusers_version": np.array(diffusers.__version__)}
    for n, mu in SCHED_CASES:
        s = diffusers.FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True,
                                                      base_shift=0.5, max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096)
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu, device="cpu")
        out[f"timesteps.{n}.{mu}"], out[f"sigmas.{n}.{mu}"] = _np(s.timesteps), _np(s.sigmas)
        assert s.order == 1
        if (n, mu) == SCHED_CASES[0]:
            for dt in ("float32", "bfloat16"):
                v, x = (t.to(getattr(torch, dt)) for t in sched_step_inputs())
                s.set_begin_index(0)
                for i, t in enumerate(s.timesteps[:3]):            # three consecutive steps: the step index advances by itself
                    x = s.step(v, t, x, return_dict=False)[0]
                    out[f"step{i}.{dt}"] = _np(x)
                    assert x.dtype == v.dtype
                s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu, device="cpu")   # reset the step index
    np.savez(os.path.join(OUT, "diffusers_sched.npz"), **out)


def main():
    try:
        import diffusers
    except ImportError:
        raise SystemExit("diffusers is not installed: run this where `pip install diffusers==0.32.2` is possible "
                         "(the fixtures stay absent and the oracle stays 'parity unpinned')")
    if not diffusers.__version__.startswith("0.32"):
        print(f"WARNING: diffusers {diffusers.__version__}, the reference pins 0.32.2 (requirements.txt:22)")
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    golden_mmdit(diffusers)
    golden_vae(diffusers)
    golden_sched(diffusers)
    print("wrote", ", ".join(f"tests/golden/diffusers_{n}.npz" for n in ("mmdit", "vae", "sched")))


if __name__ == "__main__":
    main()
