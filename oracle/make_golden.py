"""Generate the committed golden vectors under tests/golden/.  TEST INFRASTRUCTURE.

Run ONLY in the build container (needs /root/reference, which does not exist on the GPU box):

    python oracle/make_golden.py

G1-G3  the four pure helpers of the reference pipeline driver, executed from the reference file
       itself: the functions are lifted out of ``univa/utils/flux_pipeline.py`` with ``ast`` (the
       module cannot be imported -- it needs diffusers) and exec'd with only ``torch`` in scope.
G4     ``univa/utils/anyres_util.py`` imported normally from the reference tree.
G5     torch's own CPU definitions of the building-block ops the HIP kernels implement.
G7     ``prepare_condition_images`` and ``update_size`` of ``univa/serve/cli.py`` (:82-116), lifted with ``ast`` like
       G1-G3 (the module imports diffusers/flash-attn models) and executed on PNG files written here.
G8     ``encode_prompt`` of ``univa/utils/denoiser_prompt_embedding_flux.py`` (imported normally) on the tiny seeded
       T5 / CLIP models of ``tests/tiny_text_encoders.py``: both encoders, T5 only, CLIP only.
G9     the pure pieces of the training step, lifted out of ``train_denoiser.py`` with ``ast`` (the script imports
       diffusers / deepspeed): ``get_trainable_params`` + ``check_param_is_in_components`` (:70-122) applied to the full
       FLUX key list, the nested ``calculate_shift`` / ``apply_flux_schedule_shift`` (:960-986) and ``get_sigmas``
       (:779-788) with stub scheduler / accelerator objects.
G6     end-to-end outputs of THIS repo's oracle on tiny configs (self-pinned, labelled as such;
       guards the oracle against accidental edits -- it is not evidence about diffusers).

Only inputs and outputs (data) are written; no reference source text is stored.
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))


def lift_reference_helpers():
    src = open(os.path.join(REF, "univa/utils/flux_pipeline.py")).read()
    wanted = {"calculate_shift", "_prepare_latent_image_ids", "_pack_latents", "_unpack_latents"}
    ns = {"torch": torch}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            node.decorator_list = []
            exec(compile(ast.Module(body=[node], type_ignores=[]), "<reference>", "exec"), ns)
    assert wanted <= set(ns), sorted(wanted - set(ns))
    return ns


def g_helpers():
    ns = lift_reference_helpers()
    g = torch.Generator().manual_seed(1234)
    out = {}
    # G1 pack / unpack
    x = torch.randn(2, 16, 8, 12, generator=g)
    packed = ns["_pack_latents"](x, 2, 16, 8, 12)
    out["pack_in"] = x.numpy()
    out["pack_out"] = packed.numpy()
    out["unpack_out"] = ns["_unpack_latents"](packed, 64, 96, 8).numpy()
    y = torch.randn(1, 6, 64, generator=g)  # odd geometry: 2x3 patches
    out["unpack2_in"] = y.numpy()
    out["unpack2_out"] = ns["_unpack_latents"](y, 32, 48, 8).numpy()
    # G2 ids
    out["ids_4x6"] = ns["_prepare_latent_image_ids"](1, 4, 6, "cpu", torch.float32).numpy()
    out["ids_64x64_bf16"] = ns["_prepare_latent_image_ids"](3, 64, 64, "cpu", torch.bfloat16).float().numpy()
    # G3 shift
    seqs = np.array([256, 1024, 4096, 6000, 1, 3600], dtype=np.int64)
    out["shift_seq"] = seqs
    out["shift_mu"] = np.array([ns["calculate_shift"](int(s)) for s in seqs], dtype=np.float64)
    out["shift_mu_custom"] = np.array(
        [ns["calculate_shift"](int(s), 256, 4096, 0.5, 1.16) for s in seqs], dtype=np.float64)
    np.savez(os.path.join(OUT, "helpers.npz"), **out)


def g_cli():
    import tempfile

    from PIL import Image
    sys.path.insert(0, REF)
    from univa.utils.anyres_util import dynamic_resize  # the lifted functions' only non-stdlib dependency
    src = open(os.path.join(REF, "univa/serve/cli.py")).read()
    wanted = {"prepare_condition_images", "update_size"}
    ns = {"torch": torch, "np": np, "Image": Image, "dynamic_resize": dynamic_resize}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            exec(compile(ast.Module(body=[node], type_ignores=[]), "<reference>", "exec"), ns)
    assert wanted <= set(ns)
    rng = np.random.default_rng(77)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        imgs = rng.integers(0, 256, size=(2, 12, 20, 3), dtype=np.uint8)
        paths = []
        for i, a in enumerate(imgs):
            fn = os.path.join(d, f"c{i}.png")
            Image.fromarray(a).save(fn)
            paths.append(fn)
        out["cond_u8"] = imgs
        out["cond_f32"] = ns["prepare_condition_images"](paths, "cpu").numpy()
        # update_size(i1, i2, anyres, anchor_pixels): (h, w) for 0 / 1 / 2 images
        sizes = [(333, 500), (1024, 768)]       # (w, h) of the two files
        spaths = []
        for i, (w, h) in enumerate(sizes):
            fn = os.path.join(d, f"s{i}.png")
            Image.fromarray(np.zeros((h, w, 3), dtype=np.uint8)).save(fn)
            spaths.append(fn)
        rows = []
        for anchor in (512 * 512, 1024 * 1024):
            rows.append([0, anchor, *ns["update_size"](None, None, "any_11ratio", anchor)])
            rows.append([1, anchor, *ns["update_size"](spaths[0], None, "any_11ratio", anchor)])
            rows.append([2, anchor, *ns["update_size"](spaths[0], spaths[1], "any_11ratio", anchor)])
        out["update_size_wh"] = np.array(sizes, dtype=np.int64)
        out["update_size_rows"] = np.array(rows, dtype=np.int64)   # n_images, anchor, new_h, new_w
    np.savez(os.path.join(OUT, "cli.npz"), **out)


def g_prompt():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from tiny_text_encoders import build
    from univa.utils.denoiser_prompt_embedding_flux import encode_prompt, tokenize_prompt
    encoders, tokenizers = build()
    prompts = ["replace the sky with a sunset", "make it snow"]
    out = {}
    with torch.no_grad():
        pe, pp = encode_prompt(encoders, tokenizers, prompts, 24, device="cpu", num_images_per_prompt=2)
        out["both_prompt_embeds"], out["both_pooled"] = pe.numpy(), pp.numpy()
        pe, pp = encode_prompt([None, encoders[1]], [None, tokenizers[1]], prompts[0], 256, device="cpu")
        assert pp is None
        out["t5only_prompt_embeds"] = pe.numpy()
        pe, pp = encode_prompt([encoders[0], None], [tokenizers[0], None], prompts[1], 256, device="cpu")
        assert pe is None
        out["cliponly_pooled"] = pp.numpy()
        out["ids_t5_24"] = tokenize_prompt(tokenizers[1], prompts, 24).numpy()
    np.savez(os.path.join(OUT, "prompt.npz"), **out)


def g_train():
    import math
    import types
    from typing import List
    from gpt_image_edit_amd import flux_spec
    tree = ast.parse(open(os.path.join(REF, "train_denoiser.py")).read())
    want = {"get_trainable_params", "check_param_is_in_components", "calculate_shift", "apply_flux_schedule_shift", "get_sigmas"}
    sched_cfg = types.SimpleNamespace(base_image_seq_len=256, max_image_seq_len=4096, base_shift=0.5, max_shift=1.15,
                                      num_train_timesteps=1000)
    sched = types.SimpleNamespace(config=sched_cfg, timesteps=torch.arange(1000, 0, -1, dtype=torch.float32),
                                  sigmas=torch.linspace(1.0, 0.001, 1000) ** 1.5)
    ns = {"torch": torch, "math": math, "List": List, "noise_scheduler_copy": sched,
          "accelerator": types.SimpleNamespace(device="cpu")}
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in want and node.name not in found:
            found[node.name] = node
    assert set(found) == want, sorted(found)
    for name in ("calculate_shift", "get_trainable_params", "check_param_is_in_components", "apply_flux_schedule_shift", "get_sigmas"):
        exec(compile(ast.Module(body=[found[name]], type_ignores=[]), "<reference>", "exec"), ns)
    out = {}
    keys = sorted(flux_spec.flux_param_shapes(flux_spec.FLUX_KONTEXT_CONFIG))
    for tag, kw in (("img", dict(only_img_branch=True)), ("all", dict(only_img_branch=False)),
                    ("img_layers_0_1_12_19_30", dict(layers_to_train=[0, 1, 12, 19, 30], only_img_branch=True))):
        comps = ns["get_trainable_params"](**kw)
        mask = [ns["check_param_is_in_components"]("denoise_tower.denoiser." + k, comps) for k in keys]
        out[f"trainable_{tag}"] = np.array(mask, dtype=np.bool_)
        out[f"n_components_{tag}"] = np.array(len(comps))
    out["n_keys"] = np.array(len(keys))
    g = torch.Generator().manual_seed(11)
    sig = torch.sigmoid(torch.randn(16, generator=g))
    rows = []
    for (h, w) in ((64, 64), (128, 128), (96, 160), (32, 48)):
        rows.append(ns["apply_flux_schedule_shift"](sig.clone(), torch.zeros(1, 16, h, w)).numpy())
    out["shift_in"], out["shift_hw"], out["shift_out"] = sig.numpy(), np.array([(64, 64), (128, 128), (96, 160), (32, 48)]), np.stack(rows)
    ts = sched.timesteps[torch.tensor([0, 5, 999, 500])]
    out["get_sigmas_t"] = ts.numpy()
    out["get_sigmas_out"] = ns["get_sigmas"](ts, n_dim=4, dtype=torch.float32).numpy()
    out["sched_timesteps"], out["sched_sigmas"] = sched.timesteps.numpy(), sched.sigmas.numpy()
    # self-pin of the step oracle (guards oracle/train.py against accidental edits; not evidence about the reference)
    from oracle import train as otrain
    from gpt_image_edit_amd import training
    cfg, sd, batch = _tiny_train_case()
    names = training.trainable_names(sorted(sd))
    res = otrain.train_step(sd, names, batch, {}, flux_config=cfg, lr=1e-3)
    out["step_loss"] = res["loss"].numpy()
    out["step_grad_norm"] = res["grad_norm"].numpy()
    out["step_grad_to_q"] = res["grads"]["transformer_blocks.0.attn.to_q.weight"][:4, :8].numpy()
    out["step_new_norm_q"] = res["params"]["single_transformer_blocks.0.attn.norm_q.weight"].numpy()
    np.savez_compressed(os.path.join(OUT, "train.npz"), **out)


def _tiny_train_case():
    from gpt_image_edit_amd import flux_spec
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1, num_attention_heads=2,
               joint_attention_dim=64, pooled_projection_dim=32)
    sd = flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=3)
    g = torch.Generator().manual_seed(2)
    B, h, w = 2, 4, 6
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g), cond_latents=torch.randn(B, 16, h, w, generator=g),
                 noise=torch.randn(B, 16, h, w, generator=g), sigmas=torch.tensor([0.3, 0.8]),
                 prompt_embeds=torch.randn(B, 5, 64, generator=g), pooled=torch.randn(B, 32, generator=g))
    return cfg, sd, batch


def g_anyres():
    sys.path.insert(0, REF)
    from univa.utils import anyres_util as ref  # importable: needs only PIL + math

    rows = []
    sizes = [(512, 512), (1024, 1024), (768, 1024), (1024, 768), (480, 854), (1080, 1920), (333, 1000),
             (1000, 333), (448, 448), (1568, 672), (700, 500), (37, 41)]
    modes = ["any_17ratio", "any_11ratio", "any_9ratio", "any_7ratio", "any_5ratio", "any_1ratio"]
    for h, w in sizes:
        for mi, m in enumerate(modes):
            for anchor in (512 * 512, 1024 * 1024):
                rw, rh = ref.pick_ratio(h, w, m)
                nh, nw = ref.dynamic_resize(h, w, m, anchor_pixels=anchor)
                ch, cw = ref.compute_size(rw, rh, 32, anchor_pixels=anchor)
                mh, mw = ref.compute_size(rw, rh, 32, min_pixels=256 * 256, max_pixels=768 * 768)
                rows.append([h, w, mi, anchor, rw, rh, nh, nw, ch, cw, mh, mw])
    np.savez(os.path.join(OUT, "anyres.npz"), table=np.array(rows, dtype=np.int64),
             modes=np.array(modes))


def g_torch_ops():
    g = torch.Generator().manual_seed(99)
    out = {}
    x = torch.randn(3, 5, 64, generator=g) * 2
    out["x"] = x.numpy()
    out["gelu_tanh"] = F.gelu(x, approximate="tanh").numpy()
    out["silu"] = F.silu(x).numpy()
    out["layer_norm"] = F.layer_norm(x, (64,), None, None, 1e-6).numpy()
    img = torch.randn(2, 64, 6, 5, generator=g)
    gw, gb = torch.randn(64, generator=g), torch.randn(64, generator=g)
    out["gn_in"], out["gn_w"], out["gn_b"] = img.numpy(), gw.numpy(), gb.numpy()
    out["group_norm"] = F.group_norm(img, 32, gw, gb, 1e-6).numpy()
    q, k, v = (torch.randn(2, 3, 37, 16, generator=g) for _ in range(3))
    out["q"], out["k"], out["v"] = q.numpy(), k.numpy(), v.numpy()
    out["sdpa"] = F.scaled_dot_product_attention(q, k, v).numpy()
    out["nearest2x"] = F.interpolate(img[:, :4], scale_factor=2.0, mode="nearest").numpy()
    cw, cb = torch.randn(8, 64, 3, 3, generator=g) * 0.1, torch.randn(8, generator=g)
    out["conv_w"], out["conv_b"] = cw.numpy(), cb.numpy()
    out["conv3x3"] = F.conv2d(img, cw, cb, padding=1).numpy()
    out["conv3x3_s2"] = F.conv2d(F.pad(img, (0, 1, 0, 1)), cw, cb, stride=2).numpy()
    np.savez(os.path.join(OUT, "torch_ops.npz"), **out)


def g_oracle_selfpin():
    from gpt_image_edit_amd import flux_spec
    from oracle import mmdit, scheduler, vae

    out = {}
    cfg = dict(num_layers=2, num_single_layers=2, attention_head_dim=16, num_attention_heads=4,
               joint_attention_dim=32, pooled_projection_dim=24, in_channels=16, out_channels=16,
               axes_dims_rope=(4, 6, 6))
    sd = flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=3)
    g = torch.Generator().manual_seed(5)
    hs = torch.randn(2, 12, 16, generator=g)
    enc = torch.randn(2, 5, 32, generator=g)
    pooled = torch.randn(2, 24, generator=g)
    t = torch.tensor([0.75, 0.31])
    gd = torch.tensor([3.5, 1.0])
    from oracle.helpers import prepare_latent_image_ids
    img_ids = torch.cat([prepare_latent_image_ids(2, 3), prepare_latent_image_ids(2, 3, first=1.0)])
    txt_ids = torch.zeros(5, 3)
    y = mmdit.flux_forward(sd, hs, enc, pooled, t, img_ids, txt_ids, gd, config=cfg)
    out["mmdit_tiny_out"] = y.numpy()
    cos, sin = mmdit.rope_tables(torch.cat([txt_ids, img_ids]), (4, 6, 6))
    out["rope_cos"], out["rope_sin"] = cos.numpy(), sin.numpy()
    out["sinusoid"] = mmdit.sinusoid_256(torch.tensor([0.0, 1.0, 750.0, 3504.0])).numpy()

    vcfg = dict(block_out_channels=(32, 32, 64, 64), latent_channels=4)
    vsd = flux_spec.synthetic_state(flux_spec.vae_param_shapes(vcfg), seed=4)
    z = torch.randn(1, 4, 4, 6, generator=g)
    out["vae_dec_tiny"] = vae.decode(vsd, z).numpy()
    im = torch.rand(1, 3, 32, 48, generator=g) * 2 - 1
    out["vae_enc_tiny"] = vae.encode_moments(vsd, im).numpy()

    ts, sg = scheduler.shifted_sigmas(28, 1.15)
    out["sched_timesteps_mu1.15"], out["sched_sigmas_mu1.15"] = ts.numpy(), sg.numpy()
    ts, sg = scheduler.shifted_sigmas(4, 0.6289062500000001)
    out["sched_timesteps_n4"], out["sched_sigmas_n4"] = ts.numpy(), sg.numpy()
    np.savez(os.path.join(OUT, "oracle_selfpin.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)  # deterministic reductions for the committed numbers
    g_helpers()
    g_anyres()
    g_cli()
    g_prompt()
    g_train()
    g_torch_ops()
    g_oracle_selfpin()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
