"""Checkpoint IO: diffusers FLUX directory and UniWorld model directory layouts,
sharded safetensors + index, prefix stripping, shape validation.  CPU only (no HIP calls)."""
import json
import os

import pytest
import torch

from gpt_image_edit_amd import checkpoint, flux_spec

SMALL_T = dict(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
SMALL_V = dict(block_out_channels=(32, 32, 64, 64), layers_per_block=1)


def _tcfg():
    c = dict(flux_spec.FLUX_KONTEXT_CONFIG)
    c.update(SMALL_T)
    return c


def _vcfg():
    c = dict(flux_spec.FLUX_VAE_CONFIG)
    c.update(SMALL_V)
    return c


def _state(shapes, seed):
    return flux_spec.synthetic_state(shapes, seed=seed, dtype=torch.bfloat16)


def _same(a, b):
    assert list(a.keys()) == list(b.keys()) or set(a.keys()) == set(b.keys())
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k


def test_flux_directory_roundtrip_sharded(tmp_path):
    tstate = _state(flux_spec.flux_param_shapes(_tcfg()), 1)
    vstate = _state(flux_spec.vae_param_shapes(_vcfg()), 2)
    d = str(tmp_path / "flux")
    checkpoint.save_flux_directory(d, tstate, vstate, _tcfg(), _vcfg(), max_shard_bytes=200_000)
    files = os.listdir(os.path.join(d, "transformer"))
    assert sum(f.endswith(".safetensors") for f in files) > 1 and any(f.endswith(".index.json") for f in files)
    # the index covers every key exactly once
    with open(os.path.join(d, "transformer", "diffusion_pytorch_model.safetensors.index.json")) as f:
        assert set(json.load(f)["weight_map"]) == set(tstate)
    got, cfg = checkpoint.read_flux_transformer(d)
    assert cfg["num_layers"] == 2 and cfg["num_attention_heads"] == 2 and cfg["axes_dims_rope"] == (16, 56, 56)
    _same(tstate, got)
    gotv, vcfg = checkpoint.read_vae(d)
    assert vcfg["block_out_channels"] == (32, 32, 64, 64)
    _same(vstate, gotv)
    assert checkpoint.scheduler_config(d)["max_shift"] == 1.15


def test_uniworld_directory_prefixes_and_task_head(tmp_path):
    tstate = _state(flux_spec.flux_param_shapes(_tcfg()), 3)
    proj_shapes = {k[len("denoise_projector."):]: v for k, v in flux_spec.projector_param_shapes(48, 64).items()}
    pstate = _state(proj_shapes, 4)
    head = {"0.weight": torch.randn(10240, 3584), "0.bias": torch.randn(10240), "3.weight": torch.randn(2, 10240),
            "3.bias": torch.randn(2)}
    vlm = {"model.embed_tokens.weight": torch.randn(10, 4).bfloat16(), "visual.blocks.0.attn.qkv.weight": torch.randn(6, 4).bfloat16()}
    d = str(tmp_path / "uniworld")
    checkpoint.save_uniworld_directory(d, tstate, pstate, head, extra_state=vlm, max_shard_bytes=300_000)
    got, _ = checkpoint.read_flux_transformer(d, config=_tcfg())   # no transformer/ sub-directory -> prefixed keys
    _same(tstate, got)
    _same(pstate, checkpoint.read_projector(d, 48, 64))
    th = checkpoint.read_task_head(d)
    assert torch.equal(th["3.weight"], head["3.weight"]) and set(th) == set(head)
    # selective read: only what was asked for is materialised
    only = checkpoint.read_state_dict(d, prefix=checkpoint.DENOISER_PREFIX, keys={"proj_out.weight"})
    assert list(only) == ["proj_out.weight"]


def test_layout_errors_are_complete(tmp_path):
    shapes = flux_spec.flux_param_shapes(_tcfg())
    tstate = _state(shapes, 5)
    del tstate["proj_out.bias"]
    tstate["x_embedder.weight"] = tstate["x_embedder.weight"][:, :32].contiguous()
    tstate["bogus.weight"] = torch.zeros(1, dtype=torch.bfloat16)
    d = str(tmp_path / "bad")
    checkpoint.save_flux_directory(d, tstate, None, _tcfg())
    with pytest.raises(ValueError) as e:
        checkpoint.read_flux_transformer(d)
    msg = str(e.value)
    assert "proj_out.bias" in msg and "bogus.weight" in msg and "x_embedder.weight" in msg


def test_fp32_checkpoint_is_cast_to_bf16(tmp_path):
    """make_univa_qwen2p5vl_weight.py saves fp32 (:47-75); the HIP path holds bf16 parameters."""
    shapes = flux_spec.flux_param_shapes(_tcfg())
    s32 = flux_spec.synthetic_state(shapes, seed=6, dtype=torch.float32)
    d = str(tmp_path / "fp32")
    checkpoint.save_flux_directory(d, s32, None, _tcfg())
    got, _ = checkpoint.read_flux_transformer(d)
    assert all(v.dtype == torch.bfloat16 for v in got.values())
    assert torch.equal(got["proj_out.weight"], s32["proj_out.weight"].bfloat16())
