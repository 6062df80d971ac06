"""Checkpoint IO for the HIP path: real weights in, same key names out.

Layouts understood (all safetensors, single file or sharded with a ``*.safetensors.index.json``):

* a diffusers FLUX.1-Kontext directory as the reference hands to ``FluxKontextPipeline.from_pretrained(flux_path, ...)``
  (``univa/serve/cli.py:58-76``): ``transformer/diffusion_pytorch_model*.safetensors`` (+ ``config.json``),
  ``vae/diffusion_pytorch_model.safetensors`` (+ ``config.json``), ``scheduler/scheduler_config.json``;
* a UniWorld model directory as written by ``scripts/make_univa_qwen2p5vl_weight.py:35-75`` and loaded by
  ``univa/serve/cli.py:37-49``: ``model*.safetensors`` whose keys carry the prefixes
  ``denoise_tower.denoiser.`` (the FLUX transformer, diffusers key names) and
  ``denoise_tower.denoise_projector.{0,2}.`` (``modeling_univa_denoise_tower.py:34-44``), plus
  ``task_head_final.pt`` (``nn.Sequential(Linear(3584,10240), SiLU, Dropout, Linear(10240,2))`` state dict).

Nothing here touches the GPU: tensors are read on the host (memory-mapped by safetensors), checked against the
expected shapes (``flux_spec``) and copied into the model's parameters in bf16, after which the model re-packs its
fused weights lazily.  ``save_*`` write the same layouts back (round-trip tested), so a checkpoint produced by the
reference and one produced here are interchangeable.
"""
import json
import os
from collections import OrderedDict

import torch

from . import flux_spec

DENOISER_PREFIX = "denoise_tower.denoiser."
PROJECTOR_PREFIX = "denoise_tower.denoise_projector."
WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"


def _safetensors():
    try:
        import safetensors.torch as st
        from safetensors import safe_open
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("reading checkpoints needs the `safetensors` package") from e
    return st, safe_open


def _shard_files(directory, stem=None):
    """Shard list of a checkpoint directory: the index json when present, else every *.safetensors file."""
    names = sorted(os.listdir(directory))
    for n in names:
        if n.endswith(".safetensors.index.json") and (stem is None or n.startswith(stem)):
            with open(os.path.join(directory, n)) as f:
                index = json.load(f)
            files = sorted(set(index["weight_map"].values()))
            return [os.path.join(directory, x) for x in files], index["weight_map"]
    files = [n for n in names if n.endswith(".safetensors") and (stem is None or n.startswith(stem))]
    if not files:
        raise FileNotFoundError(f"no safetensors weights under {directory}")
    return [os.path.join(directory, x) for x in files], None


def read_state_dict(path, prefix="", dtype=None, keys=None):
    """Tensors of a safetensors file / checkpoint directory whose names start with ``prefix`` (prefix stripped).

    ``keys`` (optional set of stripped names) restricts what is materialised -- a 60 GB UniWorld checkpoint holds
    the 7B VLM next to the denoiser and only the latter is wanted here."""
    _, safe_open = _safetensors()
    files = [path] if os.path.isfile(path) else _shard_files(path)[0]
    out = OrderedDict()
    for fn in files:
        with safe_open(fn, framework="pt", device="cpu") as f:
            for k in f.keys():
                if not k.startswith(prefix):
                    continue
                name = k[len(prefix):]
                if keys is not None and name not in keys:
                    continue
                t = f.get_tensor(k)
                out[name] = t.to(dtype) if dtype is not None else t
    return out


def check_against(shapes, state, what):
    """Raise with the complete list of problems (missing / unexpected / wrong shape), not just the first."""
    missing = [k for k in shapes if k not in state]
    unexpected = [k for k in state if k not in shapes]
    bad = [f"{k}: checkpoint {tuple(state[k].shape)} vs expected {tuple(shapes[k])}"
           for k in shapes if k in state and tuple(state[k].shape) != tuple(shapes[k])]
    if missing or unexpected or bad:
        lines = [f"{what}: checkpoint does not match the expected parameter layout"]
        if missing:
            lines.append(f"  missing ({len(missing)}): " + ", ".join(missing[:8]) + (" ..." if len(missing) > 8 else ""))
        if unexpected:
            lines.append(f"  unexpected ({len(unexpected)}): " + ", ".join(unexpected[:8]) + (" ..." if len(unexpected) > 8 else ""))
        if bad:
            lines.append("  shape mismatches: " + "; ".join(bad[:8]) + (" ..." if len(bad) > 8 else ""))
        raise ValueError("\n".join(lines))


def _config_of(directory, defaults):
    cfg = dict(defaults)
    fn = os.path.join(directory, "config.json")
    if os.path.isfile(fn):
        with open(fn) as f:
            raw = json.load(f)
        for k in cfg:
            if k in raw and raw[k] is not None:
                cfg[k] = tuple(raw[k]) if isinstance(raw[k], list) else raw[k]
    return cfg


# ---- FLUX transformer -------------------------------------------------------------------------------------
def flux_transformer_config(flux_path):
    """``transformer/config.json`` of a diffusers FLUX directory merged over the Kontext defaults."""
    return _config_of(os.path.join(flux_path, "transformer"), flux_spec.FLUX_KONTEXT_CONFIG)


def read_flux_transformer(flux_path, config=None):
    """State dict (diffusers key names, bf16) of ``<flux_path>/transformer`` or of a UniWorld model directory."""
    sub = os.path.join(flux_path, "transformer")
    cfg = config or (flux_transformer_config(flux_path) if os.path.isdir(sub) else dict(flux_spec.FLUX_KONTEXT_CONFIG))
    shapes = flux_spec.flux_param_shapes(cfg)
    if os.path.isdir(sub):
        state = read_state_dict(sub, dtype=torch.bfloat16)
    else:  # UniWorld directory: the denoiser lives under a prefix next to the VLM
        state = read_state_dict(flux_path, prefix=DENOISER_PREFIX, dtype=torch.bfloat16)
    check_against(shapes, state, f"FLUX transformer at {flux_path}")
    return state, cfg


def load_flux_transformer(model, path):
    """Fill a ``HipFluxTransformer2DModel`` from a diffusers FLUX directory or a UniWorld model directory."""
    state, _ = read_flux_transformer(path, config=vars(model.config))
    model.load_state_dict(state, strict=True)
    return model


# ---- VAE ----------------------------------------------------------------------------------------------------
def read_vae(flux_path, config=None, dtype=torch.bfloat16):
    sub = os.path.join(flux_path, "vae")
    directory = sub if os.path.isdir(sub) else flux_path
    cfg = config or _config_of(directory, flux_spec.FLUX_VAE_CONFIG)
    state = read_state_dict(directory, dtype=dtype)
    check_against(flux_spec.vae_param_shapes(cfg), state, f"AutoencoderKL at {directory}")
    return state, cfg


def load_vae(model, path, fp32=False):
    """``fp32=True``: the reference's ``vae_fp32`` (train_denoiser.py:458 loads the VAE with torch_dtype float32) -- the
    checkpoint is read in fp32 and kept beside the bf16 parameters for ``encode(fp32=True)``
    (HipAutoencoderKL.load_fp32_state_dict)."""
    state, _ = read_vae(path, config=vars(model.config), dtype=torch.float32 if fp32 else torch.bfloat16)
    if fp32:
        model.load_fp32_state_dict(state, strict=True)
    else:
        model.load_state_dict(state, strict=True)
    return model


def scheduler_config(flux_path):
    cfg = dict(flux_spec.SCHEDULER_CONFIG)
    fn = os.path.join(flux_path, "scheduler", "scheduler_config.json")
    if os.path.isfile(fn):
        with open(fn) as f:
            raw = json.load(f)
        cfg.update({k: raw[k] for k in cfg if k in raw})
    return cfg


# ---- UniWorld extras ------------------------------------------------------------------------------------------
def read_projector(model_path, input_hidden=3584, output_hidden=4096):
    """``denoise_tower.denoise_projector.{0,2}.{weight,bias}`` -> keys ``0.weight, 0.bias, 2.weight, 2.bias``."""
    state = read_state_dict(model_path, prefix=PROJECTOR_PREFIX, dtype=torch.bfloat16)
    shapes = {k[len("denoise_projector."):]: v for k, v in flux_spec.projector_param_shapes(input_hidden, output_hidden).items()}
    check_against(shapes, state, f"denoise_projector at {model_path}")
    return state


def read_task_head(model_path):
    """``task_head_final.pt`` (cli.py:42-49): keys ``0.weight [10240,3584], 0.bias, 3.weight [2,10240], 3.bias``."""
    fn = os.path.join(model_path, "task_head_final.pt") if os.path.isdir(model_path) else model_path
    state = torch.load(fn, map_location="cpu", weights_only=True)
    expect = {"0.weight": (10240, 3584), "0.bias": (10240,), "3.weight": (2, 10240), "3.bias": (2,)}
    check_against(expect, state, f"task head at {fn}")
    return state


# ---- writing ------------------------------------------------------------------------------------------------
def save_sharded(state, directory, stem="diffusion_pytorch_model", max_shard_bytes=5 << 30, prefix="", metadata=None):
    """Write ``state`` as ``<stem>-0000i-of-0000n.safetensors`` + index json (one plain file if it fits)."""
    st, _ = _safetensors()
    os.makedirs(directory, exist_ok=True)
    shards, cur, cur_bytes = [], OrderedDict(), 0
    for k, v in state.items():
        nbytes = v.numel() * v.element_size()
        if cur and cur_bytes + nbytes > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = OrderedDict(), 0
        cur[prefix + k] = v.detach().cpu().contiguous()
        cur_bytes += nbytes
    shards.append(cur)
    meta = {"format": "pt", **(metadata or {})}
    if len(shards) == 1:
        st.save_file(shards[0], os.path.join(directory, stem + ".safetensors"), metadata=meta)
        return [stem + ".safetensors"]
    names, weight_map, total = [], {}, 0
    for i, sh in enumerate(shards):
        fn = f"{stem}-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        st.save_file(sh, os.path.join(directory, fn), metadata=meta)
        names.append(fn)
        for k, v in sh.items():
            weight_map[k] = fn
            total += v.numel() * v.element_size()
    with open(os.path.join(directory, stem + ".safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
    return names


def save_flux_directory(flux_path, transformer_state=None, vae_state=None, transformer_config=None, vae_config=None,
                        max_shard_bytes=5 << 30):
    """diffusers FLUX directory layout (transformer/, vae/, scheduler/) with config jsons."""
    def dump(directory, cfg):
        with open(os.path.join(directory, "config.json"), "w") as f:
            json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, f, indent=1)
    if transformer_state is not None:
        d = os.path.join(flux_path, "transformer")
        save_sharded(transformer_state, d, max_shard_bytes=max_shard_bytes)
        dump(d, {"_class_name": "FluxTransformer2DModel", **(transformer_config or flux_spec.FLUX_KONTEXT_CONFIG)})
    if vae_state is not None:
        d = os.path.join(flux_path, "vae")
        save_sharded(vae_state, d, max_shard_bytes=max_shard_bytes)
        dump(d, {"_class_name": "AutoencoderKL", **(vae_config or flux_spec.FLUX_VAE_CONFIG)})
    d = os.path.join(flux_path, "scheduler")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "scheduler_config.json"), "w") as f:
        json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", **flux_spec.SCHEDULER_CONFIG}, f, indent=1)


def save_uniworld_directory(model_path, denoiser_state, projector_state=None, task_head_state=None, extra_state=None,
                            max_shard_bytes=5 << 30):
    """UniWorld layout: ``model*.safetensors`` with ``denoise_tower.*`` prefixes (+ optional other keys, e.g. the
    VLM's) and ``task_head_final.pt``."""
    state = OrderedDict()
    for k, v in (extra_state or {}).items():
        state[k] = v
    for k, v in denoiser_state.items():
        state[DENOISER_PREFIX + k] = v
    for k, v in (projector_state or {}).items():
        state[PROJECTOR_PREFIX + k] = v
    save_sharded(state, model_path, stem="model", max_shard_bytes=max_shard_bytes)
    if task_head_state is not None:
        torch.save(dict(task_head_state), os.path.join(model_path, "task_head_final.pt"))
