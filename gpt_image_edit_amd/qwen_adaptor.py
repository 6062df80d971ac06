"""Qwen2.5-VL prompt encoding for the FLUX-Kontext path, on the INSTALLED ``transformers``.

The reference wraps its own copy of ``Qwen2_5_VLForConditionalGeneration`` (transformers 4.50) in
``UnivaQwen2p5VLForConditionalGeneration`` (``univa/models/qwen2p5vl/modeling_univa_qwen2p5vl.py:325-536``) and calls it
twice per edit (``univa/serve/cli.py:199-234``): once as a plain LM to route the request through a small task head, once
with ``output_type="denoise_embeds"`` to turn the VLM's last hidden states into the first part of FLUX's
``prompt_embeds``.  The north star reuses the VLM forward as-is on PyTorch-ROCm; what this module adds is the thin layer
around it, written against the stock model class of the transformers version that is installed here (5.x):

  * ``UnivaQwen2p5VL``        the wrapper's forward for ``output_type`` "lvlm" / "denoise_embeds" (incl. ``only_use_t5``,
                              ``vlm_residual_image_factor`` and the image-embedding shortcut), ending in the HIP
                              ``denoise_projector`` (``HipDenoiseProjector``: two fk_gemm_bf16 calls);
  * ``TaskHead``              ``Linear(3584,10240) -> SiLU -> Dropout(0.3) -> Linear(10240,2)`` of ``cli.py:42-49`` and the
                              routing rule of ``cli.py:203-207`` (last ``assistant`` token, generate iff logit[0] < logit[1]);
  * ``encode_edit_prompt``    the two forwards + ``cat([lvlm_embeds, t5_embeds])`` of ``cli.py:199-234``;
  * ``bench_prompt_encode``   T_prompt of SURVEY.md section 8(d) on random-init 7B weights.

The VLM itself is NOT re-implemented (out of the hot path by the north star's own words); only the projector runs on
the hand-written kernels.  transformers 5.x wants ``mm_token_type_ids`` to build the 3-D rope index, where 4.50 derived
it from ``input_ids``: the adaptor derives it the 4.50 way when the processor did not supply it.
"""
import time

import torch
from torch import nn

ASSISTANT_TOKEN_ID = 77091          # "assistant" in the Qwen2 vocabulary (univa/serve/cli.py:204)


def qwen25vl_config(size="7b", **overrides):
    """``Qwen2_5_VLConfig`` of Qwen2.5-VL-7B-Instruct (the backbone UniWorld-V1 / GPT-Image-Edit builds on), or a tiny
    one with the same structure for CPU tests."""
    from transformers import Qwen2_5_VLConfig
    if size == "7b":
        text = dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                    num_key_value_heads=4, vocab_size=152064, max_position_embeddings=128000, rms_norm_eps=1e-6,
                    rope_theta=1000000.0, tie_word_embeddings=False,
                    rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]})
        vision = dict(depth=32, hidden_size=1280, intermediate_size=3420, num_heads=16, out_hidden_size=3584,
                      patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112,
                      fullatt_block_indexes=[7, 15, 23, 31], tokens_per_second=2)
    elif size == "tiny":
        text = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                    num_key_value_heads=2, vocab_size=2048, max_position_embeddings=4096, rms_norm_eps=1e-6,
                    rope_theta=10000.0, tie_word_embeddings=False,
                    rope_scaling={"type": "mrope", "mrope_section": [2, 3, 3]})
        vision = dict(depth=2, hidden_size=32, intermediate_size=64, num_heads=2, out_hidden_size=64, patch_size=14,
                      spatial_merge_size=2, temporal_patch_size=2, window_size=56, fullatt_block_indexes=[1],
                      tokens_per_second=2)
        for k, v in dict(image_token_id=2040, video_token_id=2041, vision_start_token_id=2042, vision_end_token_id=2043).items():
            overrides.setdefault(k, v)
    else:
        raise ValueError(f"unknown size {size!r}")
    text.update(overrides.pop("text_config", {}))
    vision.update(overrides.pop("vision_config", {}))
    cfg = Qwen2_5_VLConfig(text_config=text, vision_config=vision, **overrides)
    return cfg


def build_vlm(config, device="cuda", dtype=torch.bfloat16):
    """Random-init ``Qwen2_5_VLForConditionalGeneration`` created directly on ``device`` in ``dtype``."""
    from transformers import Qwen2_5_VLForConditionalGeneration
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            model = Qwen2_5_VLForConditionalGeneration(config)
    finally:
        torch.set_default_dtype(old)
    return model.eval()


def load_vlm(model_path, device="cuda", dtype=torch.bfloat16):
    """The UniWorld checkpoint directory holds the Qwen2.5-VL weights under their stock names next to
    ``denoise_tower.*`` (scripts/make_univa_qwen2p5vl_weight.py:35-75); the stock loader ignores the latter."""
    from transformers import Qwen2_5_VLForConditionalGeneration
    return Qwen2_5_VLForConditionalGeneration.from_pretrained(model_path, dtype=dtype).to(device).eval()


class TaskHead(nn.Sequential):
    """Understanding-vs-generation router of the reference cli (``univa/serve/cli.py:42-49``; weights
    ``task_head_final.pt``, read by ``checkpoint.read_task_head``).  Runs in fp32 like the reference (``.float()``)."""

    def __init__(self, hidden=3584, inner=10240, assistant_token_id=ASSISTANT_TOKEN_ID):
        super().__init__(nn.Linear(hidden, inner), nn.SiLU(), nn.Dropout(0.3), nn.Linear(inner, 2))
        self.assistant_token_id = assistant_token_id
        self.eval()

    @torch.no_grad()
    def wants_generation(self, last_hidden_state, input_ids):
        """cli.py:202-207: hidden state of the LAST ``assistant`` token -> 2 logits -> generate iff [0] < [1]."""
        mask = input_ids == self.assistant_token_id
        if not bool(mask.any()):
            raise ValueError(f"the chat template's `assistant` token (id {self.assistant_token_id}) is missing from input_ids")
        vec = last_hidden_state[mask][-1:]
        logits = self(vec.float())[0]
        return bool(logits[0] < logits[1]), logits


def text_position_ids(attention_mask):
    """Text-only branch of the reference's ``get_rope_index`` (``modeling_univa_qwen2p5vl.py:300-303``): position =
    (number of real tokens before) for real tokens, 1 for padding, the same on the three rope axes -> [3, B, L]."""
    pos = attention_mask.long().cumsum(-1) - 1
    pos = pos.masked_fill(attention_mask == 0, 1)
    return pos.unsqueeze(0).expand(3, -1, -1)


def _find_true_blocks(mask_1d):
    """(start, length) of every run of True in a 1-D bool tensor (the image-token runs of one sample)."""
    m = mask_1d.to(torch.int8)
    d = torch.diff(torch.cat([m.new_zeros(1), m, m.new_zeros(1)]))
    starts = (d == 1).nonzero().flatten().tolist()
    ends = (d == -1).nonzero().flatten().tolist()
    return [(s, e - s) for s, e in zip(starts, ends)]


class UnivaQwen2p5VL(nn.Module):
    """``UnivaQwen2p5VLForConditionalGeneration.forward`` for inference (reference :325-536) over the stock VLM.

    ``vlm``: a ``Qwen2_5_VLForConditionalGeneration``; ``denoise_projector``: ``HipDenoiseProjector`` (any callable
    [B,L,3584] -> [B,L,4096]); ``shortcut_image_embeds`` / ``shortcut_image_embeds_scale``: the two config switches of
    ``configuration_univa_qwen2p5vl.py`` (off in the shipped checkpoints).
    """

    def __init__(self, vlm, denoise_projector, shortcut_image_embeds=False, shortcut_image_embeds_scale=0.5):
        super().__init__()
        self.vlm = vlm
        self.denoise_projector = denoise_projector
        self.shortcut_image_embeds = shortcut_image_embeds
        self.shortcut_image_embeds_scale = shortcut_image_embeds_scale

    @property
    def config(self):
        return self.vlm.config

    def _mm_token_type_ids(self, input_ids):
        c = self.vlm.config
        t = torch.zeros_like(input_ids, dtype=torch.int32)
        t[input_ids == c.image_token_id] = 1
        t[input_ids == c.video_token_id] = 2
        return t

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None,
                output_type="lvlm", only_use_t5=False, vlm_residual_image_factor=0.0, mm_token_type_ids=None,
                **kwargs):
        if output_type == "lvlm":
            if pixel_values is not None and mm_token_type_ids is None:
                mm_token_type_ids = self._mm_token_type_ids(input_ids)
            return self.vlm(input_ids=input_ids, attention_mask=attention_mask, pixel_values=pixel_values,
                            image_grid_thw=image_grid_thw, mm_token_type_ids=mm_token_type_ids, **kwargs)
        if output_type not in ("denoise_embeds", "denoise_model_pred"):
            raise ValueError(f"Unknown output_type: {output_type}.")
        if output_type == "denoise_model_pred":
            raise NotImplementedError("output_type='denoise_model_pred' is the training forward: "
                                      "gpt_image_edit_amd.training drives the denoiser directly")
        if only_use_t5:        # reference :380, :498-500: the VLM is skipped and there is nothing to project
            return None
        core = self.vlm.model   # Qwen2_5_VLModel: vision tower + language model (no lm_head)
        embeds = core.get_input_embeddings()(input_ids)
        image_embeds = image_mask = None
        if pixel_values is not None:
            feats = core.get_image_features(pixel_values, image_grid_thw).pooler_output
            image_embeds = torch.cat(list(feats), dim=0).to(embeds.device, embeds.dtype)
            n_tok = int((input_ids == self.config.image_token_id).sum())
            if n_tok != image_embeds.shape[0]:
                raise ValueError(f"Image features and image tokens do not match: tokens: {n_tok}, "
                                 f"features {image_embeds.shape[0]}")
            image_mask = (input_ids == self.config.image_token_id)[..., None].expand_as(embeds)
            embeds = embeds.masked_scatter(image_mask, image_embeds)
            if mm_token_type_ids is None:
                mm_token_type_ids = self._mm_token_type_ids(input_ids)
        # The reference derives the 3-D positions from (input_ids, attention_mask) in EVERY case (:451-466 ->
        # get_rope_index :139-318): with images from the id layout, without them from the mask (:300-303: cumsum - 1,
        # padding -> 1).  The stock 5.x model only does the former and falls back to arange for a padded text batch.
        position_ids = None
        if pixel_values is not None:
            position_ids, _ = core.get_rope_index(input_ids, mm_token_type_ids=mm_token_type_ids,
                                                  image_grid_thw=image_grid_thw, attention_mask=attention_mask)
        elif attention_mask is not None and attention_mask.dim() == 2:
            position_ids = text_position_ids(attention_mask)
        hidden = core.language_model(input_ids=None, position_ids=position_ids, attention_mask=attention_mask,
                                     inputs_embeds=embeds, use_cache=False).last_hidden_state
        if vlm_residual_image_factor > 0.0 and image_embeds is not None:      # :502-505
            old = hidden[image_mask[:, :, 0]]
            blended = old * (1 - vlm_residual_image_factor) + image_embeds * vlm_residual_image_factor
            hidden = hidden.masked_scatter(image_mask, blended.to(hidden.dtype))
        if self.shortcut_image_embeds and image_embeds is not None:            # :506-517
            s, used = self.shortcut_image_embeds_scale, 0
            hidden = hidden.clone()
            for b in range(input_ids.shape[0]):
                for start, length in _find_true_blocks(input_ids[b] == self.config.image_token_id):
                    hidden[b, start:start + length] = (s * image_embeds[used:used + length]
                                                       + (1 - s) * hidden[b, start:start + length])
                    used += length
        return self.denoise_projector(hidden)                                   # :519-523


@torch.no_grad()
def encode_edit_prompt(model, task_head, inputs, t5_prompt_embeds=None, joint_with_t5=True):
    """The prompt stage of one cli turn (``univa/serve/cli.py:199-234``).

    ``inputs``: the processor's output (``input_ids``, ``attention_mask``, optional ``pixel_values`` /
    ``image_grid_thw``).  Returns ``dict(generate, task_logits, prompt_embeds)``; ``prompt_embeds`` is None when the
    task head routes the turn to text understanding."""
    get = (lambda k: inputs.get(k)) if isinstance(inputs, dict) else (lambda k: getattr(inputs, k, None))
    kw = dict(input_ids=get("input_ids"), attention_mask=get("attention_mask"), pixel_values=get("pixel_values"),
              image_grid_thw=get("image_grid_thw"))
    out = model(**kw, output_type="lvlm", return_dict=True, output_hidden_states=True)
    generate, logits = task_head.wants_generation(out.hidden_states[-1], kw["input_ids"])
    if not generate:
        return dict(generate=False, task_logits=logits, prompt_embeds=None)
    lvlm = model(**kw, output_type="denoise_embeds")
    assert lvlm.shape[0] == 1
    embeds = lvlm
    if joint_with_t5 and t5_prompt_embeds is not None:
        embeds = torch.cat([lvlm, t5_prompt_embeds.to(lvlm.device, lvlm.dtype)], dim=1)
    return dict(generate=True, task_logits=logits, prompt_embeds=embeds)


def synthetic_turn(config, device, n_text=44, image_hw=(448, 448), batch=1, seed=0, assistant_token_id=ASSISTANT_TOKEN_ID):
    """Processor-shaped inputs for one user turn with one ``min_pixels = max_pixels = 448 * 448`` image (cli.py:166):
    ``pixel_values`` [(448/14)^2, 3*2*14*14] = [1024, 1176], grid (1, 32, 32), 256 image tokens after the 2x2 merge,
    plus ``n_text`` text tokens ending in the ``assistant`` token."""
    vc = config.vision_config
    gh, gw = image_hw[0] // vc.patch_size, image_hw[1] // vc.patch_size
    n_img = gh * gw // vc.spatial_merge_size ** 2
    g = torch.Generator().manual_seed(seed)
    vocab = config.text_config.vocab_size
    hi = min(vocab, config.image_token_id, config.video_token_id, assistant_token_id) - 1
    ids = torch.randint(10, hi, (batch, n_text + n_img), generator=g)
    ids[:, 4:4 + n_img] = config.image_token_id
    ids[:, -1] = assistant_token_id
    patch_dim = 3 * vc.temporal_patch_size * vc.patch_size ** 2
    pix = torch.randn(batch * gh * gw, patch_dim, generator=g)
    return dict(input_ids=ids.to(device), attention_mask=torch.ones_like(ids).to(device),
                pixel_values=pix.to(device), image_grid_thw=torch.tensor([[1, gh, gw]] * batch).to(device))


def bench_prompt_encode(device, batch=1, repeats=3):
    """T_prompt: Qwen2.5-VL-7B (random init, bf16, reused as-is on PyTorch-ROCm) run the way
    the cli runs it per edit -- LM forward + task head, then the ``denoise_embeds`` forward + HIP projector -- on one
    448^2 image + ~44 text tokens (L = 300), plus (round 5) the T5-XXL / CLIP-L ``encode_prompt`` call on random-init
    encoders of the FLUX.1 shapes (``prompt_embedding.bench_text_encoders``; reference
    ``denoiser_prompt_embedding_flux.py:107-144``).  T_prompt_s = the sum of the two medians."""
    from .projector import HipDenoiseProjector
    cfg = qwen25vl_config("7b")
    vlm = build_vlm(cfg, device)
    model = UnivaQwen2p5VL(vlm, HipDenoiseProjector(device=device, init="synthetic", seed=3))
    head = TaskHead().to(device)
    head[3].bias.data = torch.tensor([0.0, 1.0], device=device)      # route to generation
    inputs = synthetic_turn(cfg, device, batch=batch)
    t5 = torch.randn(batch, 256, 4096, device=device, dtype=torch.bfloat16)
    times = []
    for i in range(repeats + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = encode_edit_prompt(model, head, inputs, t5)
        torch.cuda.synchronize()
        if i:
            times.append(time.perf_counter() - t0)
    times.sort()
    L = int(inputs["input_ids"].shape[1])
    assert r["generate"] and r["prompt_embeds"].shape == (batch, L + 256, 4096)
    n_par = sum(p.numel() for p in vlm.parameters())
    t_vlm = times[len(times) // 2]
    del model, vlm
    torch.cuda.empty_cache()
    try:
        from .prompt_embedding import bench_text_encoders
        te = bench_text_encoders(device, batch=batch)
    except Exception as e:   # the Qwen figure must survive a missing encoder class
        te = {"T_t5_clip_s": 0.0, "error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return {"T_prompt_s": t_vlm + te["T_t5_clip_s"], "T_qwen_s": t_vlm, "T_t5_clip_s": te["T_t5_clip_s"], "runs_s": times,
            "vlm_tokens": L, "vlm_params": n_par, "text_encoders": te,
            "what": "random-init Qwen2.5-VL-7B (cli.py:199-234, 448^2 image + 44 tokens) + T5-XXL (256 tokens) / CLIP-L encode_prompt"}
