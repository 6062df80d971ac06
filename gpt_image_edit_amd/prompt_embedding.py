"""T5 / CLIP prompt embeddings for the FLUX-Kontext path (host plumbing, no HIP involved).

Counterpart of the reference's ``univa/utils/denoiser_prompt_embedding_flux.py`` (``tokenize_prompt`` :1-13,
``encode_prompt`` :107-144 and its two per-encoder helpers :16-104) -- what
``univa/eval/imgedit/step1_gen_samples_T5_only.py:164-171`` and the training script call to turn an instruction into
``prompt_embeds [B, L, 4096]`` (T5-XXL last hidden state) and ``pooled_prompt_embeds [B, 768]`` (CLIP pooler output).
The encoder models are reused as they come from ``transformers``; this module only
reproduces the calling convention: tokenizer arguments, which output is taken, dtype, per-prompt duplication, and the
quirk that an encoder without a tokenizer is skipped even when token ids are supplied.

``encode_prompt`` / ``tokenize_prompt`` keep the reference's argument order and defaults so that its call sites read
the same here; everything else is organised as ONE encoder routine parameterised by what to take from the output.
"""
from dataclasses import dataclass
from typing import Callable, Optional

import torch

__all__ = ["tokenize_prompt", "encode_prompt", "EncoderRole", "CLIP_POOLED", "T5_SEQUENCE", "build_text_encoders",
           "SeededTokenizer", "bench_text_encoders"]

_TOKENIZER_KW = dict(padding="max_length", truncation=True, return_length=False, return_overflowing_tokens=False,
                     return_tensors="pt")


@dataclass(frozen=True)
class EncoderRole:
    """How one text encoder is used: the padded length, the part of its output that is kept, and how that is repeated
    for ``num_images_per_prompt`` images."""
    max_length: Optional[int]                       # None: taken from the caller (T5's max_sequence_length)
    forward_kw: dict
    take: Callable                                  # model output -> tensor
    pooled: bool


CLIP_POOLED = EncoderRole(77, dict(output_hidden_states=False), lambda out: out.pooler_output, True)
T5_SEQUENCE = EncoderRole(None, {}, lambda out: out[0], False)


def tokenize_prompt(tokenizer, prompt, max_sequence_length):
    """ids [B, max_sequence_length]: padded to the maximum length and truncated there."""
    return tokenizer(prompt, max_length=max_sequence_length, **_TOKENIZER_KW).input_ids


def _owner_dtype(encoder):
    # a DDP / FSDP wrapper keeps the model under .module
    return getattr(encoder, "module", encoder).dtype


def _run_encoder(role, encoder, tokenizer, prompts, max_length, device, copies, token_ids):
    if tokenizer is not None:
        token_ids = tokenize_prompt(tokenizer, prompts, role.max_length or max_length)
    elif token_ids is None:
        raise ValueError("text_input_ids must be provided when the tokenizer is not specified")
    feats = role.take(encoder(token_ids.to(device), **role.forward_kw)).to(dtype=_owner_dtype(encoder), device=device)
    n = len(prompts) * copies
    # Both branches repeat with THREE factors, as the reference does.  On the 3-D T5 tensor that duplicates every
    # prompt's rows in place (b0, b0, b1, b1); on the 2-D pooled tensor it tiles the whole batch (b0, b1, b0, b1), so
    # for B > 1 and copies > 1 the two outputs disagree in order.  Its callers use B = 1 or copies = 1; kept for parity.
    feats = feats.repeat(1, copies, 1)
    return feats.view(n, -1) if role.pooled else feats.view(n, feats.shape[1] // copies, -1)


def encode_prompt(text_encoders, tokenizers, prompt, max_sequence_length, device=None, num_images_per_prompt=1,
                  text_input_ids_list=None):
    """(prompt_embeds, pooled_prompt_embeds) from ``text_encoders = [clip, t5]`` / ``tokenizers = [clip_tok, t5_tok]``.

    An entry is ``None`` when its encoder OR its tokenizer is ``None``, which is how the T5-only and the CLIP-only
    callers use it."""
    prompts = [prompt] if isinstance(prompt, str) else prompt
    if device is None:
        device = text_encoders[1].device
    results = []
    for slot, role in ((1, T5_SEQUENCE), (0, CLIP_POOLED)):
        if text_encoders[slot] is None or tokenizers[slot] is None:
            results.append(None)
            continue
        ids = text_input_ids_list[slot] if text_input_ids_list else None
        results.append(_run_encoder(role, text_encoders[slot], tokenizers[slot], prompts, max_sequence_length, device,
                                    num_images_per_prompt, ids))
    return results[0], results[1]


# ---- the encoders themselves, random-init (no checkpoints offline): T_prompt's T5 / CLIP part ---------------------------------
T5_XXL_ENCODER = dict(vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
                      relative_attention_num_buckets=32, relative_attention_max_distance=128, dropout_rate=0.0,
                      layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu", is_encoder_decoder=False, use_cache=False,
                      tie_word_embeddings=False)          # google/t5-v1_1-xxl, the text_encoder_2 of FLUX.1
CLIP_L_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768, bos_token_id=49406,
                   eos_token_id=49407)                      # openai/clip-vit-large-patch14 text tower, the text_encoder of FLUX.1


def build_text_encoders(device="cuda", dtype=torch.bfloat16, t5=None, clip=None):
    """[clip, t5] -- ``CLIPTextModel`` and ``T5EncoderModel`` of the FLUX.1 shapes (or the given config overrides),
    random-init, created directly on ``device``: the models ``encode_prompt`` drives, reused as they come from transformers."""
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            enc_clip = CLIPTextModel(CLIPTextConfig(**dict(CLIP_L_TEXT, **(clip or {}))))
            enc_t5 = T5EncoderModel(T5Config(**dict(T5_XXL_ENCODER, **(t5 or {}))))
    finally:
        torch.set_default_dtype(old)
    return [enc_clip.eval(), enc_t5.eval()]


class SeededTokenizer:
    """Stands in for a tokenizer where none can be downloaded: seeded ids of the padded length, last position = eos (CLIP
    pools the hidden state at the eos id).  Same call signature as the tokenizers ``tokenize_prompt`` drives."""

    def __init__(self, vocab_size, eos_token_id, seed=0):
        self.vocab_size, self.eos, self.seed = vocab_size, eos_token_id, seed

    def __call__(self, prompt, max_length, **kw):
        n = 1 if isinstance(prompt, str) else len(prompt)
        g = torch.Generator().manual_seed(self.seed)
        hi = self.eos if self.eos > 8 else self.vocab_size        # ordinary ids stay below a large eos id (CLIP pools AT the eos)
        ids = torch.randint(3, hi - 1, (n, max_length), generator=g)
        ids[:, -1] = self.eos
        return type("Enc", (), {"input_ids": ids})()


def bench_text_encoders(device, batch=1, max_sequence_length=256, repeats=3, t5=None, clip=None):
    """Time ``encode_prompt`` (reference ``denoiser_prompt_embedding_flux.py:107-144``) on random-init T5-XXL + CLIP-L text
    encoders: tokenise (seeded ids), T5 last hidden state [B, L, 4096], CLIP pooled [B, 768].  Median of ``repeats``."""
    import time
    encoders = build_text_encoders(device, t5=t5, clip=clip)
    toks = [SeededTokenizer(encoders[0].config.vocab_size, encoders[0].config.eos_token_id, 1),
            SeededTokenizer(encoders[1].config.vocab_size, 1, 2)]
    prompts = ["replace the sky with a sunset"] * batch
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    times = []
    with torch.no_grad():
        for i in range(repeats + 1):
            sync()
            t0 = time.perf_counter()
            pe, pooled = encode_prompt(encoders, toks, prompts, max_sequence_length, device=device)
            sync()
            if i:
                times.append(time.perf_counter() - t0)
    times.sort()
    assert pe.shape == (batch, max_sequence_length, encoders[1].config.d_model) and pooled.shape == (batch, encoders[0].config.hidden_size)
    return {"T_t5_clip_s": times[len(times) // 2], "runs_s": times, "t5_tokens": max_sequence_length,
            "t5_params": sum(p.numel() for p in encoders[1].parameters()), "clip_params": sum(p.numel() for p in encoders[0].parameters())}
