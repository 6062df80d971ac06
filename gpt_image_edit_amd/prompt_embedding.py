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

__all__ = ["tokenize_prompt", "encode_prompt", "EncoderRole", "CLIP_POOLED", "T5_SEQUENCE"]

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
