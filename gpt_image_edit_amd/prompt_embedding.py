"""T5 / CLIP prompt embeddings for the FLUX-Kontext path (host plumbing, no HIP involved).

Counterpart of the reference's ``univa/utils/denoiser_prompt_embedding_flux.py`` (``tokenize_prompt`` :1-13,
``_encode_prompt_with_t5`` :16-56, ``_encode_prompt_with_clip`` :59-104, ``encode_prompt`` :107-144) -- what
``univa/eval/imgedit/step1_gen_samples_T5_only.py:164-171`` and the training script call to turn an instruction into
``prompt_embeds [B, L, 4096]`` (T5-XXL last hidden state) and ``pooled_prompt_embeds [B, 768]`` (CLIP pooler output).
The encoders themselves are reused as they come from ``transformers``; this module
only reproduces the calling convention: tokenizer arguments, which output is taken, dtype, per-prompt duplication, and
the quirk that an encoder without a tokenizer is skipped even if ``text_input_ids_list`` is given.

Argument order and defaults follow the reference so that its call sites read the same here.
"""
import torch

__all__ = ["tokenize_prompt", "encode_prompt"]


def _as_list(prompt):
    return [prompt] if isinstance(prompt, str) else prompt


def _tokenize(tokenizer, prompt, max_length):
    return tokenizer(prompt, padding="max_length", max_length=max_length, truncation=True, return_length=False,
                     return_overflowing_tokens=False, return_tensors="pt").input_ids


def tokenize_prompt(tokenizer, prompt, max_sequence_length):
    """ids [B, max_sequence_length]: padded to the maximum length and truncated there (:1-13)."""
    return _tokenize(tokenizer, prompt, max_sequence_length)


def _module_dtype(encoder):
    # (a DDP / FSDP wrapper keeps the model under .module, :44-47)
    return encoder.module.dtype if hasattr(encoder, "module") else encoder.dtype


def _encode_prompt_with_t5(text_encoder, tokenizer, max_sequence_length=512, prompt=None, num_images_per_prompt=1,
                           device=None, text_input_ids=None):
    prompt = _as_list(prompt)
    batch_size = len(prompt)
    if tokenizer is not None:
        text_input_ids = _tokenize(tokenizer, prompt, max_sequence_length)
    elif text_input_ids is None:
        raise ValueError("text_input_ids must be provided when the tokenizer is not specified")
    prompt_embeds = text_encoder(text_input_ids.to(device))[0]          # last hidden state
    prompt_embeds = prompt_embeds.to(dtype=_module_dtype(text_encoder), device=device)
    seq_len = prompt_embeds.shape[1]
    # each prompt's block of rows repeated num_images_per_prompt times, prompt-major
    prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1)
    return prompt_embeds.view(batch_size * num_images_per_prompt, seq_len, -1)


def _encode_prompt_with_clip(text_encoder, tokenizer, prompt, device=None, text_input_ids=None,
                             num_images_per_prompt=1):
    prompt = _as_list(prompt)
    batch_size = len(prompt)
    if tokenizer is not None:
        text_input_ids = _tokenize(tokenizer, prompt, 77)
    elif text_input_ids is None:
        raise ValueError("text_input_ids must be provided when the tokenizer is not specified")
    out = text_encoder(text_input_ids.to(device), output_hidden_states=False)
    pooled = out.pooler_output.to(dtype=_module_dtype(text_encoder), device=device)
    # as in the reference (:99-101): a 3-argument repeat of the 2-D pooled tensor tiles the whole batch, so for
    # B > 1 and num_images_per_prompt > 1 the rows come out batch-major (b0, b1, b0, b1) while the T5 rows above are
    # prompt-major (b0, b0, b1, b1).  Its callers use B = 1 or num_images_per_prompt = 1; kept for parity.
    pooled = pooled.repeat(1, num_images_per_prompt, 1)
    return pooled.view(batch_size * num_images_per_prompt, -1)


def encode_prompt(text_encoders, tokenizers, prompt, max_sequence_length, device=None, num_images_per_prompt=1,
                  text_input_ids_list=None):
    """(prompt_embeds, pooled_prompt_embeds) from ``text_encoders = [clip, t5]`` / ``tokenizers = [clip_tok, t5_tok]``.

    An entry is ``None`` when its encoder OR its tokenizer is ``None`` (:117-141), which is how the T5-only and the
    CLIP-only callers use it."""
    prompt = _as_list(prompt)
    device = device if device is not None else text_encoders[1].device
    pooled_prompt_embeds = prompt_embeds = None
    if text_encoders[0] is not None and tokenizers[0] is not None:
        pooled_prompt_embeds = _encode_prompt_with_clip(
            text_encoders[0], tokenizers[0], prompt, device=device, num_images_per_prompt=num_images_per_prompt,
            text_input_ids=text_input_ids_list[0] if text_input_ids_list else None)
    if text_encoders[1] is not None and tokenizers[1] is not None:
        prompt_embeds = _encode_prompt_with_t5(
            text_encoders[1], tokenizers[1], max_sequence_length=max_sequence_length, prompt=prompt,
            num_images_per_prompt=num_images_per_prompt, device=device,
            text_input_ids=text_input_ids_list[1] if text_input_ids_list else None)
    return prompt_embeds, pooled_prompt_embeds
