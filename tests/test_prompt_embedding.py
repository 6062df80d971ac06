"""Prompt-embedding host plumbing (gpt_image_edit_amd/prompt_embedding.py, FluxKontextPipeline.encode_prompt) against
golden vectors produced by the reference's own ``encode_prompt`` (oracle/make_golden.py::g_prompt) on the tiny seeded
T5 / CLIP models of tests/tiny_text_encoders.py.  CPU only."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny_text_encoders import build  # noqa: E402

from gpt_image_edit_amd import prompt_embedding  # noqa: E402

PROMPTS = ["replace the sky with a sunset", "make it snow"]
TOL = dict(rtol=1e-5, atol=1e-6)   # same torch ops on the same seeded weights; only BLAS threading may differ


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "prompt.npz"))


def test_encode_prompt_matches_reference(golden):
    encoders, tokenizers = build()
    with torch.no_grad():
        pe, pp = prompt_embedding.encode_prompt(encoders, tokenizers, PROMPTS, 24, device="cpu", num_images_per_prompt=2)
    np.testing.assert_allclose(pe.numpy(), golden["both_prompt_embeds"], **TOL)
    np.testing.assert_allclose(pp.numpy(), golden["both_pooled"], **TOL)
    # duplication order: T5 rows prompt-major, CLIP rows batch-major (the reference's 3-argument repeat of a 2-D tensor)
    assert torch.equal(pe[0], pe[1]) and not torch.equal(pe[0], pe[2])
    assert torch.equal(pp[0], pp[2]) and not torch.equal(pp[0], pp[1])
    assert np.array_equal(prompt_embedding.tokenize_prompt(tokenizers[1], PROMPTS, 24).numpy(), golden["ids_t5_24"])


def test_t5_only_and_clip_only_variants(golden):
    # eval/imgedit/step1_gen_samples_T5_only.py:164-171: prompt = 256 T5 tokens only
    encoders, tokenizers = build()
    with torch.no_grad():
        pe, pp = prompt_embedding.encode_prompt([None, encoders[1]], [None, tokenizers[1]], PROMPTS[0], 256, device="cpu")
        assert pp is None and pe.shape == (1, 256, 48)
        np.testing.assert_allclose(pe.numpy(), golden["t5only_prompt_embeds"], **TOL)
        pe, pp = prompt_embedding.encode_prompt([encoders[0], None], [tokenizers[0], None], PROMPTS[1], 256, device="cpu")
        assert pe is None
        np.testing.assert_allclose(pp.numpy(), golden["cliponly_pooled"], **TOL)
        # an encoder without its tokenizer is skipped even when ids are supplied (reference behaviour, :117-141)
        ids = tokenizers[1](PROMPTS[0], max_length=16).input_ids
        assert prompt_embedding.encode_prompt([None, encoders[1]], [None, None], PROMPTS[0], 16, device="cpu",
                                              text_input_ids_list=[None, ids]) == (None, None)
        with pytest.raises(ValueError, match="text_input_ids must be provided"):
            prompt_embedding._run_encoder(prompt_embedding.T5_SEQUENCE, encoders[1], None, [PROMPTS[0]], 16, "cpu", 1, None)


def test_pipeline_encode_prompt_string_path():
    # FluxKontextPipeline.encode_prompt (flux_pipeline.py:361-438): CLIP pooled of `prompt`, T5 of `prompt_2 or prompt`,
    # prompt-major duplication for BOTH, zero text ids; embeddings given by the caller pass through
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    encoders, tokenizers = build()
    vae = types.SimpleNamespace(config=types.SimpleNamespace(block_out_channels=(1, 2, 3, 4), latent_channels=16))
    tr = types.SimpleNamespace(dtype=torch.bfloat16, device=torch.device("cpu"))
    pipe = FluxKontextPipeline(tr, vae, text_encoder=encoders[0], tokenizer=tokenizers[0],
                               text_encoder_2=encoders[1], tokenizer_2=tokenizers[1])
    with torch.no_grad():
        pe, pp, ids = pipe.encode_prompt(PROMPTS, None, device="cpu", num_images_per_prompt=2, max_sequence_length=24)
        ref_pe, _ = prompt_embedding.encode_prompt(encoders, tokenizers, PROMPTS, 24, device="cpu", num_images_per_prompt=2)
        ref_pp = encoders[0](tokenizers[0](PROMPTS, max_length=77).input_ids).pooler_output
    assert torch.equal(pe, ref_pe)
    assert torch.equal(pp, ref_pp.repeat_interleave(2, dim=0))
    assert ids.shape == (24, 3) and float(ids.abs().sum()) == 0
    e, p, _ = pipe.encode_prompt(None, None, device="cpu", prompt_embeds=pe, pooled_prompt_embeds=pp)
    assert e is pe and p is pp
    pipe2 = FluxKontextPipeline(tr, vae)
    with pytest.raises(ValueError, match="text_encoder"):
        pipe2.encode_prompt("x", None, device="cpu")
