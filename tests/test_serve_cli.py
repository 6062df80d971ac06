"""Front-end functions of gpt_image_edit_amd/serve/cli.py around the pipeline call -- CPU only, stub pipeline.

Reference: ``univa/serve/cli.py:118-267`` (chat loop, generation call :236-248),
``univa/eval/imgedit/step1_gen_samples_T5_only.py:140-183`` (T5-only edit).  ``update_size`` /
``prepare_condition_images`` are golden-pinned elsewhere (tests/test_oracle_golden.py, cli.npz); here: what reaches
the pipeline (argument names, sizes, condition pixels, prompt shapes), the chat loop's bookkeeping (conversation,
history of image paths, system-turn removal, output naming, routing) and the vision pre-processing
(``process_vision_info`` -> ``smart_resize``).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny_text_encoders import build  # noqa: E402

from gpt_image_edit_amd.serve import cli  # noqa: E402


class StubPipe:
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def __call__(self, **kw):
        from PIL import Image
        self.calls.append(kw)
        n = kw.get("num_images_per_prompt", 1)
        return types.SimpleNamespace(images=[Image.new("RGB", (kw["width"], kw["height"]), (i, 2, 3)) for i in range(n)])


class StubGenerator:
    def __init__(self, device=None):
        self.device, self.seed = device, None

    def manual_seed(self, s):
        self.seed = s
        return self


def _png(path, w, h, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path)
    return str(path)


def _args(**kw):
    d = dict(height=512, width=512, num_inference_steps=4, guidance_scale=3.5, no_joint_with_t5=False, ocr_enhancer=False,
             model_path="unused", flux_path="unused")
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_generate_image_call(tmp_path, monkeypatch):
    monkeypatch.setattr(cli.torch, "Generator", StubGenerator)
    pipe = StubPipe()
    p1, p2 = _png(tmp_path / "a.png", 48, 32, 1), _png(tmp_path / "b.png", 48, 32, 2)
    pe, pp = torch.randn(1, 7, 16), torch.randn(1, 8)
    img = cli.generate_image(pipe, pe, pp, [p1, p2], 64, 96, _args())
    kw = pipe.calls[0]
    assert set(kw) == {"image", "prompt_embeds", "pooled_prompt_embeds", "height", "width", "num_inference_steps",
                       "guidance_scale", "generator"}                      # cli.py:239-248
    assert (kw["height"], kw["width"], kw["num_inference_steps"], kw["guidance_scale"]) == (64, 96, 4, 3.5)
    assert kw["prompt_embeds"] is pe and kw["pooled_prompt_embeds"] is pp
    assert kw["generator"].device == "cuda" and kw["generator"].seed == 42  # cli.py:20-26, :247
    assert kw["image"].dtype == torch.uint8 and tuple(kw["image"].shape) == (2, 32, 48, 3)     # ALL history images, in order
    ref = cli.prepare_condition_images([p1, p2], "cpu")                     # the reference's float route (cli.py:99-116)
    assert torch.equal((kw["image"].permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5, ref)
    assert img.size == (96, 64)
    cli.generate_image(pipe, pe, pp, [], 64, 64, _args())
    assert pipe.calls[1]["image"] is None                                    # text-to-image turn
    cli.generate_image(pipe, pe, pp, [p1], 64, 64, _args(), fused_pixels=False)
    assert pipe.calls[2]["image"].dtype == torch.float32 and tuple(pipe.calls[2]["image"].shape) == (1, 3, 32, 48)


def test_run_t5_only(tmp_path):
    encoders, tokenizers = build()
    pipe = StubPipe()
    p1 = _png(tmp_path / "in.png", 300, 200, 3)
    imgs = cli.run_t5_only(pipe, encoders, tokenizers, "make it snow", p1, None, _args(height=256, width=256))
    kw = pipe.calls[0]
    h, w = cli.update_size(p1, None, "any_11ratio", anchor_pixels=256 * 256)   # step1_gen_samples_T5_only.py:147-150
    assert (kw["height"], kw["width"]) == (h, w) and h * w <= 256 * 256 * 1.2 and w > h
    assert kw["image"].dtype == torch.uint8 and tuple(kw["image"].shape) == (1, h, w, 3)   # resized to the edit size (:154-161)
    from PIL import Image
    want = np.asarray(Image.open(p1).convert("RGB").resize((w, h), Image.BILINEAR))
    assert np.array_equal(kw["image"][0].numpy(), want)
    assert tuple(kw["prompt_embeds"].shape) == (1, 256, 48) and tuple(kw["pooled_prompt_embeds"].shape) == (1, 32)   # :164-171
    assert "generator" not in kw and kw["num_images_per_prompt"] == 1 and kw["num_inference_steps"] == 4
    assert len(imgs) == 1 and imgs[0].size == (w, h)
    cli.run_t5_only(pipe, encoders, tokenizers, "a red cube", None, None, _args(height=128, width=128))
    assert pipe.calls[1]["image"] is None and (pipe.calls[1]["height"], pipe.calls[1]["width"]) == (128, 128)


def test_smart_resize_and_vision_inputs(tmp_path):
    budget = 448 * 448
    for h, w in ((448, 448), (1024, 1024), (100, 100), (600, 800), (3000, 500), (37, 911)):
        rh, rw = cli.smart_resize(h, w, 28, budget, budget)
        assert rh % 28 == 0 and rw % 28 == 0 and rh >= 28 and rw >= 28
        if round(h / 28) * round(w / 28) * 784 > budget:
            assert rh * rw <= budget                                                         # shrunk: floor in both directions
        else:
            assert rh * rw >= budget                                                         # grown: ceil in both directions
        assert abs(np.log((rh / rw) / (h / w))) < 0.35                                       # aspect kept up to rounding
    assert cli.smart_resize(1024, 1024, 28, budget, budget) == (448, 448)                   # -> grid (1, 32, 32), 256 tokens
    assert cli.smart_resize(600, 800, 28, budget, budget) == (364, 504)
    with pytest.raises(ValueError):
        cli.smart_resize(10, 4000)
    p = _png(tmp_path / "x.png", 800, 600)
    conv = [{"role": "user", "content": [{"type": "text", "text": "hi"},
                                         {"type": "image", "image": p, "min_pixels": budget, "max_pixels": budget}]},
            {"role": "assistant", "content": [{"type": "image", "image": p}]}]
    ims = cli.vision_inputs(conv)
    assert [im.size for im in ims] == [(504, 364), (812, 588)] and all(im.mode == "RGB" for im in ims)
    assert cli.vision_inputs([{"role": "user", "content": [{"type": "text", "text": "hi"}]}]) is None


def test_chat_loop_bookkeeping(tmp_path, monkeypatch):
    """Two turns: an edit request with one image (routed to generation), then a question (routed to text)."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(cli.torch, "Generator", StubGenerator)
    p1 = _png(tmp_path / "in.png", 640, 480, 5)
    seen = dict(templates=[], processor=[], encode=[], t5=[])

    class Batch(dict):                       # BatchFeature: a mapping with attribute access and .to()
        __getattr__ = dict.get

        def to(self, device):
            return self

    class Proc:
        def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=True):
            seen["templates"].append([dict(role=m["role"], n=len(m["content"])) for m in conversation])
            body = "".join(f"<|im_start|>{m['role']}\ncontent<|im_end|>\n" for m in conversation)
            return "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n" + body + "<|im_start|>assistant\n"

        def __call__(self, text, images, padding, return_tensors):
            seen["processor"].append((text[0], None if images is None else [im.size for im in images]))
            grid = None if images is None else torch.tensor([[1, im.size[1] // 14, im.size[0] // 14] for im in images])
            return Batch(input_ids=torch.tensor([[5, 6, 7]]), image_grid_thw=grid)

        def batch_decode(self, ids, **kw):
            return ["a cat"]

    vlm = types.SimpleNamespace(generate=lambda **kw: torch.tensor([[5, 6, 7, 8, 9]]))
    model = types.SimpleNamespace(vlm=vlm)
    monkeypatch.setattr(cli, "load_main_model_and_processor", lambda path, device: (model, object(), Proc()))
    routes = iter([True, False])

    def fake_encode_edit_prompt(m, head, inputs, t5, joint_with_t5=True):
        seen["encode"].append(joint_with_t5)
        gen = next(routes)
        return dict(generate=gen, task_logits=None, prompt_embeds=torch.zeros(1, 9, 4) if gen else None)
    import gpt_image_edit_amd.prompt_embedding as pe_mod
    import gpt_image_edit_amd.qwen_adaptor as qa_mod
    monkeypatch.setattr(qa_mod, "encode_edit_prompt", fake_encode_edit_prompt)

    def fake_encode_prompt(encoders, toks, text, n, device, k):
        seen["t5"].append((text, n, k))
        return torch.zeros(1, n, 4), torch.zeros(1, 8)
    monkeypatch.setattr(pe_mod, "encode_prompt", fake_encode_prompt)
    answers = iter(["make the sky red", p1, "what is in the image?", "", "", ""])
    monkeypatch.setattr("builtins.input", lambda prompt="": next(answers))
    pipe = StubPipe()
    cli.chat(_args(height=512, width=512), pipe, [None, None], [None, None], "cpu")
    # turn 1: one user message (text + image); the image was pre-sized by smart_resize to the 448^2 budget
    assert seen["templates"][0] == [dict(role="user", n=2)]
    text1, sizes1 = seen["processor"][0]
    assert not text1.startswith("<|im_start|>system") and text1.startswith("<|im_start|>user")      # cli.py:186
    rh, rw = cli.smart_resize(480, 640, 28, 448 * 448, 448 * 448)
    assert sizes1 == [(rw, rh)]
    assert seen["t5"][0] == ("make the sky red", 256, 1) and seen["encode"] == [True, True]
    kw = pipe.calls[0]
    h, w = cli.update_size(p1, None, "any_11ratio", anchor_pixels=512 * 512)
    assert (kw["height"], kw["width"]) == (h, w) and tuple(kw["image"].shape) == (1, 480, 640, 3)
    assert os.path.isfile(tmp_path / "generate_image_0.png")                                          # cli.py:250-254
    # turn 2: the conversation now holds user, assistant (the generated image), user; both images go to the VLM again
    assert seen["templates"][1] == [dict(role="user", n=2), dict(role="assistant", n=1), dict(role="user", n=1)]
    assert len(seen["processor"][1][1]) == 2
    assert len(pipe.calls) == 1                                                                       # routed to text


def test_chat_accepts_the_processors_second_resize(tmp_path, monkeypatch):
    """A 4:3 input: ``process_vision_info`` makes it 364 x 504 (area below the 448^2 budget), and the REAL Qwen2-VL image
    processor -- loaded with min_pixels = max_pixels = 448 * 448 like the reference (cli.py:30-35) -- then scales it up
    again to 392 x 532 (grid 28 x 38).  The reference goes on silently with that grid; so must this loop."""
    from transformers import Qwen2VLImageProcessor
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(cli.torch, "Generator", StubGenerator)
    p1 = _png(tmp_path / "in43.png", 800, 600, 9)
    image_proc = Qwen2VLImageProcessor(min_pixels=448 * 448, max_pixels=448 * 448)
    grids = []

    class Batch(dict):
        __getattr__ = dict.get

        def to(self, device):
            return self

    class Proc:
        def apply_chat_template(self, conversation, tokenize=False, add_generation_prompt=True):
            return "<|im_start|>system\nx<|im_end|>\n<|im_start|>user\ncontent<|im_end|>\n<|im_start|>assistant\n"

        def __call__(self, text, images, padding, return_tensors):
            assert [im.size for im in images] == [(504, 364)]                   # what vision_inputs hands over
            feats = image_proc(images=images, return_tensors="pt")
            grids.append(feats["image_grid_thw"].tolist())
            return Batch(input_ids=torch.tensor([[5, 6, 7]]), image_grid_thw=feats["image_grid_thw"])

    monkeypatch.setattr(cli, "load_main_model_and_processor",
                        lambda path, device: (types.SimpleNamespace(vlm=None), object(), Proc()))
    import gpt_image_edit_amd.prompt_embedding as pe_mod
    import gpt_image_edit_amd.qwen_adaptor as qa_mod
    monkeypatch.setattr(qa_mod, "encode_edit_prompt", lambda m, head, inputs, t5, joint_with_t5=True: dict(
        generate=True, task_logits=None, prompt_embeds=torch.zeros(1, 9, 4)))
    monkeypatch.setattr(pe_mod, "encode_prompt", lambda enc, toks, text, n, device, k: (torch.zeros(1, n, 4), torch.zeros(1, 8)))
    answers = iter(["make it night", p1, "", ""])
    monkeypatch.setattr("builtins.input", lambda prompt="": next(answers))
    pipe = StubPipe()
    cli.chat(_args(height=512, width=512), pipe, [None, None], [None, None], "cpu")
    assert grids == [[[1, 28, 38]]]                                             # 392 x 532: the processor resized again
    assert len(pipe.calls) == 1 and os.path.isfile(tmp_path / "generate_image_0.png")
