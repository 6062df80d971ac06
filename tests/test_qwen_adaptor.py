"""CPU tests of the Qwen2.5-VL adaptor (gpt_image_edit_amd/qwen_adaptor.py) on a tiny random-init VLM.

The adaptor's job is the reference wrapper's forward for ``output_type`` "lvlm" / "denoise_embeds"
(univa/models/qwen2p5vl/modeling_univa_qwen2p5vl.py:325-536) plus the cli's routing (univa/serve/cli.py:199-234) on
the installed transformers.  Checked here: its hand-assembled forward (embeddings + vision features scattered at the
image tokens + 3-D rope index derived from input_ids + language model) equals the stock ``Qwen2_5_VLModel.forward``,
the optional blends follow the reference's formulas, and the prompt assembly order is [VLM tokens | T5 tokens].
The projector is a torch stand-in here (the HIP projector has its own GPU parity test)."""
import pytest
import torch
from torch import nn

from gpt_image_edit_amd import qwen_adaptor as qa

ASSIST = 1999


@pytest.fixture(scope="module")
def tiny():
    torch.manual_seed(0)
    cfg = qa.qwen25vl_config("tiny")
    vlm = qa.build_vlm(cfg, device="cpu", dtype=torch.float32)
    proj = nn.Sequential(nn.Linear(64, 192), nn.SiLU(), nn.Linear(192, 128)).eval()
    inputs = qa.synthetic_turn(cfg, "cpu", n_text=12, image_hw=(56, 84), batch=1, assistant_token_id=ASSIST)
    return cfg, vlm, proj, inputs


def test_denoise_embeds_equals_stock_model_plus_projector(tiny):
    cfg, vlm, proj, inp = tiny
    model = qa.UnivaQwen2p5VL(vlm, proj)
    got = model(**inp, output_type="denoise_embeds")
    n_img = int((inp["input_ids"] == cfg.image_token_id).sum())
    assert n_img == (56 // 14) * (84 // 14) // 4 and got.shape == (1, 12 + n_img, 128)
    with torch.no_grad():
        mm = model._mm_token_type_ids(inp["input_ids"])
        ref_hidden = vlm.model(**inp, mm_token_type_ids=mm).last_hidden_state
        ref = proj(ref_hidden)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    # text-only turn: no vision tower, plain positions
    txt = dict(input_ids=inp["input_ids"][:, -8:].clone(), attention_mask=inp["attention_mask"][:, -8:])
    got_t = model(**txt, output_type="denoise_embeds")
    with torch.no_grad():
        ref_t = proj(vlm.model(**txt).last_hidden_state)
    torch.testing.assert_close(got_t, ref_t, rtol=1e-5, atol=1e-6)


def test_only_use_t5_and_bad_output_type(tiny):
    _, vlm, proj, inp = tiny
    model = qa.UnivaQwen2p5VL(vlm, proj)
    assert model(**inp, output_type="denoise_embeds", only_use_t5=True) is None
    with pytest.raises(ValueError, match="Unknown output_type"):
        model(**inp, output_type="nope")
    bad = dict(inp, input_ids=inp["input_ids"].clone())
    bad["input_ids"][0, 4] = 11          # one image token fewer than image features
    with pytest.raises(ValueError, match="do not match"):
        model(**bad, output_type="denoise_embeds")


def test_residual_and_shortcut_blends(tiny):
    cfg, vlm, proj, inp = tiny
    ident = nn.Identity()
    base = qa.UnivaQwen2p5VL(vlm, ident)(**inp, output_type="denoise_embeds")
    with torch.no_grad():
        feats = torch.cat(list(vlm.model.get_image_features(inp["pixel_values"], inp["image_grid_thw"]).pooler_output))
    is_img = inp["input_ids"][0] == cfg.image_token_id
    f = 0.25
    res = qa.UnivaQwen2p5VL(vlm, ident)(**inp, output_type="denoise_embeds", vlm_residual_image_factor=f)
    torch.testing.assert_close(res[0, is_img], base[0, is_img] * (1 - f) + feats * f, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res[0, ~is_img], base[0, ~is_img])
    sc = qa.UnivaQwen2p5VL(vlm, ident, shortcut_image_embeds=True, shortcut_image_embeds_scale=0.3)(
        **inp, output_type="denoise_embeds")
    torch.testing.assert_close(sc[0, is_img], 0.3 * feats + 0.7 * base[0, is_img], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sc[0, ~is_img], base[0, ~is_img])
    assert qa._find_true_blocks(torch.tensor([0, 1, 1, 0, 1, 0, 0, 1, 1, 1], dtype=torch.bool)) == [(1, 2), (4, 1), (7, 3)]


def test_task_head_routing_and_prompt_assembly(tiny):
    cfg, vlm, proj, inp = tiny
    model = qa.UnivaQwen2p5VL(vlm, proj)
    head = qa.TaskHead(hidden=64, inner=96, assistant_token_id=ASSIST)
    assert [type(m).__name__ for m in head] == ["Linear", "SiLU", "Dropout", "Linear"] and head[2].p == 0.3
    assert not head.training
    with torch.no_grad():
        head[3].weight.zero_()
        head[3].bias.copy_(torch.tensor([1.0, 0.0]))       # logit[0] > logit[1]: understanding
    r = qa.encode_edit_prompt(model, head, inp, torch.randn(1, 5, 128))
    assert r["generate"] is False and r["prompt_embeds"] is None
    with torch.no_grad():
        head[3].bias.copy_(torch.tensor([0.0, 1.0]))       # generation
    t5 = torch.randn(1, 5, 128)
    r = qa.encode_edit_prompt(model, head, inp, t5)
    L = inp["input_ids"].shape[1]
    assert r["generate"] is True and r["prompt_embeds"].shape == (1, L + 5, 128)
    torch.testing.assert_close(r["prompt_embeds"][:, L:], t5)                       # VLM tokens first, then T5
    torch.testing.assert_close(r["prompt_embeds"][:, :L], model(**inp, output_type="denoise_embeds"))
    r2 = qa.encode_edit_prompt(model, head, inp, t5, joint_with_t5=False)           # --no_joint_with_t5
    assert r2["prompt_embeds"].shape == (1, L, 128)
    # the routing vector is the hidden state of the LAST assistant token
    out = model(**inp, output_type="lvlm", return_dict=True, output_hidden_states=True)
    two = inp["input_ids"].clone()
    two[0, 1] = ASSIST
    gen, logits = head.wants_generation(out.hidden_states[-1], two)
    ref = head(out.hidden_states[-1][0, -1:].float())[0]
    torch.testing.assert_close(logits, ref)
    with pytest.raises(ValueError, match="assistant"):
        head.wants_generation(out.hidden_states[-1], torch.zeros_like(two))


def test_7b_config_matches_the_backbone():
    cfg = qa.qwen25vl_config("7b")
    t, v = cfg.text_config, cfg.vision_config
    assert (t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size) == \
        (3584, 28, 28, 4, 18944)
    assert (v.depth, v.hidden_size, v.out_hidden_size, v.patch_size, v.spatial_merge_size) == (32, 1280, 3584, 14, 2)
    inp = qa.synthetic_turn(cfg, "cpu")
    assert inp["pixel_values"].shape == (1024, 1176) and inp["input_ids"].shape == (1, 300)     # SURVEY a13 sizes
    assert int((inp["input_ids"] == cfg.image_token_id).sum()) == 256 and inp["input_ids"][0, -1] == 77091


def test_padded_text_batch_takes_positions_from_the_mask(tiny):
    """Reference ``get_rope_index`` text-only branch (modeling_univa_qwen2p5vl.py:300-303): positions follow the
    attention mask; the stock 5.x model would use arange for a padded batch."""
    cfg, vlm, proj, inp = tiny
    model = qa.UnivaQwen2p5VL(vlm, proj)
    ids = torch.cat([inp["input_ids"][:, -8:], inp["input_ids"][:, -8:]]).clone()
    am = torch.ones_like(ids)
    am[1, :3] = 0                                           # second sample left-padded by 3
    pos = qa.text_position_ids(am)
    assert pos.shape == (3, 2, 8) and pos[0, 0].tolist() == list(range(8)) and pos[0, 1].tolist() == [1, 1, 1, 0, 1, 2, 3, 4]
    got = model(input_ids=ids, attention_mask=am, output_type="denoise_embeds")
    with torch.no_grad():
        ref = proj(vlm.model(input_ids=ids, attention_mask=am, position_ids=pos).last_hidden_state)
        plain = proj(vlm.model(input_ids=ids, attention_mask=am).last_hidden_state)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    # the un-padded sample is unaffected; the padded one's real tokens see the same RELATIVE positions either way
    torch.testing.assert_close(got[0], plain[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got[1, 3:], plain[1, 3:], rtol=1e-4, atol=1e-5)
