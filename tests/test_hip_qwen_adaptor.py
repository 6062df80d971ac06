"""a13 on the GPU: the prompt stage of one cli turn (``univa/serve/cli.py:199-234``,
``modeling_univa_qwen2p5vl.py:498-536``) with the HIP ``denoise_projector`` at its end.

tests/test_qwen_adaptor.py checks the adaptor's forward on the CPU with a torch stand-in projector; here the same
tiny random-init VLM runs on the GPU in bf16 (stock transformers model, reused as-is on PyTorch-ROCm) and the
projector is ``HipDenoiseProjector`` (two fk_gemm_bf16 calls).  Checker: the stock ``Qwen2_5_VLModel.forward`` of the
installed transformers on the same inputs + ``oracle.mmdit.denoise_projector`` on the host.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ASSIST = 1999


def test_encode_edit_prompt_with_hip_projector():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conftest import bf16_ulp_diff
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd import qwen_adaptor as qa
    from gpt_image_edit_amd.projector import HipDenoiseProjector
    from oracle import mmdit as omm
    torch.manual_seed(0)
    cfg = qa.qwen25vl_config("tiny")
    vlm = qa.build_vlm(cfg, device="cuda", dtype=BF)
    hid, out_dim = cfg.text_config.hidden_size, 128
    proj = HipDenoiseProjector(input_hidden_size=hid, output_hidden_size=out_dim, device="cuda", init="synthetic", seed=5)
    sd = {"denoise_projector." + k: v.detach().cpu() for k, v in proj.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(flux_spec.projector_param_shapes(hid, out_dim))
    model = qa.UnivaQwen2p5VL(vlm, proj)
    head = qa.TaskHead(hidden=hid, inner=96, assistant_token_id=ASSIST).cuda()
    with torch.no_grad():
        head[3].weight.zero_()
        head[3].bias.copy_(torch.tensor([0.0, 1.0]))       # route to generation
    inp = qa.synthetic_turn(cfg, "cuda", n_text=12, image_hw=(56, 84), batch=1, assistant_token_id=ASSIST)
    t5 = torch.randn(1, 5, out_dim, device="cuda", dtype=BF)
    r = qa.encode_edit_prompt(model, head, inp, t5)
    L = inp["input_ids"].shape[1]
    assert r["generate"] is True and r["prompt_embeds"].shape == (1, L + 5, out_dim) and r["prompt_embeds"].dtype == BF
    assert torch.equal(r["prompt_embeds"][:, L:], t5)                                 # [VLM tokens | T5 tokens] (cli.py:232)
    # checker: the stock model's forward (the adaptor re-assembles it by hand) ...
    with torch.no_grad():
        mm = model._mm_token_type_ids(inp["input_ids"])
        ref_hidden = vlm.model(**inp, mm_token_type_ids=mm).last_hidden_state
    assert ref_hidden.dtype == BF
    # ... then the oracle's projector on the host: bf16 with torch's rounding points, and fp32
    ref_bf = omm.denoise_projector(sd, ref_hidden.cpu())
    ref_32 = omm.denoise_projector({k: v.float() for k, v in sd.items()}, ref_hidden.float().cpu())
    got = r["prompt_embeds"][:, :L].cpu()
    ulps = bf16_ulp_diff(got, ref_bf)
    frac1 = (ulps <= 1).float().mean().item()
    e_hip = (got.float() - ref_32).abs().max().item()
    e_floor = (ref_bf.float() - ref_32).abs().max().item()
    print(f"[a13] prompt_embeds[:, :L]: {frac1 * 100:.2f}% within 1 bf16 ulp of the bf16 oracle (max {int(ulps.max())}); "
          f"max|hip - fp32| {e_hip:.3e}, bf16-oracle floor {e_floor:.3e}")
    assert frac1 >= 0.99 and int(ulps.max()) <= 4
    assert e_hip <= 2.0 * e_floor + 1e-6
    # text-only turn with a padded attention mask: positions must follow the mask (reference derives them from it)
    ids = inp["input_ids"][:, -8:].clone()
    am = torch.ones_like(ids)
    am[:, :3] = 0                                                                   # left padding
    got_t = model(input_ids=ids, attention_mask=am, output_type="denoise_embeds")
    pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1).unsqueeze(0).expand(3, -1, -1)   # modeling_univa_qwen2p5vl.py:300-303
    with torch.no_grad():
        ref_t = vlm.model(input_ids=ids, attention_mask=am, position_ids=pos).last_hidden_state
    want_t = omm.denoise_projector(sd, ref_t.cpu())
    u = bf16_ulp_diff(got_t[:, 3:].cpu(), want_t[:, 3:])
    assert (u <= 1).float().mean().item() >= 0.99 and int(u.max()) <= 4
    # understanding turn: no prompt embeds, no projector call
    with torch.no_grad():
        head[3].bias.copy_(torch.tensor([1.0, 0.0]))
    r0 = qa.encode_edit_prompt(model, head, inp, t5)
    assert r0["generate"] is False and r0["prompt_embeds"] is None
