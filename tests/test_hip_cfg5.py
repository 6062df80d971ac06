"""cfg 5 (BASELINE.json configs[4]) at ITS OWN size: the stage-2 optimisation step at 1024 x 1024, batch 1 --
S = 512 text + 4096 target + 4096 condition tokens = 8704 (reference ``train_denoiser.py:1095-1181``).

The small-sequence tests (tests/test_hip_train_step.py, S = 192) pin every gradient to autograd; what they cannot reach
is the size-dependent machinery: the attention backward over 34 query blocks / 136 key tiles per head, the 256 x 256
GEMM tiles with K = tokens in the weight gradients (K = 8704), the stored-activation path (14 / 11 units of 53 MB per
block) against recomputation, buffers addressed beyond 2^31 bytes.  Checked here:
  * attention backward at S = 8704 against fp32 autograd (host), errors relative to each gradient's scale;
  * one optimisation step's forward + backward on a full-width 1 + 1 block model at S = 8704: stored activations and
    recomputation give the same bits, two calls give the same bits, every gradient is finite and non-zero,
    and the attention-free part of the step agrees with the S = 192 path's invariants (loss == the torch loss formula
    of the reference on the HIP prediction).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _close(name, got, ref, tol):
    d = report(name, got, ref)
    scale = ref.abs().max().item()
    assert torch.isfinite(got.float()).all()
    assert d.max().item() <= tol * scale + 1e-9, f"{name}: max {d.max().item():.3e} vs scale {scale:.3e}"
    assert d.mean().item() <= 0.2 * tol * scale + 1e-10, f"{name}: mean {d.mean().item():.3e} vs scale {scale:.3e}"


def test_attention_backward_at_1024sq_sequence():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import ops
    B, H, S = 1, 2, 8704
    D = H * 128
    q, k = _randn(B, H, S, 128, seed=130), _randn(B, H, S, 128, seed=131)
    qkv = _randn(B, S, 3 * D, seed=132)
    dout = _randn(B, S, D, seed=133)
    qr, kr = q.float().requires_grad_(True), k.float().requires_grad_(True)
    vr = qkv[:, :, 2 * D:].float().reshape(B, S, H, 128).transpose(1, 2).detach().requires_grad_(True)
    o_ref = F.scaled_dot_product_attention(qr, kr, vr)
    o_ref.backward(dout.float().reshape(B, S, H, 128).transpose(1, 2))
    qd, kd, qkvd, doutd = q.cuda(), k.cuda(), qkv.cuda(), dout.cuda()
    o = torch.empty(B, S, D, device="cuda", dtype=BF)
    lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
    ops.attention_lse(qd, kd, qkvd[:, :, 2 * D:], o, lse)
    with torch.no_grad():
        ref_lse = torch.stack([torch.logsumexp((qr[0, h] @ kr[0, h].t()) / math.sqrt(128), -1) for h in range(H)])[None] / math.log(2.0)
    torch.testing.assert_close(lse.cpu(), ref_lse, rtol=1e-4, atol=3e-4)
    dsum = ops.rowdot(doutd, o, H)
    dq, dk = torch.full_like(qd, 5.0), torch.full_like(kd, 5.0)
    dqkv = torch.full_like(qkvd, 5.0)
    ops.attention_bwd(qd, kd, qkvd[:, :, 2 * D:], doutd, lse, dsum, dq, dk, dqkv[:, :, 2 * D:])
    torch.cuda.synchronize()
    # bf16 P and dS operands in the MFMA products, one bf16 rounding of each gradient: a few 2^-8 of the tensor's scale
    _close(f"attention_bwd dq S{S}", dq, qr.grad, 2.5e-2)
    _close(f"attention_bwd dk S{S}", dk, kr.grad, 2.5e-2)
    _close(f"attention_bwd dv S{S}", dqkv[:, :, 2 * D:], vr.grad.transpose(1, 2).reshape(B, S, D), 2.5e-2)
    assert (dqkv[:, :, :2 * D] == 5.0).all()
    dq2, dk2, dqkv2 = torch.empty_like(dq), torch.empty_like(dk), torch.full_like(qkvd, 5.0)
    ops.attention_bwd(qd, kd, qkvd[:, :, 2 * D:], doutd, lse, dsum, dq2, dk2, dqkv2[:, :, 2 * D:])
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dqkv, dqkv2)


def test_train_step_forward_backward_at_1024sq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec, helpers, training
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    model = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=11)
    trainable = training.trainable_names(list(model.state_dict().keys()))
    B, h, w, S_txt = 1, 128, 128, 512                         # 1024^2 pixels: 128 x 128 latents, S = 512 + 4096 + 4096
    g = torch.Generator().manual_seed(21)
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g).cuda(), cond_latents=torch.randn(B, 16, h, w, generator=g).cuda(),
                 noise=torch.randn(B, 16, h, w, generator=g).cuda(), sigmas=torch.tensor([0.6]).cuda(),
                 prompt_embeds=torch.randn(B, S_txt, 4096, generator=g).to(BF).cuda(),
                 pooled=torch.randn(B, 768, generator=g).to(BF).cuda())
    runs = {}
    for store in (True, False):
        ts = DenoiserTrainStep(model, store_activations=store)
        loss, grads, d_enc = ts.forward_backward(**batch)
        assert ts.bw._buf["ckpt"].shape[2] == 8704
        loss2, grads2, d_enc2 = ts.forward_backward(**batch)
        torch.cuda.synchronize()
        assert torch.equal(loss, loss2) and torch.equal(d_enc, d_enc2), f"store={store}: not deterministic"
        for k in trainable:
            assert torch.equal(grads[k], grads2[k]), f"store={store}: {k} differs between two calls"
        runs[store] = (loss.clone(), {k: v.clone() for k, v in grads.items()}, d_enc.clone())
        del ts
    assert torch.equal(runs[True][0], runs[False][0])
    assert torch.equal(runs[True][2], runs[False][2])
    for k in trainable:
        a, b = runs[True][1][k], runs[False][1][k]
        assert torch.equal(a, b), f"{k}: stored activations and recomputation disagree"
        assert torch.isfinite(a.float()).all() and a.float().abs().max().item() > 0, k
    print(f"[cfg5 @ S=8704] loss {runs[True][0].item():.6f}; {len(trainable)} gradients bit-identical between the stored-"
          f"activation and the recomputation path and between two calls")
    # the loss the kernel reports == the reference's torch formula (train_denoiser.py:1095-1167) on the HIP prediction
    ts = DenoiserTrainStep(model, store_activations=True)
    inp, S_tgt = ts.prepare_inputs(**batch)
    with torch.no_grad():
        pred = model(**inp, return_dict=False)[0]
    mp = helpers._unpack_latents(pred[:, :S_tgt], h * 8, w * 8, 8)
    want = ((mp.float() - (batch["noise"] - batch["model_input"])) ** 2).reshape(B, -1).mean()
    # the inference forward (fused epilogues) and the training forward (un-fused, same rounding points) agree bit for bit
    assert abs(runs[True][0].item() - want.item()) <= 1e-5 * want.item()
