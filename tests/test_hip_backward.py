"""GPU parity of the backward-pass kernels (csrc/backward_kernels.hip, csrc/attention_bwd.hip) through the C ABI.

Checker: torch autograd on the host in fp32, on the same bf16-rounded inputs (the adjoint of each fused forward kernel
is what autograd derives from the oracle's restatement of that kernel's op).  The kernels compute in fp32 and round each
output tensor to bf16 once, so bf16 outputs are held to a couple of bf16 ulps of the fp32 result relative to the tensor's
scale; fp32 reductions to rtol 1e-3.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import ops as _ops
    return _ops


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def close_bf16(name, got, ref, tol=1.5e-2):
    """got (bf16, GPU) vs ref (fp32, host): max error <= tol * max|ref| (bf16 has 2^-8 relative steps)."""
    d = report(name, got, ref)
    scale = ref.abs().max().item()
    assert torch.isfinite(got.float()).all()
    assert d.max().item() <= tol * scale + 1e-6, f"{name}: {d.max().item():.3e} vs scale {scale:.3e}"
    assert d.mean().item() <= 0.2 * tol * scale + 1e-7


@pytest.mark.parametrize("D,R", [(3072, 37), (512, 300)])
def test_ln_modulate_bwd(ops, D, R):
    B, S_txt = 2, 11
    joint = randn(B, S_txt + R, D, seed=1, scale=2.0) + 0.5          # the stream is a view of the joint [B, S, D] buffer
    mod = randn(B, 6 * D, seed=2, scale=0.3)
    dn = randn(B, R, D, seed=3)
    dx_in = randn(B, S_txt + R, D, seed=4, scale=0.5)
    x = joint[:, S_txt:]
    xr = x.float().requires_grad_(True)
    sh = mod[:, :D].float().requires_grad_(True)
    sc = mod[:, D:2 * D].float().requires_grad_(True)
    n = F.layer_norm(xr, (D,), eps=1e-6) * (1 + sc[:, None]) + sh[:, None]
    n.backward(dn.float())
    jd, md, dxd = joint.cuda(), mod.cuda(), dx_in.cuda()
    dmod = torch.zeros(B, 6 * D, device="cuda", dtype=torch.float32)
    out = torch.zeros_like(jd)
    ops.ln_modulate_bwd(jd[:, S_txt:], dn.cuda(), md[:, D:2 * D], out[:, S_txt:], dmod[:, :2 * D], dx_in=dxd[:, S_txt:])
    torch.cuda.synchronize()
    close_bf16("ln_bwd dx (+dx_in)", out[:, S_txt:], xr.grad + dx_in[:, S_txt:].float())
    assert out[:, :S_txt].abs().max().item() == 0
    torch.testing.assert_close(dmod[:, :D].cpu(), sh.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(dmod[:, D:2 * D].cpu(), sc.grad, rtol=1e-3, atol=2e-3)
    assert dmod[:, 2 * D:].abs().max().item() == 0
    # without an incoming gradient, written (not accumulated) and deterministic
    o2, o3 = torch.full_like(jd, 7.0), torch.full_like(jd, -3.0)
    ops.ln_modulate_bwd(jd[:, S_txt:], dn.cuda(), md[:, D:2 * D], o2[:, S_txt:], dmod[:, :2 * D])
    ops.ln_modulate_bwd(jd[:, S_txt:], dn.cuda(), md[:, D:2 * D], o3[:, S_txt:], dmod[:, :2 * D])
    close_bf16("ln_bwd dx", o2[:, S_txt:], xr.grad)
    assert torch.equal(o2[:, S_txt:], o3[:, S_txt:])


def test_gate_res_and_gelu_and_colsum_bwd(ops):
    B, R, N = 2, 75, 3072
    dout, y = randn(B, R, N, seed=5), randn(B, R, N, seed=6, scale=2.0)
    mod = randn(B, 3 * N, seed=7, scale=0.5)
    gate = mod[:, N:2 * N]
    dy = torch.empty(B, R, N, device="cuda", dtype=BF)
    dg = torch.zeros(B, 2 * N, device="cuda", dtype=torch.float32)
    ops.gate_res_bwd(dout.cuda(), y.cuda(), mod.cuda()[:, N:2 * N], dy, dg[:, N:])
    close_bf16("gate_res_bwd dy", dy, dout.float() * gate.float()[:, None], tol=5e-3)
    torch.testing.assert_close(dg[:, N:].cpu(), (dout.float() * y.float()).sum(1), rtol=1e-3, atol=1e-3)
    assert dg[:, :N].abs().max().item() == 0
    # GELU(tanh)'
    h = randn(4, 100, 1024, seed=8, scale=2.5)
    df = randn(4, 100, 1024, seed=9)
    hr = h.float().requires_grad_(True)
    F.gelu(hr, approximate="tanh").backward(df.float())
    close_bf16("gelu_bwd", ops.gelu_bwd(h.cuda(), df.cuda()), hr.grad, tol=5e-3)
    # column sums over a strided [B, R, N] view
    wide = randn(B, R + 5, 2 * N, seed=10)
    got = ops.colsum(wide.cuda()[:, 5:, N:])
    torch.testing.assert_close(got.cpu(), wide[:, 5:, N:].float().sum((0, 1)), rtol=1e-3, atol=2e-3)


def test_qkv_post_bwd(ops):
    from oracle import mmdit
    from oracle.helpers import prepare_latent_image_ids
    B, H, S_txt, hh, ww = 2, 3, 21, 6, 9
    S = S_txt + hh * ww          # 75: not a multiple of 64
    D = H * 128
    qkv = randn(B, S, 3 * D, seed=14)
    w = [(1 + randn(128, seed=15 + i, scale=0.1).float()).to(BF) for i in range(4)]   # q_img k_img q_txt k_txt
    ids = torch.cat([torch.zeros(S_txt, 3), prepare_latent_image_ids(hh, ww)])
    cos, sin = mmdit.rope_tables(ids)
    dq, dk = randn(B, H, S, 128, seed=20), randn(B, H, S, 128, seed=21)
    x = qkv.float().requires_grad_(True)
    wf = [t.float().requires_grad_(True) for t in w]

    def post(xx, w_t, w_i):
        xh = mmdit.heads(xx, H)
        xh = torch.cat([mmdit.rms_norm(xh[:, :, :S_txt], w_t), mmdit.rms_norm(xh[:, :, S_txt:], w_i)], dim=2)
        return mmdit.apply_rope(xh, cos, sin)
    q = post(x[..., :D], wf[2], wf[0])
    k = post(x[..., D:2 * D], wf[3], wf[1])
    (q * dq.float()).sum().add((k * dk.float()).sum()).backward()
    dqkv = torch.full((B, S, 3 * D), 9.0, device="cuda", dtype=BF)
    dw = ops.qkv_post_bwd(dq.cuda(), dk.cuda(), qkv.cuda(), dqkv, w[0].cuda(), w[1].cuda(), w[2].cuda(), w[3].cuda(),
                          cos.cuda(), sin.cuda(), S_txt)
    close_bf16("qkv_post_bwd d(q|k raw)", dqkv[..., :2 * D], x.grad[..., :2 * D])
    assert (dqkv[..., 2 * D:] == 9.0).all()           # the v third belongs to the attention backward
    ref_dw = torch.stack([torch.stack([wf[0].grad, wf[2].grad]), torch.stack([wf[1].grad, wf[3].grad])])
    torch.testing.assert_close(dw.cpu(), ref_dw, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,H,S", [(1, 2, 64), (2, 3, 75), (1, 2, 300), (1, 2, 1000), (1, 3, 2560)])
def test_attention_backward(ops, B, H, S):
    D = H * 128
    q, k = randn(B, H, S, 128, seed=30), randn(B, H, S, 128, seed=31)
    qkv = randn(B, S, 3 * D, seed=32)
    dout = randn(B, S, D + 64, seed=33)[:, :, :D]                      # strided gradient rows
    qr, kr = q.float().requires_grad_(True), k.float().requires_grad_(True)
    vr = qkv[:, :, 2 * D:].float().reshape(B, S, H, 128).transpose(1, 2).detach().requires_grad_(True)
    o_ref = F.scaled_dot_product_attention(qr, kr, vr)                  # [B,H,S,128]
    o_ref.backward(dout.float().reshape(B, S, H, 128).transpose(1, 2))
    qd, kd, qkvd = q.cuda(), k.cuda(), qkv.cuda()
    doutd = torch.zeros(B, S, D + 64, device="cuda", dtype=BF)
    doutd[:, :, :D] = dout.cuda()
    o = torch.empty(B, S, D, device="cuda", dtype=BF)
    lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
    ops.attention_lse(qd, kd, qkvd[:, :, 2 * D:], o, lse)
    scores = (qr.detach() @ kr.detach().transpose(-1, -2)) / math.sqrt(128)
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(scores, -1) / math.log(2.0), rtol=1e-4, atol=2e-4)
    o2 = torch.empty_like(o)
    ops.attention(qd, kd, qkvd[:, :, 2 * D:], o2)
    assert torch.equal(o, o2)                                            # saving lse does not change the output
    dsum = ops.rowdot(doutd[:, :, :D], o, H)
    ref_dsum = (dout.float().reshape(B, S, H, 128).transpose(1, 2) * o.float().cpu().reshape(B, S, H, 128).transpose(1, 2)).sum(-1)
    torch.testing.assert_close(dsum.cpu(), ref_dsum, rtol=1e-3, atol=1e-3)
    dq, dk = torch.full_like(qd, 5.0), torch.full_like(kd, 5.0)
    dqkv = torch.full_like(qkvd, 5.0)
    ops.attention_bwd(qd, kd, qkvd[:, :, 2 * D:], doutd[:, :, :D], lse, dsum, dq, dk, dqkv[:, :, 2 * D:])
    torch.cuda.synchronize()
    close_bf16(f"attention_bwd dq B{B} H{H} S{S}", dq, qr.grad, tol=2e-2)
    close_bf16(f"attention_bwd dk B{B} H{H} S{S}", dk, kr.grad, tol=2e-2)
    close_bf16(f"attention_bwd dv B{B} H{H} S{S}", dqkv[:, :, 2 * D:], vr.grad.transpose(1, 2).reshape(B, S, D), tol=2e-2)
    assert (dqkv[:, :, :2 * D] == 5.0).all()
    # deterministic: no atomics anywhere
    dq2, dk2, dqkv2 = torch.empty_like(dq), torch.empty_like(dk), torch.full_like(qkvd, 5.0)
    ops.attention_bwd(qd, kd, qkvd[:, :, 2 * D:], doutd[:, :, :D], lse, dsum, dq2, dk2, dqkv2[:, :, 2 * D:])
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dqkv, dqkv2)
    # the three-pass form (fk_attention_bwd_set_mode(0)): dQ and dV bit for bit, dK to the last bf16 bit (it forms
    # p (dP - D) from the fp32 p, the paired pass from the bf16 p that also enters dV) and as close to autograd
    ops.attention_bwd_set_mode(0)
    try:
        dq3, dk3, dqkv3 = torch.empty_like(dq), torch.empty_like(dk), torch.full_like(qkvd, 5.0)
        ops.attention_bwd(qd, kd, qkvd[:, :, 2 * D:], doutd[:, :, :D], lse, dsum, dq3, dk3, dqkv3[:, :, 2 * D:])
        torch.cuda.synchronize()
    finally:
        ops.attention_bwd_set_mode(1)
    assert torch.equal(dq, dq3) and torch.equal(dqkv, dqkv3)
    close_bf16(f"attention_bwd dk (three passes) B{B} H{H} S{S}", dk3, kr.grad, tol=2e-2)
    scale_k = kr.grad.abs().max().item()
    assert (dk.float() - dk3.float()).abs().max().item() <= 2.0 ** -6 * scale_k


@pytest.mark.parametrize("B,H,S,split", [(1, 2, 2048, 5), (2, 3, 1500, 7), (1, 4, 4100, 3)])
def test_attention_backward_stream_k_dq_pass(ops, B, H, S, split):
    """Rounds 4 / 5: the dQ pass and the paired dK / dV pass as stream-K grids (whole rounds + dealt-out tail of tiles; the two
    parts of a cut item -- 256 query rows / 128 keys -- ADD their fp32 accumulators through the workspace).  Forced small grids
    (attention_set_split(n >= 2), the test hook) against the plain grid: every gradient deterministic, bit-identical for every
    item that is not cut and within bf16 rounding for the cut ones, and against fp32 autograd."""
    D = H * 128
    q, k = randn(B, H, S, 128, seed=40), randn(B, H, S, 128, seed=41)
    qkv = randn(B, S, 3 * D, seed=42)
    dout = randn(B, S, D, seed=43)
    qr, kr = q.float().requires_grad_(True), k.float().requires_grad_(True)
    vr = qkv[:, :, 2 * D:].float().reshape(B, S, H, 128).transpose(1, 2).detach().requires_grad_(True)
    F.scaled_dot_product_attention(qr, kr, vr).backward(dout.float().reshape(B, S, H, 128).transpose(1, 2))
    qd, kd, qkvd, doutd = q.cuda(), k.cuda(), qkv.cuda(), dout.cuda()
    o = torch.empty(B, S, D, device="cuda", dtype=BF)
    lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
    res = []
    try:
        ops.attention_set_split(0)
        ops.attention_lse(qd, kd, qkvd[:, :, 2 * D:], o, lse)
        dsum = ops.rowdot(doutd, o, H)
        for mode in (0, split, split):
            ops.attention_set_split(mode)
            dq, dk, dqkv = torch.empty_like(qd), torch.empty_like(kd), torch.zeros_like(qkvd)
            ops.attention_bwd(qd, kd, qkvd[:, :, 2 * D:], doutd, lse, dsum, dq, dk, dqkv[:, :, 2 * D:])
            res.append((dq, dk, dqkv[:, :, 2 * D:].reshape(B, S, H, 128).transpose(1, 2)))
    finally:
        ops.attention_set_split(1)
    torch.cuda.synchronize()
    (dq0, dk0, dv0), (dq1, dk1, dv1), (dq2, dk2, dv2) = res
    assert torch.equal(dq1, dq2) and torch.equal(dk1, dk2) and torch.equal(dv1, dv2)      # deterministic
    for name, g0, g1, rows in (("dQ", dq0, dq1, 256), ("dK", dk0, dk1, 128), ("dV", dv0, dv1, 128)):
        same = (g0 == g1).all(dim=-1)                    # [B, H, S]
        frac = same.float().mean().item()
        print(f"[parity] stream-K {name} B{B} H{H} S{S} on {split} workgroups: rows bit-identical to the plain grid {frac:.4f}, "
              f"max |d| {(g0.float() - g1.float()).abs().max().item():.3e}", flush=True)
        # at most split - 1 items are cut; a cut item's rows may all differ, everything else must not
        assert frac >= 1.0 - (split - 1) * rows / (B * H * S) - 1e-9, name
        assert (g0.float() - g1.float()).abs().max().item() <= 2.0 ** -7 * g0.float().abs().max().item(), name
    assert not torch.equal(dq0, dq1)      # the seams were exercised (a pass whose cuts all snap onto item boundaries has none)
    close_bf16(f"attention_bwd stream-K dk B{B} H{H} S{S}", dk1, kr.grad, tol=2e-2)
    close_bf16(f"attention_bwd stream-K dv B{B} H{H} S{S}", dv1, vr.grad, tol=2e-2)
    close_bf16(f"attention_bwd stream-K dq B{B} H{H} S{S}", dq1, qr.grad, tol=2e-2)
