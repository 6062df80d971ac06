"""GPU parity tests of the individual HIP kernels, called through the C ABI (libfk.so).

Oracle = the CPU restatement under oracle/ (torch fp32 / bf16 on the host), fed the same seeded
bf16-rounded inputs.  Tolerances:
  * fp32 debug outputs vs the fp32 oracle: rtol 1e-3 / atol 1e-4 (the tolerance BASELINE.json states);
  * bf16 outputs vs the oracle evaluated with the reference's bf16 rounding points: <= 1 bf16 ulp
    on (nearly) all elements -- two correct bf16 implementations with different accumulation order
    cannot agree tighter than that.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import bf16_ulp_diff, report

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import ops as _ops
    from gpt_image_edit_amd import libfk
    print("libfk:", libfk.load().fk_version().decode())
    return _ops


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def assert_bf16_close(name, got, ref, max_ulp=1, frac_exact=0.0, max_bad_frac=0.0):
    got, ref = got.cpu(), ref.cpu()
    report(name, got, ref)
    ulp = bf16_ulp_diff(got, ref)
    bad = (ulp > max_ulp).float().mean().item()
    exact = (ulp == 0).float().mean().item()
    print(f"[parity] {name}: exact={exact:.4f} frac(>{max_ulp}ulp)={bad:.2e} max_ulp={int(ulp.max())}", flush=True)
    assert not torch.isnan(got.float()).any()
    assert bad <= max_bad_frac, f"{name}: {bad:.3e} of elements differ by more than {max_ulp} bf16 ulp"
    assert exact >= frac_exact


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 192), (77, 64, 3072),
                                   (1, 3072, 256), (2560, 3072, 3072), (1024, 64, 3072), (4096, 3072, 64)])
def test_gemm_fp32_out(ops, M, N, K):
    a, w, bias = randn(M, K, seed=1), randn(N, K, seed=2, scale=0.05), randn(N, seed=3, scale=0.1)
    got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), out_fp32=True)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T + bias.float()
    report(f"gemm_f32 {M}x{N}x{K}", got, ref)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-4)


# the five M = 2560 launch classes of the 512^2 edit + the M = 8704 MLP-up shape, on the kernels that carry the FLOPs
HOT_SHAPES = [(2560, 9216, 3072, "fused QKV"), (2560, 12288, 3072, "MLP up"), (2560, 3072, 12288, "MLP down"),
              (2560, 3072, 15360, "single proj_out"), (2560, 3072, 3072, "out projection"), (8704, 12288, 3072, "MLP up @1024^2"),
              (8704, 3072, 15360, "single proj_out @1024^2")]


@pytest.mark.parametrize("M,N,K,what", HOT_SHAPES)
def test_hot_gemm_kernels_at_the_stated_tolerance(ops, M, N, K, what):
    """BASELINE.json's tolerance (rtol 1e-3 / atol 1e-4) on the kernels that carry 99 % of the GEMM time: the fp32-output
    build of gemm8 / gemm9 / gemm_mix / split-K gemm8 (fk_gemm_args.out_fp32 = 2: the SAME main loops, fp32(acc + bias)
    stored from the accumulator registers) against a.float() @ w.float().T in fp64-accumulated fp32 on the host --
    every launch form that applies to the shape, and the form the launch plan picks by itself."""
    from gpt_image_edit_amd import libfk
    lib = libfk.load()
    a, w, bias = randn(M, K, seed=71), randn(N, K, seed=72, scale=0.05), randn(N, seed=73, scale=0.1)
    ref = (a.double() @ w.double().T + bias.double()).float()
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    seen = {}
    try:
        for force in (0, 128, 256, 384, 512, 640):
            ops.gemm_set_variant(force)
            got = ops.gemm(ad, wd, bd, out_fp32=2)
            torch.cuda.synchronize()
            v = ops.gemm_last_variant()
            if v in seen and force != 0:
                continue                      # the forced form does not apply to this shape (fell back to one already checked)
            d = report(f"hot gemm f32 [{what}] {M}x{N}x{K} force={force} -> variant {v}", got, ref)
            torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-4)
            if v in seen:
                assert torch.equal(got, seen[v])
            seen.setdefault(v, got)
    finally:
        ops.gemm_set_variant(0)
    assert {128, 256} <= set(seen)
    if K >= 6144 and M == 2560:
        assert 512 in seen                   # the split-K pair form ran (M = 2560, N = 3072: 120 tiles)
    if K >= 6144 and M == 8704:
        assert 640 in seen                   # the stream-K ranges ran when forced (408 tiles = 1.59 rounds; the planner leaves them off: measured slower)
    # 128 / 256 / mixed accumulate over K in the same order: identical fp32 bits; the split-K pair adds two half sums
    assert torch.equal(seen[128], seen[256]) and (384 not in seen or torch.equal(seen[384], seen[256]))


def test_gemm_tile_order_does_not_change_the_bits(ops):
    """fk_gemm_set_group_m only changes WHICH workgroup (and XCD) computes a tile: every depth -- incl. the one that gives
    each XCD a column range over all rows -- must give the default order's bits, in every launch form, grouped or not."""
    from gpt_image_edit_amd import libfk
    lib = libfk.load()
    a, a2 = randn(2560, 3072, seed=81).cuda(), randn(512, 3072, seed=82).cuda()
    w, bias = randn(9216, 3072, seed=83, scale=0.05).cuda(), randn(9216, seed=84, scale=0.1).cuda()
    try:
        for force in (128, 256, 384):
            ops.gemm_set_variant(force)
            ops.gemm_set_group_m(0)
            ref = ops.gemm(a, w, bias).clone()
            ref_g = [t.clone() for t in ops.gemm_grouped([dict(a=a2, w=w, bias=bias), dict(a=a, w=w, bias=bias)])]
            for depth in (1, 3, 16, 4096):
                ops.gemm_set_group_m(depth)
                assert torch.equal(ops.gemm(a, w, bias), ref), (force, depth)
                got_g = ops.gemm_grouped([dict(a=a2, w=w, bias=bias), dict(a=a, w=w, bias=bias)])
                assert all(torch.equal(x, y) for x, y in zip(got_g, ref_g)), (force, depth)
    finally:
        ops.gemm_set_variant(0)
        ops.gemm_set_group_m(0)
    with pytest.raises(ValueError):
        ops.gemm_set_group_m(-1)                     # refused


def test_gemm_layout_is_transpose_detecting(ops):
    # A = identity-like selector with an asymmetric W: catches row/col swaps of the MFMA C layout
    M = N = 128
    K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 64.0
    got = ops.gemm(a.to(BF).cuda(), w.to(BF).cuda(), None, out_fp32=True).cpu()
    ref = a.to(BF).float() @ w.to(BF).float().T
    torch.testing.assert_close(got, ref, rtol=0, atol=0)


@pytest.mark.parametrize("epi", ["none", "gelu", "silu", "scale"])
def test_gemm_bf16_epilogues(ops, epi):
    M, N, K = 384, 512, 256
    a, w, bias = randn(M, K, seed=4), randn(N, K, seed=5, scale=0.06), randn(N, seed=6, scale=0.2)
    y = (a.float() @ w.float().T + bias.float())
    if epi == "none":
        got = ops.gemm(a.cuda(), w.cuda(), bias.cuda())
        ref = y.to(BF)
    elif epi == "gelu":
        got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), epilogue=ops.FK_EPI_GELU_TANH)
        ref = F.gelu(y.to(BF), approximate="tanh")
    elif epi == "silu":
        got = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), epilogue=ops.FK_EPI_SILU)
        ref = F.silu(y.to(BF))
    else:
        got = ops.gemm(a.cuda(), w.cuda(), None, epilogue=ops.FK_EPI_SCALE, alpha=0.125)
        ref = (0.125 * (a.float() @ w.float().T)).to(BF)
    assert_bf16_close(f"gemm_bf16[{epi}]", got, ref, max_ulp=1, max_bad_frac=1e-4)


def test_gemm_gate_residual_strided_views(ops):
    # text / image streams living in one joint [B, S, D] buffer, like the double block uses them
    B, S_txt, S_img, D, K = 2, 40, 216, 256, 128
    S = S_txt + S_img
    a_joint = randn(B, S, K, seed=7)
    w, bias = randn(D, K, seed=8, scale=0.08), randn(D, seed=9, scale=0.1)
    res = randn(B, S_img, D, seed=10)
    mod = randn(B, 6 * D, seed=11, scale=0.5)
    gate = mod[:, 2 * D:3 * D]
    a_dev, res_dev, mod_dev = a_joint.cuda(), res.cuda(), mod.cuda()
    out = res_dev  # in place, like h = h + gate * proj(o_img)
    ops.gemm(a_dev[:, S_txt:], w.cuda(), bias.cuda(), out=out, epilogue=ops.FK_EPI_GATE_RES, res=res_dev,
             gate=mod_dev[:, 2 * D:3 * D])
    y = (a_joint[:, S_txt:].float() @ w.float().T + bias.float()).to(BF)
    ref = res + gate[:, None] * y
    assert_bf16_close("gemm_gate_res", out, ref, max_ulp=1, max_bad_frac=2e-3)
    # plain residual + write into a column slice of a wider buffer (single block's [attn | mlp] buffer)
    wide = torch.zeros(B, S, 3 * D, dtype=BF, device="cuda")
    ops.gemm(a_dev, w.cuda(), bias.cuda(), out=wide[:, :, D:2 * D], epilogue=ops.FK_EPI_GELU_TANH)
    ref2 = F.gelu((a_joint.float() @ w.float().T + bias.float()).to(BF), approximate="tanh")
    assert_bf16_close("gemm_col_slice", wide[:, :, D:2 * D].contiguous(), ref2, max_ulp=1, max_bad_frac=1e-4)
    assert wide[:, :, :D].abs().max().item() == 0 and wide[:, :, 2 * D:].abs().max().item() == 0


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (2560, 3072, 3072), (2048 + 3, 9216, 128), (700, 200, 192)])
def test_gemm_large_tile_kernel(ops, M, N, K):
    # M >= 192 routes to the 256x128 LDS-DMA kernel; compare bf16 outputs with the exact fp32 product
    a, w, bias = randn(M, K, seed=41), randn(N, K, seed=42, scale=0.05), randn(N, seed=43, scale=0.1)
    got = ops.gemm(a.cuda(), w.cuda(), bias.cuda())
    ref = (a.float() @ w.float().T + bias.float()).to(BF)
    assert_bf16_close(f"gemm_large {M}x{N}x{K}", got, ref, max_ulp=1, max_bad_frac=1e-4)


def _last_variant():
    from gpt_image_edit_amd import ops
    return ops.gemm_last_variant()


@pytest.mark.parametrize("B,S,N,K,want", [(1, 2560, 12288, 3072, 256), (1, 2560, 9216, 3072, 384),
                                          (3, 100, 512, 128, 128), (2, 1200, 3072, 1024, 128), (1, 2560, 3072, 12288, 512),
                                          (1, 8704, 3072, 12288, 256)])
def test_gemm_tile_choice_and_batched_epilogue(ops, B, S, N, K, want):
    # the launcher picks the launch form with the shortest makespan over 256 CUs: 256 x 256 tiles where their higher rate
    # survives the round quantisation, one round of them + 256 x 128 tiles for the rest (384) where that beats both pure
    # grids, two half-K workgroups per 256 x 256 tile (512) for a long-K GEMM that fills at most half the chip;
    # both kernels address batched [B, S, :] operands per tile (one division per tile, a compare per row; batches
    # shorter than a tile take the reciprocal path) -- gated residual in place, like the blocks use it
    x = randn(B, S, K, seed=64)
    w, bias = randn(N, K, seed=65, scale=0.02), randn(N, seed=66, scale=0.1)
    res = randn(B, S, N, seed=67)
    mod = randn(B, 3 * N, seed=68, scale=0.5)
    xd, rd, md = x.cuda(), res.cuda(), mod.cuda()
    ops.gemm(xd, w.cuda(), bias.cuda(), out=rd, epilogue=ops.FK_EPI_GATE_RES, res=rd, gate=md[:, N:2 * N])
    assert _last_variant() == want
    y = (x.float() @ w.float().T + bias.float()).to(BF)
    ref = res + mod[:, None, N:2 * N] * y
    assert_bf16_close(f"gate_res B{B} S{S} N{N}", rd, ref, max_ulp=1, max_bad_frac=2e-3)
    if K <= 3072:   # (at K = 12288 the pre-activations reach -8: GELU outputs of 1e-14, where one ulp means nothing)
        got = ops.gemm(xd, w.cuda(), bias.cuda(), epilogue=ops.FK_EPI_GELU_TANH)
        assert_bf16_close(f"gelu B{B} S{S} N{N}", got, F.gelu(y, approximate="tanh"), max_ulp=1, max_bad_frac=1e-4)
    else:           # the split-K pair is deterministic: fp32 addition of the two partial tiles commutes
        first = rd.clone()
        for _ in range(3):
            rd2 = res.cuda()
            ops.gemm(xd, w.cuda(), bias.cuda(), out=rd2, epilogue=ops.FK_EPI_GATE_RES, res=rd2, gate=md[:, N:2 * N])
            assert torch.equal(rd2, first)


def test_gemm_grouped(ops):
    # text + image stream linears of a double block in one launch: different weights and row counts
    B, S_txt, S_img, D, K = 2, 100, 412, 384, 256
    joint_in = randn(B, S_txt + S_img, K, seed=44)
    w_t, w_i = randn(D, K, seed=45, scale=0.06), randn(D, K, seed=46, scale=0.06)
    b_t, b_i = randn(D, seed=47, scale=0.1), randn(D, seed=48, scale=0.1)
    res = randn(B, S_txt + S_img, D, seed=49)
    mod = randn(B, 2 * D, seed=50, scale=0.5)
    x, r, md = joint_in.cuda(), res.cuda(), mod.cuda()
    ops.gemm_grouped([dict(a=x[:, S_txt:], w=w_i.cuda(), bias=b_i.cuda(), out=r[:, S_txt:], res=r[:, S_txt:], gate=md[:, :D]),
                      dict(a=x[:, :S_txt], w=w_t.cuda(), bias=b_t.cuda(), out=r[:, :S_txt], res=r[:, :S_txt], gate=md[:, D:])],
                     epilogue=ops.FK_EPI_GATE_RES)
    y_i = (joint_in[:, S_txt:].float() @ w_i.float().T + b_i.float()).to(BF)
    y_t = (joint_in[:, :S_txt].float() @ w_t.float().T + b_t.float()).to(BF)
    ref = res.clone()
    ref[:, S_txt:] = res[:, S_txt:] + mod[:, None, :D] * y_i
    ref[:, :S_txt] = res[:, :S_txt] + mod[:, None, D:] * y_t
    assert_bf16_close("gemm_grouped gate_res", r, ref, max_ulp=1, max_bad_frac=2e-3)
    outs = ops.gemm_grouped([dict(a=x[:, S_txt:], w=w_i.cuda(), bias=b_i.cuda()), dict(a=x[:, :S_txt], w=w_t.cuda(), bias=b_t.cuda())],
                            epilogue=ops.FK_EPI_GELU_TANH)
    assert_bf16_close("gemm_grouped gelu img", outs[0], F.gelu(y_i, approximate="tanh"), max_ulp=1, max_bad_frac=1e-4)
    assert_bf16_close("gemm_grouped gelu txt", outs[1], F.gelu(y_t, approximate="tanh"), max_ulp=1, max_bad_frac=1e-4)


def test_gemm_fused_qkv_epilogue(ops):
    """FK_EPI_QKV (RMSNorm + RoPE + head-major layout inside the projection GEMM) must equal the unfused
    projection followed by fk_qkv_post_bf16 bit for bit, for a grouped text + image launch."""
    from oracle import mmdit
    from oracle.helpers import prepare_latent_image_ids
    B, H, S_txt, hh, ww, K = 2, 2, 70, 14, 20, 256
    S_img = hh * ww
    S, D = S_txt + S_img, H * 128
    x = randn(B, S, K, seed=60).cuda()
    w_i, w_t = randn(3 * D, K, seed=61, scale=0.06).cuda(), randn(3 * D, K, seed=62, scale=0.06).cuda()
    b_i, b_t = randn(3 * D, seed=63, scale=0.1).cuda(), randn(3 * D, seed=64, scale=0.1).cuda()
    nw = [(1 + randn(128, seed=65 + i, scale=0.1).float()).to(BF).cuda() for i in range(4)]  # q_i k_i q_t k_t
    ids = torch.cat([torch.zeros(S_txt, 3), prepare_latent_image_ids(hh, ww)])
    cos, sin = (t.cuda() for t in mmdit.rope_tables(ids))
    # unfused reference path (already parity-tested against the oracle)
    qkv_ref = torch.empty(B, S, 3 * D, dtype=BF, device="cuda")
    ops.gemm_grouped([dict(a=x[:, S_txt:], w=w_i, bias=b_i, out=qkv_ref[:, S_txt:]),
                      dict(a=x[:, :S_txt], w=w_t, bias=b_t, out=qkv_ref[:, :S_txt])])
    q_ref = torch.empty(B, H, S, 128, dtype=BF, device="cuda")
    k_ref = torch.empty_like(q_ref)
    ops.qkv_post(qkv_ref, q_ref, k_ref, nw[0], nw[1], nw[2], nw[3], cos, sin, S_txt)
    # fused
    qkv = torch.zeros(B, S, 3 * D, dtype=BF, device="cuda")
    q, k = torch.zeros_like(q_ref), torch.zeros_like(q_ref)
    ops.gemm_grouped([dict(a=x[:, S_txt:], w=w_i, bias=b_i, out=qkv[:, S_txt:],
                           qkv=dict(q_out=q, k_out=k, wq=nw[0], wk=nw[1], cos=cos, sin=sin, s_offset=S_txt)),
                      dict(a=x[:, :S_txt], w=w_t, bias=b_t, out=qkv[:, :S_txt],
                           qkv=dict(q_out=q, k_out=k, wq=nw[2], wk=nw[3], cos=cos, sin=sin, s_offset=0))],
                     epilogue=ops.FK_EPI_QKV)
    torch.cuda.synchronize()
    assert torch.equal(q, q_ref) and torch.equal(k, k_ref)
    assert torch.equal(qkv[:, :, 2 * D:], qkv_ref[:, :, 2 * D:])     # V third stored in place
    # single (non-grouped) call, one batch-row addressing, 256-wide N tiles forced through the env are covered
    q1, k1 = torch.zeros_like(q_ref), torch.zeros_like(q_ref)
    qkv1 = torch.zeros_like(qkv)
    ops.gemm(x, w_i, b_i, out=qkv1, epilogue=ops.FK_EPI_QKV,
             qkv=dict(q_out=q1, k_out=k1, wq=nw[0], wk=nw[1], cos=cos, sin=sin, s_offset=0))
    qkv2 = ops.gemm(x, w_i, b_i)
    q2, k2 = torch.empty_like(q_ref), torch.empty_like(q_ref)
    ops.qkv_post(qkv2, q2, k2, nw[0], nw[1], None, None, cos, sin, 0)
    assert torch.equal(q1, q2) and torch.equal(k1, k2) and torch.equal(qkv1[:, :, 2 * D:], qkv2[:, :, 2 * D:])


@pytest.mark.parametrize("mfma", [32, 16])
def test_gemm_k_major_operands(ops, mfma):
    """fk_gemm_args.layout 1 / 2: the data gradient reads the weight as stored ([K, N]), the weight gradient both operands
    token-major ([K, M], [K, N]) -- the same sums, bit for bit, as the row-major kernel on physically transposed copies
    (the K-major fragments come through ds_read_b64_tr_b16 in the b128 path's k-slot order), on BOTH MFMA shapes: the K-major
    forms follow fk_gemm_args.mfma since round 6 (16 x 16 x 32: every 16-lane group of the transpose read is one k-octet)."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(11)
    rnd = lambda *sh, sc=1.0: ((torch.rand(*sh, device=dev, generator=g) * 2 - 1) * sc).to(BF)  # noqa: E731
    ops.gemm_set_plan(1)     # no split-K pairs in the row-major reference (they differ in the last bits by design)
    ops.gemm_set_mfma(mfma)  # the row-major reference on the same MFMA shape
    try:
        _k_major_cases(ops, rnd, dev)
    finally:
        ops.gemm_set_plan(3)
        ops.gemm_set_mfma(0)


def _k_major_cases(ops, rnd, dev):
    # ---- layout 1: dX = dY W, W [N_out (= K), K_in (= N)], whole and as a column slice of a wider weight ----------------
    for (M, K, N, wide) in [(2560, 3072, 3072, 0), (300, 768, 512, 0), (8704, 3072, 3072, 15360), (777, 12288, 3072, 0)]:
        dy = rnd(M, K)
        Wfull = rnd(K, wide or N, sc=0.05)
        W = Wfull[:, 256:256 + N] if wide else Wfull
        ref = ops.gemm(dy, W.t().contiguous())
        got = ops.gemm(dy, W, layout=1)
        assert torch.equal(ref, got), f"layout 1 {M}x{N}x{K}: max diff {(ref.float() - got.float()).abs().max().item()}"
        acc = rnd(M, N)
        ref2 = ops.gemm(dy, W.t().contiguous(), out=acc.clone(), epilogue=ops.FK_EPI_RES, res=acc)
        a2 = acc.clone()
        got2 = ops.gemm(dy, W, out=a2, epilogue=ops.FK_EPI_RES, res=a2, layout=1)
        assert torch.equal(ref2, got2)
    # grouped (image / text rows of one [B, S, *] buffer, two weights), 3-D views with a batch stride
    B, S, S_txt, K, N = 2, 640, 128, 1024, 768
    dy, out_r, out_g = rnd(B, S, K), torch.zeros(B, S, N, device=dev, dtype=BF), torch.zeros(B, S, N, device=dev, dtype=BF)
    Wi, Wt = rnd(K, N, sc=0.05), rnd(K, N, sc=0.05)
    ops.gemm_grouped([dict(a=dy[:, S_txt:], w=Wi.t().contiguous(), out=out_r[:, S_txt:]), dict(a=dy[:, :S_txt], w=Wt.t().contiguous(), out=out_r[:, :S_txt])])
    ops.gemm_grouped([dict(a=dy[:, S_txt:], w=Wi, out=out_g[:, S_txt:], layout=1), dict(a=dy[:, :S_txt], w=Wt, out=out_g[:, :S_txt], layout=1)])
    assert torch.equal(out_r, out_g)
    # ---- layout 2: dW = dY^T X over the token rows of [1, S, *] views (column slices of wider buffers included) -----------
    for (T, M, N) in [(8704, 3072, 3072), (512, 768, 1024), (2560, 12288, 3072), (4096, 3072, 12288)]:
        dyb, xb = rnd(1, T + 64, M + 128), rnd(1, T + 64, N + 256)
        dy, x = dyb[:, 64:, 128:], xb[:, :T, 256:]
        ref = ops.gemm(dy[0].t().contiguous(), x[0].t().contiguous())
        got = ops.gemm(dy, x, layout=2)
        assert got.shape == (M, N) and torch.equal(ref, got), f"layout 2 {T}: {(ref.float() - got.float()).abs().max().item()}"
        exact = dy[0].float().t() @ x[0].float()
        assert_bf16_close(f"gemm layout 2 T{T} {M}x{N}", got, exact.to(BF), max_ulp=1, max_bad_frac=1e-3)
    # what the K-major path cannot take is refused, not mis-computed
    with pytest.raises(RuntimeError, match="layout"):
        ops.gemm(rnd(256, 128), rnd(128, 200), layout=1)            # N % 256
    with pytest.raises(RuntimeError, match="layout"):
        ops.gemm(rnd(2, 128, 384), rnd(2, 128, 256), layout=2)      # M % 256


def test_gemm_rejects_bad_arguments(ops):
    a, w = randn(64, 96).cuda(), randn(64, 96).cuda()  # K = 96 is not a multiple of 64
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm(a, w)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(randn(64, 64), randn(64, 64))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [3072, 512])
def test_ln_modulate(ops, D):
    from oracle import mmdit
    B, R = 2, 37
    x = randn(B, R, D, seed=12, scale=2.0) + 0.5
    mod = randn(B, 6 * D, seed=13, scale=0.3)
    shift, scale = mod[:, :D], mod[:, D:2 * D]
    md = mod.cuda()
    got = ops.ln_modulate(x.cuda(), md[:, :D], md[:, D:2 * D])
    ref = mmdit.layer_norm(x) * (1 + scale[:, None]) + shift[:, None]
    assert ref.dtype == BF
    assert_bf16_close(f"ln_modulate D={D}", got, ref, max_ulp=1, max_bad_frac=2e-3)
    ref32 = mmdit.layer_norm(x.float()) * (1 + scale.float()[:, None]) + shift.float()[:, None]
    d = report("ln_modulate vs fp32", got, ref32)
    assert d.max().item() < 0.1


def test_qkv_post(ops):
    from oracle import mmdit
    from oracle.helpers import prepare_latent_image_ids
    B, H, S_txt, hh, ww = 2, 3, 21, 6, 9
    S = S_txt + hh * ww  # 75: not a multiple of 64
    qkv = randn(B, S, 3 * H * 128, seed=14)
    wq_i, wk_i, wq_t, wk_t = [(1 + randn(128, seed=15 + i, scale=0.1).float()).to(BF) for i in range(4)]
    ids = torch.cat([torch.zeros(S_txt, 3), prepare_latent_image_ids(hh, ww)])
    cos, sin = mmdit.rope_tables(ids)
    q = torch.empty(B, H, S, 128, dtype=BF, device="cuda")
    k = torch.empty_like(q)
    qkv_dev = qkv.cuda()
    ops.qkv_post(qkv_dev, q, k, wq_i.cuda(), wk_i.cuda(), wq_t.cuda(), wk_t.cuda(), cos.cuda(), sin.cuda(), S_txt)
    D = H * 128
    def ref_qk(x, w_t, w_i):
        x = mmdit.heads(x, H)
        x = torch.cat([mmdit.rms_norm(x[:, :, :S_txt], w_t), mmdit.rms_norm(x[:, :, S_txt:], w_i)], dim=2)
        return mmdit.apply_rope(x, cos, sin)
    assert_bf16_close("qkv_post q", q, ref_qk(qkv[..., :D], wq_t, wq_i), max_ulp=1, max_bad_frac=2e-3)
    assert_bf16_close("qkv_post k", k, ref_qk(qkv[..., D:2 * D], wk_t, wk_i), max_ulp=1, max_bad_frac=2e-3)
    assert torch.equal(qkv_dev.cpu(), qkv)  # V (and the inputs) are left in place


@pytest.mark.parametrize("B,H,S", [(1, 2, 64), (2, 3, 75), (1, 2, 300), (1, 24, 2560), (1, 1, 1000), (2, 2, 257)])
def test_attention(ops, B, H, S):
    q, k = randn(B, H, S, 128, seed=20), randn(B, H, S, 128, seed=21)
    qkv = randn(B, S, 3 * H * 128, seed=22)          # V is read in place from the fused projection buffer
    v = qkv[:, :, 2 * H * 128:].reshape(B, S, H, 128).transpose(1, 2)
    if S == 300:  # spike a few keys so the running max jumps mid-sequence (online-softmax rescale)
        k[:, :, 200] = q[:, :, 17] * 2.0
        k[:, :, 290] = q[:, :, 150] * 3.0
    out = torch.zeros(B, S, H * 128 + 64, dtype=BF, device="cuda")  # wider row stride than H*128
    qkv_dev = qkv.cuda()
    ops.attention(q.cuda(), k.cuda(), qkv_dev[:, :, 2 * H * 128:], out)
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())  # fp32 oracle on bf16 inputs
    ref = ref.transpose(1, 2).reshape(B, S, H * 128)
    d = report(f"attention B{B} H{H} S{S}", out[..., :H * 128], ref)
    assert out[..., H * 128:].abs().max().item() == 0
    # P is rounded to bf16 before the PV product and O to bf16 at the end (2^-9 relative each): bounds RELATIVE to
    # the output scale (measured at S = 2560: max 0.25 %, mean 0.02 % of absmax)
    scale = ref.abs().max().item()
    assert d.max().item() <= 1e-2 * scale and d.mean().item() <= 1e-3 * scale
    # 8-byte aligned (not 16-byte) output rows take the narrow-store epilogue: same bits
    out2 = torch.zeros(B, S, H * 128 + 4, dtype=BF, device="cuda")
    ops.attention(q.cuda(), k.cuda(), qkv_dev[:, :, 2 * H * 128:], out2)
    assert torch.equal(out2[..., :H * 128], out[..., :H * 128]) and out2[..., H * 128:].abs().max().item() == 0


_TWO_KERNELS_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from gpt_image_edit_amd import ops
BF = torch.bfloat16
outs = {}
for B, H, S, lse_on in [(1, 2, 64, False), (2, 3, 75, True), (1, 2, 300, False), (2, 2, 257, True), (1, 1, 1000, False),
                        (1, 24, 2560, True), (1, 24, 5632, False), (2, 24, 4096, True)]:
    g = torch.Generator(device="cuda").manual_seed(1000 * S + B)
    q = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    k = torch.randn(B, H, S, 128, device="cuda", generator=g).to(BF)
    if S == 300:
        k[:, :, 200] = q[:, :, 17] * 2.0
        k[:, :, 290] = q[:, :, 150] * 40.0       # late large logit: the exponent-overflow restart
    qkv = torch.randn(B, S, 3 * H * 128, device="cuda", generator=g).to(BF)
    o = torch.zeros(B, S, H * 128, device="cuda", dtype=BF)
    if lse_on:
        lse = torch.zeros(B, H, S, device="cuda", dtype=torch.float32)
        ops.attention_lse(q, k, qkv[:, :, 2 * H * 128:], o, lse)
        outs[f"lse_{B}_{H}_{S}"] = lse.cpu()
    else:
        ops.attention(q, k, qkv[:, :, 2 * H * 128:], o)
    outs[f"o_{B}_{H}_{S}"] = o.cpu()
torch.save(outs, sys.argv[2])
"""


@pytest.mark.gpu
def test_attention_two_kernels_agree_bit_for_bit(tmp_path):
    """The 4-wave x 64-row forward (default) and the 8-wave x 32-row one are independent implementations of the same sums in
    the same order: every output element and every saved log-sum-exp must be IDENTICAL -- short, ragged, restart, plain-grid
    and stream-K shapes.  The library reads FK_ATTN_KERNEL once, so each kernel runs in its own process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for kern in ("4", "8"):
        out = tmp_path / f"k{kern}.pt"
        env = dict(os.environ, FK_ATTN_KERNEL=kern)
        subprocess.run([sys.executable, "-c", _TWO_KERNELS_SCRIPT, root, str(out)], check=True, env=env, timeout=600)
        res[kern] = torch.load(out)
    assert res["4"].keys() == res["8"].keys() and len(res["4"]) >= 12
    for name in res["4"]:
        a, b = res["4"][name], res["8"][name]
        assert torch.equal(a, b), f"{name}: {(a.float() - b.float()).abs().max().item()} max difference between the kernels"
        assert torch.isfinite(a.float()).all() and a.float().abs().max().item() > 0


@pytest.mark.parametrize("B,H,S", [(1, 2, 64), (2, 3, 75), (1, 2, 300), (1, 4, 2560), (2, 2, 257)])
def test_attention_fp32_debug_output_at_stated_tolerance(ops, B, H, S):
    """fk_attention_fwd_f32_debug = the same kernel with fp32 output and P entering PV as hi + lo bf16 terms: tiling,
    LDS layouts, exponent reference, masking and the key <-> MFMA k-slot binding are shared with the bf16 build, so
    this holds the kernel's arithmetic against fp32 SDPA at BASELINE.json's rtol 1e-3 / atol 1e-4."""
    q, k = randn(B, H, S, 128, seed=20), randn(B, H, S, 128, seed=21)
    qkv = randn(B, S, 3 * H * 128, seed=22)
    v = qkv[:, :, 2 * H * 128:].reshape(B, S, H, 128).transpose(1, 2)
    if S == 300:
        k[:, :, 200] = q[:, :, 17] * 2.0
        k[:, :, 290] = q[:, :, 150] * 3.0
    got = ops.attention_f32_debug(q.cuda(), k.cuda(), qkv.cuda()[:, :, 2 * H * 128:])
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float()).transpose(1, 2).reshape(B, S, H * 128)
    report(f"attention_f32_debug B{B} H{H} S{S}", got, ref)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-4)


def test_attention_value_layout_is_transpose_detecting(ops):
    # one-hot attention (huge matching logit) must copy exactly the selected V row: catches any mix-up of
    # the key <-> MFMA k-slot binding or of the LDS transpose-read addressing
    B, H, S = 1, 1, 192
    q = torch.zeros(B, H, S, 128)
    k = torch.zeros(B, H, S, 128)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(3))
    for i in range(S):
        q[0, 0, i, i % 128] = 12.0 + (i // 128)
        k[0, 0, perm[i], i % 128] = 12.0 + (i // 128)
    # rows i and i+128 share a one-hot direction; scale so the larger dot product wins decisively
    v = (torch.arange(S * 128, dtype=torch.float32).reshape(1, S, 128) % 509) / 32.0
    out = torch.zeros(B, S, 128, dtype=BF, device="cuda")
    ops.attention(q.to(BF).cuda(), k.to(BF).cuda(), v.to(BF).cuda(), out, scale=8.0)
    ref = F.scaled_dot_product_attention(q.to(BF).float(), k.to(BF).float(), v.to(BF).float()[:, None], scale=8.0)[:, 0]
    d = report("attention one-hot", out, ref)
    assert d.max().item() < 2e-2


# ---------------------------------------------------------------------------------------------------
def test_small_kernels(ops):
    from oracle import mmdit, scheduler
    x = randn(4, 3072, seed=30, scale=3.0)
    assert_bf16_close("silu", ops.silu(x.cuda()), F.silu(x), max_ulp=1, max_bad_frac=1e-3)
    a, b, c = randn(2, 3072, seed=31), randn(2, 3072, seed=32), randn(2, 3072, seed=33)
    assert_bf16_close("add3", ops.add3(a.cuda(), b.cuda(), c.cuda()), (a + b) + c, max_ulp=0)
    # timestep projection: model does timestep.to(bf16) * 1000 then the fp32 sinusoid, cast to bf16
    t = torch.tensor([1.0, 0.8516, 0.03125, 0.5], dtype=BF)
    freqs = torch.exp(-math.log(10000.0) * torch.arange(128, dtype=torch.float32) / 128)
    got = ops.timestep_proj(t.cuda(), freqs.cuda())
    ref = mmdit.sinusoid_256(t * 1000).to(BF)
    assert_bf16_close("timestep_proj bf16", got, ref, max_ulp=1, max_bad_frac=0.02)
    g = torch.tensor([3.5, 1.0, 2.25, 7.0])
    got = ops.timestep_proj(g.cuda(), freqs.cuda())
    ref = mmdit.sinusoid_256(g.to(BF) * 1000).to(BF)
    assert_bf16_close("timestep_proj fp32", got, ref, max_ulp=1, max_bad_frac=0.02)
    # Euler step with the slice fused
    xs = randn(2, 24, 64, seed=34)
    v = randn(2, 24, 64, seed=35)
    ts, sg = scheduler.shifted_sigmas(28, 1.15)
    x_dev = xs.cuda()
    ops.euler_step(x_dev, v.cuda(), 16, float(sg[4] - sg[3]))
    ref = xs.clone()
    ref[:, :16] = scheduler.euler_step(v[:, :16], sg[3], sg[4], xs[:, :16])
    assert_bf16_close("euler", x_dev, ref, max_ulp=0)
    # transpose
    m = randn(3, 70, 130, seed=36)
    dst = torch.empty(3, 130, 70, dtype=BF, device="cuda")
    ops.transpose(m.cuda(), dst)
    assert torch.equal(dst.cpu(), m.transpose(1, 2))
    # row softmax (fp32 scores -> bf16 probabilities)
    for n in (64, 1000, 4096, 16384):
        s = torch.randn(5, n, generator=torch.Generator().manual_seed(n)) * 4
        got = ops.softmax_rows(s.cuda())
        assert_bf16_close(f"softmax n={n}", got, torch.softmax(s, -1).to(BF), max_ulp=1, max_bad_frac=1e-3)


@pytest.mark.parametrize("gain", [6.0, 12.0])
def test_attention_restart_on_late_large_logit(ops, gain):
    """The kernel keeps a fixed exponent reference per row (row max of the first 32 keys + 24 log2 units).  A later
    score ~98 log2 units above it (gain 6) still fits fp32's exponent range and needs no second pass; ~196 units
    (gain 12) turns the row's sums into inf / NaN, which the kernel notices after the pass and repairs by repeating it
    with exact row maxima.  Both must match fp32 SDPA."""
    B, H, S = 1, 2, 640
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, H, S, 128, generator=g).to(BF)
    k = torch.randn(B, H, S, 128, generator=g).to(BF)
    qkv = torch.randn(B, S, 3 * H * 128, generator=g).to(BF)
    k[0, 0, 600] = q[0, 0, 5] * gain         # logit ~ gain * |q|^2 / sqrt(128) ~ 11 * gain nats, in the last tile
    k[0, 1, 321] = q[0, 1, 400] * (gain - 1.0)
    v = qkv[:, :, 2 * H * 128:].reshape(B, S, H, 128).transpose(1, 2)
    out = torch.empty(B, S, H * 128, device="cuda", dtype=BF)
    ops.attention(q.cuda(), k.cuda(), qkv.cuda()[:, :, 2 * H * 128:], out)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    ref = ref.transpose(1, 2).reshape(B, S, H * 128)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    # the hit rows copy (almost exactly) one V row
    assert (got[0, 5, :128] - v[0, 0, 600].float()).abs().max().item() <= 2e-2 * v.abs().max().item()


def test_attention_moderate_logit_growth_needs_no_restart_and_stays_accurate(ops):
    """Scores that grow by ~30 nats (43 log2 units) after the first keys stay far inside the exponent window
    (reference = first-block maximum + 24 log2 units, fp32 range beyond): same accuracy as the plain case."""
    B, H, S = 1, 1, 512
    g = torch.Generator().manual_seed(78)
    q = torch.randn(B, H, S, 128, generator=g).to(BF)
    k = torch.randn(B, H, S, 128, generator=g).to(BF)
    qkv = torch.randn(B, S, 3 * H * 128, generator=g).to(BF)
    k[0, 0, 300:310] = q[0, 0, 40:50] * 2.6      # ~ 2.6 * 128 / sqrt(128) ~ 29 nats on ten (query, key) pairs
    v = qkv[:, :, 2 * H * 128:].reshape(B, S, H, 128).transpose(1, 2)
    out = torch.empty(B, S, H * 128, device="cuda", dtype=BF)
    ops.attention(q.cuda(), k.cuda(), qkv.cuda()[:, :, 2 * H * 128:], out)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float()).transpose(1, 2).reshape(B, S, H * 128)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [16, 32])
def test_gemm_both_mfma_shapes_every_form(ops, shape):
    """The large-tile GEMM kernels on v_mfma_f32_16x16x32_bf16 (the default since round 5) and on v_mfma_f32_32x32x16_bf16
    (fk_gemm_set_mfma; FragMap in gemm_pingpong_bf16.hip: other fragment addressing, other accumulator-register ->
    (row, column) map in the epilogue): every check of this file under EACH shape explicitly -- the stated tolerance on the
    fp32-output build of every launch form, bit-equality between the forms, transpose detection on the large-tile kernels,
    all epilogues, grouped launches, the fused QKV epilogue against the unfused path."""
    ops.gemm_set_mfma(shape)
    try:
        for shape in HOT_SHAPES:
            test_hot_gemm_kernels_at_the_stated_tolerance(ops, *shape)
        # transpose detection ON the large-tile kernels (exact: a selector matrix against an asymmetric weight), every form
        from gpt_image_edit_amd import libfk
        lib = libfk.load()
        M, N, K = 512, 768, 128
        a = torch.zeros(M, K)
        a[torch.arange(M), (torch.arange(M) * 7) % K] = 1.0
        w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 64.0
        ref = a.to(BF).float() @ w.to(BF).float().T
        for force in (128, 256, 384):
            ops.gemm_set_variant(force)
            try:
                got = ops.gemm(a.to(BF).cuda(), w.to(BF).cuda(), None, out_fp32=2).cpu()
            finally:
                ops.gemm_set_variant(0)
            torch.testing.assert_close(got, ref, rtol=0, atol=0)
        for epi in ("none", "gelu", "silu", "scale"):
            test_gemm_bf16_epilogues(ops, epi)
        test_gemm_gate_residual_strided_views(ops)
        for args in [(256, 128, 64), (2560, 3072, 3072), (2048 + 3, 9216, 128), (700, 200, 192)]:
            test_gemm_large_tile_kernel(ops, *args)
        for args in [(1, 2560, 12288, 3072, 256), (1, 2560, 9216, 3072, 384), (2, 1200, 3072, 1024, 128), (1, 2560, 3072, 12288, 512)]:
            test_gemm_tile_choice_and_batched_epilogue(ops, *args)
        test_gemm_grouped(ops)
        test_gemm_fused_qkv_epilogue(ops)
        test_gemm_tile_order_does_not_change_the_bits(ops)
    finally:
        ops.gemm_set_mfma(0)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 512, 192), (2560, 3072, 3072), (2048 + 3, 1280 - 8, 256),
                                   (700, 200, 320), (2560, 9216, 3072)])
def test_gemm10_hand_placed_kernel_equals_gemm8_bit_for_bit(ops, M, N, K):
    """gemm10_kernel (4 waves, 128 x 128 per wave, K loop as one hand-placed asm statement: csrc/gemm10_gen.py) stages the
    same LDS image, reads the same fragments and walks K in the same order as gemm8_kernel on v_mfma_f32_16x16x32_bf16: the
    two must agree BIT FOR BIT -- on the fp32-output build (raw accumulators + bias) and through the shared bf16 epilogue --
    for 1, 2, 3 and many K-tiles (prologue / odd / even loop exits), ragged M and N edges, and the fp32 build must meet the
    stated tolerance against an fp64 product."""
    a, w, bias = randn(M, K, seed=171), randn(N, K, seed=172, scale=0.05), randn(N, seed=173, scale=0.1)
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    ops.gemm_set_mfma(16)
    try:
        got = {}
        for force in (256, 1024):
            ops.gemm_set_variant(force)
            got[force] = (ops.gemm(ad, wd, bd, out_fp32=2).clone(), ops.gemm(ad, wd, bd, epilogue=ops.FK_EPI_GELU_TANH).clone())
            torch.cuda.synchronize()
            used = ops.gemm_last_variant()
            if N % 256 == 0:
                assert used == force, f"forced {force}, ran {used}"
    finally:
        ops.gemm_set_variant(0)
        ops.gemm_set_mfma(0)
    ref = (a.double() @ w.double().T + bias.double()).float()
    report(f"gemm10 f32 {M}x{N}x{K}", got[1024][0], ref)
    torch.testing.assert_close(got[1024][0].cpu(), ref, rtol=1e-3, atol=1e-4)
    if N % 256 == 0:       # otherwise both forced forms fall back to the same 256 x 128 kernel
        assert torch.equal(got[256][0], got[1024][0]), "gemm10 accumulators differ from gemm8's"
        assert torch.equal(got[256][1], got[1024][1]), "gemm10 bf16 epilogue output differs from gemm8's"


def test_gemm10_batched_rows_gate_residual_and_grouped(ops):
    """gemm10 behind the batched [B, S, :] addressing (a batch boundary inside a 256-row tile), the gated-residual epilogue in
    place, and a grouped launch of two problems: equal to gemm8's output bit for bit."""
    B, S, N, K = 3, 200, 512, 256
    x = randn(B, S, K, seed=181)
    w, bias = randn(N, K, seed=182, scale=0.02), randn(N, seed=183, scale=0.1)
    res = randn(B, S, N, seed=184)
    mod = randn(B, 3 * N, seed=185, scale=0.5)
    outs = {}
    ops.gemm_set_mfma(16)
    try:
        for force in (256, 1024):
            ops.gemm_set_variant(force)
            rd = res.cuda().clone()
            ops.gemm(x.cuda(), w.cuda(), bias.cuda(), out=rd, epilogue=ops.FK_EPI_GATE_RES, res=rd, gate=mod.cuda()[:, N:2 * N])
            assert ops.gemm_last_variant() == force
            a1, a2 = randn(300, K, seed=186).cuda(), randn(1000, K, seed=187).cuda()
            w2 = randn(N, K, seed=188, scale=0.02).cuda()
            g = ops.gemm_grouped([dict(a=a1, w=w.cuda(), bias=bias.cuda()), dict(a=a2, w=w2, bias=None)], epilogue=ops.FK_EPI_NONE)
            assert ops.gemm_last_variant() == force
            outs[force] = (rd.clone(), g[0].clone(), g[1].clone())
    finally:
        ops.gemm_set_variant(0)
        ops.gemm_set_mfma(0)
    for u, v in zip(outs[256], outs[1024]):
        assert torch.equal(u, v)


def test_gemm10_one_workgroup_per_cu_form_and_the_plan_that_picks_it(ops):
    """A grid of >= 4 tiles per CU runs gemm10 as one workgroup per CU walking the tile list (same tile code, an LDS-only barrier
    between tiles), and the launch plan itself turns a pure 256 x 256 grid with K >= 6144 that deep into gemm10: both equal to
    gemm8's result bit for bit."""
    ops.gemm_set_mfma(16)
    try:
        a, w, bias = randn(8192, 128, seed=191).cuda(), randn(8192, 128, seed=192, scale=0.05).cuda(), randn(8192, seed=193, scale=0.1).cuda()
        got = {}
        for force in (256, 1024):           # 32 x 32 = 1024 tiles on 256 CUs: the walking form; K = 128: two K-tiles per tile
            ops.gemm_set_variant(force)
            got[force] = ops.gemm(a, w, bias, epilogue=ops.FK_EPI_SILU).clone()
            assert ops.gemm_last_variant() == force
        assert torch.equal(got[256], got[1024])
        a, w = randn(16384, 6144, seed=194).cuda(), randn(4096, 6144, seed=195, scale=0.02).cuda()
        ops.gemm_set_variant(0)
        ops.gemm_set_plan(1)                 # batch-invariant plan: no split-K, so the choice is between the pure grids
        auto = ops.gemm(a, w, None).clone()
        assert ops.gemm_last_variant() == 1024, ops.gemm_last_variant()
        ops.gemm_set_variant(256)
        assert torch.equal(auto, ops.gemm(a, w, None))
        ref = (a[:256].double() @ w.double().T).float()
        assert_bf16_close("gemm10 via the plan 16384x4096x6144 (first 256 rows)", auto[:256], ref.to(BF), max_ulp=1, max_bad_frac=1e-4)
    finally:
        ops.gemm_set_variant(0)
        ops.gemm_set_plan(3)
        ops.gemm_set_mfma(0)
