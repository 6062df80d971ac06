"""GPU parity of the whole edit (FluxKontextPipeline on HIP) against the CPU oracle pipeline, plus the
size-independent properties used at BASELINE sizes."""
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_smoke_entry():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.smoke()


def test_edit_matches_oracle_pipeline():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    from oracle import pipeline as opipe

    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=2)
    sd_f = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=21).items()}
    sd_v = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=22).items()}
    tr = HipFluxTransformer2DModel(cfg, device="cuda"); tr.load_state_dict(sd_f)
    vae = HipAutoencoderKL(device="cuda"); vae.load_state_dict(sd_v)
    pipe = FluxKontextPipeline(tr, vae)
    g = torch.Generator().manual_seed(7)
    B, H, W, Hc, Wc = 2, 64, 96, 96, 64      # target and condition of different shapes
    cond = torch.rand(B, 3, Hc, Wc, generator=g) * 2 - 1
    emb = torch.randn(B, 40, 4096, generator=g).to(BF)
    pooled = torch.randn(B, 768, generator=g).to(BF)
    noise = torch.randn(B, 16, H // 8, W // 8, generator=g).to(BF)
    steps = 3
    out = pipe(image=cond.cuda(), prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), height=H, width=W,
               num_inference_steps=steps, guidance_scale=4.0, latents=pipe._pack_latents(noise, B, 16, H // 8, W // 8).cuda(),
               output_type="pt_raw", max_area=H * W, _auto_resize=False)
    ref = opipe.kontext_edit(sd_f, sd_v, cond, emb, pooled, noise, H, W, num_inference_steps=steps,
                             guidance_scale=4.0, flux_config=cfg)
    sd_f32, sd_v32 = {k: v.float() for k, v in sd_f.items()}, {k: v.float() for k, v in sd_v.items()}
    ref32 = opipe.kontext_edit(sd_f32, sd_v32, cond.to(BF).float(), emb.float(), pooled.float(), noise.float(), H, W,
                               num_inference_steps=steps, guidance_scale=4.0, flux_config=cfg)
    d_l = report("edit latents vs bf16-oracle", out.latents, ref["latents"])
    d_l32 = report("edit latents vs fp32-oracle", out.latents, ref32["latents"])
    fl_l = report("edit latents bf16-oracle vs fp32-oracle (floor)", ref["latents"], ref32["latents"])
    d_i32 = report("edit image vs fp32-oracle", out.images, ref32["image"])
    fl_i = report("edit image bf16-oracle vs fp32-oracle (floor)", ref["image"], ref32["image"])
    assert out.images.shape == (B, 3, H, W)
    # relative to the bf16 oracle's own distance from the exact result only (no absolute arm)
    assert d_l32.max().item() <= 1.25 * fl_l.max().item() and d_l32.mean().item() <= 1.1 * fl_l.mean().item()
    assert d_i32.max().item() <= 1.25 * fl_i.max().item() and d_i32.mean().item() <= 1.1 * fl_i.mean().item()
    assert d_l.max().item() <= 1.25 * fl_l.max().item()
    # first arm (VERDICT r4 next #1a): the HIP path reproduces the bf16 rounding points, so it is held to the bf16 oracle
    # DIRECTLY, with a bound that does not scale with the bf16-vs-fp32 floor (measured r04: max 4.7e-2, mean 6.1e-3 at scale 5.7)
    _direct_bf16_bound("edit latents", d_l, ref["latents"], DIRECT_MAX, DIRECT_MEAN)
    _direct_bf16_bound("edit image", report("edit image vs bf16-oracle", out.images, ref["image"]), ref["image"],
                       DIRECT_MAX_IMG, DIRECT_MEAN_IMG)


# Direct HIP-vs-bf16-oracle bounds, as fractions of the oracle output's largest magnitude.  They do NOT scale with the
# bf16-vs-fp32 round-off floor (with random weights that floor grows to 30 % of the scale over 28 steps, so a bound relative
# to it admits a 30 x regression): <= 2 x what profiles/r04_gpu_tests.log shows for these tests (latents: max 1.1-1.2 %,
# mean 0.10-0.11 % of the scale = 1.5-2 bf16 ulps of the largest values).
DIRECT_MAX, DIRECT_MEAN = 2.0 ** -6, 2.0 ** -9
# the decoded image passes the VAE decoder (GroupNorm statistics, 3 up-sampling stages) on top of the latents' differences
DIRECT_MAX_IMG, DIRECT_MEAN_IMG = 2.0 ** -4, 2.0 ** -7


def _direct_bf16_bound(name, d, ref, c_max, c_mean):
    scale = ref.float().abs().max().item()
    mx, mn = d.max().item(), d.mean().item()
    print(f"[parity] {name}: direct bound vs bf16-oracle: max {mx:.3e} <= {c_max * scale:.3e}, mean {mn:.3e} <= "
          f"{c_mean * scale:.3e} (scale {scale:.2f})", flush=True)
    assert mx <= c_max * scale and mn <= c_mean * scale, f"{name}: HIP path disagrees with the bf16 oracle"


def test_28_step_edit_matches_oracle_pipeline():
    """the reference's default step count (28, cli.py:278) through the whole HIP edit -- condition encode,
    28 x (MMDiT + Euler) on a full-width 2 + 2-block model, decode -- against the oracle pipeline in fp32 and bf16: the
    error accumulated over 28 Euler steps stays within the bf16 oracle's own distance from the exact result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    from oracle import pipeline as opipe
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=2, num_single_layers=2)
    sd_f = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=31).items()}
    sd_v = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=32).items()}
    tr = HipFluxTransformer2DModel(cfg, device="cuda"); tr.load_state_dict(sd_f)
    vae = HipAutoencoderKL(device="cuda"); vae.load_state_dict(sd_v)
    pipe = FluxKontextPipeline(tr, vae)
    g = torch.Generator().manual_seed(11)
    B, H, W, steps = 1, 128, 128, 28
    cond = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    emb = torch.randn(B, 48, 4096, generator=g).to(BF)
    pooled = torch.randn(B, 768, generator=g).to(BF)
    noise = torch.randn(B, 16, H // 8, W // 8, generator=g).to(BF)
    out = pipe(image=cond.cuda(), prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), height=H, width=W,
               num_inference_steps=steps, guidance_scale=3.5, latents=pipe._pack_latents(noise, B, 16, H // 8, W // 8).cuda(),
               output_type="pt_raw", max_area=H * W, _auto_resize=False)
    ref = opipe.kontext_edit(sd_f, sd_v, cond, emb, pooled, noise, H, W, num_inference_steps=steps, guidance_scale=3.5,
                             flux_config=cfg)
    ref32 = opipe.kontext_edit({k: v.float() for k, v in sd_f.items()}, {k: v.float() for k, v in sd_v.items()},
                               cond.to(BF).float(), emb.float(), pooled.float(), noise.float(), H, W,
                               num_inference_steps=steps, guidance_scale=3.5, flux_config=cfg)
    for i in (0, 6, 13, 20, 27):
        a, b = ref["per_step"][i].float(), ref32["per_step"][i]
        print(f"[parity] 28-step floor growth, step {i:2d}: bf16-oracle vs fp32-oracle max {(a - b).abs().max().item():.3e} "
              f"mean {(a - b).abs().mean().item():.3e}", flush=True)
    d_l = report("28-step edit latents vs bf16-oracle", out.latents, ref["latents"])
    d_l32 = report("28-step edit latents vs fp32-oracle", out.latents, ref32["latents"])
    fl_l = report("28-step edit latents bf16-oracle vs fp32-oracle (floor)", ref["latents"], ref32["latents"])
    d_i32 = report("28-step edit image vs fp32-oracle", out.images, ref32["image"])
    fl_i = report("28-step edit image bf16-oracle vs fp32-oracle (floor)", ref["image"], ref32["image"])
    assert d_l32.max().item() <= 1.25 * fl_l.max().item() and d_l32.mean().item() <= 1.1 * fl_l.mean().item()
    assert d_i32.max().item() <= 1.25 * fl_i.max().item() and d_i32.mean().item() <= 1.1 * fl_i.mean().item()
    assert d_l.max().item() <= 1.25 * fl_l.max().item()
    # first arm: direct bound against the bf16 oracle (the floor here is 1.68 / 0.336 on latents of scale 5.2: 30 x what the
    # HIP path measures against the bf16 oracle -- r04: max 6.25e-2, mean 5.3e-3)
    _direct_bf16_bound("28-step edit latents", d_l, ref["latents"], DIRECT_MAX, DIRECT_MEAN)
    _direct_bf16_bound("28-step edit image", report("28-step edit image vs bf16-oracle", out.images, ref["image"]),
                       ref["image"], DIRECT_MAX_IMG, DIRECT_MEAN_IMG)


def test_full_size_properties():
    """At the BASELINE size (512^2 edit, S = 2560, full-width blocks) the oracle is too slow to run whole,
    so check size-independent properties of the HIP path: determinism, batch independence (a sample's
    result does not depend on its neighbours) and linearity of the Euler update."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec, ops
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=2, num_single_layers=2)
    m = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=3)
    g = torch.Generator(device="cuda").manual_seed(1)
    B, S_txt, S_img = 2, 512, 2048
    hs = torch.randn(B, S_img, 64, generator=g, device="cuda").to(BF)
    enc = torch.randn(B, S_txt, 4096, generator=g, device="cuda").to(BF)
    pooled = torch.randn(B, 768, generator=g, device="cuda").to(BF)
    t = torch.tensor([0.5, 0.25], device="cuda").to(BF)
    gd = torch.full((B,), 3.5, device="cuda")
    from gpt_image_edit_amd.helpers import _prepare_latent_image_ids as ids
    img_ids = torch.cat([ids(1, 32, 32, "cuda", BF), ids(1, 32, 32, "cuda", BF)])
    img_ids[1024:, 0] = 1
    txt_ids = torch.zeros(S_txt, 3, device="cuda", dtype=BF)
    kw = dict(txt_ids=txt_ids, img_ids=img_ids, return_dict=False)
    def run(sl):
        return m(hidden_states=hs[sl], encoder_hidden_states=enc[sl], pooled_projections=pooled[sl], timestep=t[sl],
                 guidance=gd[sl], **kw)[0].clone()
    o2, o2b = run(slice(None)), run(slice(None))
    assert torch.equal(o2, o2b), "forward is not deterministic"
    o1, o1b = run(slice(1, 2)), run(slice(1, 2))
    assert torch.equal(o1, o1b), "forward is not deterministic at batch 1 (split-K GEMMs in play)"
    assert torch.isfinite(o2.float()).all()
    # Default launch plan: at batch 1 the K = 12288 / 15360 GEMMs (120 tiles of 256 x 256) run as two half-K workgroups
    # per tile, at batch 2 (240 tiles) they do not: the two sums differ in their last bits, nothing more
    d = (o1[0].float() - o2[1].float()).abs().max().item()
    print(f"[batch independence, default plan] max|batch-of-1 - same sample in a batch of 2| = {d:.3e} at scale "
          f"{o2.float().abs().max().item():.2f}")
    assert d <= 2 ** -6 * o2.float().abs().max().item()
    # Batch-invariant plan (no split-K; mixed grids stay on, they are bit-identical; no stream-K attention grid: at batch 2
    # the 480 blocks = 1.875 rounds would run as one, cutting some blocks' keys): bit for bit
    ops.gemm_set_plan(1)
    ops.attention_set_split(0)
    try:
        i2, i1 = run(slice(None)), run(slice(1, 2))
    finally:
        ops.gemm_set_plan(3)
        ops.attention_set_split(1)
    assert torch.equal(i1[0], i2[1]), "a sample's output depends on its batch neighbours"
    # Euler: two half steps with the same velocity == one full step up to one bf16 rounding per step
    x = torch.randn(1, 1024, 64, generator=g, device="cuda").to(BF)
    v = torch.randn(1, 1024, 64, generator=g, device="cuda").to(BF)
    a = x.clone(); ops.euler_step(a, v, 1024, -0.0625)
    b = x.clone(); ops.euler_step(b, v, 1024, -0.03125); ops.euler_step(b, v, 1024, -0.03125)
    assert (a.float() - b.float()).abs().max().item() <= 2 ** -6 * (1 + x.float().abs().max().item())


def test_true_cfg_matches_oracle_and_two_pass():
    """true_cfg_scale > 1 with negative embeddings (flux_pipeline.py:928,1080-1095): the HIP pipeline runs the
    positive and negative pass as one batch of 2B + the fused combine kernel; the oracle runs the reference's two
    separate calls."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec, ops
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    from oracle import pipeline as opipe

    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    sd_f = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=41).items()}
    sd_v = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=42).items()}
    tr = HipFluxTransformer2DModel(cfg, device="cuda"); tr.load_state_dict(sd_f)
    vae = HipAutoencoderKL(device="cuda"); vae.load_state_dict(sd_v)
    pipe = FluxKontextPipeline(tr, vae)
    g = torch.Generator().manual_seed(17)
    B, H, W = 1, 64, 64
    cond = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    emb, nemb = torch.randn(B, 24, 4096, generator=g).to(BF), torch.randn(B, 24, 4096, generator=g).to(BF)
    pooled, npooled = torch.randn(B, 768, generator=g).to(BF), torch.randn(B, 768, generator=g).to(BF)
    noise = torch.randn(B, 16, H // 8, W // 8, generator=g).to(BF)
    kw = dict(image=cond.cuda(), prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), height=H, width=W,
              num_inference_steps=2, guidance_scale=4.0, latents=pipe._pack_latents(noise, B, 16, H // 8, W // 8).cuda(),
              output_type="latent", max_area=H * W, _auto_resize=False)
    out = pipe(negative_prompt_embeds=nemb.cuda(), negative_pooled_prompt_embeds=npooled.cuda(), true_cfg_scale=2.5, **kw)
    plain = pipe(**kw)
    assert not torch.equal(out.latents, plain.latents), "true CFG had no effect"
    # scale <= 1 or missing negatives: the reference's do_true_cfg is False
    same = pipe(negative_prompt_embeds=nemb.cuda(), negative_pooled_prompt_embeds=npooled.cuda(), true_cfg_scale=1.0, **kw)
    assert torch.equal(same.latents, plain.latents)
    ref = opipe.kontext_edit(sd_f, sd_v, cond, emb, pooled, noise, H, W, num_inference_steps=2, guidance_scale=4.0,
                             flux_config=cfg, decode=False, negative_prompt_embeds=nemb, negative_pooled=npooled,
                             true_cfg_scale=2.5)
    f32 = lambda d: {k: v.float() for k, v in d.items()}  # noqa: E731
    ref32 = opipe.kontext_edit(f32(sd_f), f32(sd_v), cond.to(BF).float(), emb.float(), pooled.float(), noise.float(), H, W,
                               num_inference_steps=2, guidance_scale=4.0, flux_config=cfg, decode=False,
                               negative_prompt_embeds=nemb.float(), negative_pooled=npooled.float(), true_cfg_scale=2.5)
    d32 = report("true-CFG latents vs fp32-oracle", out.latents, ref32["latents"])
    floor = report("true-CFG latents bf16-oracle vs fp32-oracle (floor)", ref["latents"], ref32["latents"])
    assert d32.max().item() <= max(2.5 * floor.max().item(), 3e-2 * ref32["latents"].abs().max().item())
    # the combine kernel alone, bit-exact against the torch expression on the GPU
    a, b = torch.randn(4, 1024, 64, generator=g).to(BF).cuda(), torch.randn(4, 1024, 64, generator=g).to(BF).cuda()
    assert torch.equal(ops.true_cfg(a, b, 2.5), b + 2.5 * (a - b))


def test_string_prompt_path_equals_embedding_path():
    # `prompt="..."` runs the pipeline's own CLIP / T5 encoders (transformers' classes, tiny random stand-ins of the
    # right widths here) and must produce exactly what passing their embeddings does
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tiny_text_encoders import ToyTokenizer
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL

    torch.manual_seed(5)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=64, hidden_size=768, intermediate_size=64, num_hidden_layers=1,
                                        num_attention_heads=4, max_position_embeddings=77, eos_token_id=61,
                                        bos_token_id=62, pad_token_id=0)).eval().to(BF).cuda()
    t5 = T5EncoderModel(T5Config(vocab_size=64, d_model=4096, d_kv=8, d_ff=64, num_layers=1, num_heads=4)).eval().to(BF).cuda()
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    sd_f = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=31).items()}
    sd_v = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=32).items()}
    tr = HipFluxTransformer2DModel(cfg, device="cuda"); tr.load_state_dict(sd_f)
    vae = HipAutoencoderKL(device="cuda"); vae.load_state_dict(sd_v)
    pipe = FluxKontextPipeline(tr, vae, text_encoder=clip, tokenizer=ToyTokenizer(), text_encoder_2=t5,
                               tokenizer_2=ToyTokenizer(model_max_length=512))
    H = W = 64
    g = torch.Generator().manual_seed(9)
    cond = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).cuda()
    lat = pipe._pack_latents(torch.randn(1, 16, H // 8, W // 8, generator=g).to(BF), 1, 16, H // 8, W // 8).cuda()
    kw = dict(image=cond, height=H, width=W, num_inference_steps=2, latents=lat, output_type="pt_raw", max_area=H * W,
              _auto_resize=False, max_sequence_length=32)
    text = "turn the car red"
    with torch.no_grad():
        a = pipe(prompt=text, **kw)
        pe, pp, ids = pipe.encode_prompt(text, None, max_sequence_length=32)
        b = pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, **kw)
    assert pe.shape == (1, 32, 4096) and pp.shape == (1, 768) and ids.shape == (32, 3)
    assert torch.equal(a.latents, b.latents) and torch.equal(a.images, b.images)
    with pytest.raises(ValueError, match="Cannot forward both"):
        pipe(prompt=text, prompt_embeds=pe, pooled_prompt_embeds=pp, **kw)


def test_graph_captured_loop_gives_the_same_bits():
    """conditioning pass + the whole denoise loop as ONE hipGraph launch (FluxKontextPipeline(use_graph=
    True) / FK_GRAPH=1): captured once per call shape, replayed for the next edits with the new inputs copied into the
    graph's buffers -- bit-identical to the eager loop, also at a shape whose K-long GEMMs run as split-K pairs (their
    ticket words are monotonic, so a replay needs no reset)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=2)
    tr = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=3)
    vae = HipAutoencoderKL(device="cuda", init="synthetic", seed=4)
    eager, graphed = FluxKontextPipeline(tr, vae, use_graph=False), FluxKontextPipeline(tr, vae, use_graph=True)
    H = W = 256                                   # S = 300 + 256 + 256 = 812 rows: 4 row tiles, K-long GEMMs split

    def edit(pipe, seed, steps=6, **kw):
        g = torch.Generator().manual_seed(seed)
        cond = torch.rand(1, 3, H, W, generator=g) * 2 - 1
        emb = torch.randn(1, 300, 4096, generator=g).to(BF)
        pooled = torch.randn(1, 768, generator=g).to(BF)
        noise = torch.randn(1, 16, H // 8, W // 8, generator=g).to(BF)
        out = pipe(image=cond.cuda(), prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), height=H, width=W,
                   num_inference_steps=steps, guidance_scale=3.5, latents=pipe._pack_latents(noise, 1, 16, H // 8, W // 8).cuda(),
                   output_type="pt_raw", max_area=H * W, _auto_resize=False, **kw)
        return out.latents.clone(), out.images.clone()
    for seed in (1, 2, 3):                         # seed 1 captures, 2 and 3 replay with new inputs
        le, ie = edit(eager, seed)
        lg, ig = edit(graphed, seed)
        torch.cuda.synchronize()
        assert torch.isfinite(lg.float()).all()
        assert torch.equal(le, lg) and torch.equal(ie, ig), f"graph replay differs from the eager loop (seed {seed})"
    key0 = graphed._loop_graph[0]
    edit(graphed, 4, steps=5)                      # another schedule: the step sizes are baked in -> a new capture
    assert graphed._loop_graph[0] != key0
    l5e, _ = edit(eager, 4, steps=5)
    l5g, _ = edit(graphed, 4, steps=5)
    assert torch.equal(l5e, l5g)
    # true CFG (positive + negative pass as one batch of 2) through the graph as well
    g = torch.Generator().manual_seed(9)
    neg = dict(negative_prompt_embeds=torch.randn(1, 300, 4096, generator=g).to(BF).cuda(),
               negative_pooled_prompt_embeds=torch.randn(1, 768, generator=g).to(BF).cuda(), true_cfg_scale=2.0)
    lce, _ = edit(eager, 5, steps=3, **neg)
    lcg, _ = edit(graphed, 5, steps=3, **neg)
    assert torch.equal(lce, lcg)
    # a per-step callback cannot live inside a graph: such calls take the eager loop
    seen = []
    edit(graphed, 6, steps=3, callback_on_step_end=lambda p, i, t, kw: seen.append(i) or {})
    assert seen == [0, 1, 2]
    # results of consecutive graph calls must stay distinct (the graph's static buffer is not handed out) ...
    def latent_only(pipe, seed):
        g = torch.Generator().manual_seed(seed)
        noise = torch.randn(1, 16, H // 8, W // 8, generator=g).to(BF)
        return pipe(image=(torch.rand(1, 3, H, W, generator=g) * 2 - 1).cuda(), prompt_embeds=torch.randn(1, 300, 4096, generator=g).to(BF).cuda(),
                    pooled_prompt_embeds=torch.randn(1, 768, generator=g).to(BF).cuda(), height=H, width=W, num_inference_steps=6,
                    guidance_scale=3.5, latents=pipe._pack_latents(noise, 1, 16, H // 8, W // 8).cuda(), output_type="latent",
                    max_area=H * W, _auto_resize=False)
    first = latent_only(graphed, 11)
    keep_l, keep_i = first.latents.clone(), first.images.clone()
    second = latent_only(graphed, 12)
    torch.cuda.synchronize()
    assert first.latents.data_ptr() != second.latents.data_ptr() and first.images.data_ptr() != second.images.data_ptr()
    assert torch.equal(first.latents, keep_l) and torch.equal(first.images, keep_i) and not torch.equal(second.latents, keep_l)
    # ... and a weight rewritten in place between two graph calls must reach the replay (no stale fused QKV copy)
    before = latent_only(graphed, 13).latents.clone()
    w = tr.p("transformer_blocks.0.attn.to_q.weight")
    with torch.no_grad():
        w.mul_(0.5)                                 # in place: bumps the parameter's version, like load_state_dict / an optimiser
    after_g = latent_only(graphed, 13).latents.clone()
    after_e = latent_only(eager, 13).latents.clone()
    with torch.no_grad():
        w.mul_(2.0)
    assert not torch.equal(before, after_g) and torch.equal(after_g, after_e)
