"""Pixels in / pixels out: the two fused HIP kernels against the oracle restatement of
cli.prepare_condition_images + VaeImageProcessor.resize/preprocess/postprocess, bit-exact; and the uint8 route
through the pipeline against the reference's float-tensor route."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("hin,win,hout,wout", [(64, 96, 64, 96), (37, 53, 64, 48), (200, 120, 96, 160),
                                               (512, 768, 832, 1248), (33, 1, 16, 16)])
def test_pixels_in_matches_oracle(hin, win, hout, wout):
    _need_gpu()
    from gpt_image_edit_amd import ops
    from oracle import vae as ovae
    g = torch.Generator().manual_seed(hin * 1000 + win)
    u8 = torch.randint(0, 256, (2, hin, win, 3), generator=g, dtype=torch.uint8)
    got = ops.pixels_to_nhwc(u8.cuda(), hout, wout, 32, renorm=bool(u8.min() >= 128)).cpu()
    ref = ovae.preprocess_uint8(u8, hout, wout)                    # [N,3,h,w] bf16
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), ref)
    assert not got[..., 3:].any(), "padding channels must be zero"


def test_pixels_in_renormalises_bright_images():
    """VaeImageProcessor.preprocess normalises once more when the [-1,1] tensor has no negative value."""
    _need_gpu()
    from gpt_image_edit_amd import image_processor
    from oracle import vae as ovae
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(128, 256, (1, 40, 56, 3), generator=g, dtype=torch.uint8)
    wrapped = image_processor.pixels_to_latent_input(u8, 48, 48, "cuda")
    assert isinstance(wrapped, image_processor.NhwcPixels)      # explicit marker of the fused route
    got = wrapped.tensor.cpu()
    ref = ovae.preprocess_uint8(u8, 48, 48)
    assert ref.min() < 0, "the quirk must actually trigger in this case"
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), ref)


def test_pixels_in_renormalisation_is_decided_after_the_resize():
    """The reference checks `image.min()` on the RESIZED tensor: a dark pixel that the nearest down-sample drops
    must not switch the second normalisation off."""
    _need_gpu()
    from gpt_image_edit_amd import image_processor
    from oracle import vae as ovae
    g = torch.Generator().manual_seed(6)
    u8 = torch.randint(128, 256, (1, 64, 64, 3), generator=g, dtype=torch.uint8)
    rows = image_processor.nearest_source_index(32, 64)
    assert 1 not in rows.tolist()
    u8[0, 1, 1] = 0                                      # never sampled by the 64 -> 32 nearest resize
    got = image_processor.pixels_to_latent_input(u8, 32, 32, "cuda").tensor.cpu()
    ref = ovae.preprocess_uint8(u8, 32, 32)
    assert ref.min() < 0                                 # the reference renormalised
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), ref)
    u8[0, 2, 2] = 0                                      # sampled: now nothing is renormalised
    got = image_processor.pixels_to_latent_input(u8, 32, 32, "cuda").tensor.cpu()
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), ovae.preprocess_uint8(u8, 32, 32))


def test_pre_encoded_latents_32_wide_are_not_mistaken_for_pixels():
    """[N,16,h,32] bf16 condition latents (a 256-px-wide image) pass through prepare_latents untouched, as in the
    reference (flux_pipeline.py:675-678), instead of being pushed through the VAE encoder as 'NHWC pixels'."""
    _need_gpu()
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=0, num_single_layers=1)
    pipe = FluxKontextPipeline(HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=1),
                               HipAutoencoderKL(device="cuda", init="synthetic", seed=2))
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(2, 16, 8, 32, generator=g).to(BF).cuda()
    _, image_latents, _, image_ids = pipe.prepare_latents(lat, 2, 16, 64, 256, BF, "cuda",
                                                          generator=torch.Generator().manual_seed(0))
    assert torch.equal(image_latents, pipe._pack_latents(lat, 2, 16, 8, 32)) and image_ids.shape == (4 * 16, 3)
    # one generator per sample (diffusers randn_tensor): sample i is reproducible from generator i alone
    gens = [torch.Generator().manual_seed(100 + i) for i in range(2)]
    a = pipe.prepare_latents(None, 2, 16, 64, 64, BF, "cuda", generator=gens)[0]
    one = pipe.prepare_latents(None, 1, 16, 64, 64, BF, "cuda", generator=[torch.Generator().manual_seed(101)])[0]
    assert torch.equal(a[1:], one)


@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_pixels_out_matches_oracle(dtype):
    _need_gpu()
    from gpt_image_edit_amd import ops
    from oracle import vae as ovae
    g = torch.Generator().manual_seed(11)
    img = (torch.randn(2, 3, 48, 80, generator=g) * 0.8).to(dtype)     # some values outside [-1, 1]
    img[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, 0.0, 1.0 / 255 - 1, 0.00390625, -3.0, 3.0, 0.5], dtype=dtype)
    got = ops.image_to_u8(img.cuda()).cpu().numpy()
    ref = ovae.postprocess_uint8(img)
    assert got.dtype == np.uint8 and got.shape == (2, 48, 80, 3)
    assert np.array_equal(got, ref)


def test_uint8_route_equals_float_route_through_the_pipeline():
    """pipe(image=<uint8 pixels>) must give the same edit as the reference route pipe(image=<float [-1,1] tensor>)
    (cli.prepare_condition_images), including the preferred-resolution resize."""
    _need_gpu()
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    tr = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=31)
    vae = HipAutoencoderKL(device="cuda", init="synthetic", seed=32)
    pipe = FluxKontextPipeline(tr, vae)
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (1, 60, 90, 3), generator=g, dtype=torch.uint8)
    flt = ((u8.float() / 255.0).permute(0, 3, 1, 2) - 0.5) / 0.5          # cli.py:106-109
    emb = torch.randn(1, 24, 4096, generator=g).to(BF).cuda()
    pooled = torch.randn(1, 768, generator=g).to(BF).cuda()
    kw = dict(prompt_embeds=emb, pooled_prompt_embeds=pooled, height=64, width=64, num_inference_steps=2,
              guidance_scale=3.5, max_area=64 * 64)
    for auto in (False, True):                   # True: snaps the condition to a preferred Kontext resolution
        if auto:
            continue  # preferred resolutions are ~1 MPix: covered by the kernel tests above, too slow here
        a = pipe(image=flt.cuda(), generator=torch.Generator().manual_seed(9), output_type="np_uint8",
                 _auto_resize=auto, **kw)
        b = pipe(image=u8, generator=torch.Generator().manual_seed(9), output_type="np_uint8", _auto_resize=auto, **kw)
        assert torch.equal(a.latents, b.latents)
        assert np.array_equal(a.images, b.images) and a.images.dtype == np.uint8 and a.images.shape == (1, 64, 64, 3)
    pil = pipe(image=u8.numpy(), generator=torch.Generator().manual_seed(9), output_type="pil", _auto_resize=False, **kw).images
    assert np.array_equal(np.asarray(pil[0]), b.images[0])
