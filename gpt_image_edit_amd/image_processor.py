"""Pixel pre/post-processing either side of the path.

``VaeImageProcessor`` restates the subset of ``diffusers.image_processor.VaeImageProcessor`` (diffusers 0.32.2, not
vendored by the reference) that ``FluxKontextPipeline.__call__`` uses -- ``get_default_height_width``, ``resize``,
``preprocess`` (reference ``univa/utils/flux_pipeline.py:960-972``) and ``postprocess`` (:1130) -- for the tensor
inputs every reference caller passes (``univa/serve/cli.py:99-116``).  Two fused HIP routes replace the torch ops
when the pixels are still uint8:

* ``pixels_to_latent_input``: uint8 NHWC -> normalised, nearest-resized NHWC bf16 (``fk_pixels_u8_to_nhwc_bf16``),
  i.e. ``cli.prepare_condition_images`` + ``resize`` + ``preprocess`` + ``.to(bf16)`` in one gather;
* ``postprocess(..., "pil" | "np_uint8")``: decoder output -> uint8 NHWC (``fk_image_to_u8_nhwc``).
"""
import numpy as np
import torch

from . import ops


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=16, do_resize=True, do_normalize=True):
        self.config = type("Config", (), dict(vae_scale_factor=vae_scale_factor, do_resize=do_resize,
                                              do_normalize=do_normalize, resample="lanczos"))()

    # diffusers: height/width of the (first) image, rounded DOWN to a multiple of vae_scale_factor
    def get_default_height_width(self, image, height=None, width=None):
        if height is None:
            height = image.height if hasattr(image, "height") else (image.shape[2] if torch.is_tensor(image) else image.shape[1])
        if width is None:
            width = image.width if hasattr(image, "width") else (image.shape[3] if torch.is_tensor(image) else image.shape[2])
        m = self.config.vae_scale_factor
        return height - height % m, width - width % m

    def resize(self, image, height, width):
        """tensor: ``F.interpolate(size=(h, w))`` (mode nearest); PIL: lanczos; numpy NHWC: through the tensor path."""
        if torch.is_tensor(image):
            return torch.nn.functional.interpolate(image, size=(height, width))
        if isinstance(image, np.ndarray):
            t = torch.from_numpy(image).permute(0, 3, 1, 2)
            return torch.nn.functional.interpolate(t, size=(height, width)).permute(0, 2, 3, 1).numpy()
        from PIL import Image
        return image.resize((width, height), resample=Image.LANCZOS)

    def preprocess(self, image, height=None, width=None):
        """tensor [N,C,H,W] (or [C,H,W]): optional resize, then ``2x - 1`` iff no value is negative -- an input that
        is already in [-1, 1] is passed through (diffusers warns and skips the normalisation)."""
        if not torch.is_tensor(image):
            raise NotImplementedError("pass tensors (cli.py:99-116) or use pixels_to_latent_input for uint8 pixels")
        if image.dim() == 3:
            image = image.unsqueeze(0)
        if self.config.do_resize and height is not None and width is not None and tuple(image.shape[2:]) != (height, width):
            image = self.resize(image, height, width)
        if self.config.do_normalize and image.min() >= 0:
            image = 2.0 * image - 1.0
        return image

    @staticmethod
    def postprocess(image, output_type="pil"):
        """denormalise + clamp; 'pt' tensor, 'np' float NHWC, 'np_uint8' uint8 NHWC (HIP), 'pil' list of images."""
        if output_type in ("pt_raw", "latent"):
            return image
        if output_type == "pt":
            return (image / 2 + 0.5).clamp(0, 1)
        if output_type == "np":
            return (image / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).float().numpy()
        if output_type not in ("pil", "np_uint8"):
            raise ValueError(f"unknown output_type {output_type!r}")
        arr = ops.image_to_u8(image.contiguous()).cpu().numpy()
        if output_type == "np_uint8":
            return arr
        from PIL import Image
        return [Image.fromarray(a) for a in arr]


def as_uint8_nhwc(image):
    """uint8 pixels in any of the host forms (PIL image / list of PIL images / numpy / tensor) -> uint8 [N,H,W,3]
    tensor, or None if ``image`` is not uint8 pixels."""
    if hasattr(image, "convert"):  # a PIL image
        image = [image]
    if isinstance(image, (list, tuple)) and image and hasattr(image[0], "convert"):
        arrs = [np.asarray(im.convert("RGB"), dtype=np.uint8) for im in image]
        if any(a.shape != arrs[0].shape for a in arrs):
            raise ValueError("condition images must share one size (the reference stacks them, cli.py:112)")
        return torch.from_numpy(np.stack(arrs))
    if isinstance(image, np.ndarray) and image.dtype == np.uint8:
        image = torch.from_numpy(image)
    if torch.is_tensor(image) and image.dtype == torch.uint8:
        if image.dim() == 3:
            image = image.unsqueeze(0)
        if image.dim() != 4 or image.shape[3] != 3:
            raise ValueError("uint8 pixels must be [N, H, W, 3]")
        return image
    return None


class NhwcPixels:
    """Marker for the fused pixel route: ``tensor`` is the VAE encoder's input already in its internal layout
    (NHWC bf16, channels zero-padded), produced by :func:`pixels_to_latent_input`.  An explicit type instead of a
    shape test: a bf16 NCHW tensor whose last dimension happens to be 32 (pre-encoded latents of a 256-px-wide
    image) must not be mistaken for it."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.shape = tensor.shape

    def to(self, *args, **kwargs):
        return NhwcPixels(self.tensor.to(*args, **kwargs))


def nearest_source_index(n_out, n_in):
    """Source index of ``F.interpolate(mode="nearest")`` for every output position: min(floor(i * in / out), in-1)."""
    idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * (float(n_in) / float(n_out))).to(torch.int64)
    return idx.clamp_(max=n_in - 1)


def pixels_to_latent_input(u8, height, width, device, cpad=32):
    """uint8 [N,H,W,3] -> NHWC bf16 [N,height,width,cpad] exactly as the reference's float route would produce it
    (normalise in fp32, nearest resize, renormalise iff nothing is negative, cast to bf16).

    The reference decides the renormalisation on the RESIZED tensor (``preprocess`` runs after ``resize``,
    flux_pipeline.py:960-972), so the minimum is taken over the pixels the nearest resize actually samples."""
    rows = nearest_source_index(height, u8.shape[1]).to(u8.device)
    cols = nearest_source_index(width, u8.shape[2]).to(u8.device)
    sampled_min = u8.index_select(1, rows).index_select(2, cols).min()
    renorm = bool(sampled_min >= 128)  # (u/255 - 0.5)/0.5 >= 0 everywhere <=> u >= 128 (127.5 is not a uint8)
    return NhwcPixels(ops.pixels_to_nhwc(u8.to(device).contiguous(), height, width, cpad, renorm))
