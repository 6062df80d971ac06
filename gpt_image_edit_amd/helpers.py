"""Host-side index helpers of the FLUX-Kontext pipeline driver (product code, runs on any device).

Same behaviour as the static helpers of the reference's ``FluxKontextPipeline``
(``univa/utils/flux_pipeline.py:106-116, 561-598``) -- training code calls them directly
(``train_denoiser.py:925,1009,1021,1098``) so the names and argument orders are kept.
Checked against golden vectors produced by the reference itself (tests/golden/helpers.npz).
"""
import torch

# (width, height) pairs, univa/utils/flux_pipeline.py:85-103
PREFERRED_KONTEXT_RESOLUTIONS = [
    (672, 1568), (688, 1504), (720, 1456), (752, 1392), (800, 1328), (832, 1248), (880, 1184),
    (944, 1104), (1024, 1024), (1104, 944), (1184, 880), (1248, 832), (1328, 800), (1392, 752),
    (1456, 720), (1504, 688), (1568, 672),
]


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


def _prepare_latent_image_ids(batch_size, height, width, device, dtype):
    ids = torch.zeros(height, width, 3)
    ids[..., 1] += torch.arange(height)[:, None]
    ids[..., 2] += torch.arange(width)[None, :]
    return ids.reshape(height * width, 3).to(device=device, dtype=dtype)


def _pack_latents(latents, batch_size, num_channels_latents, height, width):
    x = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2)
    x = x.permute(0, 2, 4, 1, 3, 5)
    return x.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)


def _unpack_latents(latents, height, width, vae_scale_factor):
    batch_size, _, channels = latents.shape
    height = 2 * (int(height) // (vae_scale_factor * 2))
    width = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(batch_size, height // 2, width // 2, channels // 4, 2, 2)
    x = x.permute(0, 3, 1, 4, 2, 5)
    return x.reshape(batch_size, channels // 4, height, width)


def fit_to_max_area(height, width, max_area, multiple_of):
    """Resolution fix-up at the top of the pipeline call (flux_pipeline.py:877-885)."""
    aspect = width / height
    w = round((max_area * aspect) ** 0.5)
    h = round((max_area / aspect) ** 0.5)
    return h // multiple_of * multiple_of, w // multiple_of * multiple_of


def preferred_condition_size(image_height, image_width, multiple_of, auto_resize=True):
    """Condition-image size selection (flux_pipeline.py:960-970)."""
    if auto_resize:
        aspect = image_width / image_height
        _, image_width, image_height = min((abs(aspect - w / h), w, h) for w, h in PREFERRED_KONTEXT_RESOLUTIONS)
    return image_height // multiple_of * multiple_of, image_width // multiple_of * multiple_of
