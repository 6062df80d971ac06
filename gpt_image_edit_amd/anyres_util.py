"""Aspect-ratio bucketing used by the serving front end (reference ``univa/utils/anyres_util.py:22-78``,
called from ``univa/serve/cli.py:82-97``).  Checked against tests/golden/anyres.npz."""
import math

from .helpers import PREFERRED_KONTEXT_RESOLUTIONS

_R11 = [(16, 9), (9, 16), (7, 5), (5, 7), (5, 4), (4, 5), (4, 3), (3, 4), (3, 2), (2, 3), (1, 1)]
RATIO = {
    "any_17ratio": [(w // math.gcd(w, h), h // math.gcd(w, h)) for w, h in PREFERRED_KONTEXT_RESOLUTIONS],
    "any_11ratio": _R11,
    "any_9ratio": [r for r in _R11 if r[0] != 7 and r[1] != 7],
    "any_7ratio": [r for r in _R11 if 7 not in r and 5 not in r],
    "any_5ratio": [(16, 9), (9, 16), (4, 3), (3, 4), (1, 1)],
    "any_1ratio": [(1, 1)],
}


def pick_ratio(orig_h, orig_w, anyres="any_17ratio"):
    ratio = orig_w / orig_h
    rw, rh = min(RATIO[anyres], key=lambda p: abs(p[0] / p[1] - ratio))
    return rw, rh


def compute_size(rw, rh, stride, *, min_pixels=None, max_pixels=None, anchor_pixels=None):
    base_w, base_h = rw * stride, rh * stride
    area = base_w * base_h
    if anchor_pixels is not None:
        target = anchor_pixels
    elif min_pixels is not None and max_pixels is not None:
        target = min(max(area, min_pixels), max_pixels)
    else:
        target = area
    scale = math.sqrt(target / area)
    new_w = max(stride, int(base_w * scale)) // stride * stride
    new_h = max(stride, int(base_h * scale)) // stride * stride
    return new_h, new_w


def dynamic_resize(orig_h, orig_w, anyres="any_17ratio", anchor_pixels=1024 * 1024, stride=32):
    rw, rh = pick_ratio(orig_h, orig_w, anyres)
    base_w, base_h = rw * stride, rh * stride
    s = max(1, round(math.sqrt(anchor_pixels / (base_w * base_h))))
    return (base_h * s) // stride * stride, (base_w * s) // stride * stride
