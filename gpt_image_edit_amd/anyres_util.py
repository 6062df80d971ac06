"""Aspect-ratio bucketing used by the serving front end (reference ``univa/utils/anyres_util.py:22-78``,
called from ``univa/serve/cli.py:82-97``).  Checked against tests/golden/anyres.npz.

A request names a bucket family (``any_<n>ratio``); the image's width : height picks the nearest reduced ratio of the
family, and the output size is that ratio times ``stride`` scaled to an area budget and snapped down to the stride.
"""
import math

from .helpers import PREFERRED_KONTEXT_RESOLUTIONS


def _reduced(w, h):
    g = math.gcd(w, h)
    return w // g, h // g


def _families():
    # the 11-ratio family, landscape / portrait pairs from widest to square; smaller families drop its odd members
    eleven = []
    for a, b in ((16, 9), (7, 5), (5, 4), (4, 3), (3, 2)):
        eleven += [(a, b), (b, a)]
    eleven.append((1, 1))
    without = lambda banned: [r for r in eleven if not (set(r) & banned)]  # noqa: E731
    return {
        "any_17ratio": [_reduced(w, h) for w, h in PREFERRED_KONTEXT_RESOLUTIONS],
        "any_11ratio": eleven,
        "any_9ratio": without({7}),
        "any_7ratio": without({7, 5}),
        "any_5ratio": without({7, 5, 2}),
        "any_1ratio": [(1, 1)],
    }


RATIO = _families()


def pick_ratio(orig_h, orig_w, anyres="any_17ratio"):
    """(rw, rh) of the family member closest to orig_w / orig_h (first one wins a tie)."""
    want = orig_w / orig_h
    best, best_err = None, None
    for rw, rh in RATIO[anyres]:
        err = abs(rw / rh - want)
        if best_err is None or err < best_err:
            best, best_err = (rw, rh), err
    return best


def _snap(length, stride):
    return length // stride * stride


def compute_size(rw, rh, stride, *, min_pixels=None, max_pixels=None, anchor_pixels=None):
    """(h, w) of ratio rw : rh scaled to ``anchor_pixels`` -- or clamped into [min_pixels, max_pixels] -- with each side
    truncated to an integer, kept at least one stride long and snapped down to the stride."""
    w0, h0 = rw * stride, rh * stride
    budget = w0 * h0
    if anchor_pixels is not None:
        budget = anchor_pixels
    elif min_pixels is not None and max_pixels is not None:
        budget = min(max(budget, min_pixels), max_pixels)
    k = math.sqrt(budget / (w0 * h0))
    return tuple(_snap(max(stride, int(side * k)), stride) for side in (h0, w0))


def dynamic_resize(orig_h, orig_w, anyres="any_17ratio", anchor_pixels=1024 * 1024, stride=32):
    """(h, w): the bucket ratio times stride, times the INTEGER factor that brings its area closest to anchor_pixels."""
    rw, rh = pick_ratio(orig_h, orig_w, anyres)
    w0, h0 = rw * stride, rh * stride
    k = max(1, round(math.sqrt(anchor_pixels / (w0 * h0))))
    return _snap(h0 * k, stride), _snap(w0 * k, stride)
