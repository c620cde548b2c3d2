"""
ORACLE (test infrastructure, NOT product code): numpy restatement of the bicubic resize this
implementation defines for the post-synthesis step.

PARITY UNPINNED against the reference: the reference calls cv2.resize(..., INTER_CUBIC)
(gance/image_sources/video_common.py:416-418); OpenCV (third party, opencv-python pinned in
requirements/prod.txt:17) is not installed here and its uint8 path uses 11-bit fixed-point
coefficients. What is restated is the published filter: Keys cubic convolution with a = -0.75,
half-pixel-centre coordinates (d + 0.5) * src / dst - 0.5, replicated border, then round half up
and saturate to uint8. Computed here in float64.
"""

import numpy as np


def _weights(t: np.ndarray) -> np.ndarray:
    a = -0.75
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return np.stack([w0, w1, w2, 1 - w0 - w1 - w2], axis=-1)


def resize_bicubic_u8(images: np.ndarray, dst_side: int, dtype=np.float64) -> np.ndarray:
    """images [B, S, S, 3] uint8 -> [B, D, D, 3] uint8."""
    src = images.shape[1]
    scale = dtype(src) / dtype(dst_side)
    pos = (np.arange(dst_side, dtype=dtype) + dtype(0.5)) * scale - dtype(0.5)
    base = np.floor(pos).astype(np.int64)
    w = _weights((pos - base).astype(dtype))  # [D, 4]
    idx = np.clip(base[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)  # [D, 4]
    x = images.astype(dtype)
    rows = np.einsum("bhdtc,dt->bhdc", x[:, :, idx, :], w)  # horizontal: [B, S, D, 3]
    out = np.einsum("bdthc,dt->bdhc", rows[:, idx, :, :], w)  # vertical: [B, D, D, 3]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
