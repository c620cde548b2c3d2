"""
ORACLE (test infrastructure, NOT product code): CPU restatement of the eye-tracking overlay
gate's arithmetic.

* `phash`: imagehash.phash as called by gance/overlay/overlay_eye_tracking.py:100-108. imagehash
  is a third-party dependency that is neither vendored nor installed here (unpinned in
  requirements/prod.txt), so its published algorithm is restated on the same library calls it
  makes: PIL convert("L") + resize((32, 32), LANCZOS), scipy.fftpack.dct over both axes, the
  top-left 8x8 block compared with its median. PARITY UNPINNED against imagehash itself; pinned
  against PIL / scipy, which do all the arithmetic.
* `draw_mask_bounds` / `write_boxes_onto_image`: overlay_common.py:104-172, pinned by goldens
  captured from the reference's own functions (oracle/make_goldens.py, tests/golden/overlay.npz).
* `bounding_box_distance`: overlay_common.py:73-101; `track_length_filter`:
  vector_reduction.py:261-273; both pinned by the same golden file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from typing import List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image
from scipy import fftpack

Box = Tuple[int, int, int, int]  # x, y, width, height


def phash_bits(image_rgb: np.ndarray, box: Box) -> np.ndarray:
    """(8, 8) bool: DCT low-frequency block of the 32x32 luma thumbnail of the crop > its median."""
    x, y, w, h = box
    crop = Image.fromarray(image_rgb).crop((x, y, x + w, y + h))
    small = crop.convert("L").resize((32, 32), Image.LANCZOS)
    pixels = np.asarray(small)
    dct = fftpack.dct(fftpack.dct(pixels, axis=0), axis=1)
    low = dct[:8, :8]
    return low > np.median(low)


def phash(image_rgb: np.ndarray, box: Box) -> int:
    """The hash as the integer whose hex string imagehash prints (first bit most significant)."""
    value = 0
    for bit in phash_bits(image_rgb, box).reshape(-1):
        value = (value << 1) | int(bit)
    return value


def phash_distance(hash_a: int, hash_b: int) -> int:
    """imagehash's `a - b`: the number of differing bits."""
    return bin(int(hash_a) ^ int(hash_b)).count("1")


def draw_mask_bounds(side: int, box: Box) -> Tuple[int, int, int, int]:
    """
    Inclusive (x_left, y_upper, x_right, y_lower) of the rectangle `_draw_mask` paints for one
    box: pads 0.098 / 0.058 of the side, float corners truncated toward zero by PIL, edges drawn.
    """
    x, y, w, h = box
    y_pad = side * 0.058
    x_pad = side * 0.098
    y_center = y + (h / 2)
    return int(x - x_pad), int(y_center - y_pad), int(x + (w + x_pad)), int(y_center + y_pad)


def write_boxes_onto_image(foreground: np.ndarray, background: np.ndarray, boxes: Sequence[Box]) -> np.ndarray:
    """Foreground inside the mask rectangles, background elsewhere (square frames)."""
    side = foreground.shape[0]
    out = background.copy()
    for box in boxes:
        left, upper, right, lower = draw_mask_bounds(side, box)
        left, upper = max(left, 0), max(upper, 0)
        right, lower = min(right, side - 1), min(lower, side - 1)
        if right >= left and lower >= upper:
            out[upper : lower + 1, left : right + 1] = foreground[upper : lower + 1, left : right + 1]
    return out


def bounding_box_distance(a_boxes: Sequence[Box], b_boxes: Sequence[Box]) -> Optional[Tuple[float, Box, Box]]:
    """Minimum centre-to-centre distance over all pairs, first minimum in product order; None if a side is empty."""
    best: Optional[Tuple[float, Box, Box]] = None
    for a_box in a_boxes:
        for b_box in b_boxes:
            ax, ay = a_box[0] + a_box[2] / 2, a_box[1] + a_box[3] / 2
            bx, by = b_box[0] + b_box[2] / 2, b_box[1] + b_box[3] / 2
            dist = float(np.sqrt((ax - bx) ** 2 + (ay - by) ** 2))
            if best is None or dist < best[0]:
                best = (dist, tuple(a_box), tuple(b_box))
    return best


def track_length_filter(bool_tracks: Sequence[bool], track_length: int) -> List[bool]:
    """Runs of True shorter than `track_length` become False."""
    flags = [bool(v) for v in bool_tracks]
    out = [False] * len(flags)
    start = 0
    while start < len(flags):
        stop = start
        while stop < len(flags) and flags[stop] == flags[start]:
            stop += 1
        if flags[start] and stop - start >= track_length:
            out[start:stop] = [True] * (stop - start)
        start = stop
    return out
