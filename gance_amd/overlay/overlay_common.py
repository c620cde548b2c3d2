"""
Bounding-box helpers and the overlay write of the eye-tracking gate.

Same names, argument meaning and results as gance/overlay/overlay_common.py (`BoundingBox` :19-27,
`convert_to_pil_box` :30-43, `landmarks_to_bounding_boxes` :46-57, `bounding_box_center` :60-66,
`DistanceBoxes` + `bounding_box_distance` :69-101, `write_boxes_onto_image` :154-172,
`OverlayResult` :175-186). Differences by design:

* `write_boxes_onto_image` runs as a HIP kernel (gance_overlay_boxes_u8) and has a batched,
  device-resident twin, `write_boxes_onto_frames_device`; the reference draws a PIL polygon mask
  and composites per frame on the host.
* `landmarks_to_bounding_boxes` restates cv2.boundingRect for integer points (min corner,
  extent + 1) instead of calling OpenCV.
* `OverlayContext` lives here: its home in the reference, overlay_visualization.py, is a
  matplotlib module that is out of scope.
"""

import itertools
import math
from typing import Dict, Iterator, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from gance_amd import hip_lib
from gance_amd.gance_types import RGBInt8ImageType


class BoundingBox(NamedTuple):
    """A rectangle as OpenCV reports it."""

    x: int
    y: int
    width: int
    height: int


class OverlayContext(NamedTuple):
    """Why an overlay was or was not written for a frame (overlay_visualization.py:20-34)."""

    overlay_written: bool = False
    image_perceptual_hash_distance: Optional[float] = None
    bbox_perceptual_hash_distance: Optional[float] = None
    bbox_distance: Optional[float] = None


def convert_to_pil_box(bounding_box: BoundingBox) -> Tuple[int, int, int, int]:
    """(left, upper, right, lower), the order PIL's crop takes."""
    return (
        bounding_box.x,
        bounding_box.y,
        bounding_box.x + bounding_box.width,
        bounding_box.y + bounding_box.height,
    )


def landmarks_to_bounding_boxes(landmarks: List[Dict[str, Tuple[Tuple[int, int], ...]]]) -> List[BoundingBox]:
    """One box per face around its left-eye and right-eye keypoints."""
    boxes = []
    for landmark in landmarks:
        points = np.array(tuple(landmark["left_eye"]) + tuple(landmark["right_eye"]), dtype=np.int64)
        x_min, y_min = points.min(axis=0)
        x_max, y_max = points.max(axis=0)
        boxes.append(BoundingBox(int(x_min), int(y_min), int(x_max - x_min + 1), int(y_max - y_min + 1)))
    return boxes


def bounding_box_center(bounding_box: BoundingBox) -> Tuple[float, float]:
    """(x, y) of the centre."""
    return (bounding_box.x + bounding_box.width / 2), (bounding_box.y + bounding_box.height / 2)


class DistanceBoxes(NamedTuple):
    """Distance in pixels between the centres of two boxes, and the boxes."""

    distance: float
    a_box: BoundingBox
    b_box: BoundingBox


def bounding_box_distance(a_boxes: List[BoundingBox], b_boxes: List[BoundingBox]) -> Optional[DistanceBoxes]:
    """The closest pair of boxes between the two sets (first minimum in product order); None if either is empty."""
    candidates = []
    for a_box, b_box in itertools.product(a_boxes, b_boxes):
        (a_x, a_y), (b_x, b_y) = bounding_box_center(a_box), bounding_box_center(b_box)
        candidates.append(DistanceBoxes(distance=float(math.sqrt((a_x - b_x) ** 2 + (a_y - b_y) ** 2)), a_box=a_box, b_box=b_box))
    return min(candidates, key=lambda distance_box: distance_box.distance, default=None)


def _boxes_array(frame_boxes: Sequence[Optional[Sequence[BoundingBox]]]) -> np.ndarray:
    """[(frame, x, y, w, h), ...] for the C ABI."""
    rows = [
        (frame_index, box[0], box[1], box[2], box[3])
        for frame_index, boxes in enumerate(frame_boxes)
        if boxes
        for box in boxes
    ]
    return np.array(rows, dtype=np.int32).reshape(-1, 5)


def write_boxes_onto_frames_device(
    foreground: torch.Tensor, background: torch.Tensor, frame_boxes: Sequence[Optional[Sequence[BoundingBox]]]
) -> torch.Tensor:
    """
    Batched overlay on frames resident in HBM: foreground / background [n, S, S, 3] uint8,
    `frame_boxes[i]` the boxes of frame i (None or empty: frame i stays the background).
    """
    if foreground.shape != background.shape or foreground.dim() != 4 or foreground.shape[1] != foreground.shape[2]:
        raise ValueError("foreground and background must both be [n, S, S, 3] uint8")
    out = torch.empty_like(background)
    hip_lib.overlay_boxes_device(
        foreground.contiguous().data_ptr(), background.contiguous().data_ptr(), out.data_ptr(), int(foreground.shape[0]),
        int(foreground.shape[1]), _boxes_array(frame_boxes), torch.cuda.current_stream(foreground.device).cuda_stream,
    )
    return out


def write_boxes_onto_image(
    foreground_image: RGBInt8ImageType, background_image: RGBInt8ImageType, bounding_boxes: List[BoundingBox]
) -> RGBInt8ImageType:
    """Regions of the foreground around `bounding_boxes` written over the background; a new image."""
    foreground = torch.from_numpy(np.ascontiguousarray(foreground_image)).cuda()[None]
    background = torch.from_numpy(np.ascontiguousarray(background_image)).cuda()[None]
    return write_boxes_onto_frames_device(foreground, background, [bounding_boxes])[0].cpu().numpy()


class OverlayResult(NamedTuple):
    """The output streams of an eye-tracking overlay computation, per frame."""

    bbox_lists: Iterator[Optional[List[BoundingBox]]]
    contexts: Iterator[OverlayContext]
