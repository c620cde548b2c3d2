"""
The eye-tracking overlay gate: write the eyes of the foreground (the projection's target video)
over the synthesized background only where both show a face in about the same place and the two
eye regions look alike.

`compute_eye_tracking_overlay` keeps the reference's signature and per-frame decision
(gance/overlay/overlay_eye_tracking.py:40-147): closest pair of eye boxes, `box_flag = distance <
min_bbox_distance`, `overlay_flag = box_flag and phash distance <= min_phash_distance`. What
differs:

* the perceptual hashes are computed on the GPU for all gated frames at once
  (gance_phash_crops_u8) instead of PIL + imagehash per frame;
* the landmark detector is a parameter. The reference hard-wires `faces.FaceFinderProxy`
  (face_recognition / dlib, not installed here and not part of the path being accelerated);
  without an explicit `face_finder` the default factory tries `face_recognition` and raises
  NotImplementedError when it is missing.
"""

from typing import Callable, Dict, Iterable, List, Optional, Protocol, Tuple

import numpy as np
import torch

from gance_amd import hip_lib
from gance_amd.gance_types import RGBInt8ImageType
from gance_amd.logger_common import LOGGER
from gance_amd.overlay.overlay_common import (
    BoundingBox,
    DistanceBoxes,
    OverlayContext,
    OverlayResult,
    bounding_box_distance,
    landmarks_to_bounding_boxes,
)

Landmarks = List[Dict[str, Tuple[Tuple[int, int], ...]]]


class FaceFinder(Protocol):  # pylint: disable=too-few-public-methods
    """What the gate needs from a landmark detector (gance/faces.py: FaceFinderProxy.face_landmarks)."""

    def face_landmarks(self, face_image: RGBInt8ImageType) -> Landmarks:
        """One dict per face with at least `left_eye` and `right_eye` point tuples."""


def default_face_finder() -> FaceFinder:
    """The reference's detector, if its library is installed."""
    try:
        import face_recognition  # pylint: disable=import-outside-toplevel,import-error
    except ImportError as error:
        raise NotImplementedError(
            "the eye-tracking overlay needs a landmark detector: pass `face_finder`, or install face_recognition (dlib)"
        ) from error

    class _Finder:  # pylint: disable=too-few-public-methods
        @staticmethod
        def face_landmarks(face_image: RGBInt8ImageType) -> Landmarks:
            return face_recognition.face_landmarks(face_image)

    return _Finder()


FACE_FINDER_FACTORY: Callable[[], FaceFinder] = default_face_finder


def phash_distance(hash_a: int, hash_b: int) -> int:
    """Number of differing bits of two 64-bit perceptual hashes (imagehash's `a - b`)."""
    return bin(int(hash_a) ^ int(hash_b)).count("1")


def _as_device_frames(images: Iterable[RGBInt8ImageType]) -> torch.Tensor:
    if isinstance(images, torch.Tensor):
        return images.cuda().contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.stack(list(images)))).cuda()


_HOST_STAGING: Dict[Tuple[int, Tuple[int, ...]], torch.Tensor] = {}
# the staging buffers stay allocated (page-locked) for the life of the process: sized for the stream's chunks, not for a
# whole video handed to the one-shot form
HOST_STAGING_LIMIT_BYTES = 1 << 30


def _copy_to_host_staging(frames: torch.Tensor, slot: int) -> np.ndarray:
    """
    Starts the copy of device frames into a pinned host buffer kept per (slot, frame shape) and returns the numpy view
    of it; the caller synchronises the stream once for all copies. (`tensor.cpu()` lands in pageable memory: for two
    chunks of 64 frames of 1024 x 1024 that was most of the gate's time.)
    """
    if frames.device.type != "cuda":
        return frames.numpy()
    if frames.numel() * frames.element_size() > HOST_STAGING_LIMIT_BYTES:
        return frames.cpu().numpy()
    key = (slot, tuple(frames.shape[1:]))
    staging = _HOST_STAGING.get(key)
    if staging is None or staging.shape[0] < frames.shape[0]:
        staging = torch.empty(tuple(frames.shape), dtype=frames.dtype, pin_memory=True)
        _HOST_STAGING[key] = staging
    view = staging[: frames.shape[0]]
    view.copy_(frames, non_blocking=True)
    return view.numpy()


def compute_eye_tracking_overlay(  # pylint: disable=too-many-locals
    foreground_images: Iterable[RGBInt8ImageType],
    background_images: Iterable[RGBInt8ImageType],
    min_phash_distance: int,
    min_bbox_distance: float,
    skip_mask: Optional[List[bool]] = None,
    face_finder: Optional[FaceFinder] = None,
) -> OverlayResult:
    """
    Per frame: the foreground's eye boxes if the overlay should be written (else None), and the
    numbers behind that decision.
    :param foreground_images: frames drawn on top ([n, S, S, 3] uint8: an iterable of arrays, or
    a tensor, which stays on the GPU).
    :param background_images: frames underneath, same shape.
    :param min_phash_distance: largest perceptual-hash distance between the two eye regions for
    the overlay to be written.
    :param min_bbox_distance: the eye boxes' centres must be closer than this.
    :param skip_mask: frames flagged True are not examined.
    :param face_finder: landmark detector; default `FACE_FINDER_FACTORY()`.
    """
    finder = face_finder if face_finder is not None else FACE_FINDER_FACTORY()
    foreground = _as_device_frames(foreground_images)
    background = _as_device_frames(background_images)
    if foreground.shape != background.shape:
        raise ValueError("foreground and background frames must have the same shape")
    num_frames, side = int(foreground.shape[0]), int(foreground.shape[1])
    skips = list(skip_mask) if skip_mask is not None else [False] * num_frames

    # host: landmarks -> closest pair of boxes per frame (the detector is CPU code either way)
    foreground_host = _copy_to_host_staging(foreground, 0)
    background_host = _copy_to_host_staging(background, 1)
    if foreground.device.type == "cuda":
        torch.cuda.current_stream(foreground.device).synchronize()
    foreground_boxes: List[List[BoundingBox]] = [[] for _ in range(num_frames)]
    distance_boxes: List[Optional[DistanceBoxes]] = [None] * num_frames
    gated: List[int] = []
    for index in range(num_frames):
        if skips[index]:
            LOGGER.info(f"Skipping eye tracking overlay for frame #{index}")
            continue
        foreground_boxes[index] = landmarks_to_bounding_boxes(finder.face_landmarks(face_image=foreground_host[index]))
        background_boxes = landmarks_to_bounding_boxes(finder.face_landmarks(face_image=background_host[index]))
        distance_boxes[index] = bounding_box_distance(a_boxes=foreground_boxes[index], b_boxes=background_boxes)
        if distance_boxes[index] is not None and distance_boxes[index].distance < min_bbox_distance:
            gated.append(index)

    # GPU: the perceptual hashes of both eye regions of every frame that passed the box test
    stream = torch.cuda.current_stream(foreground.device).cuda_stream
    phash_by_frame: Dict[int, int] = {}
    if gated:
        a_crops = np.array([(i, *distance_boxes[i].a_box) for i in gated], dtype=np.int32)
        b_crops = np.array([(i, *distance_boxes[i].b_box) for i in gated], dtype=np.int32)
        a_hashes = hip_lib.phash_crops_device(foreground.data_ptr(), num_frames, side, a_crops, stream)
        b_hashes = hip_lib.phash_crops_device(background.data_ptr(), num_frames, side, b_crops, stream)
        phash_by_frame = {i: phash_distance(a, b) for i, a, b in zip(gated, a_hashes, b_hashes)}

    bbox_lists: List[Optional[List[BoundingBox]]] = []
    contexts: List[OverlayContext] = []
    for index in range(num_frames):
        if skips[index]:
            bbox_lists.append(None)
            contexts.append(OverlayContext())
            continue
        bbox_phash_dist = phash_by_frame.get(index)
        overlay_flag = bbox_phash_dist is not None and bbox_phash_dist <= min_phash_distance
        LOGGER.info(f"Computed eye tracking overlay for frame #{index}, content? {overlay_flag}")
        bbox_lists.append(foreground_boxes[index] if overlay_flag else None)
        contexts.append(
            OverlayContext(
                bbox_perceptual_hash_distance=bbox_phash_dist,
                bbox_distance=distance_boxes[index].distance if distance_boxes[index] else None,
                overlay_written=overlay_flag,
            )
        )
    return OverlayResult(bbox_lists=iter(bbox_lists), contexts=iter(contexts))
