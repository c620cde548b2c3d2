"""
GPU parity tests of the overlay gate's pixel work (gance_phash_crops_u8, gance_overlay_boxes_u8,
through the C ABI) against the CPU oracle (PIL + scipy) and the golden composite captured from
the reference's `write_boxes_onto_image`.

Bars: the 32x32 thumbnail is integer work (PIL's fixed-point resample) and the hash is 64 bits:
bit-exact. (The DCT is float64 by direct summation here and FFT-based in scipy; a bit could only
differ for a coefficient within ~1e-10 of the median, which the smooth random test images avoid.)
"""

import numpy as np
import pytest
import torch

from gance_amd import hip_lib
from gance_amd.overlay import overlay_common, overlay_eye_tracking
from oracle import overlay_ref

pytestmark = pytest.mark.gpu


def textured_frames(num_frames: int, side: int, seed: int) -> np.ndarray:
    """Smooth-ish colour fields with detail at several scales: no flat crops, no hash ties."""
    rng = np.random.RandomState(seed)
    frames = np.zeros((num_frames, side, side, 3), dtype=np.float64)
    for cells in (4, 9, 23, 61):
        coarse = rng.rand(num_frames, cells, cells, 3)
        reps = -(-side // cells)
        frames += np.kron(coarse, np.ones((1, reps, reps, 1)))[:, :side, :side, :] / np.sqrt(cells)
    frames -= frames.min()
    return (frames / frames.max() * 255).astype(np.uint8)


@pytest.mark.parametrize("side,seed", [(256, 1), (1024, 2)])
def test_phash_of_crops_is_bit_exact(side: int, seed: int) -> None:
    frames = textured_frames(3, side, seed)
    rng = np.random.RandomState(seed + 10)
    crops = []
    for _ in range(24):
        w, h = int(rng.randint(5, side)), int(rng.randint(5, side // 2))
        crops.append((int(rng.randint(0, 3)), int(rng.randint(0, side - w + 1)), int(rng.randint(0, side - h + 1)), w, h))
    crops += [(0, 0, 0, side, side), (1, 3, 5, 32, 32), (2, 7, 9, 20, 12), (0, 0, 0, 33, 31)]  # whole frame, identity, upscale
    d_frames = torch.from_numpy(frames).cuda()
    got = hip_lib.phash_crops_device(d_frames.data_ptr(), 3, side, np.array(crops, dtype=np.int32))
    for crop, value in zip(crops, got):
        want = overlay_ref.phash(frames[crop[0]], crop[1:])
        assert int(value) == want, (crop, hex(int(value)), hex(want))


def test_phash_rejects_crops_that_leave_the_frame() -> None:
    d_frames = torch.zeros((1, 64, 64, 3), dtype=torch.uint8, device="cuda")
    for bad in [(0, 60, 0, 10, 10), (0, 0, 0, 0, 5), (1, 0, 0, 5, 5), (0, -1, 0, 5, 5)]:
        with pytest.raises(hip_lib.GanceHipError):
            hip_lib.phash_crops_device(d_frames.data_ptr(), 1, 64, np.array([bad], dtype=np.int32))
    assert hip_lib.phash_crops_device(d_frames.data_ptr(), 1, 64, np.zeros((0, 5), dtype=np.int32)).shape == (0,)


def test_overlay_write_matches_the_reference_composite(golden_dir) -> None:
    golden = np.load(golden_dir / "overlay.npz")
    boxes = [overlay_common.BoundingBox(*(int(v) for v in box)) for box in golden["composite_boxes"]]
    out = overlay_common.write_boxes_onto_image(golden["composite_fg"], golden["composite_bg"], boxes)
    assert np.array_equal(out, golden["composite_out"])


def test_batched_overlay_write_matches_oracle() -> None:
    rng = np.random.RandomState(4)
    side, n = 150, 5
    fg = rng.randint(0, 256, (n, side, side, 3)).astype(np.uint8)
    bg = rng.randint(0, 256, (n, side, side, 3)).astype(np.uint8)
    frame_boxes = [
        [overlay_common.BoundingBox(10, 20, 30, 11)],
        None,
        [overlay_common.BoundingBox(0, 0, 5, 4), overlay_common.BoundingBox(100, 120, 49, 29)],
        [],
        [overlay_common.BoundingBox(140, 3, 10, 7)],
    ]
    out = overlay_common.write_boxes_onto_frames_device(torch.from_numpy(fg).cuda(), torch.from_numpy(bg).cuda(), frame_boxes)
    out = out.cpu().numpy()
    for index in range(n):
        want = overlay_ref.write_boxes_onto_image(fg[index], bg[index], [tuple(b) for b in (frame_boxes[index] or [])])
        assert np.array_equal(out[index], want), index


class FakeFaceFinder:  # pylint: disable=too-few-public-methods
    """Deterministic stand-in for the dlib detector: eye landmarks keyed by a frame's first pixel."""

    def __init__(self, by_key):
        self.by_key = by_key

    def face_landmarks(self, face_image):
        return self.by_key.get(tuple(int(v) for v in face_image[0, 0]), [])


def eyes(x: int, y: int, w: int, h: int):
    return {"left_eye": ((x, y), (x + w // 3, y + h - 1)), "right_eye": ((x + w - 1, y), (x + 2 * w // 3, y + h - 1))}


def test_gate_decisions_follow_the_reference_logic() -> None:
    """Same face position + similar eyes -> overlay; far boxes, different eyes, skip flag or no face -> none."""
    side, n = 256, 6
    fg = textured_frames(n, side, 31)
    bg = textured_frames(n, side, 32)
    bg[0] = np.roll(fg[0], (1, 2), axis=(0, 1))     # same picture moved by (2, 1): equal crops at boxes (2, 1) apart
    bg[1] = np.clip(fg[1].astype(int) + 6, 0, 255)  # nearly identical
    bg[4] = fg[4]
    for index in range(n):                          # tag frames so the fake finder can tell them apart
        fg[index, 0, 0] = (index, 0, 1)
        bg[index, 0, 0] = (index, 0, 2)
    finder = FakeFaceFinder({
        (0, 0, 1): [eyes(60, 80, 90, 30)], (0, 0, 2): [eyes(62, 81, 90, 30)],
        (1, 0, 1): [eyes(60, 80, 90, 30)], (1, 0, 2): [eyes(60, 80, 90, 30)],
        (2, 0, 1): [eyes(60, 80, 90, 30)], (2, 0, 2): [eyes(150, 200, 90, 30)],   # too far apart
        (3, 0, 1): [eyes(60, 80, 90, 30)], (3, 0, 2): [eyes(60, 80, 90, 30)],     # unrelated pictures
        (4, 0, 1): [eyes(60, 80, 90, 30)], (4, 0, 2): [eyes(60, 80, 90, 30)],     # would pass, but skipped
        (5, 0, 1): [],                                                            # no face in the foreground
    })
    result = overlay_eye_tracking.compute_eye_tracking_overlay(
        fg, bg, min_phash_distance=10, min_bbox_distance=20.0, skip_mask=[False, False, False, False, True, False], face_finder=finder
    )
    boxes = list(result.bbox_lists)
    contexts = list(result.contexts)
    assert [b is not None for b in boxes] == [True, True, False, False, False, False]
    assert boxes[0] == overlay_common.landmarks_to_bounding_boxes(finder.by_key[(0, 0, 1)])
    assert contexts[0].overlay_written and contexts[0].bbox_perceptual_hash_distance == 0
    assert contexts[0].bbox_distance == pytest.approx(np.sqrt(5.0))
    assert contexts[2].bbox_perceptual_hash_distance is None and contexts[2].bbox_distance > 20.0
    assert contexts[3].bbox_perceptual_hash_distance > 10 and not contexts[3].overlay_written
    assert contexts[4] == overlay_common.OverlayContext() and contexts[5].bbox_distance is None
    # the hash distances are the oracle's
    for index in (0, 1, 3):
        a_box = overlay_common.landmarks_to_bounding_boxes(finder.by_key[(index, 0, 1)])[0]
        b_box = overlay_common.landmarks_to_bounding_boxes(finder.by_key[(index, 0, 2)])[0]
        want = overlay_ref.phash_distance(overlay_ref.phash(fg[index], tuple(a_box)), overlay_ref.phash(bg[index], tuple(b_box)))
        assert contexts[index].bbox_perceptual_hash_distance == want


def test_default_face_finder_fails_loudly_without_its_library() -> None:
    with pytest.raises(NotImplementedError, match="landmark detector"):
        overlay_eye_tracking.compute_eye_tracking_overlay(np.zeros((1, 8, 8, 3), np.uint8), np.zeros((1, 8, 8, 3), np.uint8), 1, 1.0)
