"""
ORACLE tooling (test infrastructure, NOT product code; build container only).

Leaf-module stubs that let the audio -> latent half of the REFERENCE be imported from
/root/reference in this container, where scikit-image, librosa, cv2, ffmpeg, vidgear,
more_itertools and resampy are not installed. The stubs are OUR code;
they restate two third-party leaves whose arithmetic is pure indexing / one formula:

  * skimage.util.view_as_windows(a, (m,), step)  ==  sliding_window_view(a, m)[::step]
    (scikit-image, unpinned in requirements/prod.txt:26; call site apply_spectrogram.py:69)
  * librosa.feature.rms(y, frame_length, hop_length=512, center=False)
    == sqrt(mean(|frames|^2)) over 1 + (len - frame_length) // hop frames
    (librosa 0.8.1, requirements/prod.txt:10; call site vector_reduction.py:33-35). librosa's
    `util.frame` yields an F-ordered (frame_length, n_frames) view, so `np.abs(x)**2` keeps that
    order and the mean over axis 0 reduces along the contiguous axis; the stub reproduces the
    same layout so that numpy takes the same (pairwise) summation path in float32.

Every other stubbed module is an empty placeholder: none of them is touched by the arithmetic.
Nothing from /root/reference is copied; this file is only used by oracle/make_goldens.py and by
tests that are skipped when /root/reference is absent (it never travels to the GPU box).
"""

import sys
import types
from pathlib import Path

import numpy as np

REFERENCE_ROOT = Path("/root/reference")


def reference_available() -> bool:
    """True when the read-only reference tree is mounted (build container only)."""
    return (REFERENCE_ROOT / "gance" / "apply_spectrogram.py").exists()


def view_as_windows(arr_in: np.ndarray, window_shape, step: int = 1) -> np.ndarray:
    """skimage.util.view_as_windows for 1-D input."""
    width = window_shape[0] if isinstance(window_shape, tuple) else int(window_shape)
    return np.lib.stride_tricks.sliding_window_view(arr_in, width)[::step]


def rms(y=None, S=None, frame_length: int = 2048, hop_length: int = 512, center: bool = True, pad_mode: str = "reflect"):  # pylint: disable=invalid-name,unused-argument
    """librosa.feature.rms (0.8.1) for a 1-D signal with center=False."""
    if center or S is not None:
        raise NotImplementedError("only the call form the reference uses is restated")
    y = np.asarray(y)
    n_frames = 1 + (len(y) - frame_length) // hop_length
    # librosa.util.frame: shape (frame_length, n_frames), strides (itemsize, hop*itemsize)
    frames = np.lib.stride_tricks.as_strided(
        y, shape=(frame_length, n_frames), strides=(y.itemsize, hop_length * y.itemsize), writeable=False
    )
    power = np.mean(np.abs(frames) ** 2, axis=0, keepdims=True)
    return np.sqrt(power)


def _stub(name: str, **attributes) -> types.ModuleType:
    module = types.ModuleType(name)
    module.__dict__.update(attributes)
    sys.modules[name] = module
    return module


def install() -> None:
    """Seed sys.modules with the stubs and put the reference on sys.path (idempotent)."""
    if not reference_available():
        raise RuntimeError("/root/reference is not mounted: goldens can only be generated in the build container")
    sys.dont_write_bytecode = True
    if "skimage" not in sys.modules:
        _stub("skimage", util=_stub("skimage.util", view_as_windows=view_as_windows))
    if "librosa" not in sys.modules:
        _stub("librosa", feature=_stub("librosa.feature", rms=rms))
    if "cv2" not in sys.modules:
        cv2 = _stub("cv2")
        cv2.cv2 = cv2
        sys.modules["cv2.cv2"] = cv2
    if "ffmpeg" not in sys.modules:
        _stub("ffmpeg", nodes=_stub("ffmpeg.nodes", FilterableStream=object))
    if "vidgear" not in sys.modules:
        _stub("vidgear", gears=_stub("vidgear.gears", WriteGear=object))
    for name in ("more_itertools", "resampy", "imagehash", "face_recognition"):
        if name not in sys.modules:
            _stub(name)
    if "lz" not in sys.modules:
        _stub("lz", transposition=_stub("lz.transposition", transpose=lambda rows: zip(*rows)))
    if str(REFERENCE_ROOT) not in sys.path:
        sys.path.insert(0, str(REFERENCE_ROOT))
