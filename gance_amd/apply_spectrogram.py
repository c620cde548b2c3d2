"""
Spectrogram stages of the audio -> latent chain as stand-alone functions, on the GPU.

Same names, argument meaning, shapes and dtypes as gance/apply_spectrogram.py (line numbers cited per
function). Each function runs one `gance_vec_*` stage of libgance_hip (gance_amd/csrc/audio.hip); the hot
path does not call them one by one but runs the fused `gance_blend_run`. There is no CPU fallback.
"""

from typing import Optional, Tuple

import numpy as np

from gance_amd import hip_lib
from gance_amd.vector_sources import vector_sources_common
from gance_amd.vector_sources.vector_types import ConcatenatedVectors


def reshape_spectrogram_to_vectors(
    spectrogram_data: np.ndarray, vector_length: int, amplitude_range: Optional[Tuple[int, int]] = None
) -> ConcatenatedVectors:
    """
    (bins, frames) spectrogram -> concatenated vectors of `vector_length` (apply_spectrogram.py:20-46):
    every frame's bins are Fourier-resampled (scipy.signal.resample) to `vector_length`, then the whole
    1-D array is min-max scaled to `amplitude_range` if one is given.
    :raises ValueError: a non-finite value reaches the scaling (a silent window gives log10(0) = -inf).
    """
    swapped = np.ascontiguousarray(np.swapaxes(np.asarray(spectrogram_data, dtype=np.float64), 0, 1))
    scaled = hip_lib.vec_fourier_resample(swapped, vector_length).reshape(-1)
    if amplitude_range is not None:
        scaled = hip_lib.vec_minmax_scale(scaled, amplitude_range)
    return ConcatenatedVectors(scaled)


def compute_spectrogram(time_series_audio: np.ndarray, num_frequency_bins: int, truncate: bool = True) -> np.ndarray:
    """
    dB spectrogram of an audio stream (apply_spectrogram.py:49-82): stereo is averaged to mono; windows
    of `num_frequency_bins - 1 * 2` samples (operator precedence in the reference: bins - 2) every
    `num_frequency_bins` samples, periodic Hann, FFT, `20 log10(|X| / max |X|)` against the GLOBAL maximum.
    :param truncate: keep the first window // 2 bins (the form the path uses) or, with False, every bin of the two-sided
    spectrum (apply_spectrogram.py:75-78); the maximum is taken over the bins that are kept.
    :return: float64 (window // 2, frames), or (window, frames) with `truncate=False`.
    """
    audio = np.asarray(time_series_audio)
    if audio.ndim != 1:
        audio = np.mean(audio, axis=1)  # apply_spectrogram.py:63-66
    return hip_lib.vec_spectrogram(audio, num_frequency_bins, truncate=truncate)


def compute_spectrogram_smooth_scale(
    data: ConcatenatedVectors, vector_length: int, amplitude_range: Optional[Tuple[int, int]] = None
) -> ConcatenatedVectors:
    """
    Spectrogram -> vectors -> Savitzky-Golay (7, 3) across vectors -> (5, 3) within each vector
    (apply_spectrogram.py:85-118), stage by stage. `gance_blend_run` runs the same chain fused.
    """
    spectrogram = compute_spectrogram(data, vector_length)
    as_vectors = reshape_spectrogram_to_vectors(spectrogram, amplitude_range=amplitude_range, vector_length=vector_length)
    smoothed = vector_sources_common.smooth_across_vectors(as_vectors, vector_length, window_length=7, polyorder=3)
    return vector_sources_common.smooth_each_vector(data=smoothed, vector_length=vector_length, window_length=5, polyorder=3)
