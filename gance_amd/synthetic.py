"""
Deterministic synthetic inputs for the benchmark and the parity tests (there is no network for
datasets or checkpoints): the synthetic WAV of BASELINE.md §5 config 3 and the synthetic projected
latents. Pure numpy; no GPU.
"""

from typing import Tuple

import numpy as np


def synthetic_audio(num_frames: int, vector_length: int = 512, seed: int = 7, frames_per_second: float = 60.0) -> np.ndarray:
    """
    `num_frames * vector_length` float32 samples at `vector_length * fps` Hz (30 720 Hz at the
    defaults, so the reference's resampy step is the identity):
        0.5 * chirp(100 Hz -> 8 kHz, linear) * (0.55 + 0.45 sin(2 pi 2 t)) + 0.05 * N(0,1)
    No 510-sample window is silent, so the dB spectrogram has no -inf (apply_spectrogram.py:81).
    """
    count = num_frames * vector_length
    sample_rate = vector_length * frames_per_second
    t = np.arange(count, dtype=np.float64) / sample_rate
    duration = count / sample_rate
    f0, f1 = 100.0, 8000.0
    phase = 2.0 * np.pi * (f0 * t + 0.5 * (f1 - f0) / duration * t * t)
    envelope = 0.55 + 0.45 * np.sin(2.0 * np.pi * 2.0 * t)
    noise = np.random.RandomState(seed).randn(count)
    return (0.5 * np.cos(phase) * envelope + 0.05 * noise).astype(np.float32)


def synthetic_final_latents(num_projection_frames: int, vector_length: int = 512, depth: int = 18, seed: int = 11) -> np.ndarray:
    """
    Concatenated final latents as `final_latents_matrices_label` yields them
    (gance/projection/projection_file_reader.py:280-284): (depth, F * L) float32, every row of a
    frame identical (the reference relies on that, visualization_inputs.py:220-231).
    """
    per_frame = np.random.RandomState(seed).randn(num_projection_frames, vector_length).astype(np.float32)
    return np.tile(per_frame.reshape(1, -1), (depth, 1))


def benchmark_blend_inputs(num_frames: int = 1800, vector_length: int = 512) -> Tuple[np.ndarray, np.ndarray]:
    """30 s @ 60 fps audio + 900 projected latents (30 fps projection)."""
    return synthetic_audio(num_frames, vector_length), synthetic_final_latents(num_frames // 2, vector_length)
