"""
Read WAV files and fit them to a video: the input stage of the path.

Same names, modes, error behaviour and output contract as gance/vector_sources/music.py
(`read_wav_file` :172-209, `read_wavs_scale_for_video` :60-169, `_scale_wav_to_sample_rate`
:212-230). Two deliberate differences:

* The integer -> [-1, 1] remap is one vectorised linear map (the same `interp1d` line,
  slope * (x - lo) + out_lo) instead of a `multiprocessing.Pool().map` over every sample
  (vector_sources_common.py:59-61), which is seconds of pure overhead on a 30 s file.
* Time-stretching is `resampy.resample` (0.2.2, kaiser_best), a third-party dependency that is not installed
  here: its published algorithm is restated operation for operation as a HIP kernel
  (gance_amd/csrc/resample.hip: the 32 769-entry filter table, linear interpolation between entries, the
  running-sum time register, accumulation in the signal's dtype) and pinned by the reference's own known answer
  on this path (test/test_dynamic_model_switching.py:15-39, reproduced in tests/test_music_gpu.py). As in
  resampy, a ratio of exactly 1 still passes the samples through the low-pass filter.
"""

import pickle
from pathlib import Path
from typing import List, NamedTuple, Optional, Union

import numpy as np
import torch
from scipy.io import wavfile

from gance_amd import hip_lib
from gance_amd.logger_common import LOGGER
from gance_amd.vector_sources.vector_sources_common import pad_array

WavDataType = Union["np.ndarray[np.int16]", "np.ndarray[np.float32]", "np.ndarray[np.int32]"]

# integer PCM ranges the reference remaps from (music.py:191-196; its int8 branch is kept as is)
_INTEGER_RANGES = {
    np.dtype(np.int32): (-2147483648, 2147483647),
    np.dtype(np.int16): (-32768, 32767),
    np.dtype(np.int8): (0, 255),
}


class WavFileProperties(NamedTuple):
    """Sample rate, samples and a name (music.py:19-33)."""

    sample_rate: int
    wav_data: WavDataType
    name: str


def read_wav_file(wav_path: Path, convert_to_32bit_float: bool = True) -> WavFileProperties:
    """
    Read a `.wav`; integer PCM is mapped linearly onto [-1, 1] float32.
    :raises ValueError: for sample formats the reference does not convert either.
    """
    sample_rate, wav_data = wavfile.read(str(wav_path))
    if convert_to_32bit_float and wav_data.dtype != np.float32:
        if wav_data.dtype not in _INTEGER_RANGES:
            raise ValueError(
                f"Cannot safely convert wav data to np.float32, unknown input format: {wav_data.dtype}"
            )
        lo, hi = _INTEGER_RANGES[wav_data.dtype]
        slope = (1.0 - (-1.0)) / (float(hi) - float(lo))
        wav_data = (slope * (wav_data.astype(np.float64) - lo) + (-1.0)).astype(np.float32)
    return WavFileProperties(sample_rate=int(sample_rate), wav_data=wav_data, name=wav_path.with_suffix("").name)


def resample_audio(samples: np.ndarray, sr_orig: float, sr_new: float) -> np.ndarray:
    """
    `resampy.resample(samples, sr_orig, sr_new)` (kaiser_best) of a 1-D signal on the GPU
    (gance_resample_audio_f32 / _f64). Output length int(len(samples) * sr_new / sr_orig), output dtype = input
    dtype for float32 / float64 (the accumulation runs in it, as resampy's does); other dtypes go through float32.
    :raises ValueError: non-positive sample rates, or a signal too short to give one output sample (resampy's checks).
    """
    if sr_orig <= 0:
        raise ValueError(f"Invalid sample rate: sr_orig={sr_orig}")
    if sr_new <= 0:
        raise ValueError(f"Invalid sample rate: sr_new={sr_new}")
    ratio = float(sr_new) / sr_orig
    count = int(samples.shape[0] * ratio)
    if count < 1:
        raise ValueError(f"Input signal length={samples.shape[0]} is too small to resample from {sr_orig}->{sr_new}")
    double_precision = samples.dtype == np.float64
    dtype = np.float64 if double_precision else np.float32
    d_in = torch.from_numpy(np.ascontiguousarray(samples, dtype=dtype)).cuda()
    d_out = torch.empty((count,), dtype=d_in.dtype, device=d_in.device)
    hip_lib.resample_audio_device(
        d_in.data_ptr(), int(d_in.shape[0]), sr_orig, sr_new, d_out.data_ptr(), count,
        torch.cuda.current_stream().cuda_stream, double_precision=double_precision,
    )
    return d_out.cpu().numpy()


def _scale_wav_to_sample_rate(wav_file: WavFileProperties, new_sample_rate: float) -> WavFileProperties:
    """Speed the audio up or slow it down; the result keeps the INPUT sample rate (music.py:212-230)."""
    return WavFileProperties(
        wav_data=resample_audio(wav_file.wav_data, wav_file.sample_rate, new_sample_rate),
        sample_rate=wav_file.sample_rate,
        name=f"{wav_file.name}_scaled",
    )


def read_wavs_scale_for_video(  # pylint: disable=too-many-arguments
    wavs: Union[List[Path], List[WavFileProperties]],
    vector_length: int,
    frames_per_second: Optional[float] = None,
    target_num_vectors: Optional[int] = None,
    cache_path: Optional[Path] = None,
    pad_to_length: bool = True,
) -> WavFileProperties:
    """
    Read several WAVs, mix each to mono, concatenate, and stretch in time so that there is one
    `vector_length` vector per video frame (FPS mode) or exactly `target_num_vectors` vectors
    (target mode); zero-pad to a multiple of `vector_length`.
    :raises ValueError: both or neither mode given; differing sample rates.
    """
    if frames_per_second is not None and target_num_vectors is not None:
        raise ValueError("Can't use both FPS mode and target vector count mode.")
    if frames_per_second is None and target_num_vectors is None:
        raise ValueError("Need to use FPS mode or target vector count mode.")
    if cache_path is not None and cache_path.exists():
        LOGGER.info("Cached audio found. Loading.")
        with open(str(cache_path), "rb") as read_file:
            cached: WavFileProperties = pickle.load(read_file)
        return cached

    input_wavs = [read_wav_file(wav) if isinstance(wav, Path) else wav for wav in wavs]
    sample_rates = {wav.sample_rate for wav in input_wavs}
    if len(sample_rates) != 1:
        raise ValueError("Multiple sample rates for input audio files is unsupported.")
    sample_rate = next(iter(sample_rates))
    mono = np.concatenate([wav.wav_data.mean(axis=1) if wav.wav_data.ndim > 1 else wav.wav_data for wav in input_wavs])
    name = "_".join(wav.name for wav in input_wavs) + "_mono"
    num_samples = mono.shape[0]

    if frames_per_second is not None:
        # sr * (L * fps * duration) / samples, truncated exactly like music.py:127-132
        scaled_sample_rate: float = int(
            sample_rate * (vector_length * (frames_per_second * (num_samples / sample_rate))) / num_samples
        )
    else:
        scaled_sample_rate = float(sample_rate) * (target_num_vectors / (num_samples / vector_length))

    scaled = _scale_wav_to_sample_rate(WavFileProperties(sample_rate, mono, name), scaled_sample_rate)
    data = scaled.wav_data
    if pad_to_length:
        data = pad_array(data, int(np.ceil(data.shape[0] / vector_length) * vector_length))
    output = WavFileProperties(wav_data=data, sample_rate=sample_rate, name=f"{scaled.name}_padded")
    if cache_path is not None:
        with open(str(cache_path), "wb") as write_file:
            pickle.dump(output, write_file)
    return output
