"""
GPU tests of the audio time-stretch (gance_resample_audio_f32 / _f64 through the C ABI) against its CPU
oracle (oracle/audio_ref.resample_audio: resampy 0.2.2's kaiser_best algorithm restated in numpy), the
reference's own length rule (test/test_vector_source_music.py:13-24) and the reference's own known answer
through the resampler (test/test_dynamic_model_switching.py:15-39). Bar: BIT-EXACT — the kernel follows
resampy's tap order and its per-tap rounding to the signal's dtype, and its filter table is identical to
numpy's (tests/test_music.py).
"""

from pathlib import Path

import numpy as np
import pytest
from scipy.io import wavfile

from gance_amd import hip_lib, synthetic
from gance_amd.vector_sources import music
from oracle import audio_ref

pytestmark = pytest.mark.gpu

RESAMPLE_ATOL = 0.0  # same operations in the same order on the same table


@pytest.mark.parametrize("multiplier", [2, 1.5, 0.3, 0.1, 10])
def test_scaled_length_follows_the_reference_rule(multiplier: float) -> None:
    wav = music.WavFileProperties(44100, synthetic.synthetic_audio(8, 512, seed=1), "synthetic")
    scaled = music._scale_wav_to_sample_rate(wav, int(wav.sample_rate * multiplier))  # pylint: disable=protected-access
    assert len(scaled.wav_data) == int(len(wav.wav_data) * int(wav.sample_rate * multiplier) / wav.sample_rate)
    assert scaled.sample_rate == wav.sample_rate and scaled.wav_data.dtype == np.float32
    want = audio_ref.resample_audio(wav.wav_data, wav.sample_rate, int(wav.sample_rate * multiplier))
    np.testing.assert_allclose(scaled.wav_data, want, rtol=0, atol=RESAMPLE_ATOL)


def test_resampler_matches_oracle_on_a_long_signal_and_preserves_a_tone() -> None:
    rate = 44100
    x = synthetic.synthetic_audio(120, 512, seed=3)  # 61 440 samples
    got = music.resample_audio(x, rate, 30720)       # the down-sampling a 44.1 kHz WAV gets at 60 fps
    want = audio_ref.resample_audio(x, rate, 30720)
    assert got.shape == want.shape == (int(len(x) * 30720 / rate),)
    np.testing.assert_allclose(got, want, rtol=0, atol=RESAMPLE_ATOL)
    t = np.arange(8000) / 8000
    tone = np.sin(2 * np.pi * 440.0 * t).astype(np.float32)
    y = music.resample_audio(tone, 8000, 12000)
    inner = slice(200, len(y) - 200)
    assert np.abs(y[inner] - np.sin(2 * np.pi * 440.0 * np.arange(len(y)) / 12000)[inner]).max() < 2e-3


def test_reduce_vector_rms_alignment_known_answer_of_the_reference(golden_dir) -> None:
    """
    test/test_dynamic_model_switching.py:15-39 reproduced call for call on the product path (claps.wav is that
    test's own asset): WAV -> read_wavs_scale_for_video (resampler on the GPU, 44.1 kHz -> 60 kHz) -> first vector
    of 1000 samples -> reduce_vector_rms_rolling_max (RMS kernel) -> 0.00298562 at np.isclose's defaults.
    """
    from gance_amd.vector_sources.vector_reduction import reduce_vector_rms_rolling_max  # pylint: disable=import-outside-toplevel
    from gance_amd.vector_sources.vector_sources_common import sub_vectors  # pylint: disable=import-outside-toplevel

    vector_length = 1000
    audio = music.read_wavs_scale_for_video(
        wavs=[golden_dir / "claps.wav"], vector_length=vector_length, frames_per_second=60.0
    ).wav_data
    assert audio.dtype == np.float32 and audio.shape == (101000,)  # 100 310 samples padded to whole vectors
    single_audio_vector = sub_vectors(data=audio, vector_length=vector_length)[0]
    reduced = reduce_vector_rms_rolling_max(time_series_audio_vectors=single_audio_vector, vector_length=vector_length)
    assert reduced.result.data.shape[0] == 1
    # Known value.
    assert np.isclose(0.00298562, reduced.result.data[0])
    assert reduced.result.label == "Rolling Max" and reduced.layers[0].label == "Raw RMS Power"
    # and sample for sample the oracle's restatement of resampy
    wav = music.read_wav_file(golden_dir / "claps.wav")
    want = audio_ref.resample_audio(wav.wav_data, wav.sample_rate, 60000)
    assert np.array_equal(audio[: len(want)], want) and not audio[len(want) :].any()


def test_rms_rolling_max_matches_oracle_on_a_long_series() -> None:
    """Frames of 1000 samples every 512 (librosa's fixed hop), 200 values: a rolling maximum over two values."""
    from gance_amd.vector_sources.vector_reduction import reduce_vector_rms_rolling_max  # pylint: disable=import-outside-toplevel

    audio = synthetic.synthetic_audio(201, 512, seed=5)
    for vector_length in (1000, 512):
        got = reduce_vector_rms_rolling_max(audio, vector_length)
        raw, want = audio_ref.reduce_vector_rms_rolling_max(audio, vector_length)
        assert len(raw) == 1 + (len(audio) - vector_length) // 512 and len(raw) // 80 == 2
        assert got.layers[0].data.dtype == np.float32 and np.array_equal(got.layers[0].data, raw)
        assert np.array_equal(got.result.data, want)


def test_resampler_accumulates_float64_signals_in_float64() -> None:
    x = synthetic.synthetic_audio(40, 512, seed=9).astype(np.float64)
    for new_rate in (30720, 48000):
        got = music.resample_audio(x, 44100, new_rate)
        want = audio_ref.resample_audio(x, 44100, new_rate)
        assert got.dtype == np.float64 and np.array_equal(got, want)
    # ratio 1: still the low-pass filter
    same = music.resample_audio(x.astype(np.float32), 30720, 30720)
    assert np.array_equal(same, audio_ref.resample_audio(x.astype(np.float32), 30720, 30720)) and not np.array_equal(same, x)


def test_resampler_rejects_a_wrong_output_length() -> None:
    import torch  # pylint: disable=import-outside-toplevel

    d_in = torch.zeros(1000, dtype=torch.float32, device="cuda")
    d_out = torch.zeros(1500, dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError, match="num_out"):
        hip_lib.resample_audio_device(d_in.data_ptr(), 1000, 8000, 12000, d_out.data_ptr(), 1499)
    with pytest.raises(ValueError, match="Invalid sample rate"):
        hip_lib.resample_audio_device(d_in.data_ptr(), 1000, 0, 12000, d_out.data_ptr(), 1500)
    with pytest.raises(ValueError, match="too small"):
        music.resample_audio(np.zeros(3, dtype=np.float32), 8000, 1000)


def test_scale_for_video_stretches_to_the_requested_vector_count(tmp_path: Path) -> None:
    L, rate = 512, 30720
    mono = (np.sin(np.arange(2 * rate) / 20.0) * 20000).astype(np.int16)
    path = tmp_path / "m.wav"
    wavfile.write(str(path), rate, mono)
    by_count = music.read_wavs_scale_for_video([path], L, target_num_vectors=90)
    assert by_count.wav_data.shape == (90 * L,) and by_count.wav_data.dtype == np.float32
    unpadded = music.read_wavs_scale_for_video([path], L, target_num_vectors=7, pad_to_length=False)
    assert len(unpadded.wav_data) == 7 * L
    # the stretched signal is the oracle's
    source = music.read_wav_file(path).wav_data
    want = audio_ref.resample_audio(source, rate, float(rate) * (90 / (len(source) / L)))
    np.testing.assert_allclose(by_count.wav_data[: len(want)], want, rtol=0, atol=RESAMPLE_ATOL)
