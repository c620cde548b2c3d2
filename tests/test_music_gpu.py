"""
GPU tests of the audio time-stretch (gance_resample_audio_f32 through the C ABI) against its CPU
oracle (oracle/audio_ref.resample_audio, float64 numpy) and the reference's own length rule
(test/test_vector_source_music.py:13-24). Bar: 2e-6 absolute on [-1, 1] audio (float32 output of
float64 sums; the kernel and numpy evaluate sin / I0 with different library routines).
"""

from pathlib import Path

import numpy as np
import pytest
from scipy.io import wavfile

from gance_amd import hip_lib, synthetic
from gance_amd.vector_sources import music
from oracle import audio_ref

pytestmark = pytest.mark.gpu

RESAMPLE_ATOL = 2e-6


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


def test_resampler_rejects_a_wrong_output_length() -> None:
    import torch  # pylint: disable=import-outside-toplevel

    d_in = torch.zeros(1000, dtype=torch.float32, device="cuda")
    d_out = torch.zeros(1500, dtype=torch.float32, device="cuda")
    with pytest.raises(hip_lib.GanceHipError, match="num_out"):
        hip_lib.resample_audio_device(d_in.data_ptr(), 1000, 8000, 12000, d_out.data_ptr(), 1499)
    with pytest.raises(hip_lib.GanceHipError):
        hip_lib.resample_audio_device(d_in.data_ptr(), 1000, 0, 12000, d_out.data_ptr(), 1500)


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
