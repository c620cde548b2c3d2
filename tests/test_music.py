"""
CPU tests of the WAV input stage: reading, mono mix, modes, padding, errors,
and the ORACLE of the time-stretch (oracle/audio_ref.resample_audio). Whatever actually resamples
runs a HIP kernel and is tested in tests/test_music_gpu.py.
"""

from pathlib import Path

import numpy as np
import pytest
from scipy.io import wavfile

from gance_amd.vector_sources import music
from oracle import audio_ref


def test_resampler_oracle_reproduces_the_reference_known_answer(golden_dir) -> None:
    """
    test/test_dynamic_model_switching.py:15-39 through the ORACLE: claps.wav (the reference's own test asset,
    44.1 kHz int16) -> read_wavs_scale_for_video(vector_length=1000, frames_per_second=60) -> first vector ->
    reduce_vector_rms_rolling_max -> 0.00298562 at np.isclose's defaults. This is the one number the reference
    holds on the resampy path; it pins the restatement of resampy in oracle/audio_ref.py.
    """
    wav = music.read_wav_file(golden_dir / "claps.wav")
    assert wav.sample_rate == 44100 and wav.wav_data.shape == (73728,)
    samples = wav.wav_data.shape[0]
    new_rate = int(wav.sample_rate * (1000 * (60.0 * (samples / wav.sample_rate))) / samples)  # music.py:127-132
    assert new_rate == 60000
    scaled = audio_ref.resample_audio(wav.wav_data, wav.sample_rate, new_rate)
    assert scaled.dtype == np.float32 and scaled.shape == (int(samples * (60000 / 44100)),)
    raw, reduced = audio_ref.reduce_vector_rms_rolling_max(scaled[:1000], 1000)
    assert reduced.shape == (1,) and reduced.dtype == np.float32
    assert np.isclose(0.00298562, reduced[0])


def test_resampler_oracle_preserves_a_tone_and_filters_at_ratio_one() -> None:
    rate, tone = 8000, 440.0
    t = np.arange(8000) / rate
    x = np.sin(2 * np.pi * tone * t).astype(np.float32)
    # resampy at ratio 1 is a pass through its low-pass filter (roll-off 0.9476), not a copy
    same_rate = audio_ref.resample_audio(x, rate, rate)
    assert same_rate.shape == x.shape and same_rate.dtype == np.float32 and not np.array_equal(same_rate, x)
    assert np.abs(same_rate[200:-200] - x[200:-200]).max() < 2e-3
    for new_rate in (12000, 5000):
        y = audio_ref.resample_audio(x, rate, new_rate)
        assert len(y) == int(len(x) * new_rate / rate)
        want = np.sin(2 * np.pi * tone * np.arange(len(y)) / new_rate)
        inner = slice(200, len(y) - 200)  # away from the zero-padded ends
        assert np.abs(y[inner] - want[inner]).max() < 2e-3
    # a tone above the new Nyquist is removed, not aliased
    high = np.sin(2 * np.pi * 3500.0 * t).astype(np.float32)
    assert np.abs(audio_ref.resample_audio(high, rate, 4000)[100:-100]).max() < 5e-3
    # float64 signals are accumulated in float64 (y takes x's dtype)
    assert audio_ref.resample_audio(x.astype(np.float64), rate, 12000).dtype == np.float64
    with pytest.raises(ValueError):
        audio_ref.resample_audio(x[:3], 8000, 1000)  # int(3 / 8) = 0 output samples


def test_resampler_filter_table_is_the_published_kaiser_best_window() -> None:
    """
    The table libgance_hip builds on the host (Cephes I0, libm sin) against resampy's recipe evaluated with
    numpy / scipy (sinc_window(64, 9, kaiser(14.7697), 0.9476)): 32 769 entries, identical to the last bit here,
    bar 4 ulp (library exp / sin may differ by an ulp between builds). No GPU involved.
    """
    from gance_amd import hip_lib  # pylint: disable=import-outside-toplevel

    table = hip_lib.resample_filter_table()
    want, num_table = audio_ref.kaiser_best_half_window()
    assert num_table == 512 and table.shape == want.shape == (64 * 512 + 1,)
    assert table[0] == audio_ref.ROLLOFF and abs(table[-1]) < 1e-7  # the Kaiser taper ends at 1 / I0(beta), not at zero
    ulps = np.abs(table.view(np.int64) - want.view(np.int64))
    assert ulps[np.abs(want) > 1e-300].max() <= 4


def _write(path: Path, rate: int, data: np.ndarray) -> Path:
    wavfile.write(str(path), rate, data)
    return path


def test_read_wav_file_remaps_integers_to_unit_floats(tmp_path: Path) -> None:
    ramp16 = np.array([-32768, -16384, 0, 16383, 32767], dtype=np.int16)
    got = music.read_wav_file(_write(tmp_path / "a.wav", 22050, ramp16))
    assert got.sample_rate == 22050 and got.name == "a" and got.wav_data.dtype == np.float32
    np.testing.assert_allclose(got.wav_data, np.interp(ramp16.astype(float), [-32768, 32767], [-1, 1]), atol=1e-7)
    ramp32 = np.array([-2147483648, 0, 2147483647], dtype=np.int32)
    np.testing.assert_allclose(music.read_wav_file(_write(tmp_path / "b.wav", 8000, ramp32)).wav_data, [-1, 0, 1], atol=1e-6)
    floats = np.linspace(-0.5, 0.5, 7, dtype=np.float32)
    assert np.array_equal(music.read_wav_file(_write(tmp_path / "c.wav", 8000, floats)).wav_data, floats)
    raw = music.read_wav_file(_write(tmp_path / "d.wav", 8000, ramp16), convert_to_32bit_float=False)
    assert raw.wav_data.dtype == np.int16
    with pytest.raises(ValueError, match="unknown input format"):
        music.read_wav_file(_write(tmp_path / "e.wav", 8000, np.zeros(4, dtype=np.uint8)))


def test_scale_for_video_modes_padding_mono_and_errors(tmp_path: Path, monkeypatch) -> None:
    # host logic only: the resampling kernel (tests/test_music_gpu.py) is stood in for by its oracle
    monkeypatch.setattr(music, "resample_audio", audio_ref.resample_audio)
    L, rate = 512, 30720
    stereo = np.stack([np.full(rate, 8192, dtype=np.int16), np.full(rate, -8192, dtype=np.int16)], axis=1)
    mono = (np.sin(np.arange(rate) / 20.0) * 20000).astype(np.int16)
    paths = [_write(tmp_path / "s.wav", rate, stereo), _write(tmp_path / "m.wav", rate, mono)]
    # 2 s at 60 fps and L * fps == rate: exactly 120 vectors (ratio 1: filtered, not copied)
    by_fps = music.read_wavs_scale_for_video(paths, L, frames_per_second=60.0)
    assert by_fps.wav_data.shape == (120 * L,) and by_fps.sample_rate == rate and by_fps.name == "s_m_mono_scaled_padded"
    np.testing.assert_allclose(by_fps.wav_data[: rate - 200], 0.0, atol=2e-5)  # stereo halves cancel in the mono mix (the filter rings at the seam)
    by_count = music.read_wavs_scale_for_video(paths, L, target_num_vectors=120)  # the same ratio of 1
    assert by_count.wav_data.shape == (120 * L,) and np.array_equal(by_count.wav_data, by_fps.wav_data)
    unpadded = music.read_wavs_scale_for_video(paths[1:], L, target_num_vectors=60, pad_to_length=False)
    assert len(unpadded.wav_data) == 60 * L
    cache = tmp_path / "cache.p"
    first = music.read_wavs_scale_for_video(paths, L, target_num_vectors=120, cache_path=cache)
    assert cache.exists()
    again = music.read_wavs_scale_for_video([], L, target_num_vectors=120, cache_path=cache)  # served from the cache
    assert np.array_equal(first.wav_data, again.wav_data)
    with pytest.raises(ValueError, match="both FPS mode"):
        music.read_wavs_scale_for_video(paths, L, frames_per_second=60, target_num_vectors=3)
    with pytest.raises(ValueError, match="Need to use FPS mode"):
        music.read_wavs_scale_for_video(paths, L)
    other = _write(tmp_path / "o.wav", 8000, mono)
    with pytest.raises(ValueError, match="Multiple sample rates"):
        music.read_wavs_scale_for_video([paths[0], other], L, frames_per_second=60)


@pytest.mark.parametrize("case,rate", [("pcm16", 44100), ("pcm32", 48000), ("stereo16", 22050), ("float32", 30720)])
def test_read_wav_file_matches_the_reference(tmp_path, golden_dir, case: str, rate: int) -> None:
    """
    `read_wav_file` against what the REFERENCE's own function returned (music.py:172-209 run by
    oracle/make_goldens.py::read_wav_cases on the same generated PCM): sample values bit for bit, dtype,
    sample rate and name. The reference maps every sample through interp1d in a process pool; here one
    vectorised line.
    """
    from scipy.io import wavfile  # pylint: disable=import-outside-toplevel

    golden = np.load(golden_dir / "read_wav.npz")
    path = tmp_path / f"{case}_clip.wav"
    wavfile.write(str(path), rate, golden[case])
    result = music.read_wav_file(path)
    want_rate, want_name, want_dtype = golden[f"{case}_meta"]
    assert str(result.sample_rate) == want_rate and result.name == want_name and str(result.wav_data.dtype) == want_dtype
    assert result.wav_data.shape == golden[f"{case}_out"].shape
    assert np.array_equal(result.wav_data, golden[f"{case}_out"])


def test_read_wav_file_rejects_8_bit_pcm_like_the_reference(tmp_path) -> None:
    """scipy reads 8-bit PCM as uint8, which the reference's dtype switch does not list (music.py:188-201)."""
    from scipy.io import wavfile  # pylint: disable=import-outside-toplevel

    path = tmp_path / "eight.wav"
    wavfile.write(str(path), 8000, np.arange(0, 200, dtype=np.uint8))
    with pytest.raises(ValueError, match="unknown input format"):
        music.read_wav_file(path)
