"""
CPU tests pinning the audio -> latent ORACLE (oracle/audio_ref.py) against golden vectors captured
from the reference's own code (oracle/make_goldens.py; fixtures in tests/golden/), and against
the live reference when /root/reference is mounted (build container only).
"""

import numpy as np
import pytest

from gance_amd import synthetic
from oracle import audio_ref, ref_stubs

BLEND_CASES = [
    "blend_n60_seed0_roll_k3",
    "blend_n60_seed1_noroll_k1",
    "blend_n60_seed2_roll_k1",
    "blend_n240_seed3_roll_k3",
    "blend_n1800_seed7_roll_k3",
]
FLOAT_TOL = dict(rtol=1e-10, atol=1e-10)


def load_case(golden_dir, name):
    golden = np.load(golden_dir / f"{name}.npz")
    num_frames, vector_length, num_projection, seed, roll, num_networks, stride, depth = (int(v) for v in golden["meta"])
    alpha, amp_lo, amp_hi = (float(v) for v in golden["alpha_amp"])
    audio = synthetic.synthetic_audio(num_frames, vector_length, seed=seed)
    latents = synthetic.synthetic_final_latents(num_projection, vector_length, seed=seed + 4)
    config = dict(
        num_frames=num_frames, vector_length=vector_length, roll=bool(roll), num_networks=num_networks,
        stride=stride, depth=depth, alpha=alpha, amp=(amp_lo, amp_hi),
    )
    return golden, audio, latents, config


def check_stage(golden, key: str, array: np.ndarray, stride: int, **tol) -> None:
    flat = np.asarray(array).reshape(-1)
    tol = tol or FLOAT_TOL
    np.testing.assert_allclose(flat[::stride], golden[f"{key}_sample"], **tol)
    lo, hi, total, size = golden[f"{key}_stats"]
    assert flat.size == int(size)
    np.testing.assert_allclose([flat.min(), flat.max()], [lo, hi], **tol)
    np.testing.assert_allclose(flat.sum(dtype=np.float64), total, rtol=1e-9, atol=1e-6)


@pytest.mark.parametrize("name", BLEND_CASES)
def test_every_stage_matches_the_reference_goldens(golden_dir, name: str) -> None:
    golden, audio, latents, cfg = load_case(golden_dir, name)
    L, stride = cfg["vector_length"], cfg["stride"]
    stages = audio_ref.create_spectrogram_stages(audio, L, cfg["amp"], cfg["roll"])
    check_stage(golden, "db", stages.db, stride)
    check_stage(golden, "scaled", stages.scaled, stride)
    check_stage(golden, "smoothed_time", stages.smoothed_time, stride)
    check_stage(golden, "smoothed", stages.smoothed, stride)
    check_stage(golden, "final", stages.final, stride)
    # float32 RMS: bit-exact (same summation order as librosa's numpy call)
    assert np.array_equal(stages.raw_rms, golden["raw_rms"]) and stages.raw_rms.dtype == golden["raw_rms"].dtype
    smoothed, rolling = audio_ref.smoothed_rolling_average(stages.raw_rms, 3, 7, 3)
    np.testing.assert_allclose(rolling, golden["rolling_average"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(smoothed, golden["rolling_smoothed"], rtol=1e-13, atol=1e-16)
    # integer stages: exact
    assert np.array_equal(audio_ref.quantize_to_indices(smoothed, 3), golden["roll_values"])
    if cfg["roll"]:
        assert np.array_equal(stages.roll_values, golden["roll_values"])

    result = audio_ref.alpha_blend_projection_file(
        latents, cfg["alpha"], cfg["roll"], cfg["amp"], cfg["depth"], audio, L, list(range(cfg["num_networks"]))
    )
    assert np.array_equal(result.network_indices, golden["network_indices"])
    assert result.combined.shape == (18, cfg["num_frames"] * L) and result.combined.dtype == np.float64
    assert result.projected.dtype == np.float32
    check_stage(golden, "combined_row0", result.combined[0], stride)
    check_stage(golden, "combined_row_depth", result.combined[cfg["depth"]], stride, rtol=0, atol=0)
    check_stage(golden, "projected_row0", result.projected[0], stride, rtol=0, atol=0)
    for row in range(1, cfg["depth"]):
        assert np.array_equal(result.combined[row], result.combined[0])


NOISE_CASES = ["noise_n60_seed0_roll_k3", "noise_n600_seed5_noroll_k2"]


@pytest.mark.parametrize("name", NOISE_CASES)
def test_noise_blend_matches_the_reference_goldens(golden_dir, name: str) -> None:
    """noise-blend: the noise source is the same library calls, so float32 stages are bit-equal."""
    golden = np.load(golden_dir / f"{name}.npz")
    num_frames, vector_length, seed, roll, num_networks, stride = (int(v) for v in golden["meta"])
    alpha, amp_lo, amp_hi = (float(v) for v in golden["alpha_amp"])
    audio = synthetic.synthetic_audio(num_frames, vector_length, seed=seed)
    result = audio_ref.alpha_blend_vectors_max_rms_power_audio(
        alpha, bool(roll), (amp_lo, amp_hi), audio, vector_length, list(range(num_networks))
    )
    assert result.noise.dtype == np.float32 and result.combined.dtype == np.float64
    assert np.array_equal(result.noise[::stride], golden["noise_sample"])
    assert np.array_equal([result.noise.min(), result.noise.max()], golden["noise_stats"][:2])
    assert np.array_equal(result.network_indices, golden["network_indices"])
    check_stage(golden, "spectrogram", result.spectrogram, stride)
    check_stage(golden, "combined", result.combined, stride)
    assert np.array_equal(audio_ref.gaussian_data(vector_length, num_frames)[::stride], golden["gaussian_default_sample"])
    assert np.array_equal(audio_ref.gaussian_data(vector_length, num_frames, 3, 2)[::stride], golden["gaussian_both_sample"])


def test_array_helpers_match_the_reference_goldens(golden_dir) -> None:
    golden = np.load(golden_dir / "vector_helpers.npz")
    data = golden["data"]
    assert np.array_equal(audio_ref.rotate_vectors_over_time(data, 64, golden["rolls"]), golden["rotated"])
    np.testing.assert_allclose(audio_ref.smooth_across_vectors(data, 8, 7, 3), golden["smooth_across_7_3"], **FLOAT_TOL)
    np.testing.assert_allclose(audio_ref.smooth_each_vector(data, 64, 5, 3), golden["smooth_each_5_3"], **FLOAT_TOL)
    np.testing.assert_allclose(audio_ref.smooth_each_vector(data, 64), golden["smooth_each_default"], **FLOAT_TOL)
    assert np.array_equal(audio_ref.duplicate_to_vector_count(data, 64, 18), golden["duplicated_x3"])
    assert np.array_equal(audio_ref.sub_vectors(golden["matrices"], 32), golden["sub_vectors_matrix"])
    assert golden["sub_vectors_matrix"].shape == (5, 18, 32)  # test/test_vector_sources_common.py:66-83
    assert np.array_equal(audio_ref.sub_vectors(data, 64), golden["sub_vectors_vector"])
    assert np.array_equal(audio_ref.quantize_to_indices(golden["ramp"], 3), golden["quantized_3"])
    assert np.array_equal(audio_ref.quantize_to_indices(golden["ramp"], 5), golden["quantized_5"])
    resampled = audio_ref.reshape_spectrogram_to_vectors(golden["resample_255_input"].reshape(3, 255).T, 512, None)
    np.testing.assert_allclose(resampled, golden["resample_255_to_512"], **FLOAT_TOL)
    with pytest.raises(ValueError):
        audio_ref.duplicate_to_vector_count(data, 64, 20)  # vsc:318-331


def test_pairwise_sum_restatement_is_numpy_bit_for_bit() -> None:
    rng = np.random.RandomState(0)
    for n in (1, 7, 8, 9, 100, 128, 129, 512, 1000, 1800, 4097):
        values = (rng.randn(n) * rng.rand() * 10).astype(np.float32)
        assert audio_ref.numpy_pairwise_sum_f32(values) == np.add.reduce(values)


def test_rolling_mean_restatement_is_pandas_bit_for_bit() -> None:
    import pandas as pd

    rng = np.random.RandomState(1)
    for n in (3, 4, 10, 1800):
        values = np.abs(rng.randn(n)).astype(np.float32)
        want = pd.Series(values).rolling(3).mean().to_numpy()
        got = audio_ref.pandas_rolling_mean_kahan(values, 3)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[2:], want[2:])


def test_rms_known_answer_shape_of_reference_test() -> None:
    """Mirror of test/test_dynamic_model_switching.py:15-39: one RMS value per 1000-sample vector."""
    audio = synthetic.synthetic_audio(4, 1000, seed=3)
    rms = audio_ref.compute_raw_rms(audio[:1000], 1000)
    assert rms.shape == (1,)
    assert np.isclose(rms[0], np.sqrt(np.mean(audio[:1000].astype(np.float64) ** 2)), rtol=1e-6)


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="/root/reference only exists in the build container")
def test_live_reference_agrees_on_fresh_inputs() -> None:
    """Beyond the committed goldens: a seed that no fixture covers, straight against the reference."""
    ref_stubs.install()
    from gance.data_into_network_visualization import visualization_inputs as vi  # pylint: disable=import-error
    from gance.vector_sources.vector_types import MatricesLabel  # pylint: disable=import-error

    num_frames, L = 90, 512
    audio = synthetic.synthetic_audio(num_frames, L, seed=99)
    latents = synthetic.synthetic_final_latents(num_frames // 3, L, seed=5)
    want = vi.alpha_blend_projection_file(
        final_latents_matrices_label=MatricesLabel(latents, L, "x"), alpha=0.4, fft_roll_enabled=True,
        fft_amplitude_range=(-1, 1), blend_depth=10, time_series_audio_vectors=audio, vector_length=L,
        network_indices=[0, 1, 2, 3],
    )
    got = audio_ref.alpha_blend_projection_file(latents, 0.4, True, (-1, 1), 10, audio, L, [0, 1, 2, 3])
    np.testing.assert_allclose(got.spectrogram, want.a_vectors.data, **FLOAT_TOL)
    np.testing.assert_allclose(got.combined, want.combined.data, **FLOAT_TOL)
    assert np.array_equal(got.projected, want.b_vectors.data)
    assert np.array_equal(got.network_indices, want.network_indices.result.data)
