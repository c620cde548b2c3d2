"""
The stand-alone stage functions (gance_amd/apply_spectrogram.py, gance_amd/vector_sources/*) against goldens
captured from the REFERENCE's own functions (oracle/make_goldens.py::standalone_cases, unit_cases and the
end-to-end blend cases), plus the reference's own unit test of `scale_vectors_to_length_resample`
(test/test_vector_sources_common.py:16-63) pointed at the mirror.

Bars: float64 stages 1e-9 absolute on values of order 1..100 (observed ~1e-13: the operator tables are built
in long double, the reference goes through FFTs); float32 RMS and every integer stage bit-exact.
"""

from pathlib import Path

import numpy as np
import pytest
import torch

from gance_amd import apply_spectrogram, synthetic
from gance_amd.data_into_network_visualization.visualization_common import DataLabel, ResultLayers
from gance_amd.vector_sources import vector_reduction, vector_sources_common
from gance_amd.vector_sources.vector_types import is_vector

pytestmark = pytest.mark.gpu

TOL = dict(rtol=0.0, atol=1e-9)


@pytest.fixture(scope="module")
def standalone(golden_dir: Path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the product path has no CPU fallback")
    return np.load(golden_dir / "standalone_api.npz")


@pytest.fixture(scope="module")
def helpers(golden_dir: Path):
    return np.load(golden_dir / "vector_helpers.npz")


@pytest.mark.parametrize("original_vector_length,num_original_vectors,output_vector_length", [(10, 2, 50), (10, 1, 1000)])
def test_scale_vectors_to_length(standalone, original_vector_length: int, num_original_vectors: int, output_vector_length: int) -> None:
    """The reference's own sanity test, verbatim expectations, plus equality with what the reference returned."""
    len_original = original_vector_length * num_original_vectors
    unscaled = np.sin(np.linspace(start=0, stop=len_original - 1, num=len_original))
    scaled = vector_sources_common.scale_vectors_to_length_resample(
        data=unscaled, original_vector_length=original_vector_length, output_vector_length=output_vector_length
    )
    indexer = np.arange(start=0, stop=(output_vector_length * num_original_vectors), step=output_vector_length / original_vector_length).astype(int)
    at_points = np.array([scaled[index] for index in indexer])
    assert np.allclose(unscaled, at_points, atol=0.5)
    assert len(vector_sources_common.sub_vectors(unscaled, original_vector_length)) == len(
        vector_sources_common.sub_vectors(scaled, output_vector_length)
    )
    assert np.isclose(max(unscaled), max(scaled), atol=0.1)
    assert np.isclose(min(unscaled), min(scaled), atol=0.1)
    assert is_vector(scaled)
    np.testing.assert_allclose(scaled, standalone[f"scale_{original_vector_length}x{num_original_vectors}_to_{output_vector_length}"], **TOL)


def test_resample_and_smoothing_match_the_reference(standalone, helpers) -> None:
    vectors = standalone["vectors_11x24"]
    np.testing.assert_allclose(vector_sources_common.scale_vectors_to_length_resample(vectors, 24, 9), standalone["scale_24_to_9"], **TOL)
    np.testing.assert_allclose(vector_sources_common.scale_vectors_to_length_resample(vectors, 24, 16), standalone["scale_24_to_16"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_across_vectors(vectors, 24, window_length=5, polyorder=2), standalone["across_5_2"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_across_vectors(vectors, 24), standalone["across_default"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_each_vector(vectors, 24, window_length=9, polyorder=3), standalone["each_9_3"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_each_vector(vectors, 24, window_length=3, polyorder=1), standalone["each_3_1"], **TOL)
    # polynomial orders above 3: scipy takes any order below the window length, and so do the reference's functions
    np.testing.assert_allclose(vector_sources_common.smooth_each_vector(vectors, 24, window_length=9, polyorder=5), standalone["each_9_5"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_each_vector(vectors, 24, window_length=11, polyorder=8), standalone["each_11_8"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_across_vectors(vectors, 24, window_length=7, polyorder=4), standalone["across_7_4"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_across_vectors(vectors, 24, window_length=7, polyorder=6), standalone["across_7_6"], **TOL)
    data = helpers["data"]
    np.testing.assert_allclose(vector_sources_common.smooth_across_vectors(data, 8, 7, 3), helpers["smooth_across_7_3"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_each_vector(data, 64, 5, 3), helpers["smooth_each_5_3"], **TOL)
    np.testing.assert_allclose(vector_sources_common.smooth_each_vector(data, 64), helpers["smooth_each_default"], **TOL)
    np.testing.assert_allclose(vector_sources_common.scale_vectors_to_length_resample(data, 64, 100), helpers["resample_64_to_100"], **TOL)
    np.testing.assert_allclose(
        vector_sources_common.scale_vectors_to_length_resample(helpers["resample_255_input"], 255, 512), helpers["resample_255_to_512"], **TOL
    )


def test_smoothing_rejects_what_scipy_rejects() -> None:
    data = np.arange(4 * 8, dtype=np.float64)
    with pytest.raises(ValueError):
        vector_sources_common.smooth_across_vectors(data, 8)  # 4 vectors < window 7
    with pytest.raises(ValueError):
        vector_sources_common.smooth_each_vector(data, 8)  # vector of 8 < window 51
    with pytest.raises(ValueError):
        vector_sources_common.smooth_each_vector(data, 8, window_length=4, polyorder=2)  # even window


def test_remap_values_into_range(standalone) -> None:
    out = vector_sources_common.remap_values_into_range(standalone["remap_in"], (-2.0, 5.0), (10.0, -3.0))
    assert isinstance(out, list) and len(out) == 37
    np.testing.assert_allclose(np.array(out), standalone["remap_out"], rtol=0, atol=1e-13)
    with pytest.raises(ValueError):
        vector_sources_common.remap_values_into_range([0.0, 9.0], (0.0, 5.0), (0.0, 1.0))


def test_spectrogram_stages_match_the_reference(standalone) -> None:
    num_frames, seed = (int(v) for v in standalone["audio_frames_seed"])
    audio = synthetic.synthetic_audio(num_frames, 512, seed=seed)
    db = apply_spectrogram.compute_spectrogram(audio, 512)
    assert db.shape == (255, num_frames) and db.dtype == np.float64
    np.testing.assert_allclose(db, standalone["db"], rtol=0, atol=1e-8)  # dB of magnitudes down to 1e-9 of the maximum
    stereo = np.stack([audio, audio[::-1]], axis=1)
    np.testing.assert_allclose(apply_spectrogram.compute_spectrogram(stereo, 512), standalone["db_stereo"], rtol=0, atol=1e-8)
    full = apply_spectrogram.compute_spectrogram(audio, 512, truncate=False)  # apply_spectrogram.py:75-78: every bin, maximum over all of them
    assert full.shape == (510, num_frames) and full.dtype == np.float64
    np.testing.assert_allclose(full, standalone["db_full"], rtol=0, atol=1e-8)
    reference_db = standalone["db"]
    np.testing.assert_allclose(apply_spectrogram.reshape_spectrogram_to_vectors(reference_db, 512, None), standalone["vectors_no_range"], **TOL)
    np.testing.assert_allclose(apply_spectrogram.reshape_spectrogram_to_vectors(reference_db, 512, (0, 3)), standalone["vectors_range_0_3"], **TOL)
    np.testing.assert_allclose(apply_spectrogram.compute_spectrogram_smooth_scale(audio, 512, (-2, 2)), standalone["smooth_scale_m2_2"], rtol=0, atol=1e-8)


def test_silent_window_raises_like_the_reference() -> None:
    """log10(0) = -inf reaches minmax_scale, which raises ValueError (apply_spectrogram.py:43,81)."""
    audio = synthetic.synthetic_audio(12, 512, seed=1).copy()
    audio[512 * 3 : 512 * 4] = 0.0
    with pytest.raises(ValueError):
        apply_spectrogram.compute_spectrogram_smooth_scale(audio, 512, (-5, 5))


def test_rms_reduction_and_quantisation_match_the_reference(standalone, helpers) -> None:
    num_frames, seed = (int(v) for v in standalone["audio_frames_seed"])
    audio = synthetic.synthetic_audio(num_frames, 512, seed=seed)
    layers = vector_reduction.reduce_vector_rms_rolling_average(audio, 512, rolling_average_window=5, savgol_window_length=9, savgol_polyorder=2)
    assert [layers.result.label, layers.layers[0].label, layers.layers[1].label] == list(standalone["rms_labels"])
    assert layers.layers[1].data.dtype == np.float32 and np.array_equal(layers.layers[1].data, standalone["rms_raw"])  # bit-exact
    np.testing.assert_allclose(layers.layers[0].data, standalone["rms_rolling_5"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(layers.result.data, standalone["rms_smoothed_9_2"], rtol=1e-13, atol=0)
    order_5 = vector_reduction.reduce_vector_rms_rolling_average(audio, 512, rolling_average_window=3, savgol_window_length=9, savgol_polyorder=5)
    np.testing.assert_allclose(order_5.result.data, standalone["rms_smoothed_9_5"], rtol=1e-12, atol=0)
    quantized = vector_reduction.quantize_results_layers(layers, [0, 1, 2, 3])
    assert quantized.result.label == str(standalone["rms_quantized_label"][0])
    assert np.array_equal(quantized.result.data, standalone["rms_quantized_4"])
    assert len(quantized.layers) == 3
    ramp = ResultLayers(result=DataLabel(helpers["ramp"], "ramp"))
    assert np.array_equal(vector_reduction.quantize_results_layers(ramp, [0, 1, 2]).result.data, helpers["quantized_3"])
    assert np.array_equal(vector_reduction.quantize_results_layers(ramp, [0, 1, 2, 3, 4]).result.data, helpers["quantized_5"])


def test_stage_by_stage_equals_the_fused_blend(golden_dir: Path) -> None:
    """The stand-alone chain and `gance_blend_run` are the same arithmetic: compare on a blend golden's stages."""
    golden = np.load(golden_dir / "blend_n60_seed0_roll_k3.npz")
    audio = synthetic.synthetic_audio(60, 512, seed=0)
    smoothed = apply_spectrogram.compute_spectrogram_smooth_scale(audio, 512, (-5, 5))
    stride = int(round(smoothed.size / golden["smoothed_sample"].size))
    np.testing.assert_allclose(smoothed[::stride], golden["smoothed_sample"], rtol=0, atol=1e-7)
    layers = vector_reduction.reduce_vector_rms_rolling_average(time_series_audio_vectors=audio, vector_length=512)
    assert np.array_equal(vector_reduction.quantize_results_layers(layers, [0, 1, 2]).result.data, golden["roll_values"])
