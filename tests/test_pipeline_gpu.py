"""
GPU end-to-end test of the drop-in API: network files -> MultiNetwork, audio + projected latents ->
alpha_blend_projection_file -> vector_synthesis -> frames, against the two oracles chained the
same way (BASELINE.json configs[2] and [4] in miniature: blend + per-frame network switching).
"""

from pathlib import Path

import numpy as np
import pytest
import torch

from gance_amd import network_file, synthetic
from gance_amd.data_into_network_visualization import network_visualization, visualization_inputs
from gance_amd.network_interface import network_functions
from gance_amd.stylegan2 import spec as sg2_spec
from gance_amd.vector_sources.vector_types import MatricesLabel
from oracle import audio_ref, stylegan2_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def network_dir(tmp_path_factory) -> Path:
    """Three tiny random-init 1024-layout-compatible... no: three 64x64 generators (W = 10 rows)."""
    directory = tmp_path_factory.mktemp("networks")
    for seed in range(3):
        network_file.write_random_network(directory / f"network_{seed}.pkl", 64, seed=seed)
    return directory


def test_single_network_interface_shapes_and_dispatch(network_dir: Path) -> None:
    """test/test_network_functions.py:100-118 in this implementation: shape, dtype, non-trivial sum."""
    process = network_functions.create_network_interface_process(network_dir / "network_0.pkl")
    try:
        interface = process.network_interface
        assert interface.expected_vector_length == 512
        z = np.random.RandomState(0).randn(512)  # float64, like the reference's callers pass
        image = interface.create_image_vector(z)
        assert image.shape == (64, 64, 3) and image.dtype == np.uint8 and image.flags["C_CONTIGUOUS"] and image.sum() > 0
        assert np.array_equal(interface.create_image_generic(z), image)
        matrix = np.random.RandomState(1).randn(10, 512)
        from_matrix = interface.create_image_matrix(matrix)
        assert np.array_equal(interface.create_image_generic(matrix), from_matrix)
        variables = network_file.load_network(network_dir / "network_0.pkl").variables
        want = stylegan2_ref.convert_images_to_uint8(stylegan2_ref.synthesize_w(matrix[None].astype(np.float32), variables, 64))[0]
        assert np.abs(from_matrix.astype(int) - want.astype(int)).max() <= 1
    finally:
        process.stop_function()
        process.stop_function()  # idempotent


def test_blend_api_returns_the_reference_structure() -> None:
    num_frames, L = 64, 512
    audio = synthetic.synthetic_audio(num_frames, L, seed=31)
    latents = synthetic.synthetic_final_latents(num_frames // 2, L, seed=32)
    got = visualization_inputs.alpha_blend_projection_file(
        final_latents_matrices_label=MatricesLabel(latents, L, "projected"), alpha=0.25, fft_roll_enabled=True,
        fft_amplitude_range=(-5, 5), blend_depth=12, time_series_audio_vectors=audio, vector_length=L,
        network_indices=[0, 1, 2],
    )
    want = audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, audio, L, [0, 1, 2])
    assert got.a_vectors.data.shape == (num_frames * L,) and got.a_vectors.data.dtype == np.float64
    assert got.b_vectors.data.shape == (18, num_frames * L) and got.b_vectors.data.dtype == np.float32
    assert got.combined.data.shape == (18, num_frames * L) and got.combined.data.dtype == np.float64
    assert got.b_vectors.label == "projected"
    np.testing.assert_allclose(got.a_vectors.data, want.spectrogram, rtol=0, atol=1e-7)
    np.testing.assert_allclose(got.combined.data, want.combined, rtol=0, atol=1e-7)
    assert np.array_equal(got.b_vectors.data, want.projected)
    assert np.array_equal(got.network_indices.result.data, want.network_indices)
    with pytest.raises(ValueError, match="Cannot duplicate"):
        visualization_inputs.alpha_blend_projection_file(
            final_latents_matrices_label=MatricesLabel(latents[:, : 7 * L], L, "x"), alpha=0.25, fft_roll_enabled=True,
            fft_amplitude_range=(-5, 5), blend_depth=12, time_series_audio_vectors=audio, vector_length=L, network_indices=[0],
        )


def test_blend_then_synthesis_with_network_switching(network_dir: Path) -> None:
    """projection-file-blend in miniature: 3 networks, index chosen per frame from the audio."""
    num_frames, L, W = 24, 512, 10
    audio = synthetic.synthetic_audio(num_frames, L, seed=41)
    latents = synthetic.synthetic_final_latents(num_frames // 2, L, depth=18, seed=42)
    paths = network_functions.parse_network_paths(str(network_dir), None, None)
    with network_functions.MultiNetwork(network_paths=paths) as multi:
        data = visualization_inputs.alpha_blend_projection_file(
            final_latents_matrices_label=MatricesLabel(latents, L, "projected"), alpha=0.25, fft_roll_enabled=True,
            fft_amplitude_range=(-5, 5), blend_depth=12, time_series_audio_vectors=audio, vector_length=L,
            network_indices=multi.network_indices,
        )
        # a 64x64 generator takes W = 10 rows: feed it the first 10 rows of the 18-row matrices
        trimmed = data._replace(combined=MatricesLabel(data.combined.data[:W], L, "trimmed"))
        output = network_visualization.vector_synthesis(data=trimmed, networks=multi, enable_2d=False, enable_3d=False)
        frames = list(output.synthesized_images)
    indices = data.network_indices.result.data
    assert len(frames) == num_frames and len(set(indices.tolist())) > 1
    want = audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, audio, L, [0, 1, 2])
    assert np.array_equal(indices, want.network_indices)
    dlatents = audio_ref.sub_vectors(want.combined[:W], L).astype(np.float32)
    for frame_index in (0, 7, num_frames - 1):
        variables = network_file.load_network(paths[int(indices[frame_index])]).variables
        image = stylegan2_ref.synthesize_w(dlatents[frame_index : frame_index + 1], variables, 64)
        expected = stylegan2_ref.convert_images_to_uint8(image)[0]
        diff = np.abs(frames[frame_index].astype(int) - expected.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
