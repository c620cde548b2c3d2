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


def test_resident_networks_share_one_workspace() -> None:
    """
    Eight 1024^2 networks resident at 32 frames per call: weights per network (0.55 GB: 135 MB as trained + every kernel form's own image of them), ONE activation
    workspace per (device, resolution, max_batch) -- under 35 GB in all where private workspaces took 8 x 26 GB.
    Switching between them leaves every network's frames unchanged.
    """
    from gance_amd import hip_lib  # pylint: disable=import-outside-toplevel
    from gance_amd.stylegan2 import spec as sg2_spec  # pylint: disable=import-outside-toplevel

    resolution, batch = 1024, 32
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    engines = [hip_lib.Engine(sg2_spec.make_random_variables(resolution, seed=seed), resolution, max_batch=batch) for seed in range(2)]
    free_after_two, _ = torch.cuda.mem_get_info()
    engines += [hip_lib.Engine(sg2_spec.make_random_variables(resolution, seed=seed), resolution, max_batch=batch) for seed in range(2, 8)]
    free_after_eight, _ = torch.cuda.mem_get_info()
    try:
        assert free_before - free_after_eight < 35e9, f"8 resident networks took {(free_before - free_after_eight) / 1e9:.1f} GB"
        assert (free_after_two - free_after_eight) / 6 < 0.6e9, "each further network should cost its weights only"
        z = np.random.RandomState(3).randn(2, 512).astype(np.float32)
        first = [engine.synthesize_z(z) for engine in engines[:3]]
        assert not np.array_equal(first[0], first[1])  # different networks
        for engine, want in zip(reversed(engines[:3]), reversed(first)):  # switch back and forth
            assert np.array_equal(engine.synthesize_z(z), want)
        # two engines driven from two streams, asynchronously, still take turns on the shared scratch
        d_z = torch.from_numpy(z).cuda()
        outs = [torch.empty((2, resolution, resolution, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        for _ in range(3):
            for engine, out, stream in zip(engines[:2], outs, streams):
                engine.synthesize_z_device(d_z.data_ptr(), 2, 1.2, out.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(outs[0].cpu().numpy(), first[0]) and np.array_equal(outs[1].cpu().numpy(), first[1])
    finally:
        for engine in engines:
            engine.close()
    torch.cuda.synchronize()
    free_end, _ = torch.cuda.mem_get_info()
    assert free_before - free_end < 1e9, "closing the last engine frees the shared workspace"


def test_config_0_verbatim_256_random_z_through_the_network_interface(tmp_path: Path) -> None:
    """
    BASELINE.json configs[0] as stated: 256x256 random-init generator (seed 0), the 64
    z vectors `np.random.RandomState(1234).randn(64, 512)` (the seed of gance/vector_sources/primatives.py:17),
    each through `create_network_interface(path).create_image_vector` -- the reference's one-frame call form --
    against the oracle on the same z.
    """
    resolution = 256
    path = tmp_path / "config0.pkl"
    network_file.write_random_network(path, resolution, seed=0)
    vectors = np.random.RandomState(1234).randn(64, 512).astype(np.float32)
    interface = network_functions.create_network_interface_process(path)
    try:
        network = interface.network_interface
        assert network.expected_vector_length == 512
        frames = np.stack([network.create_image_vector(vector) for vector in vectors])
        generic = network.create_image_generic(vectors[5])
    finally:
        interface.stop_function()
    assert frames.shape == (64, resolution, resolution, 3) and frames.dtype == np.uint8
    assert np.array_equal(generic, frames[5])
    variables = sg2_spec.make_random_variables(resolution, seed=0)
    for start in (0, 60):  # the oracle (fp64 on the host cores) on the first and last four; all 64 went through the GPU path
        want = stylegan2_ref.convert_images_to_uint8(stylegan2_ref.synthesize_z(vectors[start : start + 4], variables, resolution, truncation_psi=1.2))
        diff = np.abs(frames[start : start + 4].astype(np.int16) - want.astype(np.int16))
        assert int(diff.max()) <= 1 and float((diff > 0).mean()) < 1e-3
    assert len({frame.tobytes() for frame in frames}) == 64  # 64 different z vectors, 64 different frames
    assert frames.sum() > 0  # the reference's own assertion at this boundary (test_network_functions.py:100-118)


def test_legacy_tf_pickle_loads_into_a_network_and_renders(tmp_path: Path) -> None:
    """
    SURVEY.md §8 f-1 end to end: a (G, D, Gs) pickle in the published dnnlib Network state layout (fabricated:
    no legacy pickle exists in the reference tree) -> restricted unpickler -> LoadedNetwork in HBM -> frame,
    against the oracle on the variables the pickle was written from (every term on: noise strengths, biases,
    stored noise buffers, dlatent_avg).
    """
    import importlib.util  # pylint: disable=import-outside-toplevel

    helper_spec = importlib.util.spec_from_file_location("legacy_pickle_helper", Path(__file__).resolve().parent / "test_legacy_import.py")
    helper = importlib.util.module_from_spec(helper_spec)
    helper_spec.loader.exec_module(helper)

    resolution = 64
    variables = sg2_spec.make_random_variables(resolution, seed=11, perturb=True)
    path = tmp_path / "legacy.pkl"
    helper._write_legacy_pickle(path, variables)  # pylint: disable=protected-access
    network = network_functions.LoadedNetwork(path, max_batch=2)
    try:
        rng = np.random.RandomState(12)
        z = rng.randn(2, 512).astype(np.float32)
        dlatents = rng.randn(1, network.engine.num_layers, 512).astype(np.float32)
        from_z = network.create_images_vector(z, randomize_noise=False)  # the stored buffers: what the oracle computes
        fresh = network.create_images_vector(z)  # the reference's vector path: upstream default randomize_noise=True
        assert (fresh != from_z).mean() > 0.2 and (network.create_images_vector(z) != fresh).mean() > 0.2  # (saturated pixels stay equal)
        assert np.array_equal(network.create_images_vector(z, noise_seed=7), network.create_images_vector(z, noise_seed=7))
        from_w = network.create_image_matrix(dlatents[0].astype(np.float64))  # callers pass float64; stored noise again
    finally:
        network.stop()
    want_z = stylegan2_ref.convert_images_to_uint8(stylegan2_ref.synthesize_z(z, variables, resolution, truncation_psi=1.2))
    want_w = stylegan2_ref.convert_images_to_uint8(stylegan2_ref.synthesize_w(dlatents, variables, resolution))
    for got, want in ((from_z, want_z), (from_w[None], want_w)):
        diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert int(diff.max()) <= 1 and float((diff > 0).mean()) < 1e-3
