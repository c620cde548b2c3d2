"""
Full-size, oracle-checked tests of the PRODUCT STREAM (BASELINE.json configs[2] / [3] / [4] at 1024x1024, not in
miniature): `projection_file_blend_frame_chunks` on the bench's own inputs (gance_amd/synthetic.py:
`benchmark_blend_inputs`, files on disk), 64 frames per engine call, a ragged tail chunk, sampled frames against the
CHAINED oracles (oracle/audio_ref -> oracle/stylegan2_ref in fp64 -> oracle/resize_ref); the reference's own GPU test
at this boundary run verbatim (test/test_network_functions.py:100-118: zeros z, 1024^2, through
`create_network_interface_process`); and the same stream once through RCCL (`init_process_group("nccl")` with one
rank, in process, GANCE_FORCE_COLLECTIVES=1: scatter, the asynchronous gather of every chunk, the gloo control group
beside NCCL, the reader stream's `work.wait()`).

What these cover that the stage-wise tests do not: `_WindowSynthesizer`, the engine calls writing straight into the
stream's chunk buffers (`writes_into`), the pinned host ring, the resize inside the stream and the streaming overlay,
all at 1024^2 with full batches.

Budget: the fp64 oracle at 1024^2 is ~10-20 s per frame on the host cores; the sampled frames are shared between
the native-size run, the 2160 run (= the bicubic oracle of the same native frame) and the RCCL run.
"""

import datetime
import os
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
from scipy.io import wavfile

from gance_amd import network_file, projection_file_blend, synthetic
from gance_amd.network_interface import network_functions
from gance_amd.network_interface.network_functions import MultiNetwork
from gance_amd.projection import projection_file_reader as pfr
from gance_amd.stylegan2 import spec as sg2_spec
from oracle import audio_ref, resize_ref, stylegan2_ref

pytestmark = pytest.mark.gpu

RESOLUTION, VECTOR_LENGTH, FPS_IN, FPS_OUT = 1024, 512, 30.0, 60.0
BLEND_FRAMES = 240          # frames the WAV / projection file hold (120 projected latents at 30 fps -> 240 frames at 60 fps)
FRAMES_TO_VISUALIZE = 200   # 3 full chunks of 64 + a tail chunk of 8
FRAMES_PER_CALL = 64
SAMPLES = (0, 64, 199)      # first chunk, the first frame behind a chunk border, the last frame of the ragged tail


class _Job:
    """Files on disk + the chained oracles of one blend job, shared by the tests of this module."""

    def __init__(self, directory: Path) -> None:
        self.directory = directory
        self.audio, self.latents = synthetic.benchmark_blend_inputs(BLEND_FRAMES)
        self.wav_path = directory / "audio.wav"
        wavfile.write(str(self.wav_path), int(VECTOR_LENGTH * FPS_OUT), self.audio)
        rng = np.random.RandomState(73)
        self.targets = (np.kron(rng.rand(BLEND_FRAMES // 2, 8, 8, 3), np.ones((1, 16, 16, 1))) * 255).astype(np.uint8)  # 128 x 128
        self.projection_path = directory / "projection.npz"
        pfr.write_projection_npz(
            self.projection_path, self.latents.reshape(18, BLEND_FRAMES // 2, VECTOR_LENGTH).transpose(1, 0, 2), projection_fps=FPS_IN,
            target_images=self.targets,
        )
        self.network_paths = [directory / f"net_{seed}.pkl" for seed in range(3)]
        for seed, path in enumerate(self.network_paths):
            network_file.write_random_network(path, RESOLUTION, seed=seed)
        self._blends: dict = {}
        self._frames: dict = {}

    def common(self, num_networks: int, side: int) -> dict:
        return dict(
            wav=[str(self.wav_path)], network_paths=self.network_paths[:num_networks], frames_to_visualize=FRAMES_TO_VISUALIZE,
            output_fps=FPS_OUT, output_side_length=side, alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5),
            projection_file_path=str(self.projection_path), blend_depth=12, frames_per_call=FRAMES_PER_CALL,
        )

    def blend(self, num_networks: int):
        """oracle/audio_ref on the same files' contents: (dlatents [N][18][L] float32, network indices [N])."""
        if num_networks not in self._blends:
            sample_rate = int(VECTOR_LENGTH * FPS_OUT)
            stretched = audio_ref.resample_audio(self.audio, sample_rate, float(sample_rate) * (BLEND_FRAMES / (len(self.audio) / VECTOR_LENGTH)))
            want = audio_ref.alpha_blend_projection_file(self.latents, 0.25, True, (-5, 5), 12, stretched, VECTOR_LENGTH, list(range(num_networks)))
            self._blends[num_networks] = (audio_ref.sub_vectors(want.combined, VECTOR_LENGTH).astype(np.float32), np.asarray(want.network_indices))
        return self._blends[num_networks]

    def oracle_frame(self, frame_index: int, network_index: int) -> np.ndarray:
        """uint8 [1024][1024][3]: the fp64 oracle of one frame (its dlatents do not depend on the number of networks)."""
        key = (frame_index, network_index)
        if key not in self._frames:
            dlatents, _ = self.blend(1)
            variables = network_file.load_network(self.network_paths[network_index]).variables
            image = stylegan2_ref.synthesize_w(dlatents[frame_index : frame_index + 1], variables, RESOLUTION)
            self._frames[key] = stylegan2_ref.convert_images_to_uint8(image)[0]
        return self._frames[key]


@pytest.fixture(scope="module")
def job(tmp_path_factory) -> _Job:
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the product path has no CPU fallback")
    return _Job(tmp_path_factory.mktemp("full_size_stream"))


def _run_stream(job: _Job, num_networks: int, side: int, **extra):
    """The stream, every chunk consumed on the host: {frame index: copy} of the sampled frames + the chunk starts."""
    kept, firsts, received = {}, [], 0
    wanted = extra.pop("keep", SAMPLES)
    for first, total, frames in projection_file_blend.projection_file_blend_frame_chunks(**job.common(num_networks, side), **extra):
        assert total == FRAMES_TO_VISUALIZE and frames.dtype == np.uint8 and frames.shape[1:] == (side, side, 3)
        firsts.append(first)
        received += len(frames)
        for index in wanted:
            if first <= index < first + len(frames):
                kept[index] = frames[index - first].copy()
    assert received == FRAMES_TO_VISUALIZE
    assert firsts == list(range(0, FRAMES_TO_VISUALIZE, FRAMES_PER_CALL)), firsts  # 0, 64, 128, 192: the tail chunk holds 8
    return kept


def _assert_close(got: np.ndarray, want: np.ndarray, max_lsb: int, share: float) -> None:
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert int(diff.max()) <= max_lsb and float((diff > 0).mean()) < share, (int(diff.max()), float((diff > 0).mean()))


def test_blend_stream_1024_native_and_2160_against_the_chained_oracles(job: _Job) -> None:
    """configs[2] and configs[3] (on one GPU) through the product stream at full size, sampled frames vs the chained oracles."""
    dlatents, indices = job.blend(1)
    assert dlatents.shape == (BLEND_FRAMES, 18, VECTOR_LENGTH) and not indices.any()
    native = _run_stream(job, 1, RESOLUTION)
    for index in SAMPLES:
        _assert_close(native[index], job.oracle_frame(index, 0), 1, 1e-3)
    resized = _run_stream(job, 1, 2160)
    for index in SAMPLES:
        want = resize_ref.resize_bicubic_u8(job.oracle_frame(index, 0)[None], 2160)[0]
        _assert_close(resized[index], want, 2, 5e-3)  # 1 LSB of the synthesis x the bicubic kernel's overshoot


def test_blend_stream_1024_three_networks_with_the_streaming_overlay(job: _Job) -> None:
    """
    configs[4] on one GPU at full size: three resident networks switched by the RMS index (network-major windows of 12
    pieces), the overlay gate evaluated chunk by chunk. Sampled frames: the background against the chained oracles with
    the network the oracle's index picks; the overlay regions bit-equal to the scaled target picture.
    """
    from bench import SyntheticFaceFinder  # pylint: disable=import-outside-toplevel  (the stand-in detector the bench uses)
    from gance_amd.overlay import overlay_common  # pylint: disable=import-outside-toplevel
    from oracle import overlay_ref  # pylint: disable=import-outside-toplevel

    _, indices = job.blend(3)
    assert set(np.unique(indices[:FRAMES_TO_VISUALIZE])) == {0, 1, 2}
    # the sampled frames of the one-network test that network 0 renders here too (their oracle frames exist already: the
    # dlatents of a frame do not depend on the number of networks), plus the first frame behind the second chunk border
    # that another network renders, in the middle of a network-major window
    samples = [index for index in SAMPLES if indices[index] == 0]
    samples.append(next(i for i in range(128, FRAMES_TO_VISUALIZE) if indices[i] != 0))
    if len(samples) == 1:
        samples.append(next(i for i in range(FRAMES_TO_VISUALIZE) if indices[i] == 0))

    background = _run_stream(job, 3, RESOLUTION, keep=samples)
    for index in samples:
        _assert_close(background[index], job.oracle_frame(index, int(indices[index])), 1, 1e-3)

    finder = SyntheticFaceFinder(RESOLUTION)
    overlay = projection_file_blend.OverlayParameters(phash_distance=64, bbox_distance=5.0, track_length=5, face_finder=finder)
    timings: dict = {}
    blended = _run_stream(job, 3, RESOLUTION, keep=samples, overlay=overlay, timings=timings)
    assert 1 <= timings["overlay_chunks_held_max"] <= 3 and timings["overlays_written"] > 0
    # the overlay's foreground: the projection's target pictures scaled to the output side (bicubic), each shown twice
    box = tuple(overlay_common.landmarks_to_bounding_boxes(finder.face_landmarks(np.full((1, 1, 3), 255)))[0])
    written = 0
    for index in samples:
        foreground = resize_ref.resize_bicubic_u8(job.targets[index // 2][None], RESOLUTION)[0]
        with_box = overlay_ref.write_boxes_onto_image(foreground, background[index], [box])
        if np.array_equal(blended[index], background[index]):
            continue  # (the gate did not pass on this frame, or its run was shorter than track_length)
        written += 1
        region = with_box != background[index]
        # inside the written region the frame is the scaled target picture (<= 1 LSB: float32 / float64 bicubic), outside it the background
        _assert_close(blended[index][region], with_box[region], 1, 2e-2)
        assert np.array_equal(blended[index][~region], background[index][~region])
    assert written >= 1, "the test should sample at least one frame with an overlay written"


def test_the_references_own_gpu_test_verbatim_zeros_z_1024(job: _Job) -> None:
    """
    /root/reference/test/test_network_functions.py:100-118, statement for statement: a 1024^2 network through
    `create_network_interface_process`, `create_image_vector(np.zeros((expected_vector_length,)))`, shape
    (1024, 1024, 3), sum > 0, stop -- plus the oracle on the same z (which the reference cannot assert).
    """
    network_interface_process = network_functions.create_network_interface_process(network_path=job.network_paths[0])
    image = network_interface_process.network_interface.create_image_vector(
        data=np.zeros((network_interface_process.network_interface.expected_vector_length,))
    )
    assert image.shape == (1024, 1024, 3)
    assert np.sum(image) > 0
    network_interface_process.stop_function()
    variables = sg2_spec.make_random_variables(RESOLUTION, seed=0)
    want = stylegan2_ref.convert_images_to_uint8(stylegan2_ref.synthesize_z(np.zeros((1, 512), np.float32), variables, RESOLUTION, truncation_psi=1.2))[0]
    _assert_close(image, want, 1, 1e-3)


def test_blend_stream_once_through_rccl_with_one_rank(job: _Job, monkeypatch) -> None:
    """
    The collective path of the stream on RCCL, in this process (no child, no exec): process group "nccl" of one rank,
    GANCE_FORCE_COLLECTIVES=1 -> `scatter_for_stream` (dist.scatter of device tensors), one asynchronous `dist.gather` of
    uint8 device tensors per chunk, the gloo control group beside NCCL (one status word per chunk), `work.wait()` on the
    reader stream, both drains. The frames must equal the short-cut path's (checked against the oracle above).
    """
    from gance_amd import frame_sharding  # pylint: disable=import-outside-toplevel

    assert not dist.is_initialized()
    store = dist.FileStore(str(job.directory / "rccl_store"), 1)
    monkeypatch.setenv("GANCE_FORCE_COLLECTIVES", "1")
    device = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=device, timeout=datetime.timedelta(seconds=120))
    calls = {"gather": 0, "scatter": 0}
    real_gather, real_scatter = dist.gather, dist.scatter

    def counting_gather(tensor, *args, **kwargs):
        assert tensor.is_cuda and tensor.dtype == torch.uint8
        calls["gather"] += 1
        return real_gather(tensor, *args, **kwargs)

    def counting_scatter(tensor, *args, **kwargs):
        assert tensor.is_cuda
        calls["scatter"] += 1
        return real_scatter(tensor, *args, **kwargs)

    monkeypatch.setattr(dist, "gather", counting_gather)
    monkeypatch.setattr(dist, "scatter", counting_scatter)
    try:
        assert dist.get_backend() == "nccl" and frame_sharding.collectives_forced()
        control = frame_sharding.control_group()
        assert control is not None and dist.get_backend(control) == "gloo"
        networks = MultiNetwork(network_paths=job.network_paths[:1], load=True, max_batch=FRAMES_PER_CALL)
        try:
            through_rccl = _run_stream(job, 1, RESOLUTION, networks=networks)
            assert calls == {"gather": 4, "scatter": 2}, calls  # one gather per chunk; latents + indices scattered
            per_rank = _run_stream(job, 1, RESOLUTION, networks=networks, drain="per-rank")
            assert calls["gather"] == 4  # (no gather with the per-rank drain)
            monkeypatch.setenv("GANCE_FORCE_COLLECTIVES", "0")
            short_cut = _run_stream(job, 1, RESOLUTION, networks=networks)
        finally:
            networks.unload()
        for index in SAMPLES:
            assert np.array_equal(through_rccl[index], short_cut[index]) and np.array_equal(per_rank[index], short_cut[index])
            _assert_close(short_cut[index], job.oracle_frame(index, 0), 1, 1e-3)
    finally:
        frame_sharding._CONTROL_GROUP[0] = None  # pylint: disable=protected-access
        dist.destroy_process_group()
