"""
GPU tests of the post-synthesis resize kernel and of the end-to-end
`projection_file_blend_api` harness (BASELINE.json configs[2] in miniature): WAV file + projection
file + network files -> resized uint8 frames, against the chained oracles.
"""

from pathlib import Path

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from gance_amd import hip_lib, network_file, projection_file_blend, synthetic
from gance_amd.projection import projection_file_reader as pfr
from oracle import audio_ref, resize_ref, stylegan2_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,dst,batch", [(64, 135, 3), (1024, 2160, 1), (256, 100, 2), (32, 32, 1)])
def test_bicubic_resize_matches_oracle(src: int, dst: int, batch: int) -> None:
    """Float bicubic a=-0.75: <= 1 LSB from the float64 restatement (fp32 vs fp64 rounding at .5)."""
    images = np.random.RandomState(src + dst).randint(0, 256, (batch, src, src, 3)).astype(np.uint8)
    d_in = torch.from_numpy(images).cuda()
    d_out = torch.empty((batch, dst, dst, 3), dtype=torch.uint8, device="cuda")
    hip_lib.resize_bicubic_u8_device(d_in.data_ptr(), batch, src, d_out.data_ptr(), dst, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    want = resize_ref.resize_bicubic_u8(images, dst)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3
    if src == dst:
        assert np.array_equal(got, images)  # identity at scale 1


def test_resize_keeps_flat_images_flat() -> None:
    flat = torch.full((2, 64, 64, 3), 200, dtype=torch.uint8, device="cuda")
    out = torch.empty((2, 150, 150, 3), dtype=torch.uint8, device="cuda")
    hip_lib.resize_bicubic_u8_device(flat.data_ptr(), 2, 64, out.data_ptr(), 150)
    torch.cuda.synchronize()
    assert int(out.min()) == 200 and int(out.max()) == 200


def test_projection_file_blend_api_end_to_end(tmp_path: Path) -> None:
    L, num_projection, fps_in, fps_out, side, out_side = 512, 8, 15.0, 30.0, 64, 96
    num_frames = int(num_projection * fps_out / fps_in)
    # inputs on disk: a float32 WAV at L * fps Hz (stretch ratio 1: resampy then still low-pass filters it), a
    # projection file, two network files
    audio = synthetic.synthetic_audio(num_frames, L, seed=51, frames_per_second=fps_out)
    wav_path = tmp_path / "audio.wav"
    wavfile.write(str(wav_path), int(L * fps_out), audio)
    latents = synthetic.synthetic_final_latents(num_projection, L, seed=52)
    projection_path = tmp_path / "projection.npz"
    pfr.write_projection_npz(projection_path, latents.reshape(18, num_projection, L).transpose(1, 0, 2), projection_fps=fps_in)
    network_paths = []
    for seed in range(2):
        path = tmp_path / f"net_{seed}.pkl"
        network_file.write_random_network(path, side, seed=seed)
        network_paths.append(path)
    out_path = tmp_path / "frames.npy"
    projection_file_blend.projection_file_blend_api(
        wav=[str(wav_path)], output_path=str(out_path), network_paths=network_paths, frames_to_visualize=None,
        output_fps=fps_out, output_side_length=out_side, debug_path=None, debug_window=None, debug_side_length=None,
        alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5), projection_file_path=str(projection_path),
        blend_depth=12, complexity_change_rolling_sum_window=None, complexity_change_threshold=None,
        phash_distance=None, bbox_distance=None, track_length=None,
    )
    frames = np.load(out_path)
    assert frames.shape == (num_frames, out_side, out_side, 3) and frames.dtype == np.uint8

    # the same pipeline through the oracles (the WAV stage: resampy at the ratio read_wavs_scale_for_video computes)
    stretched = audio_ref.resample_audio(audio, int(L * fps_out), float(int(L * fps_out)) * (num_frames / (len(audio) / L)))
    want_blend = audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, stretched, L, [0, 1])
    dlatents = audio_ref.sub_vectors(want_blend.combined, L).astype(np.float32)
    rows = int(np.log2(side)) * 2 - 2
    for frame_index in (0, num_frames // 2, num_frames - 1):
        variables = network_file.load_network(network_paths[int(want_blend.network_indices[frame_index])]).variables
        image = stylegan2_ref.synthesize_w(dlatents[frame_index : frame_index + 1, :rows], variables, side)
        native = stylegan2_ref.convert_images_to_uint8(image)
        expected = resize_ref.resize_bicubic_u8(native, out_side)[0]
        diff = np.abs(frames[frame_index].astype(int) - expected.astype(int))
        assert diff.max() <= 2 and (diff > 0).mean() < 5e-3  # 1 LSB synthesis x bicubic overshoot

    with pytest.raises(ValueError, match="Overlay music mask"):
        projection_file_blend.projection_file_blend_api(
            [str(wav_path)], None, network_paths, None, fps_out, out_side, None, None, None, 0.25, True, (-5, 5),
            str(projection_path), 12, 3, 4, None, None, None,
        )
    with pytest.raises(ValueError, match="Cannot evenly divide"):
        projection_file_blend.projection_file_blend_frames(
            [str(wav_path)], network_paths, None, 40.0, out_side, 0.25, True, (-5, 5), str(projection_path), 12
        )
    incomplete = tmp_path / "incomplete.npz"
    pfr.write_projection_npz(incomplete, latents.reshape(18, num_projection, L).transpose(1, 0, 2), projection_fps=fps_in, complete=False)
    with pytest.raises(ValueError, match="Invalid Projection File"):
        projection_file_blend.projection_file_blend_frames(
            [str(wav_path)], network_paths, None, fps_out, out_side, 0.25, True, (-5, 5), str(incomplete), 12
        )


def test_noise_blend_end_to_end(tmp_path: Path) -> None:
    """WAV + two networks -> frames through the z entry, against the oracles on sampled frames."""
    from gance_amd import noise_blend  # pylint: disable=import-outside-toplevel

    L, num_frames, fps, side, out_side = 512, 24, 30.0, 64, 48
    audio = synthetic.synthetic_audio(num_frames, L, seed=61, frames_per_second=fps)
    wav_path = tmp_path / "audio.wav"
    wavfile.write(str(wav_path), int(L * fps), audio)
    network_paths = []
    for seed in range(2):
        path = tmp_path / f"net_{seed}.pkl"
        network_file.write_random_network(path, side, seed=seed + 5)
        network_paths.append(path)
    frames = noise_blend.noise_blend_frames([str(wav_path)], network_paths, 20, fps, out_side, 0.25, True, (-5, 5))
    assert frames.shape == (20, out_side, out_side, 3) and frames.dtype == np.uint8

    stretched = audio_ref.resample_audio(audio, int(L * fps), int(L * fps))  # FPS mode at L * fps Hz: ratio 1, still filtered
    want = audio_ref.alpha_blend_vectors_max_rms_power_audio(0.25, True, (-5, 5), stretched, L, [0, 1])
    z = audio_ref.sub_vectors(want.combined, L).astype(np.float32)
    for frame_index in (0, 9, 19):
        variables = network_file.load_network(network_paths[int(want.network_indices[frame_index])]).variables
        image = stylegan2_ref.synthesize_z(z[frame_index : frame_index + 1], variables, side, truncation_psi=1.2)
        expected = resize_ref.resize_bicubic_u8(stylegan2_ref.convert_images_to_uint8(image), out_side)[0]
        diff = np.abs(frames[frame_index].astype(int) - expected.astype(int))
        assert diff.max() <= 2 and (diff > 0).mean() < 5e-3


def test_projection_file_blend_with_eye_tracking_overlay(tmp_path: Path) -> None:
    """The overlay stage end to end with a stand-in landmark detector, against the oracles on the host."""
    from gance_amd.overlay import overlay_common  # pylint: disable=import-outside-toplevel
    from oracle import overlay_ref  # pylint: disable=import-outside-toplevel

    L, num_projection, fps_in, fps_out, side, out_side = 512, 6, 15.0, 30.0, 64, 128
    num_frames = int(num_projection * fps_out / fps_in)
    audio = synthetic.synthetic_audio(num_frames, L, seed=71, frames_per_second=fps_out)
    wav_path = tmp_path / "audio.wav"
    wavfile.write(str(wav_path), int(L * fps_out), audio)
    latents = synthetic.synthetic_final_latents(num_projection, L, seed=72)
    rng = np.random.RandomState(73)
    targets = (np.kron(rng.rand(num_projection, 8, 8, 3), np.ones((1, 16, 16, 1))) * 255).astype(np.uint8)  # 128 x 128
    projection_path = tmp_path / "projection.npz"
    pfr.write_projection_npz(
        projection_path, latents.reshape(18, num_projection, L).transpose(1, 0, 2), projection_fps=fps_in, target_images=targets
    )
    network_path = tmp_path / "net.pkl"
    network_file.write_random_network(network_path, side, seed=9)

    class EveryFrameHasAFace:  # pylint: disable=too-few-public-methods
        """Same eye landmarks in every frame except frames whose first pixel is dark (no face)."""

        @staticmethod
        def face_landmarks(face_image):
            if int(face_image[0, 0].sum()) < 96:
                return []
            return [{"left_eye": ((30, 40), (50, 52)), "right_eye": ((70, 41), (95, 55))}]

    common = dict(
        wav=[str(wav_path)], network_paths=[network_path], frames_to_visualize=None, output_fps=fps_out,
        output_side_length=out_side, alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5),
        projection_file_path=str(projection_path), blend_depth=12,
    )
    background = projection_file_blend.projection_file_blend_frames(**common)
    overlay = projection_file_blend.OverlayParameters(
        phash_distance=64, bbox_distance=5.0, track_length=3, face_finder=EveryFrameHasAFace()
    )
    blended = projection_file_blend.projection_file_blend_frames(**common, overlay=overlay)
    assert blended.shape == background.shape == (num_frames, out_side, out_side, 3)

    # host restatement: which frames have a face in both pictures, runs >= 3, write the eye box
    foreground = np.repeat(targets, int(fps_out / fps_in), axis=0)
    box = overlay_common.landmarks_to_bounding_boxes(EveryFrameHasAFace.face_landmarks(np.full((1, 1, 3), 255)))[0]
    has_face = [
        bool(EveryFrameHasAFace.face_landmarks(fg) and EveryFrameHasAFace.face_landmarks(bg)) for fg, bg in zip(foreground, background)
    ]
    keep = overlay_ref.track_length_filter(has_face, 3)
    assert any(keep), "the test should exercise at least one written overlay"
    for index in range(num_frames):
        want = overlay_ref.write_boxes_onto_image(foreground[index], background[index], [tuple(box)] if keep[index] else [])
        assert np.array_equal(blended[index], want), index

    with pytest.raises(NotImplementedError, match="landmark detector"):
        projection_file_blend.projection_file_blend_api(
            [str(wav_path)], None, [network_path], None, fps_out, out_side, None, None, None, 0.25, True, (-5, 5),
            str(projection_path), 12, None, None, 10, 5.0, 3,
        )


def test_streaming_overlay_with_three_networks_equals_the_one_shot_overlay(tmp_path: Path) -> None:
    """
    configs[4] in miniature through the STREAM: three resident networks switched by the RMS index, engine calls
    batched by network across a window of pieces, the overlay gate evaluated chunk by chunk (4 frames per chunk, runs
    of gated frames shorter than 5 dropped, so decisions span chunk borders). Frames must equal the one-shot form
    (every frame resident, `apply_eye_tracking_overlay` on the lot), and the stage may never hold more than the chunks
    an undecided run spans.
    """
    L, num_projection, fps_in, fps_out, side = 512, 30, 15.0, 30.0, 64
    num_frames = int(num_projection * fps_out / fps_in)
    audio = synthetic.synthetic_audio(num_frames, L, seed=81, frames_per_second=fps_out)
    wav_path = tmp_path / "audio.wav"
    wavfile.write(str(wav_path), int(L * fps_out), audio)
    latents = synthetic.synthetic_final_latents(num_projection, L, seed=82)
    rng = np.random.RandomState(83)
    targets = (np.kron(rng.rand(num_projection, 8, 8, 3), np.ones((1, 8, 8, 1))) * 255).astype(np.uint8)  # 64 x 64
    projection_path = tmp_path / "projection.npz"
    pfr.write_projection_npz(
        projection_path, latents.reshape(18, num_projection, L).transpose(1, 0, 2), projection_fps=fps_in, target_images=targets
    )
    network_paths = []
    for seed in range(3):
        network_paths.append(tmp_path / f"net_{seed}.pkl")
        network_file.write_random_network(network_paths[-1], side, seed=20 + seed)

    class FaceWhereThePictureIsBright:  # pylint: disable=too-few-public-methods
        @staticmethod
        def face_landmarks(face_image):
            if int(face_image[0, 0].sum()) < 200:
                return []
            return [{"left_eye": ((12, 20), (24, 26)), "right_eye": ((36, 21), (50, 28))}]

    common = dict(
        wav=[str(wav_path)], network_paths=network_paths, frames_to_visualize=None, output_fps=fps_out,
        output_side_length=side, alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5),
        projection_file_path=str(projection_path), blend_depth=12,
    )
    overlay = projection_file_blend.OverlayParameters(
        phash_distance=64, bbox_distance=5.0, track_length=5, face_finder=FaceWhereThePictureIsBright()
    )
    background = projection_file_blend.projection_file_blend_frames(**common)  # one chunk, no overlay
    want = projection_file_blend.apply_eye_tracking_overlay(
        torch.from_numpy(background).cuda(), targets, int(fps_out / fps_in), overlay, audio, L
    ).cpu().numpy()
    assert (want != background).any(), "the test should exercise at least one written overlay"

    timings: dict = {}
    streamed = np.zeros_like(background)
    firsts = []
    for first, total, frames in projection_file_blend.projection_file_blend_frame_chunks(**common, frames_per_call=4, overlay=overlay, timings=timings):
        assert total == num_frames
        streamed[first : first + len(frames)] = frames
        firsts.append(first)
    assert firsts == list(range(0, num_frames, 4))

    def same_frames(a: np.ndarray, b: np.ndarray) -> bool:
        # engine calls of 4 frames and of 20 pick different kernel forms for some layers: 1 LSB on a few pixels
        diff = np.abs(a.astype(np.int16) - b.astype(np.int16))
        return int(diff.max()) <= 1 and float((diff > 0).mean()) < 1e-3

    assert same_frames(streamed, want)
    # the overlay regions themselves are copies of the target pictures: wherever the one-shot form wrote one, so did the stream
    written = (want != background).any(axis=(1, 2, 3))
    assert written.any() and np.array_equal((np.abs(streamed.astype(np.int16) - background.astype(np.int16)) > 1).any(axis=(1, 2, 3)), written)
    assert 1 <= timings["overlay_chunks_held_max"] <= 3 and timings["overlays_written"] == int(written.sum())
    # and the network-major windows (2 x 3 pieces) gave the frames of the piece-by-piece path
    assert same_frames(projection_file_blend.projection_file_blend_frames(**common, frames_per_call=4), background)


def test_blend_from_a_real_hdf5_projection_file(tmp_path: Path, golden_dir: Path) -> None:
    """The reference's own container end to end: an h5py-written projection file (12 projected frames, 15 fps) -> 24 frames at 30 fps."""
    L, fps_out, side = 512, 30.0, 64
    expected = np.load(golden_dir / "projection_hdf5_expected.npz")
    latents = np.concatenate(list(expected["projection_v2_latents"]), axis=-1)  # (18, F*L), what final_latents_matrices_label builds
    num_frames = 24
    audio = synthetic.synthetic_audio(num_frames, L, seed=71, frames_per_second=fps_out)
    wav_path = tmp_path / "audio.wav"
    wavfile.write(str(wav_path), int(L * fps_out), audio)
    network_path = tmp_path / "net.pkl"
    network_file.write_random_network(network_path, side, seed=3)
    frames = projection_file_blend.projection_file_blend_frames(
        wav=[str(wav_path)], network_paths=[network_path], frames_to_visualize=None, output_fps=fps_out, output_side_length=side,
        alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5), projection_file_path=str(golden_dir / "projection_v2.hdf5"),
        blend_depth=12,
    )
    assert frames.shape == (num_frames, side, side, 3)
    stretched = audio_ref.resample_audio(audio, int(L * fps_out), float(int(L * fps_out)) * (num_frames / (len(audio) / L)))
    want_blend = audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, stretched, L, [0])
    dlatents = audio_ref.sub_vectors(want_blend.combined, L).astype(np.float32)
    rows = int(np.log2(side)) * 2 - 2
    variables = network_file.load_network(network_path).variables
    want = stylegan2_ref.convert_images_to_uint8(stylegan2_ref.synthesize_w(dlatents[:, :rows, :], variables, side))
    diff = np.abs(frames.astype(np.int16) - want.astype(np.int16))
    assert int(diff.max()) <= 1 and float((diff > 0).mean()) < 1e-3
