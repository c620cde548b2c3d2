"""
projection-file-blend end to end: WAV(s) + projection file + network(s) -> frames.

`projection_file_blend_api` keeps the reference's parameter list and checks
(gance/projection_file_blend.py:56-343) for the synthesis path. What differs, by scope:
* the blend runs on the GPU and its per-frame latent matrices never leave HBM;
* synthesis is batched, every network is resident, and with `torch.distributed` initialised the
  frames are sharded across ranks and gathered in order on rank 0 (gance_amd/frame_sharding.py);
* the resize to `output_side_length` is the HIP bicubic kernel, on the frames still in HBM;
* the eye-tracking overlay runs on rank 0 on the gathered frames while they are still in HBM
  (perceptual hashes and the overlay write are HIP kernels; the landmark detector is external);
* video encoding (ffmpeg / x264) and the matplotlib debug video are out of scope here: frames are
  returned / written as a `.npy` uint8 array [N][S][S][3], and asking for the debug video raises
  NotImplementedError.
"""

from pathlib import Path
from typing import Iterator, List, NamedTuple, Optional, Tuple

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist

from gance_amd import divisor, frame_sharding, hip_lib, torch_ops  # noqa: F401  (torch_ops registers torch.ops.gance.*)
from gance_amd.data_into_network_visualization import visualization_inputs
from gance_amd.data_into_network_visualization.visualization_common import DataLabel, ResultLayers
from gance_amd.logger_common import LOGGER
from gance_amd.network_interface.network_functions import DEFAULT_MAX_BATCH, TRUNCATION_PSI, MultiNetwork
from gance_amd.projection import projection_file_reader
from gance_amd.overlay import overlay_common, overlay_eye_tracking
from gance_amd.vector_sources import music, vector_reduction
from gance_amd.vector_sources.vector_sources_common import underlying_length


def _common_output_side(networks: MultiNetwork, indices: np.ndarray, output_side_length: Optional[int]) -> int:
    """
    Side of the frames a run produces: `output_side_length`, or the networks' own resolution, which must then
    be the same for every network the indices select (the reference resizes to one side before writing,
    video_common.py:416-429; without a target side there is nothing to resize mismatching networks to).
    :raises ValueError: networks of different resolutions and no `output_side_length`.
    """
    if output_side_length is not None:
        return int(output_side_length)
    sides = {int(networks._network_at(int(index)).engine.resolution) for index in np.unique(indices)}  # pylint: disable=protected-access
    if len(sides) > 1:
        raise ValueError(f"networks of different resolutions {sorted(sides)} need an output_side_length to resize to")
    return sides.pop() if sides else 0


def synthesize_device_frames(  # pylint: disable=too-many-locals
    dlatents: torch.Tensor,
    network_indices: torch.Tensor,
    networks: MultiNetwork,
    output_side_length: Optional[int] = None,
    batch: int = DEFAULT_MAX_BATCH,
) -> Iterator[torch.Tensor]:
    """
    dlatents [n, W18, L] float32 and network_indices [n] int32 on the GPU -> uint8 frame batches
    [<=batch, S, S, 3] on the GPU, in frame order. Rows beyond what a network takes (a 256^2
    generator reads 14 of the 18 rows) are dropped, like feeding `combined[:W]`.
    A 2-D `dlatents` [n, L] holds z vectors and takes the network's vector entry (mapping +
    truncation psi 1.2 + synthesis, network_functions.py:144-158), as `is_vector` data does in
    the reference's `create_image_generic`.
    """
    indices = network_indices.cpu().numpy()
    out_side = _common_output_side(networks, indices, output_side_length)
    for start in range(0, dlatents.shape[0], batch):
        stop = min(dlatents.shape[0], start + batch)
        chunk = indices[start:stop]
        frames = torch.empty((stop - start, out_side, out_side, 3), dtype=torch.uint8, device=dlatents.device)
        for network_index in np.unique(chunk):
            engine = networks._network_at(int(network_index)).engine  # pylint: disable=protected-access
            members = torch.from_numpy(np.nonzero(chunk == network_index)[0] + start).to(dlatents.device)
            if dlatents.dim() == 2:
                images = torch.ops.gance.synthesize_z(dlatents.index_select(0, members), engine.op_handle, TRUNCATION_PSI)
            else:
                images = torch.ops.gance.synthesize_w(dlatents.index_select(0, members)[:, : engine.num_layers, :], engine.op_handle)
            if engine.resolution != out_side:
                images = torch.ops.gance.resize_bicubic(images, out_side)
            frames.index_copy_(0, members - start, images)
        yield frames


def synthesize_device_frames_network_major(  # pylint: disable=too-many-locals
    dlatents: torch.Tensor,
    network_indices: torch.Tensor,
    networks: MultiNetwork,
    output_side_length: Optional[int] = None,
    batch: int = DEFAULT_MAX_BATCH,
) -> torch.Tensor:
    """
    The same frames as `synthesize_device_frames`, as one tensor [n, S, S, 3], computed network by
    network: every engine call is a full batch however often the index switches (the reference
    sorts by network for the same reason, network_visualization.py:653-674: there a switch costs a
    process restart, here only a short batch). Frames land at their own positions, so the order of
    the result is the frame order.
    """
    device = dlatents.device
    indices = network_indices.cpu().numpy()
    num_frames = int(dlatents.shape[0])
    out_side = _common_output_side(networks, indices, output_side_length)
    out = torch.empty((num_frames, out_side, out_side, 3), dtype=torch.uint8, device=device)
    for network_index in np.unique(indices):
        engine = networks._network_at(int(network_index)).engine  # pylint: disable=protected-access
        side = engine.resolution
        frames_of_network = np.nonzero(indices == network_index)[0]
        for start in range(0, len(frames_of_network), batch):
            host_members = frames_of_network[start : start + batch]
            count = len(host_members)
            first = int(host_members[0])
            # a run of consecutive frames (always, with one network) is produced in place
            in_place = int(host_members[-1]) - first + 1 == count
            members = torch.from_numpy(host_members).to(device)
            native = out[first : first + count] if in_place and out_side == side else torch.empty(
                (count, side, side, 3), dtype=torch.uint8, device=device
            )
            if dlatents.dim() == 2:
                selected = dlatents[first : first + count] if in_place else dlatents.index_select(0, members)
                torch.ops.gance.synthesize_z_out(selected, engine.op_handle, TRUNCATION_PSI, native)
            else:
                selected = (dlatents[first : first + count] if in_place else dlatents.index_select(0, members))[:, : engine.num_layers, :]
                torch.ops.gance.synthesize_w_out(selected, engine.op_handle, native)
            if out_side != side:
                target = out[first : first + count] if in_place else torch.empty(
                    (count, out_side, out_side, 3), dtype=torch.uint8, device=device
                )
                torch.ops.gance.resize_bicubic_out(native, target)
                native = target
            if not in_place:
                out.index_copy_(0, members, native)
    return out


def shard_synthesize_gather(  # pylint: disable=too-many-arguments
    dlatents: Optional[torch.Tensor],
    indices: Optional[torch.Tensor],
    num_frames: int,
    networks: MultiNetwork,
    output_side_length: int,
    device: torch.device,
    keep_on_device: bool = False,
):
    """
    Rank 0 holds the per-frame network inputs (and `num_frames`); every rank synthesises its
    contiguous share and rank 0 gets the frames back in order (None elsewhere): a numpy array,
    or the uint8 tensor still in HBM with `keep_on_device`.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    if world_size > 1:
        count = [num_frames]
        dist.broadcast_object_list(count, src=0)
        num_frames = count[0]
        dlatents = frame_sharding.scatter_latents(dlatents, num_frames, device)
        indices = frame_sharding.scatter_latents(indices, num_frames, device)
    local = synthesize_device_frames_network_major(dlatents, indices, networks, output_side_length)
    frames, _ = frame_sharding.gather_frames(local, num_frames)
    torch.cuda.synchronize(device)
    if frames is None or keep_on_device:
        return frames
    return frames.cpu().numpy()


class OverlayParameters(NamedTuple):
    """The eye-tracking overlay options of the reference CLI (projection_file_blend.py:56-76)."""

    phash_distance: int
    bbox_distance: float
    track_length: int
    complexity_change_rolling_sum_window: Optional[int] = None
    complexity_change_threshold: Optional[int] = None
    face_finder: Optional[overlay_eye_tracking.FaceFinder] = None  # default: overlay_eye_tracking.FACE_FINDER_FACTORY()


def music_complexity_skip_mask(audio: np.ndarray, vector_length: int, window: int, threshold: float) -> List[bool]:
    """
    Frames where the music's complexity is changing too fast for an overlay
    (gance/projection_file_blend.py:192-226): zlib size per vector, rolling average + savgol,
    first derivative, absolute value, rolling sum; NaN (window not yet full) counts as infinity.
    """
    smoothed_sizes = vector_reduction.reduce_vector_gzip_compression_rolling_average(audio, vector_length)
    derived = vector_reduction.derive_results_layers(smoothed_sizes, order=1).result.data
    mask = vector_reduction.rolling_sum_results_layers(
        vector_reduction.absolute_value_results_layers(
            ResultLayers(result=DataLabel(derived, "Gzipped audio, smoothed, averaged, 1st order derivation."))
        ),
        window_length=window,
    )
    return list(pd.Series(mask.result.data).fillna(np.inf) > threshold)


def apply_eye_tracking_overlay(  # pylint: disable=too-many-arguments,too-many-locals
    synthesized: torch.Tensor,
    target_images: np.ndarray,
    frame_multiplier: int,
    parameters: OverlayParameters,
    audio: np.ndarray,
    vector_length: int,
) -> torch.Tensor:
    """
    The overlay stage of gance/projection_file_blend.py:181-262 on frames in HBM: the projection's
    target images (each shown `frame_multiplier` times, scaled to the output side) are the
    foreground, the synthesized frames the background; gate per frame, drop overlay runs shorter
    than `track_length`, write the eye regions.
    """
    num_frames, side = int(synthesized.shape[0]), int(synthesized.shape[1])
    device = synthesized.device
    targets = torch.from_numpy(np.ascontiguousarray(target_images)).to(device)
    if targets.shape[1] != side:
        resized = torch.empty((targets.shape[0], side, side, 3), dtype=torch.uint8, device=device)
        hip_lib.resize_bicubic_u8_device(
            targets.data_ptr(), int(targets.shape[0]), int(targets.shape[1]), resized.data_ptr(), side,
            torch.cuda.current_stream(device).cuda_stream,
        )
        targets = resized
    foreground = targets.repeat_interleave(frame_multiplier, dim=0)[:num_frames].contiguous()
    if foreground.shape[0] != num_frames:
        raise ValueError("the projection file holds too few target images for the frames being written")
    music_mask = parameters.complexity_change_rolling_sum_window is not None and parameters.complexity_change_threshold is not None
    skip_mask = (
        music_complexity_skip_mask(
            audio, vector_length, parameters.complexity_change_rolling_sum_window, parameters.complexity_change_threshold
        )[:num_frames]
        if music_mask
        else [False] * num_frames
    )
    overlay_results = overlay_eye_tracking.compute_eye_tracking_overlay(
        foreground_images=foreground,
        background_images=synthesized,
        min_phash_distance=parameters.phash_distance,
        min_bbox_distance=parameters.bbox_distance,
        skip_mask=skip_mask,
        face_finder=parameters.face_finder,
    )
    boxes_list = list(overlay_results.bbox_lists)
    long_tracks_mask = vector_reduction.track_length_filter(
        bool_tracks=[not skip and boxes is not None for skip, boxes in zip(skip_mask, boxes_list)],
        track_length=parameters.track_length,
    )
    written = [boxes if in_long_track else None for boxes, in_long_track in zip(boxes_list, long_tracks_mask)]
    LOGGER.info(f"Eye tracking overlay written on {sum(box is not None for box in written)} of {num_frames} frames")
    return overlay_common.write_boxes_onto_frames_device(foreground, synthesized, written)


class _BlendInputs(NamedTuple):
    """What rank 0 prepares before synthesis starts (everything is None / 0 on the other ranks)."""

    dlatents: Optional[torch.Tensor]
    indices: Optional[torch.Tensor]
    num_frames: int
    target_images: Optional[np.ndarray]
    audio: Optional[np.ndarray]
    frame_multiplier: int


def _prepare_blend_inputs(  # pylint: disable=too-many-arguments,too-many-locals
    wav: List[str],
    networks: MultiNetwork,
    frames_to_visualize: Optional[int],
    output_fps: float,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
    projection_file_path: str,
    blend_depth: int,
    want_target_images: bool,
    device: torch.device,
) -> _BlendInputs:
    """Rank 0: projection file + WAV -> per-frame latent matrices and network indices in HBM (the reference's checks included)."""
    vector_length = networks.expected_vector_length
    target_images = None
    # the audio -> latent stage has global dependencies over a few MB: once, on rank 0
    with projection_file_reader.load_projection_file(Path(projection_file_path)) as reader:
        final_latents = projection_file_reader.final_latents_matrices_label(reader)
        attributes = reader.projection_attributes
        if want_target_images:
            target_images = np.stack(list(reader.target_images))
    final_latents_in_file = underlying_length(final_latents.data) / vector_length
    LOGGER.info(
        f"Reading projection file. Complete: {attributes.complete}, "
        f"Final Latent Count: {final_latents_in_file}, Processed Frames: {attributes.projection_frame_count}"
    )
    if not attributes.complete or abs(final_latents_in_file - attributes.projection_frame_count) > 2:
        raise ValueError("Invalid Projection File, cannot continue.")
    frame_multiplier = divisor.divide_no_remainder(numerator=output_fps, denominator=attributes.projection_fps)
    num_output_frames = int(frame_multiplier * final_latents_in_file)
    audio = music.read_wavs_scale_for_video(
        wavs=[Path(path) for path in wav], vector_length=vector_length, target_num_vectors=num_output_frames
    ).wav_data
    blend = visualization_inputs.alpha_blend_projection_file_device(
        final_latents.data, alpha, fft_roll_enabled, fft_amplitude_range, blend_depth, audio, vector_length,
        len(networks.network_indices), device=device.index,
    )
    dlatents, indices = blend.dlatents, blend.network_indices
    blend.blend.close()
    if frames_to_visualize is not None:
        dlatents, indices = dlatents[:frames_to_visualize], indices[:frames_to_visualize]
    return _BlendInputs(dlatents, indices, int(dlatents.shape[0]), target_images, audio, int(frame_multiplier))


def projection_file_blend_frame_chunks(  # pylint: disable=too-many-arguments,too-many-locals
    wav: List[str],
    network_paths: List[Path],
    frames_to_visualize: Optional[int],
    output_fps: float,
    output_side_length: int,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
    projection_file_path: str,
    blend_depth: int,
    frames_per_call: int = DEFAULT_MAX_BATCH,
) -> Iterator[Tuple[int, int, np.ndarray]]:
    """
    The frame stream of the reference's pipeline (gance/projection_file_blend.py:343 hands an iterator of frames to
    the video writer): a generator of (first_frame_index, total_frames, frames [n, S, S, 3] uint8) in frame order.
    Nothing holds all frames: per chunk, every rank synthesises `frames_per_call` frames, one gather lands the chunk
    in order on rank 0 while the next chunk is already being synthesised, and rank 0 drains it to a pinned host
    ring (`frames` is a view of a ring slot: consume or copy it before advancing the generator twice more).
    Collective under `torch.distributed`: every rank must exhaust the generator; only rank 0 receives chunks.
    """
    rank = dist.get_rank() if dist.is_initialized() else 0
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    device = torch.device("cuda", torch.cuda.current_device())
    networks = MultiNetwork(network_paths=network_paths, load=True, max_batch=frames_per_call)
    try:
        inputs = _BlendInputs(None, None, 0, None, None, 1)
        if rank == 0:
            inputs = _prepare_blend_inputs(
                wav, networks, frames_to_visualize, output_fps, alpha, fft_roll_enabled, fft_amplitude_range,
                projection_file_path, blend_depth, False, device,
            )
        num_frames = inputs.num_frames
        if world_size > 1:
            count = [num_frames]
            dist.broadcast_object_list(count, src=0)
            num_frames = count[0]
        dlatents = frame_sharding.scatter_for_stream(inputs.dlatents, num_frames, frames_per_call, device)
        indices = frame_sharding.scatter_for_stream(inputs.indices, num_frames, frames_per_call, device)
        side = _common_output_side(networks, np.asarray(networks.network_indices), output_side_length)

        def synthesize_piece(offset: int, count: int) -> torch.Tensor:
            return synthesize_device_frames_network_major(
                dlatents[offset : offset + count], indices[offset : offset + count], networks, side, frames_per_call
            )

        for first, frames in frame_sharding.ordered_frame_stream(synthesize_piece, num_frames, frames_per_call, (side, side, 3), device):
            yield first, num_frames, frames
        torch.cuda.synchronize(device)
    finally:
        networks.unload()


def projection_file_blend_frames(  # pylint: disable=too-many-arguments,too-many-locals
    wav: List[str],
    network_paths: List[Path],
    frames_to_visualize: Optional[int],
    output_fps: float,
    output_side_length: int,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
    projection_file_path: str,
    blend_depth: int,
    overlay: Optional[OverlayParameters] = None,
) -> Optional[np.ndarray]:
    """
    The pipeline of `projection_file_blend_api`, returning ALL frames [N][S][S][3] uint8 at once (on rank 0;
    None on other ranks when running distributed): the stream of `projection_file_blend_frame_chunks` collected
    into one array. With `overlay`, rank 0 runs the eye-tracking overlay on the gathered frames before they
    leave HBM (the run-length filter of the overlay needs every frame's gate decision before the first write,
    so that path keeps the frames resident instead of streaming them).
    """
    if overlay is None:
        collected: Optional[np.ndarray] = None
        for first, total, frames in projection_file_blend_frame_chunks(
            wav, network_paths, frames_to_visualize, output_fps, output_side_length, alpha, fft_roll_enabled,
            fft_amplitude_range, projection_file_path, blend_depth,
        ):
            if collected is None:
                collected = np.empty((total, *frames.shape[1:]), dtype=np.uint8)
            collected[first : first + len(frames)] = frames
        rank = dist.get_rank() if dist.is_initialized() else 0
        if rank == 0 and collected is None:
            collected = np.empty((0, output_side_length, output_side_length, 3), dtype=np.uint8)
        return collected
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = torch.device("cuda", torch.cuda.current_device())
    networks = MultiNetwork(network_paths=network_paths, load=True)
    try:
        inputs = _BlendInputs(None, None, 0, None, None, 1)
        if rank == 0:
            inputs = _prepare_blend_inputs(
                wav, networks, frames_to_visualize, output_fps, alpha, fft_roll_enabled, fft_amplitude_range,
                projection_file_path, blend_depth, True, device,
            )
        frames = shard_synthesize_gather(
            inputs.dlatents, inputs.indices, inputs.num_frames, networks, output_side_length, device, keep_on_device=True
        )
        if frames is None:
            return None
        blended = apply_eye_tracking_overlay(
            frames, inputs.target_images, inputs.frame_multiplier, overlay, inputs.audio, networks.expected_vector_length
        )
        return blended.cpu().numpy()
    finally:
        networks.unload()


def projection_file_blend_api(  # pylint: disable=too-many-arguments,too-many-locals
    wav: List[str],
    output_path: Optional[str],
    network_paths: List[Path],
    frames_to_visualize: Optional[int],
    output_fps: float,
    output_side_length: int,
    debug_path: Optional[str],
    debug_window: Optional[int],  # pylint: disable=unused-argument
    debug_side_length: Optional[int],  # pylint: disable=unused-argument
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
    projection_file_path: str,
    blend_depth: int,
    complexity_change_rolling_sum_window: Optional[int],
    complexity_change_threshold: Optional[int],
    phash_distance: Optional[int],
    bbox_distance: Optional[float],
    track_length: Optional[int],
) -> None:
    """
    Same parameter list as the reference API (gance/projection_file_blend.py:56-76). Frames are
    written to `output_path` as a `.npy` uint8 array (no video encoder here).
    :raises ValueError: the reference's own checks (music mask without overlay, invalid projection file,
    non-integer fps ratio).
    :raises NotImplementedError: debug video requested (out of scope), or the overlay requested
    without a landmark detector (face_recognition / dlib is not installed; see overlay_eye_tracking).
    """
    overlay_enabled = all(param is not None for param in (phash_distance, bbox_distance, track_length))
    overlay_music_mask_enabled = all(
        param is not None for param in (complexity_change_rolling_sum_window, complexity_change_threshold)
    )
    if overlay_music_mask_enabled and not overlay_enabled:
        raise ValueError("Overlay music mask without overlay being enabled is not supported!")
    if debug_path is not None:
        raise NotImplementedError("the matplotlib debug video is out of scope")
    overlay = (
        OverlayParameters(
            phash_distance, bbox_distance, track_length, complexity_change_rolling_sum_window, complexity_change_threshold
        )
        if overlay_enabled
        else None
    )
    if overlay is None:
        # frame chunks go straight from the pinned ring into the (memory-mapped) output file: nothing holds the video
        writer = None
        for first, total, frames in projection_file_blend_frame_chunks(
            wav, network_paths, frames_to_visualize, output_fps, output_side_length, alpha, fft_roll_enabled,
            fft_amplitude_range, projection_file_path, blend_depth,
        ):
            if output_path is None:
                continue
            if writer is None:
                writer = np.lib.format.open_memmap(output_path, mode="w+", dtype=np.uint8, shape=(total, *frames.shape[1:]))
            writer[first : first + len(frames)] = frames
        if writer is not None:
            writer.flush()
        return
    frames = projection_file_blend_frames(
        wav, network_paths, frames_to_visualize, output_fps, output_side_length, alpha, fft_roll_enabled,
        fft_amplitude_range, projection_file_path, blend_depth, overlay,
    )
    if frames is not None and output_path is not None:
        np.save(output_path, frames)
