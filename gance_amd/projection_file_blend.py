"""
projection-file-blend end to end: WAV(s) + projection file + network(s) -> frames.

`projection_file_blend_api` keeps the reference's parameter list and checks
(gance/projection_file_blend.py:56-343) for the synthesis path. What differs, by scope:
* the blend runs on the GPU and its per-frame latent matrices never leave HBM;
* synthesis is batched, every network is resident, and with `torch.distributed` initialised the
  frames are sharded across ranks and gathered in order on rank 0 (gance_amd/frame_sharding.py);
* the resize to `output_side_length` is the HIP bicubic kernel, on the frames still in HBM;
* the eye-tracking overlay runs on rank 0, chunk by chunk, on the gathered frames while they are still in HBM
  (perceptual hashes and the overlay write are HIP kernels; the landmark detector is external); the stream stays
  lazy with the overlay on, as the reference's iterator chain does (gance/projection_file_blend.py:223-275): the
  run-length filter only ever holds back the frames of a run that is still shorter than `track_length`;
* video encoding (ffmpeg / x264) and the matplotlib debug video are out of scope here: frames are
  returned / written as a `.npy` uint8 array [N][S][S][3], and asking for the debug video raises
  NotImplementedError.
"""

import os
import time
from pathlib import Path
from typing import Callable, Dict, Iterator, List, NamedTuple, Optional, Tuple

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist

from gance_amd import divisor, frame_sharding, hip_lib, torch_ops  # noqa: F401  (torch_ops registers torch.ops.gance.*)
from gance_amd.data_into_network_visualization import visualization_inputs
from gance_amd.data_into_network_visualization.visualization_common import DataLabel, ResultLayers
from gance_amd.logger_common import LOGGER
from gance_amd.network_interface.network_functions import TRUNCATION_PSI, MultiNetwork

# frames per engine call and rank of the frame stream: the batch the kernels are tuned for (one block per CU on every
# layer from 64^2 up; 32: -2 %, 16: -8 %); the activation workspace it needs is 51 GB of the 288 GB at 1024^2
DEFAULT_STREAM_BATCH = 64
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
    batch: Optional[int] = None,
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
    batch = batch or networks.max_batch  # (default: full engine calls, the capacity the networks were loaded with)
    for start in range(0, dlatents.shape[0], batch):
        stop = min(dlatents.shape[0], start + batch)
        chunk = indices[start:stop]
        frames = torch.empty((stop - start, out_side, out_side, 3), dtype=torch.uint8, device=dlatents.device)
        for network_index in np.unique(chunk):
            engine = networks._network_at(int(network_index)).engine  # pylint: disable=protected-access
            members = torch.from_numpy(np.nonzero(chunk == network_index)[0] + start).to(dlatents.device)
            stream = torch.cuda.current_stream(dlatents.device).cuda_stream
            if dlatents.dim() == 2:
                # the reference's vector path draws fresh noise per call, a plane per layer and sample (upstream default)
                engine.randomize_noise(count=int(members.numel()), stream=stream)
                images = torch.ops.gance.synthesize_z(dlatents.index_select(0, members), engine.op_handle, TRUNCATION_PSI)
            else:
                engine.restore_noise(stream=stream)  # its matrix path passes randomize_noise=False: the stored buffers
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
    batch: Optional[int] = None,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """
    The same frames as `synthesize_device_frames`, as one tensor [n, S, S, 3] (`out`, if given), computed network by
    network: every engine call is a full batch however often the index switches (the reference
    sorts by network for the same reason, network_visualization.py:653-674: there a switch costs a
    process restart, here only a short batch). Frames land at their own positions, so the order of
    the result is the frame order.
    """
    device = dlatents.device
    # (a host array is taken as it is: the stream hands slices of one host copy, not a device-to-host sync per window)
    indices = network_indices if isinstance(network_indices, np.ndarray) else network_indices.cpu().numpy()
    num_frames = int(dlatents.shape[0])
    out_side = _common_output_side(networks, indices, output_side_length)
    batch = batch or networks.max_batch  # (default: full engine calls, the capacity the networks were loaded with)
    if out is None:
        out = torch.empty((num_frames, out_side, out_side, 3), dtype=torch.uint8, device=device)
    elif tuple(out.shape) != (num_frames, out_side, out_side, 3) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError(f"`out` must be a contiguous uint8 tensor of shape {(num_frames, out_side, out_side, 3)}")
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
            # (only a scattered call needs its frame numbers on the device; from pinned memory, without blocking: a pageable
            # copy is synchronous IN STREAM ORDER, i.e. the host would wait for the previous engine call before it could
            # issue this one, and the GPU would idle for the launch sequence of every call)
            members = None
            if not in_place:
                members = torch.from_numpy(host_members)
                members = (members.pin_memory() if device.type == "cuda" else members).to(device, non_blocking=True)
            native = out[first : first + count] if in_place and out_side == side else torch.empty(
                (count, side, side, 3), dtype=torch.uint8, device=device
            )
            stream = torch.cuda.current_stream(device).cuda_stream
            if dlatents.dim() == 2:
                # the reference's vector path draws fresh noise per call, a plane per layer and sample (upstream default)
                engine.randomize_noise(count=count, stream=stream)
                selected = dlatents[first : first + count] if in_place else dlatents.index_select(0, members)
                torch.ops.gance.synthesize_z_out(selected, engine.op_handle, TRUNCATION_PSI, native)
            else:
                engine.restore_noise(stream=stream)  # its matrix path passes randomize_noise=False: the stored buffers
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
    timings: Optional[Dict[str, float]] = None,
) -> _BlendInputs:
    """Rank 0: projection file + WAV -> per-frame latent matrices and network indices in HBM (the reference's checks included)."""
    vector_length = networks.expected_vector_length
    target_images = None
    clock = time.perf_counter()
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
    if timings is not None:
        timings["read_projection_file_ms"] = (time.perf_counter() - clock) * 1e3
        clock = time.perf_counter()
    audio = music.read_wavs_scale_for_video(
        wavs=[Path(path) for path in wav], vector_length=vector_length, target_num_vectors=num_output_frames
    ).wav_data
    if timings is not None:
        timings["read_and_stretch_wav_ms"] = (time.perf_counter() - clock) * 1e3
        clock = time.perf_counter()
    blend_timings: Optional[Dict[str, float]] = {} if timings is not None else None
    blend = visualization_inputs.alpha_blend_projection_file_device(
        final_latents.data, alpha, fft_roll_enabled, fft_amplitude_range, blend_depth, audio, vector_length,
        len(networks.network_indices), device=device.index, timings=blend_timings,
    )
    dlatents, indices = blend.dlatents, blend.network_indices
    blend.blend.close()
    if timings is not None:
        timings["audio_to_latents_ms"] = (time.perf_counter() - clock) * 1e3
        timings["audio_to_latents_split_ms"] = blend_timings
    if frames_to_visualize is not None:
        dlatents, indices = dlatents[:frames_to_visualize], indices[:frames_to_visualize]
    return _BlendInputs(dlatents, indices, int(dlatents.shape[0]), target_images, audio, int(frame_multiplier))


# Pieces per network in a window of the multi-network stream: a window of w * networks pieces ends in up to `networks` short
# engine calls, so the loss against full batches shrinks as 1 / w; the window's frames wait in HBM (uint8, 3 MiB each at
# 1024^2: 768 frames = 2.4 GB with three networks at 4 pieces per network). GANCE_STREAM_WINDOW_PIECES overrides.
STREAM_WINDOW_PIECES_PER_NETWORK = int(os.environ.get("GANCE_STREAM_WINDOW_PIECES", "4"))


class _WindowSynthesizer:  # pylint: disable=too-few-public-methods
    """
    `synthesize_piece(offset, count)` of the frame stream, batched by network ACROSS a window of pieces: the rank's
    frames [offset, offset + window * frames_per_call) are synthesised network by network into one buffer the
    first time a piece of the window is asked for, and the pieces are served from it. With several networks and an
    RMS-driven index that switches every few frames a single piece splits into as many short engine calls as it has
    networks; over a window the calls stay (nearly) full batches -- what the reference's sort-by-network does over
    the whole video (network_visualization.py:653-674), bounded to `window` pieces of HBM.
    """

    def __init__(self, dlatents: torch.Tensor, indices: torch.Tensor, networks: MultiNetwork, side: int, frames_per_call: int, window: int) -> None:
        self._dlatents, self._indices, self._networks, self._side = dlatents, indices.cpu().numpy(), networks, side
        self._frames_per_call, self._window = frames_per_call, max(1, window)
        self._start, self._frames = 0, None

    writes_into = True  # (frame_sharding.ordered_device_chunks: the frames go straight into the stream's buffer)

    def __call__(self, offset: int, count: int, out: torch.Tensor) -> None:
        if self._window == 1:  # one network: every engine call writes its frames in place
            synthesize_device_frames_network_major(
                self._dlatents[offset : offset + count], self._indices[offset : offset + count], self._networks, self._side, self._frames_per_call, out=out
            )
            return
        if self._frames is None or not self._start <= offset < self._start + int(self._frames.shape[0]):
            stop = min(int(self._dlatents.shape[0]), offset + self._window * self._frames_per_call)
            self._start = offset
            self._frames = synthesize_device_frames_network_major(
                self._dlatents[offset:stop], self._indices[offset:stop], self._networks, self._side, self._frames_per_call
            )
        out.copy_(self._frames[offset - self._start : offset - self._start + count])


def decided_prefix(gated: List[bool], track_length: int, final: bool) -> int:
    """
    How many leading frames of a gate sequence seen SO FAR already have their final `track_length_filter` value
    (vector_reduction.py:261-274: runs of True shorter than `track_length` become False): all of them, unless the
    sequence ends in a run of True that is still shorter than `track_length` and may yet grow -- then everything
    before that run. At the end of the stream (`final`) a short trailing run is judged by the length it has.
    """
    known = len(gated)
    if final:
        return known
    run = 0
    while run < known and gated[known - 1 - run]:
        run += 1
    return known if run >= track_length or run == 0 else known - run


class _StreamingOverlay:
    """
    The overlay stage of gance/projection_file_blend.py:181-275 as a stage of the frame stream on rank 0: chunks of
    synthesized frames go in (in HBM), chunks with the eye regions written come out, in order, as soon as every frame
    of a chunk is DECIDED. The gate (landmarks, box distance, perceptual hashes) is evaluated per chunk as it arrives;
    `track_length_filter` (vector_reduction.py:261-274) drops runs of gated frames shorter than `track_length`, so a
    gated frame is undecided only while its run is still open and shorter than that: the stage holds back at most the
    chunks such a run spans (track_length - 1 frames of look-ahead), never the video.
    """

    def __init__(  # pylint: disable=too-many-arguments
        self, target_images: np.ndarray, frame_multiplier: int, parameters: OverlayParameters, skip_mask: List[bool], num_frames: int, side: int,
        device: torch.device,
    ) -> None:
        self._targets, self._multiplier, self._parameters = target_images, frame_multiplier, parameters
        self._skip_mask, self._num_frames, self._side, self._device = skip_mask, num_frames, side, device
        self._gated: List[bool] = []  # per frame seen so far: not skipped and the gate passed
        self._pending: List[Tuple[int, torch.Tensor, torch.Tensor, list]] = []  # (first, background, foreground, boxes per frame)
        self.chunks_held_max = 0
        self.overlays_written = 0
        if -(-num_frames // frame_multiplier) > len(target_images):
            raise ValueError("the projection file holds too few target images for the frames being written")

    def _foreground(self, first: int, count: int) -> torch.Tensor:
        """Target images of frames [first, first + count): each shown `frame_multiplier` times, scaled to the output side."""
        lo, hi = first // self._multiplier, (first + count - 1) // self._multiplier + 1
        targets = torch.from_numpy(np.ascontiguousarray(self._targets[lo:hi])).to(self._device)
        if targets.shape[1] != self._side:
            targets = torch.ops.gance.resize_bicubic(targets, self._side)
        repeated = targets.repeat_interleave(self._multiplier, dim=0)
        skip = first - lo * self._multiplier
        return repeated[skip : skip + count].contiguous()

    def _decided(self, final: bool) -> int:
        return decided_prefix(self._gated, self._parameters.track_length, final)

    def _release(self, final: bool) -> List[Tuple[int, torch.Tensor]]:
        decided = self._decided(final)
        # the filter is local to runs: on the frames known so far it already gives every decided frame its final value
        keep = vector_reduction.track_length_filter(bool_tracks=self._gated, track_length=self._parameters.track_length)
        out = []
        while self._pending and self._pending[0][0] + int(self._pending[0][1].shape[0]) <= decided:
            first, background, foreground, boxes_list = self._pending.pop(0)
            written = [boxes if keep[first + i] else None for i, boxes in enumerate(boxes_list)]
            self.overlays_written += sum(boxes is not None for boxes in written)
            out.append((first, overlay_common.write_boxes_onto_frames_device(foreground, background, written)))
        return out

    def push(self, first: int, frames: torch.Tensor) -> List[Tuple[int, torch.Tensor]]:
        """One gathered chunk in (a view that is only valid now: it is copied), the chunks that became ready out."""
        count = int(frames.shape[0])
        background = frames.clone()
        foreground = self._foreground(first, count)
        skips = self._skip_mask[first : first + count]
        result = overlay_eye_tracking.compute_eye_tracking_overlay(
            foreground_images=foreground, background_images=background, min_phash_distance=self._parameters.phash_distance,
            min_bbox_distance=self._parameters.bbox_distance, skip_mask=skips, face_finder=self._parameters.face_finder,
        )
        boxes_list = list(result.bbox_lists)
        self._gated.extend(not skip and boxes is not None for skip, boxes in zip(skips, boxes_list))
        self._pending.append((first, background, foreground, boxes_list))
        self.chunks_held_max = max(self.chunks_held_max, len(self._pending))
        return self._release(final=False)

    def flush(self) -> List[Tuple[int, torch.Tensor]]:
        """End of the stream: a run still open is judged by the length it has."""
        return self._release(final=True)


def projection_file_blend_frame_chunks(  # pylint: disable=too-many-arguments,too-many-locals,too-many-statements
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
    frames_per_call: int = DEFAULT_STREAM_BATCH,
    overlay: Optional[OverlayParameters] = None,
    networks: Optional[MultiNetwork] = None,
    timings: Optional[Dict[str, object]] = None,
    drain: str = "rank0",
    on_total: Optional[Callable[[int], None]] = None,
) -> Iterator[Tuple[int, int, np.ndarray]]:
    """
    The frame stream of the reference's pipeline (gance/projection_file_blend.py:343 hands an iterator of frames to
    the video writer): a generator of (first_frame_index, total_frames, frames [n, S, S, 3] uint8) in frame order.
    Nothing holds all frames: per chunk, every rank synthesises `frames_per_call` frames, one gather lands the chunk
    in order on rank 0 while the next chunk is already being synthesised, rank 0 runs the eye-tracking overlay on it
    if `overlay` is given (in HBM; see _StreamingOverlay) and drains it to a pinned host ring (`frames` is a view of
    a ring slot: consume or copy it before advancing the generator twice more). With several networks the engine
    calls are batched by network across a window of pieces (_WindowSynthesizer).
    Collective under `torch.distributed`: every rank must exhaust the generator; only rank 0 receives chunks. If a rank
    fails, every rank leaves the generator with an exception (frame_sharding.exchange_status).
    :param networks: networks already resident (not unloaded at the end); default: load `network_paths`, unload after.
    :param timings: if given, rank 0 records the wall-clock split of the call there (milliseconds).
    :param drain: "rank0" (default): the RCCL gather lands every chunk on rank 0, which alone receives chunks -- for
    consumers of the ordered stream in one place (the overlay stage; an encoder fed by rank 0). "per-rank": no gather; EVERY
    rank receives its own pieces (first_frame_index of the piece, total, frames) and drains them over its own PCIe link
    into its own pinned ring -- for host-bound legs: at 2160^2 a frame is 14 MB, one link sustains ~57 GB/s pinned,
    i.e. ~4 000 frames/s through rank 0 alone, less than four GPUs synthesise. Not with `overlay`
    (its run-length filter needs the ordered stream in one place).
    :param on_total: called once on every rank with the total frame count as soon as it is known, rank 0 first and the
    other ranks after rank 0's call has returned (so rank 0 can create an output file the others then open).
    """
    if drain not in frame_sharding.DRAIN_MODES:
        raise ValueError(f"drain must be one of {frame_sharding.DRAIN_MODES}, got {drain!r}")
    if drain == "per-rank" and overlay is not None:
        raise ValueError("the eye-tracking overlay needs the ordered stream on rank 0: drain=\"rank0\"")
    rank = dist.get_rank() if dist.is_initialized() else 0
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    device = torch.device("cuda", torch.cuda.current_device())
    own_networks = networks is None
    if own_networks:
        networks = MultiNetwork(network_paths=network_paths, load=True, max_batch=frames_per_call)
    try:
        inputs = _BlendInputs(None, None, 0, None, None, 1)
        failure = None
        if rank == 0:
            try:
                inputs = _prepare_blend_inputs(
                    wav, networks, frames_to_visualize, output_fps, alpha, fft_roll_enabled, fft_amplitude_range,
                    projection_file_path, blend_depth, overlay is not None, device, timings,
                )
            except Exception as error:  # pylint: disable=broad-except
                failure = error
        try:  # (a bad projection file or WAV on rank 0 must not leave the other ranks waiting in the broadcast below)
            frame_sharding.exchange_status(failure is not None, "preparing the blend")
        except frame_sharding.StreamRankError:
            if failure is None:
                raise
        if failure is not None:
            raise failure
        num_frames = inputs.num_frames
        if not frame_sharding.single_process():
            count = [num_frames]
            dist.broadcast_object_list(count, src=0)
            num_frames = count[0]
        if on_total is not None:
            # rank 0 first (it creates the output), then the others (they open it); a failure on ANY rank is relayed before the
            # stream's collectives start: a rank that cannot map the shared output must not leave the others in the scatter
            for turn_of_rank_0 in (True, False):
                failure = None
                if (rank == 0) == turn_of_rank_0:
                    try:
                        on_total(num_frames)
                    except Exception as error:  # pylint: disable=broad-except
                        failure = error
                try:
                    frame_sharding.exchange_status(failure is not None, "preparing the output" if turn_of_rank_0 else "opening the output")
                except frame_sharding.StreamRankError:
                    if failure is None:
                        raise
                if failure is not None:
                    raise failure
                if world_size == 1:
                    break
        clock = time.perf_counter()
        dlatents = frame_sharding.scatter_for_stream(inputs.dlatents, num_frames, frames_per_call, device)
        indices = frame_sharding.scatter_for_stream(inputs.indices, num_frames, frames_per_call, device)
        side = _common_output_side(networks, np.asarray(networks.network_indices), output_side_length)
        num_networks = len(set(networks.network_paths))
        synthesize_piece = _WindowSynthesizer(dlatents, indices, networks, side, frames_per_call, 1 if num_networks == 1 else STREAM_WINDOW_PIECES_PER_NETWORK * num_networks)
        stage = None
        if overlay is not None and rank == 0:
            music_mask = overlay.complexity_change_rolling_sum_window is not None and overlay.complexity_change_threshold is not None
            skip_mask = (
                music_complexity_skip_mask(
                    inputs.audio, networks.expected_vector_length, overlay.complexity_change_rolling_sum_window, overlay.complexity_change_threshold
                )[:num_frames]
                if music_mask
                else [False] * num_frames
            )
            stage = _StreamingOverlay(inputs.target_images, inputs.frame_multiplier, overlay, skip_mask, num_frames, side, device)
        ring = None
        bytes_to_host = 0
        for first, frames, reader_stream in frame_sharding.ordered_device_chunks(
            synthesize_piece, num_frames, frames_per_call, (side, side, 3), device, drain=drain
        ):
            if ring is None:
                ring = frame_sharding.HostRing((1 if drain == "per-rank" else world_size) * frames_per_call, (side, side, 3), device, slots=3)
            if stage is None:
                ready = [(first, frames)]
            else:
                with torch.cuda.stream(reader_stream):  # (the chunk view may only be read on the reader stream)
                    ready = stage.push(first, frames)
            for ready_first, ready_frames in ready:
                done = ring.push(ready_first, ready_frames, reader_stream)
                bytes_to_host += ready_frames.numel()
                if done is not None:
                    yield done[0], num_frames, done[1]
        if stage is not None:
            with torch.cuda.stream(reader_stream):
                ready = stage.flush()
            for ready_first, ready_frames in ready:
                done = ring.push(ready_first, ready_frames, reader_stream)
                bytes_to_host += ready_frames.numel()
                if done is not None:
                    yield done[0], num_frames, done[1]
            LOGGER.info(f"Eye tracking overlay written on {stage.overlays_written} of {num_frames} frames")
        if ring is not None:
            last = ring.flush()
            if last is not None:
                yield last[0], num_frames, last[1]
        torch.cuda.synchronize(device)
        if timings is not None and (rank == 0 or drain == "per-rank"):
            elapsed = time.perf_counter() - clock
            timings["synthesis_to_host_ms"] = elapsed * 1e3
            timings["frames"] = num_frames
            timings["bytes_to_host"] = bytes_to_host
            timings["d2h_gb_per_s"] = bytes_to_host / elapsed / 1e9 if elapsed > 0 else None
            if stage is not None:
                timings["overlay_chunks_held_max"] = stage.chunks_held_max
                timings["overlays_written"] = stage.overlays_written
    finally:
        if own_networks:
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
    frames_per_call: int = DEFAULT_STREAM_BATCH,
) -> Optional[np.ndarray]:
    """
    The pipeline of `projection_file_blend_api`, returning ALL frames [N][S][S][3] uint8 at once (on rank 0;
    None on other ranks when running distributed): the stream of `projection_file_blend_frame_chunks` (overlay
    included) collected into one host array.
    """
    collected: Optional[np.ndarray] = None
    for first, total, frames in projection_file_blend_frame_chunks(
        wav, network_paths, frames_to_visualize, output_fps, output_side_length, alpha, fft_roll_enabled,
        fft_amplitude_range, projection_file_path, blend_depth, frames_per_call=frames_per_call, overlay=overlay,
    ):
        if collected is None:
            collected = np.empty((total, *frames.shape[1:]), dtype=np.uint8)
        collected[first : first + len(frames)] = frames
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == 0 and collected is None:
        collected = np.empty((0, output_side_length, output_side_length, 3), dtype=np.uint8)
    return collected


def _npy_path(output_path: str) -> str:
    """`np.save` appends `.npy` when the suffix is missing: the streamed writer lands in the same file."""
    return output_path if output_path.endswith(".npy") else output_path + ".npy"


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
    drain: Optional[str] = None,
) -> None:
    """
    Same parameter list as the reference API (gance/projection_file_blend.py:56-76). Frames are
    written to `output_path` as a `.npy` uint8 array (no video encoder here).
    :raises ValueError: the reference's own checks (music mask without overlay, invalid projection file,
    non-integer fps ratio).
    :raises NotImplementedError: debug video requested (out of scope), or the overlay requested
    without a landmark detector (face_recognition / dlib is not installed; see overlay_eye_tracking).
    :param drain: (not a parameter of the reference) how frames reach the output file under `torch.distributed`:
    "rank0" = gathered over RCCL and written by rank 0; "per-rank" = every rank writes the pieces it synthesised into
    the same memory-mapped file over its own PCIe link. Default: GANCE_STREAM_DRAIN, else "rank0"; the overlay forces "rank0".
    """
    drain = drain or os.environ.get("GANCE_STREAM_DRAIN", "rank0")
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
    if overlay is not None:
        drain = "rank0"
    # frame chunks go straight from the pinned ring into the (memory-mapped) output file: nothing holds the video
    rank = dist.get_rank() if dist.is_initialized() else 0
    state: Dict[str, object] = {"writer": None, "total": 0}

    def open_output(total: int) -> None:
        # rank 0 creates the file (the frame side is known up front: every frame is resized to output_side_length); with
        # drain="per-rank" the other ranks then map the same file and write their own pieces into it
        state["total"] = total
        if output_path is None or (rank != 0 and drain != "per-rank"):
            return
        if rank == 0:
            state["writer"] = np.lib.format.open_memmap(
                _npy_path(output_path), mode="w+", dtype=np.uint8, shape=(total, output_side_length, output_side_length, 3)
            )
        elif total > 0:
            state["writer"] = np.load(_npy_path(output_path), mmap_mode="r+")

    for first, _total, frames in projection_file_blend_frame_chunks(
        wav, network_paths, frames_to_visualize, output_fps, output_side_length, alpha, fft_roll_enabled,
        fft_amplitude_range, projection_file_path, blend_depth, overlay=overlay, drain=drain, on_total=open_output,
    ):
        if state["writer"] is not None:
            state["writer"][first : first + len(frames)] = frames
    if state["writer"] is not None:
        state["writer"].flush()
