"""
projection-file-blend end to end: WAV(s) + projection file + network(s) -> frames.

`projection_file_blend_api` keeps the reference's parameter list and checks
(gance/projection_file_blend.py:56-343) for the synthesis path. What differs, by scope:
* the blend runs on the GPU and its per-frame latent matrices never leave HBM;
* synthesis is batched, every network is resident, and with `torch.distributed` initialised the
  frames are sharded across ranks and gathered in order on rank 0 (gance_amd/frame_sharding.py);
* the resize to `output_side_length` is the HIP bicubic kernel, on the frames still in HBM;
* video encoding (ffmpeg / x264), the eye-tracking overlay and the matplotlib debug video are out
  of scope here: frames are returned / written as a `.npy` uint8 array [N][S][S][3], and asking
  for the overlay or the debug video raises NotImplementedError.
"""

from pathlib import Path
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from gance_amd import divisor, frame_sharding, hip_lib
from gance_amd.data_into_network_visualization import visualization_inputs
from gance_amd.logger_common import LOGGER
from gance_amd.network_interface.network_functions import DEFAULT_MAX_BATCH, TRUNCATION_PSI, MultiNetwork
from gance_amd.projection import projection_file_reader
from gance_amd.vector_sources import music
from gance_amd.vector_sources.vector_sources_common import underlying_length


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
    stream = torch.cuda.current_stream(dlatents.device).cuda_stream
    for start in range(0, dlatents.shape[0], batch):
        stop = min(dlatents.shape[0], start + batch)
        chunk = indices[start:stop]
        frames: Optional[torch.Tensor] = None
        for network_index in np.unique(chunk):
            network = networks._network_at(int(network_index))  # pylint: disable=protected-access
            engine = network.engine
            members = torch.from_numpy(np.nonzero(chunk == network_index)[0] + start).to(dlatents.device)
            side = engine.resolution
            images = torch.empty((len(members), side, side, 3), dtype=torch.uint8, device=dlatents.device)
            if dlatents.dim() == 2:
                selected = dlatents.index_select(0, members).contiguous()
                engine.synthesize_z_device(selected.data_ptr(), len(members), TRUNCATION_PSI, images.data_ptr(), 0, stream)
            else:
                selected = dlatents.index_select(0, members)[:, : engine.num_layers, :].contiguous()
                engine.synthesize_w_device(selected.data_ptr(), len(members), images.data_ptr(), 0, stream)
            if frames is None:
                frames = torch.empty((stop - start, side, side, 3), dtype=torch.uint8, device=dlatents.device)
            frames.index_copy_(0, members - start, images)
        assert frames is not None
        if output_side_length is not None and output_side_length != frames.shape[1]:
            resized = torch.empty(
                (frames.shape[0], output_side_length, output_side_length, 3), dtype=torch.uint8, device=frames.device
            )
            hip_lib.resize_bicubic_u8_device(
                frames.data_ptr(), frames.shape[0], frames.shape[1], resized.data_ptr(), output_side_length, stream
            )
            frames = resized
        yield frames


def shard_synthesize_gather(  # pylint: disable=too-many-arguments
    dlatents: Optional[torch.Tensor],
    indices: Optional[torch.Tensor],
    num_frames: int,
    networks: MultiNetwork,
    output_side_length: int,
    device: torch.device,
) -> Optional[np.ndarray]:
    """
    Rank 0 holds the per-frame network inputs (and `num_frames`); every rank synthesises its
    contiguous share and rank 0 gets the frames back in order (None elsewhere).
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    if world_size > 1:
        count = [num_frames]
        dist.broadcast_object_list(count, src=0)
        num_frames = count[0]
        dlatents = frame_sharding.scatter_latents(dlatents, num_frames, device)
        indices = frame_sharding.scatter_latents(indices, num_frames, device)
    batches = list(synthesize_device_frames(dlatents, indices, networks, output_side_length))
    local = torch.cat(batches) if batches else torch.empty((0, output_side_length, output_side_length, 3), dtype=torch.uint8, device=device)
    frames, _ = frame_sharding.gather_frames(local, num_frames)
    torch.cuda.synchronize(device)
    return frames.cpu().numpy() if frames is not None else None


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
) -> Optional[np.ndarray]:
    """
    The pipeline of `projection_file_blend_api`, returning the frames [N][S][S][3] uint8 (on rank
    0; None on other ranks when running distributed).
    """
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = torch.device("cuda", torch.cuda.current_device())
    networks = MultiNetwork(network_paths=network_paths, load=True)
    try:
        vector_length = networks.expected_vector_length
        dlatents = indices = None
        num_frames = 0
        if rank == 0:
            # the audio -> latent stage has global dependencies over a few MB: once, on rank 0
            with projection_file_reader.load_projection_file(Path(projection_file_path)) as reader:
                final_latents = projection_file_reader.final_latents_matrices_label(reader)
                attributes = reader.projection_attributes
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
            num_frames = int(dlatents.shape[0])
        return shard_synthesize_gather(dlatents, indices, num_frames, networks, output_side_length, device)
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
    :raises NotImplementedError: overlay or debug video requested (next rows).
    """
    overlay_enabled = all(param is not None for param in (phash_distance, bbox_distance, track_length))
    overlay_music_mask_enabled = all(
        param is not None for param in (complexity_change_rolling_sum_window, complexity_change_threshold)
    )
    if overlay_music_mask_enabled and not overlay_enabled:
        raise ValueError("Overlay music mask without overlay being enabled is not supported!")
    if overlay_enabled:
        raise NotImplementedError("the eye-tracking overlay gate is not built yet (DESIGN.md section 9)")
    if debug_path is not None:
        raise NotImplementedError("the matplotlib debug video is out of scope")
    frames = projection_file_blend_frames(
        wav, network_paths, frames_to_visualize, output_fps, output_side_length, alpha, fft_roll_enabled,
        fft_amplitude_range, projection_file_path, blend_depth,
    )
    if frames is not None and output_path is not None:
        np.save(output_path, frames)
