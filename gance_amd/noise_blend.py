"""
noise-blend end to end: WAV(s) + network(s) -> frames, the vector-input twin of
projection-file-blend (music_into_networks.py:285-401 in the reference).

Same stages as the reference command: read the WAVs stretched to one 512-sample vector per output
frame (FPS mode), blend the spectrogram with the smoothed-noise field
(`alpha_blend_vectors_max_rms_power_audio`), feed every blended vector to the network the
rolling RMS picks, through the z entry (mapping + truncation psi 1.2 + synthesis), and scale to
`output_side_length`. Everything after the WAV read stays in HBM; with `torch.distributed`
initialised the frames are sharded across ranks exactly as in projection_file_blend.py. Video
encoding and the matplotlib debug video are out of scope: frames come back as a uint8 array.
"""

from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from gance_amd.data_into_network_visualization import visualization_inputs
from gance_amd.network_interface.network_functions import MultiNetwork
from gance_amd.projection_file_blend import shard_synthesize_gather
from gance_amd.vector_sources import music


def noise_blend_frames(  # pylint: disable=too-many-arguments
    wav: List[str],
    network_paths: List[Path],
    frames_to_visualize: Optional[int],
    output_fps: float,
    output_side_length: int,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
) -> Optional[np.ndarray]:
    """
    The synthesis pipeline of the reference's `noise_blend` command with its parameter names.
    :return: frames [N][S][S][3] uint8 on rank 0; None on other ranks when running distributed.
    """
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = torch.device("cuda", torch.cuda.current_device())
    networks = MultiNetwork(network_paths=network_paths, load=True)
    try:
        vector_length = networks.expected_vector_length
        vectors = indices = None
        num_frames = 0
        if rank == 0:
            audio = music.read_wavs_scale_for_video(
                wavs=[Path(path) for path in wav], vector_length=vector_length, frames_per_second=output_fps
            ).wav_data
            blend = visualization_inputs.alpha_blend_vectors_max_rms_power_audio_device(
                alpha, fft_roll_enabled, fft_amplitude_range, audio, vector_length, len(networks.network_indices),
                device=device.index,
            )
            vectors, indices = blend.vectors, blend.network_indices
            blend.blend.close()
            if frames_to_visualize is not None:
                vectors, indices = vectors[:frames_to_visualize], indices[:frames_to_visualize]
            num_frames = int(vectors.shape[0])
        return shard_synthesize_gather(vectors, indices, num_frames, networks, output_side_length, device)
    finally:
        networks.unload()
