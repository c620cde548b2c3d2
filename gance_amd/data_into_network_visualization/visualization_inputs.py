"""
Audio -> `VisualizationInput`, on the GPU.

`alpha_blend_vectors_max_rms_power_audio` (the `noise-blend` source, reference :94-166) is the
same spectrogram chain blended with the smoothed-noise field of
gance_amd/vector_sources/primatives.py instead of projected latents; its output feeds the
network's z input.

`alpha_blend_projection_file` keeps the reference's signature, argument meaning, return type,
dtypes and error behaviour (gance/data_into_network_visualization/visualization_inputs.py:169-270)
but runs as six HIP kernels (gance_amd/csrc/audio.hip) instead of per-vector Python loops over
scipy / pandas / librosa calls. `alpha_blend_projection_file_device` is the same computation left
in HBM for the frame-sharded synthesis pipeline (no 133 MB float64 `combined` round trip).
"""

import time
from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from gance_amd import hip_lib
from gance_amd.data_into_network_visualization.visualization_common import DataLabel, ResultLayers, VisualizationInput
from gance_amd.vector_sources import primatives, vector_sources_common
from gance_amd.vector_sources.vector_types import ConcatenatedMatrices, ConcatenatedVectors, MatricesLabel, VectorsLabel

LATENT_ROWS = 18  # hard-coded in the reference's concatenate (visualization_inputs.py:241-243)
NOISE_SIGMAS = primatives.Sigmas(across_vectors=50, within_vectors=0)  # visualization_inputs.py:139
NOISE_RANGE = (-4, 4)  # visualization_inputs.py:141
NOISE_INDEX_SAVGOL = (7, 3)  # reduce_vector_rms_rolling_average defaults (vector_reduction.py:102-108), :146-151


class DeviceBlend(NamedTuple):
    """Blend results resident in HBM."""

    dlatents: torch.Tensor  # [N, 18, L] float32: what each frame feeds the network
    network_indices: torch.Tensor  # [N] int32
    blend: hip_lib.Blend  # owner of the intermediates (spectrogram, blend row ...); close() when done


def alpha_blend_projection_file_device(
    final_latents: np.ndarray,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Optional[Tuple[float, float]],
    blend_depth: int,
    time_series_audio_vectors: np.ndarray,
    vector_length: int,
    num_networks: int,
    device: int = 0,
    keep_stages: bool = False,
    timings: Optional[Dict[str, float]] = None,
) -> DeviceBlend:
    """
    Run the blend on `device` and leave the per-frame latent matrices there.
    :param final_latents: (depth, F*L) float32 concatenated final latents; only row 0 is read,
    as in the reference (visualization_inputs.py:220-231).
    :param timings: if given, filled with the wall-clock split of the call in milliseconds: `create_ms` (operator
    tables -- cached per process after the first call -- and the workspace allocation), `h2d_ms` (audio + latent
    row to HBM), `kernels_ms` (the six kernels, stream drained), `check_ms` (the 24-byte read-back of the extrema).
    :raises ValueError: if the frame count is not a multiple of the projected-latent count.
    """
    if vector_length != 512:
        raise ValueError("vector_length must be 512 (the reference's RMS hop is librosa's fixed 512)")
    audio = np.ascontiguousarray(time_series_audio_vectors, dtype=np.float32)
    if audio.ndim != 1:
        audio = np.ascontiguousarray(np.mean(audio, axis=1), dtype=np.float32)  # apply_spectrogram.py:63-66
    num_frames = int(audio.shape[0] / vector_length)
    row0 = np.ascontiguousarray(final_latents[0], dtype=np.float32)
    num_projection = int(row0.shape[0] / vector_length)
    clock = [time.perf_counter()]

    def lap(name: str) -> None:
        now = time.perf_counter()
        if timings is not None:
            timings[name] = (now - clock[0]) * 1e3
        clock[0] = now

    blend = hip_lib.Blend(
        num_frames, num_projection, alpha, fft_roll_enabled, fft_amplitude_range, blend_depth, num_networks,
        vector_length=vector_length, latent_depth=int(final_latents.shape[0]), device=device,
    )
    lap("create_ms")
    cuda = torch.device("cuda", device)
    stream = torch.cuda.current_stream(cuda)
    d_audio = torch.from_numpy(audio).to(cuda)
    d_row0 = torch.from_numpy(row0).to(cuda)
    dlatents = torch.empty((num_frames, int(final_latents.shape[0]), vector_length), dtype=torch.float32, device=cuda)
    indices = torch.empty((num_frames,), dtype=torch.int32, device=cuda)
    if timings is not None:
        stream.synchronize()
    lap("h2d_ms")
    blend.run_device(
        d_audio.data_ptr(), audio.size, d_row0.data_ptr(), dlatents.data_ptr(), indices.data_ptr(),
        debug_stages=keep_stages, stream=stream.cuda_stream,
    )
    stream.synchronize()  # d_audio / d_row0 go out of scope here
    lap("kernels_ms")
    try:
        blend.check_finite()
    except ValueError:
        blend.close()
        raise
    lap("check_ms")
    return DeviceBlend(dlatents, indices, blend)


class DeviceNoiseBlend(NamedTuple):
    """`noise-blend` results resident in HBM."""

    vectors: torch.Tensor  # [N, L] float32: the z vector each frame feeds the network
    noise: torch.Tensor  # [N, L] float32: the min-max scaled noise field
    network_indices: torch.Tensor  # [N] int32
    blend: hip_lib.Blend  # owner of the intermediates; close() when done


def alpha_blend_vectors_max_rms_power_audio_device(  # pylint: disable=too-many-arguments
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Optional[Tuple[float, float]],
    time_series_audio_vectors: np.ndarray,
    vector_length: int,
    num_networks: int,
    device: int = 0,
    keep_stages: bool = False,
) -> DeviceNoiseBlend:
    """
    Run the noise blend on `device` and leave the per-frame z vectors there. The noise field is
    one "projected latent" per frame with a single row, so the blend kernels are the ones
    `alpha_blend_projection_file_device` runs: float32(noise * float32(1 - alpha)) widened, plus
    spectrogram * alpha in float64 (the reference's dtypes, visualization_inputs.py:135-144).
    """
    if vector_length != 512:
        raise ValueError("vector_length must be 512 (the reference's RMS hop is librosa's fixed 512)")
    audio = np.ascontiguousarray(time_series_audio_vectors, dtype=np.float32)
    if audio.ndim != 1:
        audio = np.ascontiguousarray(np.mean(audio, axis=1), dtype=np.float32)  # apply_spectrogram.py:63-66
    num_frames = int(audio.shape[0] / vector_length)
    blend = hip_lib.Blend(
        num_frames, num_frames, alpha, fft_roll_enabled, fft_amplitude_range, 1, num_networks,
        vector_length=vector_length, latent_depth=1, device=device, index_savgol=NOISE_INDEX_SAVGOL,
    )
    try:
        cuda = torch.device("cuda", device)
        noise = primatives.gaussian_data_device(vector_length, num_frames, NOISE_SIGMAS, None, NOISE_RANGE, device)
        d_audio = torch.from_numpy(audio).to(cuda)
        vectors = torch.empty((num_frames, 1, vector_length), dtype=torch.float32, device=cuda)
        indices = torch.empty((num_frames,), dtype=torch.int32, device=cuda)
        stream = torch.cuda.current_stream(cuda)
        blend.run_device(
            d_audio.data_ptr(), audio.size, noise.data_ptr(), vectors.data_ptr(), indices.data_ptr(),
            debug_stages=keep_stages, stream=stream.cuda_stream,
        )
        stream.synchronize()
        blend.check_finite()
    except Exception:
        blend.close()
        raise
    return DeviceNoiseBlend(vectors.reshape(num_frames, vector_length), noise, indices, blend)


def alpha_blend_vectors_max_rms_power_audio(  # pylint: disable=too-many-arguments
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
    time_series_audio_vectors: ConcatenatedVectors,
    vector_length: int,
    network_indices: List[int],
) -> VisualizationInput:
    """
    Drop-in for the reference function of the same name (visualization_inputs.py:94-166): same
    arguments, same NamedTuple back (a_vectors = spectrogram (N*L,) float64, b_vectors = noise
    (N*L,) float32, combined (N*L,) float64, network_indices = ResultLayers with int indices).
    """
    result = alpha_blend_vectors_max_rms_power_audio_device(
        alpha, fft_roll_enabled, fft_amplitude_range, time_series_audio_vectors, vector_length, len(network_indices)
    )
    try:
        spectrogram = ConcatenatedVectors(result.blend.read_stage("final").reshape(-1))
        combined = ConcatenatedVectors(result.blend.read_stage("blend_row").reshape(-1))
        index_smoothed = result.blend.read_stage("index_smoothed")
        indices = result.network_indices.cpu().numpy().astype(int)
        noise = ConcatenatedVectors(result.noise.cpu().numpy().reshape(-1))
    finally:
        result.blend.close()
    return VisualizationInput(
        a_vectors=VectorsLabel(data=spectrogram, vector_length=vector_length, label="Audio Spectrogram"),
        b_vectors=VectorsLabel(data=noise, vector_length=vector_length, label="Gaussian Smoothed Noise"),
        combined=VectorsLabel(data=combined, vector_length=vector_length, label=f"Combined w/ Alpha Blending, a={alpha}"),
        network_indices=ResultLayers(
            result=DataLabel(indices, "Savgol Smoothing Filter (window=7, polyorder=3) Scaled, Quantized"),
            layers=[DataLabel(index_smoothed, "Savgol Smoothing Filter (window=7, polyorder=3)")],
        ),
    )


def alpha_blend_projection_file(  # pylint: disable=too-many-locals
    final_latents_matrices_label: MatricesLabel,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[int, int],
    blend_depth: int,
    time_series_audio_vectors: ConcatenatedVectors,
    vector_length: int,
    network_indices: List[int],
) -> VisualizationInput:
    """
    Drop-in for the reference function of the same name: same arguments, same NamedTuple back
    (a_vectors = rolled spectrogram (N*L,) float64, b_vectors = projected latents (depth, N*L)
    float32, combined (18, N*L) float64, network_indices = ResultLayers with int indices).
    """
    latents = np.asarray(final_latents_matrices_label.data)
    result = alpha_blend_projection_file_device(
        latents, alpha, fft_roll_enabled, fft_amplitude_range, blend_depth, time_series_audio_vectors,
        vector_length, len(network_indices),
    )
    try:
        spectrogram = ConcatenatedVectors(result.blend.read_stage("final").reshape(-1))
        blend_row = result.blend.read_stage("blend_row").reshape(-1)
        index_smoothed = result.blend.read_stage("index_smoothed")
        indices = result.network_indices.cpu().numpy().astype(int)
    finally:
        result.blend.close()
    num_vectors = int(vector_sources_common.underlying_length(spectrogram) / vector_length)
    projected: ConcatenatedMatrices = vector_sources_common.promote_to_matrix_duplicate(
        data=vector_sources_common.duplicate_to_vector_count(
            data=vector_sources_common.demote_to_vector_select(latents, index_to_take=0),
            vector_length=vector_length,
            target_vector_count=num_vectors,
        ),
        target_depth=latents.shape[0],
    )
    combined = np.concatenate(
        (vector_sources_common.promote_to_matrix_duplicate(ConcatenatedVectors(blend_row), blend_depth), projected[blend_depth:LATENT_ROWS])
    )
    return VisualizationInput(
        a_vectors=VectorsLabel(data=spectrogram, vector_length=vector_length, label="Rolled Audio Spectrogram"),
        b_vectors=MatricesLabel(data=projected, vector_length=vector_length, label=final_latents_matrices_label.label),
        combined=MatricesLabel(data=combined, vector_length=vector_length, label=f"Combined w/ Alpha Blending, a={alpha}"),
        network_indices=ResultLayers(
            result=DataLabel(indices, "Savgol Smoothing Filter (window=3, polyorder=2) Scaled, Quantized"),
            layers=[DataLabel(index_smoothed, "Savgol Smoothing Filter (window=3, polyorder=2)")],
        ),
    )
