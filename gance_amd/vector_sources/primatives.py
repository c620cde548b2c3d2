"""
The smoothed-noise vector source: `gaussian_data`, the random-z source of
`noise-blend` and of the plumbing benchmark configs.

Same name, arguments, defaults, seed and output contract as gance/vector_sources/primatives.py:17,
36-74. The standard-normal draws come from the caller's `np.random.RandomState` (numpy's MT19937
stream is the contract: seed 1234 gives the reference's numbers); the Gaussian filter, the RMS
normalisation and (for `gaussian_data_device`) the min-max scaling run as HIP kernels
(gance_amd/csrc/noise.hip) instead of scipy.ndimage on the host. The reference's demo shapes
(line sweeps, square / sine waves) are out of scope.
"""

from typing import NamedTuple, Optional, Tuple

import numpy as np
import torch

from gance_amd import hip_lib
from gance_amd.vector_sources.vector_types import ConcatenatedVectors

DEFAULT_RANDOM_SEED = 1234


class Sigmas(NamedTuple):
    """Smoothing strengths of the noise field (primatives.py:36-46)."""

    # How alike one point will be to the same point in the following vector.
    across_vectors: int
    # How alike neighbouring points of one vector will be.
    within_vectors: int


def gaussian_data_device(  # pylint: disable=too-many-arguments
    vector_length: int,
    num_vectors: int,
    sigmas: Sigmas = Sigmas(20, 0),
    random_state: Optional["np.random.RandomState"] = None,
    feature_range: Optional[Tuple[float, float]] = None,
    device: int = 0,
) -> torch.Tensor:
    """
    The noise field as a [num_vectors, vector_length] float32 tensor left in HBM, optionally
    min-max scaled to `feature_range` (what `noise-blend` does next with it,
    visualization_inputs.py:135-142) in the same pass.
    """
    if random_state is None:
        random_state = np.random.RandomState(DEFAULT_RANDOM_SEED)  # pylint: disable=no-member
    draws = random_state.randn(num_vectors, 1, vector_length).astype(np.float32)
    cuda = torch.device("cuda", device)
    with torch.cuda.device(cuda):
        d_draws = torch.from_numpy(draws.reshape(num_vectors, vector_length)).to(cuda)
        d_field = torch.empty_like(d_draws)
        hip_lib.gaussian_noise_device(
            d_draws.data_ptr(), num_vectors, vector_length, sigmas.across_vectors, sigmas.within_vectors, feature_range,
            d_field.data_ptr(), torch.cuda.current_stream(cuda).cuda_stream,
        )
    return d_field


def gaussian_data(
    vector_length: int,
    num_vectors: int,
    sigmas: Sigmas = Sigmas(20, 0),
    random_state: Optional["np.random.RandomState"] = None,
) -> ConcatenatedVectors:
    """
    Gaussian field data shaped like WAV data: `num_vectors` vectors of `vector_length` float32
    samples, smoothed across / within vectors, unit RMS, concatenated.
    :return: A (vector_length * num_vectors,) float32 array.
    """
    field = gaussian_data_device(vector_length, num_vectors, sigmas, random_state)
    return ConcatenatedVectors(field.cpu().numpy().reshape(vector_length * num_vectors))
