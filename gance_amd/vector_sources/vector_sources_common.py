"""
Array plumbing on "concatenated vectors" (no arithmetic: pure reshapes, repeats and gathers).
Same names and semantics as gance/vector_sources/vector_sources_common.py (line numbers cited per
function). The arithmetic members of that module (savgol smoothing, Fourier resampling) are not
re-exposed one vector at a time: on this path they run fused inside libgance_hip's blend kernels
(gance_amd/csrc/audio.hip).
"""

from typing import Union

import numpy as np

from gance_amd import divisor
from gance_amd.vector_sources.vector_types import (
    ConcatenatedMatrices,
    ConcatenatedVectors,
    DividedMatrices,
    DividedVectors,
    SingleMatrix,
    SingleVector,
    is_vector,
)


def pad_array(array: np.ndarray, size: int) -> np.ndarray:
    """Zero-pad a 1-D array at the end to `size` (vsc:32-41)."""
    return np.concatenate([array, np.zeros(size - len(array), dtype=array.dtype)])


def sub_vectors(
    data: Union[ConcatenatedMatrices, ConcatenatedVectors], vector_length: int
) -> Union[DividedMatrices, DividedVectors]:
    """(N*L,) -> (N, L); (W, N*L) -> (N, W, L)  (vsc:86-101)."""
    if len(data.shape) >= 2:
        count = int(data.shape[-1] / vector_length)
        return DividedMatrices(np.stack(np.split(data, count, axis=-1)))
    return DividedVectors(np.reshape(data, (-1, vector_length)))


def underlying_length(data: Union[SingleVector, SingleMatrix, ConcatenatedVectors, ConcatenatedMatrices]) -> int:
    """Length of a vector, or of the vectors inside a matrix (vsc:124-133)."""
    return int(data.shape[0] if is_vector(data) else data.shape[1])


def duplicate_to_vector_count(
    data: ConcatenatedVectors, vector_length: int, target_vector_count: int
) -> ConcatenatedVectors:
    """
    Repeat every vector the same whole number of times (vsc:298-345).
    :raises ValueError: if target_vector_count is not a multiple of the vector count.
    """
    vectors = sub_vectors(data, vector_length)
    try:
        factor = divisor.divide_no_remainder(numerator=target_vector_count, denominator=len(vectors))
    except ValueError as error:
        raise ValueError(
            f"Cannot duplicate the input vectors (count {len(vectors)}) to the desired count {target_vector_count}."
        ) from error
    return ConcatenatedVectors(np.repeat(vectors, factor, axis=0).reshape(-1))


def promote_to_matrix_duplicate(data: ConcatenatedVectors, target_depth: int) -> ConcatenatedMatrices:
    """(N*L,) -> (target_depth, N*L) by row duplication (vsc:348-365)."""
    if len(data.shape) != 1:
        raise ValueError("Undefined behavior!")
    return ConcatenatedMatrices(np.tile(data, (target_depth, 1)))


def demote_to_vector_select(
    data: Union[SingleMatrix, ConcatenatedMatrices], index_to_take: int = 0
) -> Union[SingleVector, ConcatenatedVectors]:
    """Take one row of a matrix (vsc:380-394)."""
    return ConcatenatedVectors(data[index_to_take])


def rotate_vectors_over_time(
    data: Union[ConcatenatedVectors, ConcatenatedMatrices], vector_length: int, roll_values: np.ndarray
) -> np.ndarray:
    """
    out_t = np.roll(in_t, -cumsum(roll)_t) (vsc:408-428), as one integer gather. Host helper for
    callers outside the fused blend; the blend kernel does the same gather on the GPU.
    """
    vectors = sub_vectors(data, vector_length)
    shift = np.cumsum(np.asarray(roll_values, dtype=np.int64))
    index = (np.arange(vector_length, dtype=np.int64) + shift[:, None]) % vector_length
    if vectors.ndim == 3:  # (N, W, L): the reference rolls the flattened (W, L) block
        flat = vectors.reshape(len(vectors), -1)
        width = flat.shape[1]
        index = (np.arange(width, dtype=np.int64) + shift[:, None]) % width
        return np.concatenate(list(np.take_along_axis(flat, index, axis=1).reshape(vectors.shape)))
    return np.take_along_axis(vectors, index, axis=1).reshape(-1)
