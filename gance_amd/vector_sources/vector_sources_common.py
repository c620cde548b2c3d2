"""
Array plumbing on "concatenated vectors" (no arithmetic: pure reshapes, repeats and gathers).
Same names and semantics as gance/vector_sources/vector_sources_common.py (line numbers cited per
function). The arithmetic members (Savitzky-Golay smoothing, Fourier resampling, range remap) run on
the GPU through the `gance_vec_*` entry points of libgance_hip (gance_amd/csrc/audio.hip), one stage per
call; the hot path runs the same arithmetic fused inside `gance_blend_run`.
"""

from typing import Iterable, List, Tuple, Union

import numpy as np

from gance_amd import divisor, hip_lib
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


def remap_values_into_range(
    data: Iterable[Union[float, int]],
    input_range: Tuple[Union[float, int], Union[float, int]],
    output_range: Tuple[Union[float, int], Union[float, int]],
) -> List[float]:
    """
    Linear map of values from one range into another (vsc:44-61: `interp1d(input_range, output_range)`
    applied per value by a multiprocessing Pool; here one kernel).
    :raises ValueError: a value lies outside `input_range` (interp1d's bounds error).
    """
    values = np.asarray(list(data), dtype=np.float64)
    lo, hi = float(min(input_range)), float(max(input_range))
    if values.size and (values.min() < lo or values.max() > hi):
        raise ValueError("A value in x_new is outside the interpolation range.")
    if values.size == 0:
        return []
    return list(hip_lib.vec_remap(values, input_range, output_range))


def smooth_across_vectors(data: ConcatenatedVectors, vector_length: int, window_length: int = 7, polyorder: int = 3) -> ConcatenatedVectors:
    """
    Savitzky-Golay along TIME for each position of the vector (vsc:136-166): makes one vector similar to
    the next. `scipy.signal.savgol_filter` defaults (mode "interp").
    :raises ValueError: scipy's own argument errors (window longer than the number of vectors, ...).
    """
    vectors = np.ascontiguousarray(sub_vectors(data, vector_length), dtype=np.float64)
    return ConcatenatedVectors(hip_lib.vec_savgol(vectors, 0, window_length, polyorder).reshape(-1))


def smooth_each_vector(data: ConcatenatedVectors, vector_length: int, window_length: int = 51, polyorder: int = 2) -> ConcatenatedVectors:
    """Savitzky-Golay within each vector, vectors independent of one another (vsc:169-188)."""
    vectors = np.ascontiguousarray(sub_vectors(data, vector_length), dtype=np.float64)
    return ConcatenatedVectors(hip_lib.vec_savgol(vectors, 1, window_length, polyorder).reshape(-1))


def scale_vectors_to_length_resample(data: ConcatenatedVectors, original_vector_length: int, output_vector_length: int) -> ConcatenatedVectors:
    """Every vector Fourier-resampled (`scipy.signal.resample`) to a new length (vsc:211-230)."""
    vectors = np.ascontiguousarray(sub_vectors(data, original_vector_length), dtype=np.float64)
    return ConcatenatedVectors(hip_lib.vec_fourier_resample(vectors, output_vector_length).reshape(-1))


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
