"""
Array type aliases of the vector pipeline. Names and fields follow
gance/vector_sources/vector_types.py:20-68 so callers written against the reference type-check
and unpack the same way.
"""

from typing import NamedTuple, NewType, Union

import numpy as np

SingleVector = NewType("SingleVector", "np.ndarray[np.float32]")  # (L,)  # type: ignore
ConcatenatedVectors = NewType("ConcatenatedVectors", "np.ndarray[np.float32]")  # (N*L,)  # type: ignore
DividedVectors = NewType("DividedVectors", "np.ndarray[np.float32]")  # (N, L)  # type: ignore
SingleMatrix = NewType("SingleMatrix", "np.ndarray[np.float32]")  # (W, L)  # type: ignore
ConcatenatedMatrices = NewType("ConcatenatedMatrices", "np.ndarray[np.float32]")  # (W, N*L)  # type: ignore
DividedMatrices = NewType("DividedMatrices", "np.ndarray[np.float32]")  # (N, W, L)  # type: ignore


class VectorsLabel(NamedTuple):
    """A concatenated vector array and its display label."""

    data: ConcatenatedVectors
    vector_length: int
    label: str


class MatricesLabel(NamedTuple):
    """A concatenated matrix array and its display label."""

    data: ConcatenatedMatrices
    vector_length: int
    label: str


def is_vector(data: Union[SingleVector, SingleMatrix, np.ndarray]) -> bool:
    """True for 0-D / 1-D input, False for anything with two or more axes."""
    return len(data.shape) < 2
