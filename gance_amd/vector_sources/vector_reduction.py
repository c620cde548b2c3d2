"""
Per-frame scalar reductions that steer the overlay (host logic over N-element series).

Same names and results as gance/vector_sources/vector_reduction.py for `track_length_filter`
(:261-273), `rolling_sum_results_layers` (:243-258), `absolute_value_results_layers` (:227-240),
`derive_results_layers` (:210-224) and `reduce_vector_gzip_compression_rolling_average`
(:138-158), the chain projection_file_blend.py:192-217 builds the music-complexity skip mask
from. `reduce_vector_rms_rolling_average` (:102-124), `reduce_vector_rms_rolling_max` (:38-58) and
`quantize_results_layers` (:161-194), the reductions that pick networks and roll amounts, run on the GPU (`gance_vec_*` of gance_amd/csrc/audio.hip:
numpy's float32 pairwise RMS sums and pandas' rolling-mean update order are followed there, so the integers
come out identical); the blend runs the same arithmetic fused.

`track_length_filter` is a run-length pass instead of the reference's pandas
diff / cumsum / groupby pipeline; the compressed sizes come from a plain loop instead of a
`multiprocessing.Pool` (zlib releases the GIL and a vector is 2 KiB).
"""

import zlib
from typing import List, Sequence

import numpy as np
import pandas as pd
from scipy.interpolate import UnivariateSpline
from scipy.signal import savgol_filter

from gance_amd import hip_lib
from gance_amd.data_into_network_visualization.visualization_common import DataLabel, ResultLayers
from gance_amd.vector_sources.vector_sources_common import sub_vectors
from gance_amd.vector_sources.vector_types import ConcatenatedVectors


def track_length_filter(bool_tracks: Sequence[bool], track_length: int) -> List[bool]:
    """
    Reject periods of True shorter than `track_length` (inclusive).
    :return: The input tracks with the shorter tracks replaced with all False.
    """
    flags = [bool(value) for value in bool_tracks]
    output = [False] * len(flags)
    start = 0
    while start < len(flags):
        stop = start
        while stop < len(flags) and flags[stop] == flags[start]:
            stop += 1
        if flags[start] and stop - start >= track_length:
            output[start:stop] = [True] * (stop - start)
        start = stop
    return output


def _smoothed_rolling_average(
    input_values: DataLabel, rolling_average_window: int = 3, savgol_window_length: int = 7, savgol_polyorder: int = 3
) -> ResultLayers:
    """Rolling mean (NaN head filled with the series mean) then a Savitzky-Golay filter (:61-99)."""
    as_series = pd.Series(input_values.data)
    rolling_average = as_series.rolling(rolling_average_window).mean().fillna(as_series.mean()).to_numpy()
    smoothed_average = savgol_filter(x=rolling_average, window_length=savgol_window_length, polyorder=savgol_polyorder)
    return ResultLayers(
        result=DataLabel(smoothed_average, f"Savgol Smoothing Filter (window={savgol_window_length}, polyorder={savgol_polyorder})"),
        layers=[DataLabel(rolling_average, f"Rolling Average (window={rolling_average_window})"), input_values],
    )


def reduce_vector_rms_rolling_average(
    time_series_audio_vectors: ConcatenatedVectors,
    vector_length: int,
    rolling_average_window: int = 3,
    savgol_window_length: int = 7,
    savgol_polyorder: int = 3,
) -> ResultLayers:
    """
    One RMS value per frame of audio (librosa.feature.rms, frame `vector_length`, librosa's default hop of
    512, center=False), rolling mean with the NaN head filled by the series mean, Savitzky-Golay
    (vector_reduction.py:102-124 -> :22-35, :61-99). Same layers and labels as the reference.
    """
    raw_rms, rolling_average, smoothed_average = hip_lib.vec_rms_rolling_average(
        time_series_audio_vectors, vector_length, rolling_average_window, savgol_window_length, savgol_polyorder
    )
    return ResultLayers(
        result=DataLabel(smoothed_average, f"Savgol Smoothing Filter (window={savgol_window_length}, polyorder={savgol_polyorder})"),
        layers=[DataLabel(rolling_average, f"Rolling Average (window={rolling_average_window})"), DataLabel(raw_rms, "Raw RMS Power")],
    )


def reduce_vector_rms_rolling_max(time_series_audio_vectors: ConcatenatedVectors, vector_length: int) -> ResultLayers:
    """
    One RMS value per hop of audio (librosa.feature.rms, frame `vector_length`, hop 512, center=False), then a
    rolling maximum over len // 80 values (scipy.ndimage.maximum_filter1d) when that is at least one value
    (vector_reduction.py:38-58 -> :22-35). Same layers and labels as the reference.
    """
    raw_rms, output = hip_lib.vec_rms_rolling_max(time_series_audio_vectors, vector_length)
    return ResultLayers(result=DataLabel(output, "Rolling Max"), layers=[DataLabel(raw_rms, "Raw RMS Power")])


def quantize_results_layers(results_layers: ResultLayers, network_indices: List[int]) -> ResultLayers:
    """
    Scale a reduction's result into the range of the candidate indices and round to integers
    (vector_reduction.py:161-194: interp1d remap of [min, max] onto [0, K - 1], np.rint, astype(int)).
    """
    quantized = hip_lib.vec_quantize(np.asarray(results_layers.result.data, dtype=np.float64), len(network_indices)).astype(int)
    return ResultLayers(
        result=DataLabel(quantized, f"{results_layers.result.label} Scaled, Quantized"),
        layers=[results_layers.result] + results_layers.layers,
    )


def reduce_vector_gzip_compression_rolling_average(
    time_series_audio_vectors: ConcatenatedVectors, vector_length: int
) -> ResultLayers:
    """Each vector reduced to the size of its zlib-compressed bytes, then averaged and smoothed."""
    compressed_sizes = [
        len(zlib.compress(np.ascontiguousarray(vector).tobytes()))
        for vector in sub_vectors(data=time_series_audio_vectors, vector_length=vector_length)
    ]
    return _smoothed_rolling_average(DataLabel(np.array(compressed_sizes), "Gzipped Audio"))


def _derive_data(data: np.ndarray, order: int) -> np.ndarray:
    """nth derivative of a smoothing spline through the data; NaN counts as zero."""
    data = np.nan_to_num(data)
    x_axis = np.arange(len(data))
    return UnivariateSpline(x=x_axis, y=data).derivative(n=order)(x_axis)


def derive_results_layers(results_layers: ResultLayers, order: int) -> ResultLayers:
    """Derivative of the result, pushed onto the layers."""
    return ResultLayers(
        result=DataLabel(_derive_data(data=results_layers.result.data, order=order), f"Derevation order={order}"),
        layers=[results_layers.result] + results_layers.layers,
    )


def absolute_value_results_layers(results_layers: ResultLayers) -> ResultLayers:
    """Absolute value of the result, pushed onto the layers."""
    return ResultLayers(
        result=DataLabel(np.abs(results_layers.result.data), "Absolute Value"),
        layers=[results_layers.result] + results_layers.layers,
    )


def rolling_sum_results_layers(results_layers: ResultLayers, window_length: int) -> ResultLayers:
    """Rolling sum of the result (NaN until the window fills), pushed onto the layers."""
    series = pd.Series(results_layers.result.data)
    return ResultLayers(
        result=DataLabel(np.array(series.rolling(window_length).sum()), f"Rolling Sum (window={window_length})"),
        layers=[results_layers.result] + results_layers.layers,
    )
