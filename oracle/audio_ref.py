"""
ORACLE (test infrastructure, NOT product code): numpy / scipy restatement of GANce's
audio -> latent chain, the half of the hot path that IS under /root/reference.

Pinned: tests/test_oracle_audio.py checks every function here against golden vectors captured
from the reference's own code (oracle/make_goldens.py imports it with the leaf stubs of
oracle/ref_stubs.py), and, when /root/reference is mounted, against the live reference.

Third-party leaves: scipy.signal.resample / savgol_filter, sklearn minmax_scale and pandas rolling
mean are restated from their published algorithms (cited per function); librosa.feature.rms as in
oracle/ref_stubs.py. resampy 0.2.2 (music.py:222-227; absent here) is restated from its published algorithm
and pinned by the reference's known answer through it (test/test_dynamic_model_switching.py:15-39).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from typing import List, NamedTuple, Optional, Tuple

import numpy as np
import pandas as pd
import scipy.signal.windows
from scipy.ndimage import gaussian_filter, maximum_filter1d
from scipy.signal import resample, savgol_filter


# ----------------------------------------------------------------------------------------------
# a3  compute_spectrogram                                   gance/apply_spectrogram.py:49-82
# ----------------------------------------------------------------------------------------------
def compute_spectrogram(audio: np.ndarray, vector_length: int) -> np.ndarray:
    """
    Windows of m = L - 2 samples (`num_frequency_bins - 1 * 2`, :68) every L samples (:69), periodic
    Hann (:70), FFT along the window (:73), first m // 2 bins (:76), 20 log10(|X| / global max) (:80-81).
    float32 audio is promoted to float64 by the window multiply. Returns (m // 2, N) float64.
    """
    m = vector_length - 1 * 2
    frames = np.lib.stride_tricks.sliding_window_view(np.asarray(audio), m)[::vector_length]
    window = np.hanning(m + 1)[:-1]
    spectrum = np.fft.fft((frames * window).T, axis=0)[: m // 2]
    magnitude = np.abs(spectrum)
    with np.errstate(divide="ignore"):
        return 20 * np.log10(magnitude / np.max(magnitude))


# ----------------------------------------------------------------------------------------------
# a4  reshape_spectrogram_to_vectors                        gance/apply_spectrogram.py:20-46
# ----------------------------------------------------------------------------------------------
def minmax_scale_1d(values: np.ndarray, feature_range: Tuple[float, float]) -> np.ndarray:
    """
    sklearn.preprocessing.minmax_scale on a 1-D array (one global min / max), MinMaxScaler algebra:
    scale = (hi - lo) / (max - min); min_ = lo - min * scale; X * scale + min_   (two roundings).
    """
    lo, hi = feature_range
    data_min, data_max = np.min(values), np.max(values)
    data_range = data_max - data_min
    if data_range == 0.0:
        data_range = 1.0  # sklearn _handle_zeros_in_scale
    scale = (hi - lo) / data_range
    min_ = lo - data_min * scale
    out = values * scale
    out += min_
    return out


def reshape_spectrogram_to_vectors(
    spectrogram: np.ndarray, vector_length: int, amplitude_range: Optional[Tuple[float, float]]
) -> np.ndarray:
    """Transpose to (N, 255), Fourier-resample every frame to L (vector_sources_common.py:222-230), flatten, min-max."""
    scaled = resample(np.transpose(spectrogram), vector_length, axis=1).reshape(-1)
    return minmax_scale_1d(scaled, amplitude_range) if amplitude_range is not None else scaled


# ----------------------------------------------------------------------------------------------
# a5 / a6  savgol smoothing                       gance/vector_sources/vector_sources_common.py:136-188
# ----------------------------------------------------------------------------------------------
def smooth_across_vectors(data: np.ndarray, vector_length: int, window_length: int = 7, polyorder: int = 3) -> np.ndarray:
    """savgol_filter (mode 'interp') along TIME for every bin (vsc:158-163)."""
    return savgol_filter(data.reshape(-1, vector_length), window_length, polyorder, axis=0).reshape(-1)


def smooth_each_vector(data: np.ndarray, vector_length: int, window_length: int = 51, polyorder: int = 2) -> np.ndarray:
    """savgol_filter (mode 'interp') along the BINS of every frame; defaults 51 / 2 (vsc:169-171)."""
    return savgol_filter(data.reshape(-1, vector_length), window_length, polyorder, axis=1).reshape(-1)


def compute_spectrogram_smooth_scale(audio: np.ndarray, vector_length: int, amplitude_range) -> np.ndarray:
    """apply_spectrogram.py:85-118: spectrogram -> vectors -> savgol(7,3) over time -> savgol(5,3) over bins."""
    vectors = reshape_spectrogram_to_vectors(compute_spectrogram(audio, vector_length), vector_length, amplitude_range)
    return smooth_each_vector(smooth_across_vectors(vectors, vector_length, 7, 3), vector_length, 5, 3)


# ----------------------------------------------------------------------------------------------
# a7  RMS per frame                                 gance/vector_sources/vector_reduction.py:22-35
# ----------------------------------------------------------------------------------------------
def numpy_pairwise_sum_f32(values: np.ndarray) -> np.float32:
    """
    numpy's float32 add.reduce over a contiguous run (pairwise summation, block 128, 8 lanes):
    the order librosa's `np.mean(np.abs(x) ** 2, axis=0)` sums each frame in. Restated so the HIP
    kernel can reproduce the float32 RMS bit for bit.
    """
    a = np.asarray(values, dtype=np.float32)
    n = len(a)
    if n < 8:
        res = np.float32(0.0)
        for value in a:
            res = np.float32(res + value)
        return res
    if n <= 128:
        r = [np.float32(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = np.float32(r[j] + a[i + j])
            i += 8
        res = np.float32(
            np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3]))
            + np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7]))
        )
        while i < n:
            res = np.float32(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return np.float32(numpy_pairwise_sum_f32(a[:n2]) + numpy_pairwise_sum_f32(a[n2:]))


def compute_raw_rms(audio: np.ndarray, vector_length: int) -> np.ndarray:
    """
    librosa.feature.rms(y, frame_length=L, center=False)[0] with librosa's DEFAULT hop 512
    (vector_reduction.py:33-35): sqrt(mean(|frame|^2)) in the input dtype (float32 audio ->
    float32 sums, pairwise order, see numpy_pairwise_sum_f32).
    """
    y = np.asarray(audio)
    hop = 512
    n_frames = 1 + (len(y) - vector_length) // hop
    frames = np.lib.stride_tricks.as_strided(
        y, shape=(vector_length, n_frames), strides=(y.itemsize, hop * y.itemsize), writeable=False
    )
    return np.sqrt(np.mean(np.abs(frames) ** 2, axis=0))


# ----------------------------------------------------------------------------------------------
# a8  rolling mean + savgol                         gance/vector_sources/vector_reduction.py:61-124
# ----------------------------------------------------------------------------------------------
def pandas_rolling_mean_kahan(values: np.ndarray, window: int) -> np.ndarray:
    """
    pandas `Series.rolling(window).mean()` (pandas/_libs/window/aggregations.pyx roll_mean): online
    add / remove with separate Kahan compensations, NaN until `window` observations; values are
    taken as float64. Restated so the HIP kernel can follow the same operation order.
    """
    x = np.asarray(values, dtype=np.float64)
    out = np.full(len(x), np.nan)
    nobs, neg_ct, sum_x = 0, 0, 0.0
    comp_add = comp_remove = 0.0
    same_count, prev = 0, np.nan
    for i in range(len(x)):
        if i >= window:  # remove the value leaving the window
            val = x[i - window]
            nobs -= 1
            y = -val - comp_remove
            t = sum_x + y
            comp_remove = t - sum_x - y
            sum_x = t
            if np.signbit(val):
                neg_ct -= 1
        val = x[i]
        nobs += 1
        y = val - comp_add
        t = sum_x + y
        comp_add = t - sum_x - y
        sum_x = t
        if np.signbit(val):
            neg_ct += 1
        if val == prev:
            same_count += 1
        else:
            same_count = 1
        prev = val
        if nobs >= window:
            result = sum_x / nobs
            if same_count >= nobs:
                result = prev
            elif neg_ct == 0 and result < 0:
                result = 0.0
            elif neg_ct == nobs and result > 0:
                result = 0.0
            out[i] = result
    return out


def smoothed_rolling_average(raw: np.ndarray, rolling_average_window: int = 3, savgol_window_length: int = 7, savgol_polyorder: int = 3) -> Tuple[np.ndarray, np.ndarray]:
    """
    vector_reduction.py:78-87: pandas rolling mean, NaN head filled with the mean of the RAW series
    (float32 series -> float32 pairwise mean), then savgol_filter. Returns (smoothed, rolling).
    """
    series = pd.Series(raw)
    rolling = series.rolling(rolling_average_window).mean().fillna(series.mean()).to_numpy()
    return savgol_filter(rolling, savgol_window_length, savgol_polyorder), rolling


# ----------------------------------------------------------------------------------------------
# a9  quantize_results_layers                       gance/vector_sources/vector_reduction.py:161-194
# ----------------------------------------------------------------------------------------------
def quantize_to_indices(values: np.ndarray, num_indices: int) -> np.ndarray:
    """
    remap [min, max] -> [0, K-1] with scipy interp1d's linear formula slope * (x - x_lo) + y_lo
    (vector_sources_common.py:59, scipy.interpolate._interpolate._call_linear), then np.rint
    (half to even) and astype(int) (:189). min == max is undefined in the reference (0/0).
    """
    x_lo, x_hi = min(values), max(values)
    y_lo, y_hi = 0.0, float(num_indices - 1)
    slope = (y_hi - y_lo) / (x_hi - x_lo)
    return np.rint(slope * (np.asarray(values, dtype=np.float64) - x_lo) + y_lo).astype(int)


# ----------------------------------------------------------------------------------------------
# a10 rotate_vectors_over_time                     gance/vector_sources/vector_sources_common.py:408-428
# ----------------------------------------------------------------------------------------------
def rotate_vectors_over_time(data: np.ndarray, vector_length: int, roll_values: np.ndarray) -> np.ndarray:
    """out_t[i] = in_t[(i + cumsum(roll)_t) mod L]  (np.roll(v, -c))."""
    vectors = data.reshape(-1, vector_length)
    shift = np.cumsum(np.asarray(roll_values, dtype=np.int64))
    index = (np.arange(vector_length, dtype=np.int64)[None, :] + shift[:, None]) % vector_length
    return np.take_along_axis(vectors, index, axis=1).reshape(-1)


# ----------------------------------------------------------------------------------------------
# visualization_inputs._create_spectrogram            visualization_inputs.py:53-91
# ----------------------------------------------------------------------------------------------
class SpectrogramStages(NamedTuple):
    """Every intermediate of `_create_spectrogram`, for stage-by-stage parity."""

    db: np.ndarray  # (255, N)
    scaled: np.ndarray  # after resample + minmax, (N*L,)
    smoothed_time: np.ndarray
    smoothed: np.ndarray  # compute_spectrogram_smooth_scale output
    raw_rms: np.ndarray
    roll_values: Optional[np.ndarray]
    rolled: Optional[np.ndarray]
    final: np.ndarray


def create_spectrogram_stages(audio: np.ndarray, vector_length: int, amplitude_range, fft_roll_enabled: bool) -> SpectrogramStages:
    """`_create_spectrogram` with its intermediates exposed."""
    db = compute_spectrogram(audio, vector_length)
    scaled = reshape_spectrogram_to_vectors(db, vector_length, amplitude_range)
    smoothed_time = smooth_across_vectors(scaled, vector_length, 7, 3)
    smoothed = smooth_each_vector(smoothed_time, vector_length, 5, 3)
    raw_rms = compute_raw_rms(audio, vector_length)
    roll_values = rolled = None
    final = smoothed
    if fft_roll_enabled:
        roll_values = quantize_to_indices(smoothed_rolling_average(raw_rms, 3, 7, 3)[0], 3)  # network_indices=[0,1,2] (:79)
        rolled = rotate_vectors_over_time(smoothed, vector_length, roll_values)
        final = smooth_each_vector(rolled, vector_length)  # defaults 51 / 2 (:82-89)
    return SpectrogramStages(db, scaled, smoothed_time, smoothed, raw_rms, roll_values, rolled, final)


# ----------------------------------------------------------------------------------------------
# a11 alpha_blend_projection_file                     visualization_inputs.py:169-270
# ----------------------------------------------------------------------------------------------
class BlendResult(NamedTuple):
    """What `alpha_blend_projection_file` returns, as plain arrays."""

    spectrogram: np.ndarray  # a_vectors.data (N*L,) float64
    projected: np.ndarray  # b_vectors.data (depth, N*L) float32
    combined: np.ndarray  # combined.data (18 or depth..., N*L) float64
    network_indices: np.ndarray  # (N,) int


def duplicate_to_vector_count(data: np.ndarray, vector_length: int, target_vector_count: int) -> np.ndarray:
    """np.repeat of every vector, exact factor required (vsc:298-345, divisor.py:19-24)."""
    vectors = data.reshape(-1, vector_length)
    factor, remainder = divmod(target_vector_count, len(vectors))
    if remainder != 0:
        raise ValueError(f"Cannot duplicate the input vectors (count {len(vectors)}) to the desired count {target_vector_count}.")
    return np.repeat(vectors, factor, axis=0).reshape(-1)


def alpha_blend_projection_file(
    final_latents: np.ndarray,
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[float, float],
    blend_depth: int,
    audio: np.ndarray,
    vector_length: int,
    network_indices: List[int],
) -> BlendResult:
    """
    final_latents: (depth=18, F*L) float32 (projection_file_reader.py:280-284). Row 0 only is used
    (:224-226), repeated N/F times per frame and tiled to `depth` rows (:220-231); blend row =
    row0 * (1 - alpha) [float32] + spectrogram * alpha [float64] (:235-236) tiled blend_depth
    times, concatenated with projected[blend_depth:18] (the 18 is hard-coded, :241-243).
    """
    spectrogram = create_spectrogram_stages(audio, vector_length, fft_amplitude_range, fft_roll_enabled).final
    num_vectors = int(spectrogram.shape[0] / vector_length)
    depth = final_latents.shape[0]
    row0 = duplicate_to_vector_count(final_latents[0], vector_length, num_vectors)
    projected = np.tile(row0, (depth, 1))
    blend_row = projected[0] * (1.0 - alpha) + spectrogram * alpha
    combined = np.concatenate((np.tile(blend_row, (blend_depth, 1)), projected[blend_depth:18]))
    raw_rms = compute_raw_rms(audio, vector_length)
    indices = quantize_to_indices(smoothed_rolling_average(raw_rms, 3, 3, 2)[0], len(network_indices))
    return BlendResult(spectrogram, projected, combined, indices)


# ----------------------------------------------------------------------------------------------
# noise-blend: gaussian_data                          gance/vector_sources/primatives.py:49-74
#              alpha_blend_vectors_max_rms_power_audio   visualization_inputs.py:94-166
# ----------------------------------------------------------------------------------------------
def gaussian_data(vector_length: int, num_vectors: int, sigma_across: float = 20, sigma_within: float = 0, seed: int = 1234) -> np.ndarray:
    """
    float32 N(0,1) draws from RandomState(seed) shaped (N, 1, L), Gaussian-filtered with mode
    "wrap" over vectors / within vectors, divided by their RMS (all float32), flattened.
    """
    draws = np.random.RandomState(seed).randn(num_vectors, 1, vector_length).astype(np.float32)  # pylint: disable=no-member
    field = gaussian_filter(input=draws, sigma=(sigma_across, 0, sigma_within), mode="wrap")
    field /= np.sqrt(np.mean(np.square(field)))
    return field.reshape(vector_length * num_vectors)


class NoiseBlendResult(NamedTuple):
    """What `alpha_blend_vectors_max_rms_power_audio` returns, as plain arrays."""

    spectrogram: np.ndarray  # a_vectors.data (N*L,) float64
    noise: np.ndarray  # b_vectors.data (N*L,) float32
    combined: np.ndarray  # combined.data (N*L,) float64
    network_indices: np.ndarray  # (N,) int


def alpha_blend_vectors_max_rms_power_audio(
    alpha: float,
    fft_roll_enabled: bool,
    fft_amplitude_range: Tuple[float, float],
    audio: np.ndarray,
    vector_length: int,
    network_indices: List[int],
) -> NoiseBlendResult:
    """
    noise = minmax_scale(gaussian_data(sigma across 50), (-4, 4)) stays float32 (:135-142);
    combined = noise * (1 - alpha) [float32] + spectrogram * alpha [float64] (:144).
    """
    spectrogram = create_spectrogram_stages(audio, vector_length, fft_amplitude_range, fft_roll_enabled).final
    num_vectors = int(spectrogram.shape[0] / vector_length)
    noise = minmax_scale_1d(gaussian_data(vector_length, num_vectors, 50, 0), (-4, 4))
    combined = noise * (1.0 - alpha) + spectrogram * alpha
    raw_rms = compute_raw_rms(audio, vector_length)
    indices = quantize_to_indices(smoothed_rolling_average(raw_rms, 3, 7, 3)[0], len(network_indices))  # defaults (:146-151)
    return NoiseBlendResult(spectrogram, noise, combined, indices)


# ----------------------------------------------------------------------------------------------
# a2  time-stretch (music.py:212-230 -> resampy.resample). resampy (pinned 0.2.2, requirements/prod.txt:25)
#     is a third-party dependency that is NOT under /root/reference and is not installed here: its published
#     algorithm (filters.sinc_window + core.resample + interpn.resample_f) is restated below, vectorised over
#     output samples but with resampy's tap order and its per-tap rounding to the signal's dtype.
#     Pinned by the reference's own known answer on this path (test/test_dynamic_model_switching.py:15-39:
#     claps.wav, 60 fps, L = 1000 -> RMS of the first vector 0.00298562): tests/test_oracle_audio.py.
# ----------------------------------------------------------------------------------------------
SINC_ZERO_CROSSINGS = 64
SINC_PRECISION = 9
KAISER_BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596


def kaiser_best_half_window() -> Tuple[np.ndarray, int]:
    """
    resampy.filters.sinc_window(num_zeros=64, precision=9, window=kaiser(beta), rolloff): the right wing of the
    windowed sinc, 512 entries per zero crossing (the array resampy ships as data/kaiser_best.npz).
    """
    num_bits = 2**SINC_PRECISION
    n = num_bits * SINC_ZERO_CROSSINGS
    sinc_win = ROLLOFF * np.sinc(ROLLOFF * np.linspace(0, SINC_ZERO_CROSSINGS, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, KAISER_BETA)[n:]
    return taper * sinc_win, num_bits


def resample_audio(samples: np.ndarray, sr_orig: float, sr_new: float) -> np.ndarray:
    """
    resampy.resample(samples, sr_orig, sr_new, filter="kaiser_best") of a 1-D signal: output length
    int(len * ratio) in the input's dtype; interp_win scaled by the ratio when down-sampling; per output sample the
    running-sum time register, left wing then right wing, linear interpolation between table entries, and
    `y[t] += weight * x[...]` rounded to y's dtype after every tap (resample_f).
    """
    if sr_orig <= 0 or sr_new <= 0:
        raise ValueError("Invalid sample rate")
    x = np.asarray(samples)
    if not np.issubdtype(x.dtype, np.floating):
        x = x.astype(np.float32)
    sample_ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * sample_ratio)
    if n_out < 1:
        raise ValueError("Input signal length is too small to resample")
    interp_win, num_table = kaiser_best_half_window()
    if sample_ratio < 1:
        interp_win = interp_win * sample_ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, sample_ratio)
    time_increment = 1.0 / sample_ratio
    index_step = int(scale * num_table)
    # time_register += time_increment per output sample (np.cumsum adds sequentially, like the loop)
    time_register = np.concatenate([[0.0], np.cumsum(np.full(n_out - 1, time_increment))])
    n = time_register.astype(np.int64)
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    y = np.zeros(n_out, dtype=x.dtype)

    def wing(frac: np.ndarray, limit: np.ndarray, sign: int, first: int) -> None:
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        count = np.minimum(limit, (nwin - offset) // index_step)
        for i in range(int(count.max(initial=0))):
            live = i < count
            index = offset[live] + i * index_step
            weight = interp_win[index] + eta[live] * interp_delta[index]
            y[live] = (y[live].astype(np.float64) + weight * x[n[live] + sign * i + first].astype(np.float64)).astype(x.dtype)

    frac = scale * (time_register - n)
    wing(frac, n + 1, -1, 0)  # x[n - i]
    wing(scale - frac, n_orig - n - 1, +1, 1)  # x[n + k + 1]
    return y


def reduce_vector_rms_rolling_max(audio: np.ndarray, vector_length: int) -> Tuple[np.ndarray, np.ndarray]:
    """vector_reduction.py:38-58: raw RMS (hop 512) and its maximum_filter1d over len // 80 values (or itself)."""
    raw = compute_raw_rms(audio, vector_length)
    feature_length = int(len(raw) / 80)
    return raw, (maximum_filter1d(input=raw, size=feature_length) if feature_length > 0 else raw)


def sub_vectors(data: np.ndarray, vector_length: int) -> np.ndarray:
    """vsc:86-101: (N*L,) -> (N, L); (depth, N*L) -> (N, depth, L)."""
    if data.ndim >= 2:
        return np.array(np.split(data, int(data.shape[-1] / vector_length), axis=-1))
    return np.reshape(data, (-1, vector_length))
