// Audio -> latent half of the hot path on the GPU, behind the gance_blend_* entry points of
// include/gance_hip.h. Restates, MI355X-first, what the reference computes with per-vector Python
// loops over numpy / scipy / pandas / librosa calls:
//
//   compute_spectrogram                gance/apply_spectrogram.py:49-82
//   reshape_spectrogram_to_vectors     gance/apply_spectrogram.py:20-46
//   smooth_across_vectors / each       gance/vector_sources/vector_sources_common.py:136-188
//   _compute_raw_rms, _smoothed_rolling_average, quantize_results_layers
//                                      gance/vector_sources/vector_reduction.py:22-35,61-99,161-194
//   rotate_vectors_over_time           gance/vector_sources/vector_sources_common.py:408-428
//   alpha_blend_projection_file        gance/data_into_network_visualization/visualization_inputs.py:169-270
//
// Everything the reference computes in float64 is computed in float64 here (the path is a few MB:
// launch-latency bound, never ALU bound); the float32 RMS follows numpy's pairwise summation
// order bit for bit and the rolling mean follows pandas' Kahan add/remove order, so that the
// integer roll amounts and network indices (np.rint of a remapped float) come out identical.
// Every smoothing / resampling step is a fixed linear operator whose table is built once on the
// host in long double (Savitzky-Golay interior taps + edge polynomial-fit matrices, the Dirichlet
// kernel of scipy.signal.resample, the periodic Hann window, the DFT twiddles).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace gance_audio {

// ------------------------------------------------------------------------------------------
// host: operator tables
// ------------------------------------------------------------------------------------------

// Savitzky-Golay (window w, order p < w, deriv 0): interior taps c[0..w-1] (symmetric) and the edge
// matrix E[i][j], i < w/2: value at position i of the polynomial fitted to the first w samples
// (scipy.signal.savgol_filter mode='interp', _fit_edges_polyfit). Table layout: c[w] then E[h][w].
// The least-squares fit of a polynomial of degree p to w equally spaced samples is the projection onto the
// first p + 1 discrete orthogonal polynomials over those points (built here by modified Gram-Schmidt on the
// monomials, in long double, points scaled to [-1, 1]): weight(i, j) = sum_k Q_k(x_i) Q_k(x_j), Q_k
// orthonormal. Any polyorder the reference's callers pass (vector_sources_common.py:136-188 hands it to
// scipy unchanged); rounds 1 - 4 inverted the 4 x 4 normal matrix, which stopped at polyorder 3.
static std::vector<double> savgol_table(int w, int p) {
    const int h = w / 2;
    const long double hc = (w - 1) / 2.0L;
    const long double scale = hc > 0 ? hc : 1.0L;
    std::vector<std::vector<long double>> q((size_t)p + 1, std::vector<long double>((size_t)w));
    for (int k = 0; k <= p; ++k) {
        for (int j = 0; j < w; ++j) q[k][j] = powl(((long double)j - hc) / scale, k);
        for (int pass = 0; pass < 2; ++pass)  // (re-orthogonalised once: the monomials are nearly dependent at high orders)
            for (int l = 0; l < k; ++l) {
                long double dot = 0;
                for (int j = 0; j < w; ++j) dot += q[k][j] * q[l][j];
                for (int j = 0; j < w; ++j) q[k][j] -= dot * q[l][j];
            }
        long double norm = 0;
        for (int j = 0; j < w; ++j) norm += q[k][j] * q[k][j];
        norm = sqrtl(norm);
        for (int j = 0; j < w; ++j) q[k][j] /= norm;
    }
    std::vector<double> table((size_t)w + (size_t)h * w);
    auto weight = [&](int i, int j) {
        long double s = 0;
        for (int k = 0; k <= p; ++k) s += q[k][i] * q[k][j];
        return s;
    };
    for (int j = 0; j < w; ++j) table[j] = (double)weight(h, j);  // the fitted value at the window's centre sample
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) table[(size_t)w + (size_t)i * w + j] = (double)weight(i, j);
    return table;
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned long long order_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// wavefront (64 lanes) reductions by butterfly shuffles; every lane ends up with the result
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) {
        const unsigned long long other = __shfl_xor(k, offset, 64);
        k = other > k ? other : k;
    }
    return k;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) v = fmin(v, __shfl_xor(v, offset, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) v = fmax(v, __shfl_xor(v, offset, 64));
    return v;
}

// Savitzky-Golay at position i of a line x[0..n-1] (stride in elements), mode 'interp'.
// Interior: folded symmetric sum in scipy.ndimage.correlate1d's order, no fused multiply-add.
template <typename Load>
__device__ __forceinline__ double savgol_at(Load x, int i, int n, const double* __restrict__ table, int w) {
    const int h = w >> 1;
    if (i >= h && i < n - h) {
        double acc = __dmul_rn(x(i), table[h]);
        for (int l = 1; l <= h; ++l) acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(x(i + l), x(i - l)), table[h + l]));
        return acc;
    }
    const double* e = table + w;
    double acc = 0.0;
    if (i < h) {
        for (int j = 0; j < w; ++j) acc = fma(e[i * w + j], x(j), acc);
    } else {
        const int r = n - 1 - i;
        for (int j = 0; j < w; ++j) acc = fma(e[r * w + j], x(n - 1 - j), acc);
    }
    return acc;
}

// numpy float32 pairwise summation (add.reduce over a contiguous run), exact operation order.
template <typename Load>
__device__ float pairwise_sum_f32(Load a, int lo, int n) {
    // explicit stack instead of recursion: (offset, length) pairs, results combined left to right
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, a(lo + i));
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a(lo + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a(lo + i + j));
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, a(lo + i));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float left = pairwise_sum_f32(a, lo, n2);
    const float right = pairwise_sum_f32(a, lo + n2, n - n2);
    return __fadd_rn(left, right);
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// a3: windowed DFT magnitude. One block per frame, thread k = frequency bin. |X| max -> *max_key: the block's
// maximum by wavefront shuffles, then ONE global atomic per block.
__global__ __launch_bounds__(256) void dft_magnitude_kernel(const float* __restrict__ audio, int L, int m,
                                                            const double* __restrict__ window,
                                                            const double* __restrict__ twiddle,  // cos[m], sin[m]
                                                            double* __restrict__ mag, unsigned long long* max_key, int bins) {
    // bins: m / 2 (the one-sided spectrum the path uses, apply_spectrogram.py:75-76) or m (truncate=False, :77-78)
    extern __shared__ double lds[];
    double* xs = lds;          // [m]
    double* tc = lds + m;      // [m]
    double* ts = lds + 2 * m;  // [m]
    __shared__ unsigned long long wave_best[4];
    const int t = blockIdx.x;
    for (int n = threadIdx.x; n < m; n += blockDim.x) {
        xs[n] = __dmul_rn((double)audio[(size_t)t * L + n], window[n]);
        tc[n] = twiddle[n];
        ts[n] = twiddle[m + n];
    }
    __syncthreads();
    unsigned long long best = 0ull;
    for (int k = threadIdx.x; k < bins; k += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < m; ++n) {
            re = fma(xs[n], tc[idx], re);
            im = fma(-xs[n], ts[idx], im);
            idx += k;
            if (idx >= m) idx -= m;
        }
        const double a = hypot(re, im);
        mag[(size_t)t * bins + k] = a;
        const unsigned long long key = order_key(a);
        best = key > best ? key : best;
    }
    best = wave_max_key(best);
    if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = wave_best[w] > best ? wave_best[w] : best;
        atomicMax(max_key, best);
    }
}

// a3 tail + a4: dB against the global max, Fourier resample bins -> L (matrix RT[bins][L]),
// running global min / max of the resampled values. FR frames per block, thread j = output bin.
constexpr int kResampleFrames = 4;
__global__ __launch_bounds__(512) void db_resample_kernel(const double* __restrict__ mag, int N, int bins, int L,
                                                          const unsigned long long* __restrict__ max_key,
                                                          const double* __restrict__ RT, double* __restrict__ db_out,
                                                          double* __restrict__ resampled,
                                                          unsigned long long* minmax_key /*[0]=min via ~, [1]=max*/) {
    extern __shared__ double lds[];  // [FR][bins]
    const int t0 = blockIdx.x * kResampleFrames;
    const double gmax = key_value(*max_key);
    for (int i = threadIdx.x; i < kResampleFrames * bins; i += blockDim.x) {
        const int f = i / bins, k = i - f * bins;
        double v = 0.0;
        if (t0 + f < N) {
            v = 20.0 * log10(mag[(size_t)(t0 + f) * bins + k] / gmax);
            if (db_out != nullptr) db_out[(size_t)(t0 + f) * bins + k] = v;
            // a silent window gives log10(0) = -inf (the reference then fails inside minmax_scale);
            // fmin / fmax below drop NaNs, so record the non-finite value in the running minimum
            if (!isfinite(v)) atomicMin(&minmax_key[0], order_key(-INFINITY));
        }
        lds[i] = v;
    }
    __syncthreads();
    double lo = INFINITY, hi = -INFINITY;
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        double acc[kResampleFrames];
        for (int f = 0; f < kResampleFrames; ++f) acc[f] = 0.0;
        for (int k = 0; k < bins; ++k) {
            const double r = RT[(size_t)k * L + j];
            for (int f = 0; f < kResampleFrames; ++f) acc[f] = fma(r, lds[f * bins + k], acc[f]);
        }
        for (int f = 0; f < kResampleFrames; ++f)
            if (t0 + f < N) {
                resampled[(size_t)(t0 + f) * L + j] = acc[f];
                lo = fmin(lo, acc[f]);
                hi = fmax(hi, acc[f]);
            }
    }
    // the wavefront's extrema by shuffles, one pair of global atomics per wavefront
    lo = wave_min_f64(lo);
    hi = wave_max_f64(hi);
    if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&minmax_key[0], order_key(lo));
        atomicMax(&minmax_key[1], order_key(hi));
    }
}

// minmax scale (sklearn algebra) + a5 savgol(7,3) over time + a6 savgol(5,3) over bins.
// One block per frame; the time pass lands in LDS, the bin pass reads it from there.
__global__ __launch_bounds__(512) void smooth_kernel(const double* __restrict__ resampled, int N, int L,
                                                     const unsigned long long* __restrict__ minmax_key,
                                                     double amp_lo, double amp_hi, int has_range,
                                                     const double* __restrict__ sg_time, int w_time,
                                                     const double* __restrict__ sg_bins, int w_bins,
                                                     double* __restrict__ scaled_out,
                                                     double* __restrict__ time_out, double* __restrict__ spec) {
    extern __shared__ double line[];  // [L]
    const int t = blockIdx.x;
    double scale = 1.0, offset = 0.0;
    if (has_range) {
        const double dmin = key_value(minmax_key[0]), dmax = key_value(minmax_key[1]);
        double range = dmax - dmin;
        if (range == 0.0) range = 1.0;
        scale = (amp_hi - amp_lo) / range;
        offset = amp_lo - dmin * scale;
    }
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        auto x = [&](int tt) {
            const double v = resampled[(size_t)tt * L + j];
            return has_range ? __dadd_rn(__dmul_rn(v, scale), offset) : v;
        };
        if (scaled_out != nullptr) scaled_out[(size_t)t * L + j] = x(t);
        const double v = savgol_at(x, t, N, sg_time, w_time);
        line[j] = v;
        if (time_out != nullptr) time_out[(size_t)t * L + j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        auto x = [&](int jj) { return line[jj]; };
        spec[(size_t)t * L + j] = savgol_at(x, j, L, sg_bins, w_bins);
    }
}

// a7: float32 RMS per frame, numpy pairwise order (librosa hop is 512 whatever L is). Generic form: one thread per frame.
__global__ void rms_kernel(const float* __restrict__ audio, size_t num_samples, int L, int n_frames,
                           float* __restrict__ rms) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_frames) return;
    const float* frame = audio + (size_t)t * 512;
    auto sq = [&](int i) {
        const float v = frame[i];
        return __fmul_rn(v, v);
    };
    const float total = pairwise_sum_f32(sq, 0, L);
    // np.mean: the float32 sum divided by the count in float32 (exact when L is a power of two); correctly
    // rounded float32 sqrt: sqrt in float64 then one rounding (53 >= 2*24+2 bits: the double rounding is innocuous)
    rms[t] = (float)sqrt((double)__fdiv_rn(total, (float)L));
}

// The same for L = 512 (the blend's only frame length), two frames per wavefront. numpy's pairwise sum of 512
// values is a fixed tree: four runs of 128, each accumulated into 8 interleaved partial sums r[j] += a[i + j]
// (i = 8, 16, ... 120), the eight combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), the four runs as
// (b0 + b1) + (b2 + b3). That is 32 independent chains of 16 additions: one per lane of half a wavefront, and the
// tree is five butterfly shuffles (x + y is commutative bit for bit, so both partners hold the same sum).
__global__ __launch_bounds__(256) void rms512_wave_kernel(const float* __restrict__ audio, int n_frames, float* __restrict__ rms) {
    const int lane = threadIdx.x & 63;
    const int frame = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const int j = lane & 7, run = (lane >> 3) & 3;
    float r = 0.f;
    if (frame < n_frames) {
        const float* a = audio + (size_t)frame * 512 + run * 128 + j;
        const float v0 = a[0];
        r = __fmul_rn(v0, v0);
#pragma unroll
        for (int i = 8; i < 128; i += 8) {
            const float v = a[i];
            r = __fadd_rn(r, __fmul_rn(v, v));
        }
    }
#pragma unroll
    for (int offset = 1; offset <= 16; offset <<= 1) r = __fadd_rn(r, __shfl_xor(r, offset, 64));
    if (frame < n_frames && (lane & 31) == 0) rms[frame] = (float)sqrt((double)__fdiv_rn(r, 512.f));
}

// scipy.ndimage.maximum_filter1d(series, size) (mode "reflect", origin 0) of reduce_vector_rms_rolling_max
// (vector_reduction.py:38-58): out[i] = max over j in [i - size / 2, i - size / 2 + size) of the reflected series
__global__ void rolling_max_kernel(const float* __restrict__ in, int n, int size, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = -INFINITY;
    bool nan = false;
    for (int k = 0; k < size; ++k) {
        int j = i - size / 2 + k;
        // reflect about the half-sample edges: (d c b a | a b c d | d c b a), period 2 n
        j %= 2 * n;
        if (j < 0) j += 2 * n;
        if (j >= n) j = 2 * n - 1 - j;
        const float v = in[j];
        nan |= v != v;
        best = fmaxf(best, v);
    }
    out[i] = nan ? NAN : best;
}

// a8 + a9: one thread walks the per-frame series: float32 mean (fill value), pandas rolling mean
// (Kahan add/remove), savgol, min/max, linear remap, rint, and (for the roll) the running sum mod L.
struct ChainArgs {
    const float* rms;
    int n;
    int rolling_window;
    const double* sg;  // savgol table
    int w;
    int num_indices;  // K
    int cumulative_mod;  // > 0: write cumsum(values) mod this instead of the raw values' copy
    double* rolling;     // [n] scratch / debug
    double* smoothed;    // [n] scratch / debug
    int* values;         // [n] quantised integers
    int* cumulative;     // [n] or nullptr
};
// One block of 256 threads per series. The float32 mean (numpy's pairwise order) and pandas' rolling mean (a Kahan
// sum carried from window to window) are sequential by definition: one lane each, in two different wavefronts so
// that they run side by side. Everything after them is data parallel: the Savitzky-Golay filter per element, the
// series' extrema by wavefront shuffles, the remap + rint per element, and the running sum of the roll amounts as
// a block-wide scan (per-thread runs, shuffle scan inside a wavefront, wavefront totals through LDS).
constexpr int kChainThreads = 256;
constexpr int kChainStaged = 15360;  // series up to this many values are staged in LDS (60 KB)
__global__ __launch_bounds__(kChainThreads) void reduce_chain_kernel(ChainArgs a0, ChainArgs a1) {
    const ChainArgs a = blockIdx.x == 0 ? a0 : a1;
    if (a.n <= 0) return;
    const int n = a.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double fill_shared;
    __shared__ double wave_lo[kChainThreads / 64], wave_hi[kChainThreads / 64];
    __shared__ long long wave_total[kChainThreads / 64];
    // the raw series goes to LDS first (all threads): the two sequential lanes below then read it at LDS latency instead of
    // one dependent global load per element (0.58 ms of a 0.92 ms chain at N = 1800 went there); longer series stay in HBM
    extern __shared__ float rms_lds[];
    const bool staged = n <= kChainStaged;
    if (staged) {
        for (int i = tid; i < n; i += kChainThreads) rms_lds[i] = a.rms[i];
        __syncthreads();
    }
    const float* const series = staged ? rms_lds : a.rms;
    if (tid == 64) {  // wavefront 1: fill value = float32 mean of the raw series
        auto r = [&](int i) { return series[i]; };
        fill_shared = (double)__fdiv_rn(pairwise_sum_f32(r, 0, n), (float)n);
    }
    if (tid == 0) {  // wavefront 0: pandas roll_mean, fixed window, min_periods = window (head entries set below)
        int nobs = 0, neg_ct = 0, same = 0;
        double sum_x = 0.0, comp_add = 0.0, comp_remove = 0.0, prev = NAN;
        for (int i = 0; i < n; ++i) {
            if (i >= a.rolling_window) {
                const double val = (double)series[i - a.rolling_window];
                nobs -= 1;
                const double y = __dsub_rn(-val, comp_remove);
                const double t = __dadd_rn(sum_x, y);
                comp_remove = __dsub_rn(__dsub_rn(t, sum_x), y);
                sum_x = t;
                if (signbit(val)) neg_ct -= 1;
            }
            const double val = (double)series[i];
            nobs += 1;
            const double y = __dsub_rn(val, comp_add);
            const double t = __dadd_rn(sum_x, y);
            comp_add = __dsub_rn(__dsub_rn(t, sum_x), y);
            sum_x = t;
            if (signbit(val)) neg_ct += 1;
            same = (val == prev) ? same + 1 : 1;
            prev = val;
            if (nobs >= a.rolling_window) {
                double result = __ddiv_rn(sum_x, (double)nobs);
                if (same >= nobs) result = prev;
                else if (neg_ct == 0 && result < 0) result = 0.0;
                else if (neg_ct == nobs && result > 0) result = 0.0;
                a.rolling[i] = result;
            }
        }
    }
    __syncthreads();
    // NaN head -> fillna(series.mean())
    for (int i = tid; i < n && i < a.rolling_window - 1; i += kChainThreads) a.rolling[i] = fill_shared;
    __threadfence_block();
    __syncthreads();
    double lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < n; i += kChainThreads) {
        auto x = [&](int ii) { return a.rolling[ii]; };
        const double v = savgol_at(x, i, n, a.sg, a.w);
        a.smoothed[i] = v;
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    lo = wave_min_f64(lo);
    hi = wave_max_f64(hi);
    if (lane == 0) {
        wave_lo[wave] = lo;
        wave_hi[wave] = hi;
    }
    __syncthreads();
    for (int w = 0; w < kChainThreads / 64; ++w) {
        lo = fmin(lo, wave_lo[w]);
        hi = fmax(hi, wave_hi[w]);
    }
    // scipy interp1d linear: slope * (x - x_lo) + y_lo with y_lo = 0, y_hi = K - 1
    const double slope = __ddiv_rn((double)(a.num_indices - 1), __dsub_rn(hi, lo));
    // contiguous runs per thread, so that the running sum is a scan over threads
    const int per_thread = (n + kChainThreads - 1) / kChainThreads;
    const int first = tid * per_thread, last = min(n, first + per_thread);
    long long local = 0;
    for (int i = first; i < last; ++i) {
        const int v = (int)rint(__dadd_rn(__dmul_rn(slope, __dsub_rn(a.smoothed[i], lo)), 0.0));
        a.values[i] = v;
        local += v;
    }
    if (a.cumulative == nullptr) return;  // (uniform over the block)
    long long inclusive = local;
#pragma unroll
    for (int offset = 1; offset < 64; offset <<= 1) {
        const long long other = __shfl_up(inclusive, offset, 64);
        if (lane >= offset) inclusive += other;
    }
    if (lane == 63) wave_total[wave] = inclusive;
    __syncthreads();
    long long running = inclusive - local;  // exclusive prefix of this thread's run
    for (int w = 0; w < wave; ++w) running += wave_total[w];
    for (int i = first; i < last; ++i) {
        running += a.values[i];
        a.cumulative[i] = (int)(running % a.cumulative_mod);
    }
}

// a10 + second a6 + a11: roll the frame, savgol(51,2) over bins, alpha blend, write latents.
struct BlendArgs {
    const double* spec;       // [N][L] compute_spectrogram_smooth_scale
    const int* cumulative;    // [N] cumsum(roll) mod L, or nullptr when the roll is off
    const double* sg;         // savgol(51,2) table
    int w;
    const float* latent_row0;  // [F][L]
    int N, L, F, depth, blend_depth;
    double alpha;
    double* rolled_out;       // [N][L] or nullptr (debug)
    double* final_out;        // [N][L] or nullptr
    double* blend_row_out;    // [N][L] or nullptr
    float* dlatents_out;      // [N][depth][L] or nullptr
};
__global__ __launch_bounds__(512) void roll_blend_kernel(const BlendArgs a) {
    extern __shared__ double line[];  // [L]
    const int t = blockIdx.x;
    const int L = a.L;
    const bool roll = a.cumulative != nullptr;
    const int shift = roll ? a.cumulative[t] : 0;
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        int src = j + shift;
        if (src >= L) src -= L;
        const double v = a.spec[(size_t)t * L + src];
        line[j] = v;
        if (a.rolled_out != nullptr) a.rolled_out[(size_t)t * L + j] = v;
    }
    __syncthreads();
    const int mult = a.N / a.F;
    const float one_minus_alpha = (float)(1.0 - a.alpha);
    const float* proj = a.latent_row0 + (size_t)(t / mult) * L;
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        double v = line[j];
        if (roll) {
            auto x = [&](int jj) { return line[jj]; };
            v = savgol_at(x, j, L, a.sg, a.w);
        }
        if (a.final_out != nullptr) a.final_out[(size_t)t * L + j] = v;
        const float p = proj[j];
        // float32 row0 * float32 (1 - alpha), then float64 + spectrogram * alpha (two roundings)
        const double blend = __dadd_rn((double)__fmul_rn(p, one_minus_alpha), __dmul_rn(v, a.alpha));
        if (a.blend_row_out != nullptr) a.blend_row_out[(size_t)t * L + j] = blend;
        if (a.dlatents_out != nullptr) {
            float* dst = a.dlatents_out + (size_t)t * a.depth * L + j;
            const float bf = (float)blend;
            for (int row = 0; row < a.depth; ++row) dst[(size_t)row * L] = row < a.blend_depth ? bf : p;
        }
    }
}


// ------------------------------------------------------------------------------------------
// stand-alone forms of the stages (gance_vec_* entry points): the same arithmetic as above on
// caller-shaped arrays, for gance_amd/apply_spectrogram.py and gance_amd/vector_sources/*
// ------------------------------------------------------------------------------------------

// scipy.signal.resample of a real line, n_in -> n_out, as the matrix M[n][j] (y = x M), following the
// library's own steps: keep min(n_in, n_out)//2 + 1 rfft terms, double (down-sampling) or halve (up-sampling)
// the term at N/2 when N is even, irfft to n_out points (whose own Nyquist term is real and counted once),
// scale by n_out / n_in.
static std::vector<double> fourier_resample_matrix(int n_in, int n_out) {
    std::vector<double> M((size_t)n_in * n_out);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    const int N = n_in < n_out ? n_in : n_out;
    const int nyq = N / 2 + 1;
    // twiddles of the kept terms: X_k of the unit impulse at n is exp(-2 pi i k n / n_in), the output term of Y_k at j is
    // Re(Y_k exp(+2 pi i k j / n_out)); angles reduced modulo the period before the long-double cos / sin
    std::vector<long double> xin_re((size_t)nyq * n_in), xin_im((size_t)nyq * n_in), out_c((size_t)nyq * n_out), out_s((size_t)nyq * n_out);
    for (int k = 0; k < nyq; ++k) {
        for (int n = 0; n < n_in; ++n) {
            const long double angle = two_pi * (long double)(((long long)k * n) % n_in) / n_in;
            xin_re[(size_t)k * n_in + n] = cosl(angle);
            xin_im[(size_t)k * n_in + n] = -sinl(angle);
        }
        for (int j = 0; j < n_out; ++j) {
            const long double angle = two_pi * (long double)(((long long)k * j) % n_out) / n_out;
            out_c[(size_t)k * n_out + j] = cosl(angle);
            out_s[(size_t)k * n_out + j] = sinl(angle);
        }
    }
    for (int n = 0; n < n_in; ++n)
        for (int j = 0; j < n_out; ++j) {
            long double acc = 1.0L;  // k = 0
            for (int k = 1; k < nyq; ++k) {
                long double re = xin_re[(size_t)k * n_in + n], im = xin_im[(size_t)k * n_in + n];
                if (N % 2 == 0 && k == N / 2) {
                    if (n_out < n_in) { re *= 2.0L; im *= 2.0L; }
                    else if (n_in < n_out) { re *= 0.5L; im *= 0.5L; }
                }
                const long double c = out_c[(size_t)k * n_out + j], sn = out_s[(size_t)k * n_out + j];
                const bool out_nyquist = (n_out % 2 == 0) && (k == n_out / 2);  // the irfft's own Nyquist bin: real, counted once
                acc += out_nyquist ? re * c : 2.0L * (re * c - im * sn);
            }
            M[(size_t)n * n_out + j] = (double)(acc / n_in);
        }
    return M;
}

__global__ void vec_savgol_kernel(const double* __restrict__ in, int N, int L, int axis, const double* __restrict__ table, int w,
                                  double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * L) return;
    const int t = (int)(i / L), j = (int)(i % L);
    if (axis == 0) {
        auto x = [&](int tt) { return in[(size_t)tt * L + j]; };
        out[i] = savgol_at(x, t, N, table, w);
    } else {
        auto x = [&](int jj) { return in[(size_t)t * L + jj]; };
        out[i] = savgol_at(x, j, L, table, w);
    }
}

__global__ void vec_resample_kernel(const double* __restrict__ in, int N, int Lin, int Lout, const double* __restrict__ M,
                                    double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * Lout) return;
    const int t = (int)(i / Lout), j = (int)(i % Lout);
    double acc = 0.0;
    for (int n = 0; n < Lin; ++n) acc = fma(in[(size_t)t * Lin + n], M[(size_t)n * Lout + j], acc);
    out[i] = acc;
}

// 20 log10(|X| / max |X|), written transposed: [bins][N] like the reference's array
__global__ void vec_db_transpose_kernel(const double* __restrict__ mag, int N, int bins, const unsigned long long* __restrict__ max_key,
                                        double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * bins) return;
    const int t = (int)(i / bins), k = (int)(i % bins);
    out[(size_t)k * N + t] = 20.0 * log10(mag[i] / key_value(*max_key));
}

__global__ void vec_minmax_reduce_kernel(const double* __restrict__ data, size_t count, unsigned long long* keys /*[0] min, [1] max*/) {
    double lo = INFINITY, hi = -INFINITY;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const double v = data[i];
        if (!isfinite(v)) bad = true;
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    if (bad) atomicMin(&keys[0], order_key(-INFINITY));
    if (lo <= hi) {
        atomicMin(&keys[0], order_key(lo));
        atomicMax(&keys[1], order_key(hi));
    }
}

// sklearn.preprocessing.minmax_scale on the whole array: X * scale + (lo - min * scale)
__global__ void vec_minmax_apply_kernel(double* __restrict__ data, size_t count, const unsigned long long* __restrict__ keys,
                                        double amp_lo, double amp_hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double dmin = key_value(keys[0]), dmax = key_value(keys[1]);
    double range = dmax - dmin;
    if (range == 0.0) range = 1.0;
    const double scale = (amp_hi - amp_lo) / range;
    const double offset = amp_lo - dmin * scale;
    data[i] = __dadd_rn(__dmul_rn(data[i], scale), offset);
}

// scipy.interpolate.interp1d (linear, two knots): slope * (x - x_lo) + y_lo
__global__ void vec_remap_kernel(const double* __restrict__ in, size_t count, double x_lo, double x_hi, double y_lo, double y_hi,
                                 double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double slope = __ddiv_rn(__dsub_rn(y_hi, y_lo), __dsub_rn(x_hi, x_lo));
    out[i] = __dadd_rn(__dmul_rn(slope, __dsub_rn(in[i], x_lo)), y_lo);
}

// quantize_results_layers on an arbitrary series: remap [min, max] -> [0, K-1], round half to even
__global__ void vec_quantize_kernel(const double* __restrict__ in, int n, int num_indices, long long* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < n; ++i) {
        lo = fmin(lo, in[i]);
        hi = fmax(hi, in[i]);
    }
    const double slope = __ddiv_rn((double)(num_indices - 1), __dsub_rn(hi, lo));
    for (int i = 0; i < n; ++i) out[i] = (long long)rint(__dadd_rn(__dmul_rn(slope, __dsub_rn(in[i], lo)), 0.0));
}

}  // namespace gance_audio

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

namespace {
int audio_fail(int code, const std::string& message) { return gance::set_last_error(code, message); }
#define GANCE_AUDIO_CHECK(expr)                                                              \
    do {                                                                                     \
        hipError_t gance_err_ = (expr);                                                      \
        if (gance_err_ != hipSuccess)                                                        \
            return audio_fail(gance_err_ == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY    \
                                                                : GANCE_ERR_HIP,             \
                              std::string(#expr) + ": " + hipGetErrorString(gance_err_));    \
    } while (0)
}  // namespace

// Operator tables depend on (vector length, filter parameters) only, never on the audio: built once per process and
// device (the Dirichlet resampling matrix alone is 130 000 long-double sines) and shared by every gance_blend; a
// warm gance_blend_create is one hipMalloc. Never freed (about 1.1 MB per device).
namespace {
struct TableKey {
    int device, kind, a, b;
    bool operator<(const TableKey& o) const { return std::tie(device, kind, a, b) < std::tie(o.device, o.kind, o.a, o.b); }
};
enum TableKind { kTableWindow = 0, kTableTwiddle = 1, kTableResample = 2, kTableSavgol = 3 };
std::mutex g_tables_mutex;
std::map<TableKey, double*> g_tables;

template <typename Build>
hipError_t cached_table(int device, int kind, int a, int b, Build build, double** out) {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    const TableKey key{device, kind, a, b};
    auto found = g_tables.find(key);
    if (found != g_tables.end()) {
        *out = found->second;
        return hipSuccess;
    }
    const std::vector<double> host = build();
    double* ptr = nullptr;
    hipError_t e = hipMalloc((void**)&ptr, host.size() * sizeof(double));
    if (e != hipSuccess) return e;
    if ((e = hipMemcpy(ptr, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
        hipFree(ptr);
        return e;
    }
    g_tables[key] = ptr;
    *out = ptr;
    return hipSuccess;
}
}  // namespace

struct gance_blend {
    gance_blend_config cfg{};
    int device = 0;
    int m = 0, bins = 0;
    int index_w = 3;
    // tables
    double *window = nullptr, *twiddle = nullptr, *RT = nullptr;
    double *sg_time = nullptr, *sg_bins = nullptr, *sg_roll_bins = nullptr, *sg_chain_roll = nullptr,
           *sg_chain_index = nullptr;
    // workspace
    double *mag = nullptr, *db = nullptr, *resampled = nullptr, *scaled = nullptr, *time_smoothed = nullptr,
           *spec = nullptr, *rolled = nullptr, *final_spec = nullptr, *blend_row = nullptr;
    double *rolling[2] = {nullptr, nullptr}, *smoothed[2] = {nullptr, nullptr};
    float* rms = nullptr;
    int *roll_values = nullptr, *cumulative = nullptr, *net_indices = nullptr;
    unsigned long long* keys = nullptr;  // [0] max |X|, [1] min resampled, [2] max resampled
    void* workspace = nullptr;           // ONE allocation behind every pointer of the group above
};

extern "C" {

void gance_blend_destroy(gance_blend* b) {
    if (!b) return;
    hipFree(b->workspace);  // the operator tables belong to the process-wide cache (cached_table)
    delete b;
}

int gance_blend_create(const gance_blend_config* config, int32_t device, gance_blend** out) {
    if (config == nullptr || out == nullptr) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    const gance_blend_config& c = *config;
    if (c.vector_length != 512)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT,
                          "vector_length must be 512: the reference's RMS hop is librosa's fixed 512 "
                          "(vector_reduction.py:33-35), frames only line up with vectors at L = 512");
    if (c.num_frames < 7) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "num_frames must be >= 7 (savgol window over time)");
    if (c.num_projection_frames < 1 || c.num_frames % c.num_projection_frames != 0)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT,
                          "Cannot duplicate the input vectors (count " + std::to_string(c.num_projection_frames) +
                              ") to the desired count " + std::to_string(c.num_frames) + ".");
    if (c.latent_depth < 1 || c.blend_depth < 0 || c.blend_depth > c.latent_depth)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "blend_depth must be in [0, latent_depth]");
    if (c.num_networks < 1) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "num_networks must be >= 1");
    if (c.index_savgol_window_length != 0 &&
        (c.index_savgol_window_length < 3 || c.index_savgol_window_length > 7 || c.index_savgol_window_length % 2 == 0 ||
         c.index_savgol_polyorder < 0 || c.index_savgol_polyorder >= c.index_savgol_window_length))
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "index savgol: window_length odd in [3, 7], polyorder in [0, window_length)");
    int device_count = 0;
    const hipError_t count_err = hipGetDeviceCount(&device_count);
    if (count_err != hipSuccess || device_count < 1)
        return audio_fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    if (device < 0 || device >= device_count) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    gance::DeviceGuard guard(device);
    GANCE_AUDIO_CHECK(guard.status());

    const int L = c.vector_length, N = c.num_frames;
    const int m = L - 2, bins = m / 2;
    if (bins % 2 == 0) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "vector_length with an even bin count is not supported");
    const int index_w = c.index_savgol_window_length == 0 ? 3 : c.index_savgol_window_length;
    const int index_p = c.index_savgol_window_length == 0 ? 2 : c.index_savgol_polyorder;

    gance_blend* b = new gance_blend();
    b->cfg = c;
    b->device = device;
    b->m = m;
    b->bins = bins;
    b->index_w = index_w;
    const long double two_pi = 6.283185307179586476925286766559L;
    auto fail_tables = [&](hipError_t e) {
        gance_blend_destroy(b);
        return audio_fail(e == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY : GANCE_ERR_HIP,
                          std::string("operator tables: ") + hipGetErrorString(e));
    };
    hipError_t te = cached_table(device, kTableWindow, m, 0, [&] {
        std::vector<double> window(m);
        for (int n = 0; n < m; ++n) window[n] = (double)(0.5L - 0.5L * cosl(two_pi * n / m));  // np.hanning(m + 1)[:-1]
        return window;
    }, &b->window);
    if (te != hipSuccess) return fail_tables(te);
    te = cached_table(device, kTableTwiddle, m, 0, [&] {
        std::vector<double> twiddle(2 * (size_t)m);
        for (int n = 0; n < m; ++n) {
            twiddle[n] = (double)cosl(two_pi * n / m);
            twiddle[m + n] = (double)sinl(two_pi * n / m);
        }
        return twiddle;
    }, &b->twiddle);
    if (te != hipSuccess) return fail_tables(te);
    te = cached_table(device, kTableResample, bins, L, [&] {
        // scipy.signal.resample of a real length-`bins` line to L points: keep bins//2+1 rfft terms,
        // y[j] = (1/bins) * sum_n x[n] * D(theta), D = sin((M + 1/2) theta) / sin(theta / 2),
        // M = (bins - 1) / 2 (bins odd), theta = 2 pi (j / L - n / bins)
        std::vector<double> RT((size_t)bins * L);
        const int M = (bins - 1) / 2;
        for (int n = 0; n < bins; ++n)
            for (int j = 0; j < L; ++j) {
                const long double theta = two_pi * ((long double)j / L - (long double)n / bins);
                const long double s = sinl(theta / 2);
                const long double d = fabsl(s) < 1e-18L ? 2.0L * M + 1.0L : sinl((M + 0.5L) * theta) / s;
                RT[(size_t)n * L + j] = (double)(d / bins);
            }
        return RT;
    }, &b->RT);
    if (te != hipSuccess) return fail_tables(te);
    struct { int w, p; double** out; } const filters[] = {
        {7, 3, &b->sg_time}, {5, 3, &b->sg_bins}, {51, 2, &b->sg_roll_bins}, {7, 3, &b->sg_chain_roll}, {index_w, index_p, &b->sg_chain_index}};
    for (const auto& f : filters) {
        te = cached_table(device, kTableSavgol, f.w, f.p, [&] { return gance_audio::savgol_table(f.w, f.p); }, f.out);
        if (te != hipSuccess) return fail_tables(te);
    }

    // workspace: one allocation, carved into 256-byte aligned pieces
    const size_t NL = (size_t)N * L;
    size_t total = 0;
    auto reserve = [&](size_t bytes) {
        const size_t at = total;
        total += (bytes + 255) / 256 * 256;
        return at;
    };
    const size_t o_mag = reserve((size_t)N * bins * 8), o_db = reserve((size_t)N * bins * 8), o_resampled = reserve(NL * 8),
                 o_scaled = reserve(NL * 8), o_time = reserve(NL * 8), o_spec = reserve(NL * 8), o_rolled = reserve(NL * 8),
                 o_final = reserve(NL * 8), o_blend = reserve(NL * 8), o_rolling0 = reserve((size_t)N * 8), o_rolling1 = reserve((size_t)N * 8),
                 o_smoothed0 = reserve((size_t)N * 8), o_smoothed1 = reserve((size_t)N * 8), o_rms = reserve((size_t)N * 4),
                 o_roll = reserve((size_t)N * 4), o_cumulative = reserve((size_t)N * 4), o_net = reserve((size_t)N * 4), o_keys = reserve(3 * 8);
    if (hipMalloc(&b->workspace, total) != hipSuccess) {
        gance_blend_destroy(b);
        return audio_fail(GANCE_ERR_OUT_OF_MEMORY, "hipMalloc failed in gance_blend_create");
    }
    char* const base = (char*)b->workspace;
    b->mag = (double*)(base + o_mag);
    b->db = (double*)(base + o_db);
    b->resampled = (double*)(base + o_resampled);
    b->scaled = (double*)(base + o_scaled);
    b->time_smoothed = (double*)(base + o_time);
    b->spec = (double*)(base + o_spec);
    b->rolled = (double*)(base + o_rolled);
    b->final_spec = (double*)(base + o_final);
    b->blend_row = (double*)(base + o_blend);
    b->rolling[0] = (double*)(base + o_rolling0);
    b->rolling[1] = (double*)(base + o_rolling1);
    b->smoothed[0] = (double*)(base + o_smoothed0);
    b->smoothed[1] = (double*)(base + o_smoothed1);
    b->rms = (float*)(base + o_rms);
    b->roll_values = (int*)(base + o_roll);
    b->cumulative = (int*)(base + o_cumulative);
    b->net_indices = (int*)(base + o_net);
    b->keys = (unsigned long long*)(base + o_keys);
    *out = b;
    return GANCE_OK;
}

int gance_blend_run(gance_blend* b, const float* d_audio, uint64_t num_samples, const float* d_latent_row0,
                    float* d_dlatents, int32_t* d_network_indices, int32_t debug_stages, void* stream_) {
    if (b == nullptr || d_audio == nullptr || d_latent_row0 == nullptr)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_blend_run");
    const gance_blend_config& c = b->cfg;
    const int L = c.vector_length, N = c.num_frames, m = b->m, bins = b->bins;
    if (num_samples < (uint64_t)N * L)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "audio has fewer than num_frames * vector_length samples");
    gance::DeviceGuard guard(b->device);
    GANCE_AUDIO_CHECK(guard.status());
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned long long init_keys[3] = {0ull, ~0ull, 0ull};
    GANCE_AUDIO_CHECK(hipMemcpyAsync(b->keys, init_keys, sizeof(init_keys), hipMemcpyHostToDevice, stream));

    hipLaunchKernelGGL(gance_audio::dft_magnitude_kernel, dim3(N), dim3(256), 3 * (size_t)m * sizeof(double), stream,
                       d_audio, L, m, b->window, b->twiddle, b->mag, b->keys, bins);
    hipLaunchKernelGGL(gance_audio::db_resample_kernel,
                       dim3((N + gance_audio::kResampleFrames - 1) / gance_audio::kResampleFrames), dim3(512),
                       (size_t)gance_audio::kResampleFrames * bins * sizeof(double), stream, b->mag, N, bins, L,
                       b->keys, b->RT, debug_stages ? b->db : nullptr, b->resampled, b->keys + 1);
    const int has_range = c.has_amplitude_range ? 1 : 0;
    hipLaunchKernelGGL(gance_audio::smooth_kernel, dim3(N), dim3(512), (size_t)L * sizeof(double), stream,
                       b->resampled, N, L, b->keys + 1, c.amplitude_lo, c.amplitude_hi, has_range, b->sg_time, 7,
                       b->sg_bins, 5, debug_stages ? b->scaled : nullptr, debug_stages ? b->time_smoothed : nullptr,
                       b->spec);
    hipLaunchKernelGGL(gance_audio::rms512_wave_kernel, dim3((N + 7) / 8), dim3(256), 0, stream, d_audio, N, b->rms);
    gance_audio::ChainArgs roll{b->rms, c.fft_roll_enabled ? N : 0, 3, b->sg_chain_roll, 7, 3, L,
                                b->rolling[0], b->smoothed[0], b->roll_values, b->cumulative};
    gance_audio::ChainArgs index{b->rms, N, 3, b->sg_chain_index, b->index_w, c.num_networks, 0,
                                 b->rolling[1], b->smoothed[1], b->net_indices, nullptr};
    hipLaunchKernelGGL(gance_audio::reduce_chain_kernel, dim3(2), dim3(gance_audio::kChainThreads), (N <= gance_audio::kChainStaged ? (size_t)N : 0) * sizeof(float), stream, roll, index);
    gance_audio::BlendArgs blend{};
    blend.spec = b->spec;
    blend.cumulative = c.fft_roll_enabled ? b->cumulative : nullptr;
    blend.sg = b->sg_roll_bins;
    blend.w = 51;
    blend.latent_row0 = d_latent_row0;
    blend.N = N;
    blend.L = L;
    blend.F = c.num_projection_frames;
    blend.depth = c.latent_depth;
    blend.blend_depth = c.blend_depth;
    blend.alpha = c.alpha;
    blend.rolled_out = debug_stages ? b->rolled : nullptr;
    blend.final_out = b->final_spec;
    blend.blend_row_out = b->blend_row;
    blend.dlatents_out = d_dlatents;
    hipLaunchKernelGGL(gance_audio::roll_blend_kernel, dim3(N), dim3(512), (size_t)L * sizeof(double), stream, blend);
    if (d_network_indices != nullptr)
        GANCE_AUDIO_CHECK(hipMemcpyAsync(d_network_indices, b->net_indices, (size_t)N * sizeof(int),
                                         hipMemcpyDeviceToDevice, stream));
    GANCE_AUDIO_CHECK(hipGetLastError());
    return GANCE_OK;
}

int gance_blend_read_stage(gance_blend* b, int32_t stage, void* h_out, uint64_t num_bytes) {
    if (b == nullptr || h_out == nullptr) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument");
    const size_t N = b->cfg.num_frames, L = b->cfg.vector_length, bins = b->bins;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (stage) {
        case GANCE_STAGE_DB: src = b->db; bytes = N * bins * 8; break;
        case GANCE_STAGE_SCALED: src = b->scaled; bytes = N * L * 8; break;
        case GANCE_STAGE_SMOOTHED_TIME: src = b->time_smoothed; bytes = N * L * 8; break;
        case GANCE_STAGE_SMOOTHED: src = b->spec; bytes = N * L * 8; break;
        case GANCE_STAGE_ROLLED: src = b->rolled; bytes = N * L * 8; break;
        case GANCE_STAGE_FINAL: src = b->final_spec; bytes = N * L * 8; break;
        case GANCE_STAGE_BLEND_ROW: src = b->blend_row; bytes = N * L * 8; break;
        case GANCE_STAGE_RAW_RMS: src = b->rms; bytes = N * 4; break;
        case GANCE_STAGE_ROLL_VALUES: src = b->roll_values; bytes = N * 4; break;
        case GANCE_STAGE_ROLL_CUMULATIVE: src = b->cumulative; bytes = N * 4; break;
        case GANCE_STAGE_NETWORK_INDICES: src = b->net_indices; bytes = N * 4; break;
        case GANCE_STAGE_ROLLING_AVERAGE: src = b->rolling[0]; bytes = N * 8; break;
        case GANCE_STAGE_ROLLING_SMOOTHED: src = b->smoothed[0]; bytes = N * 8; break;
        case GANCE_STAGE_INDEX_SMOOTHED: src = b->smoothed[1]; bytes = N * 8; break;
        case GANCE_STAGE_MINMAX: {
            if (num_bytes != 24) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "GANCE_STAGE_MINMAX holds 24 bytes");
            unsigned long long keys[3];
            gance::DeviceGuard stage_guard(b->device);
            GANCE_AUDIO_CHECK(stage_guard.status());
            GANCE_AUDIO_CHECK(hipDeviceSynchronize());
            GANCE_AUDIO_CHECK(hipMemcpy(keys, b->keys, sizeof(keys), hipMemcpyDeviceToHost));
            double* out = (double*)h_out;
            for (int i = 0; i < 3; ++i) {  // inverse of the order-preserving key transform
                const unsigned long long k = keys[i];
                const unsigned long long bits = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
                std::memcpy(&out[i], &bits, 8);
            }
            return GANCE_OK;
        }
        default: return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "unknown stage");
    }
    if (num_bytes != bytes)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "stage holds " + std::to_string(bytes) + " bytes, buffer has " +
                                                          std::to_string(num_bytes));
    gance::DeviceGuard guard(b->device);
    GANCE_AUDIO_CHECK(guard.status());
    GANCE_AUDIO_CHECK(hipDeviceSynchronize());
    GANCE_AUDIO_CHECK(hipMemcpy(h_out, src, bytes, hipMemcpyDeviceToHost));
    return GANCE_OK;
}

// ---- stand-alone stages (include/gance_hip.h: gance_vec_*) ----

namespace {
struct DeviceBuffer {  // freed on every return path
    void* ptr = nullptr;
    ~DeviceBuffer() { hipFree(ptr); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&ptr, bytes ? bytes : 8); }
};
int vec_begin(const void* pointer, hipStream_t* stream, void* stream_) {
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
        return audio_fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    (void)pointer;
    *stream = (hipStream_t)stream_;
    return GANCE_OK;
}
}  // namespace

int gance_vec_savgol_f64(const double* d_in, int32_t num_vectors, int32_t vector_length, int32_t axis, int32_t window_length,
                         int32_t polyorder, double* d_out, void* stream_) {
    if (d_in == nullptr || d_out == nullptr || d_in == d_out) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL or aliased argument to gance_vec_savgol_f64");
    if (num_vectors < 1 || vector_length < 1 || (axis != 0 && axis != 1)) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "bad shape or axis");
    const int line = axis == 0 ? num_vectors : vector_length;
    // scipy.signal.savgol_filter's own argument checks (mode="interp")
    if (window_length < 1 || window_length % 2 == 0) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "window_length must be a positive odd integer.");
    if (polyorder < 0 || polyorder >= window_length) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "polyorder must be less than window_length.");
    if (window_length > line)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "If mode is 'interp', window_length must be less than or equal to the size of x.");
    hipStream_t stream;
    if (int rc = vec_begin(d_in, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));
    GANCE_AUDIO_CHECK(guard.status());
    const std::vector<double> table = gance_audio::savgol_table(window_length, polyorder);
    DeviceBuffer d_table;
    GANCE_AUDIO_CHECK(d_table.alloc(table.size() * sizeof(double)));
    GANCE_AUDIO_CHECK(hipMemcpyAsync(d_table.ptr, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    const size_t total = (size_t)num_vectors * vector_length;
    hipLaunchKernelGGL(gance_audio::vec_savgol_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_in, num_vectors,
                       vector_length, axis, (const double*)d_table.ptr, window_length, d_out);
    GANCE_AUDIO_CHECK(hipGetLastError());
    GANCE_AUDIO_CHECK(hipStreamSynchronize(stream));  // the table dies here
    return GANCE_OK;
}

int gance_vec_fourier_resample_f64(const double* d_in, int32_t num_vectors, int32_t in_length, int32_t out_length, double* d_out,
                                   void* stream_) {
    if (d_in == nullptr || d_out == nullptr || d_in == d_out) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL or aliased argument to gance_vec_fourier_resample_f64");
    if (num_vectors < 1 || in_length < 1 || out_length < 1 || in_length > 8192 || out_length > 65536)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "bad vector count or length");
    hipStream_t stream;
    if (int rc = vec_begin(d_in, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));
    GANCE_AUDIO_CHECK(guard.status());
    const std::vector<double> M = gance_audio::fourier_resample_matrix(in_length, out_length);
    DeviceBuffer d_M;
    GANCE_AUDIO_CHECK(d_M.alloc(M.size() * sizeof(double)));
    GANCE_AUDIO_CHECK(hipMemcpyAsync(d_M.ptr, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    const size_t total = (size_t)num_vectors * out_length;
    hipLaunchKernelGGL(gance_audio::vec_resample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_in, num_vectors,
                       in_length, out_length, (const double*)d_M.ptr, d_out);
    GANCE_AUDIO_CHECK(hipGetLastError());
    GANCE_AUDIO_CHECK(hipStreamSynchronize(stream));
    return GANCE_OK;
}

int gance_debug_fourier_resample_matrix(int32_t in_length, int32_t out_length, double* h_out) {
    if (h_out == nullptr || in_length < 1 || out_length < 1) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "bad argument");
    const std::vector<double> M = gance_audio::fourier_resample_matrix(in_length, out_length);
    std::memcpy(h_out, M.data(), M.size() * sizeof(double));
    return GANCE_OK;
}

int gance_vec_spectrogram2_f64(const float* d_audio, uint64_t num_samples, int32_t num_frequency_bins, int32_t truncate, double* d_out,
                               void* stream_) {
    if (d_audio == nullptr || d_out == nullptr) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_vec_spectrogram_f64");
    const int L = num_frequency_bins, m = L - 2;  // apply_spectrogram.py:68 (operator precedence)
    if (L < 4 || m % 2 != 0 || num_samples < (uint64_t)m) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "num_frequency_bins must be even and >= 4, with at least one window of samples");
    const int bins = truncate ? m / 2 : m;  // :75-78: fft[: m // 2], or every bin of the two-sided spectrum
    const int N = (int)((num_samples - m) / L + 1);  // view_as_windows(window m, step L)
    hipStream_t stream;
    if (int rc = vec_begin(d_audio, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_audio));
    GANCE_AUDIO_CHECK(guard.status());
    std::vector<double> tables((size_t)3 * m);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int n = 0; n < m; ++n) {
        tables[n] = (double)(0.5L - 0.5L * cosl(two_pi * n / m));
        tables[(size_t)m + n] = (double)cosl(two_pi * n / m);
        tables[(size_t)2 * m + n] = (double)sinl(two_pi * n / m);
    }
    DeviceBuffer d_tables, d_mag, d_key;
    GANCE_AUDIO_CHECK(d_tables.alloc(tables.size() * sizeof(double)));
    GANCE_AUDIO_CHECK(d_mag.alloc((size_t)N * bins * sizeof(double)));
    GANCE_AUDIO_CHECK(d_key.alloc(sizeof(unsigned long long)));
    GANCE_AUDIO_CHECK(hipMemcpyAsync(d_tables.ptr, tables.data(), tables.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    GANCE_AUDIO_CHECK(hipMemsetAsync(d_key.ptr, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(gance_audio::dft_magnitude_kernel, dim3(N), dim3(256), 3 * (size_t)m * sizeof(double), stream, d_audio, L, m,
                       (const double*)d_tables.ptr, (const double*)d_tables.ptr + m, (double*)d_mag.ptr, (unsigned long long*)d_key.ptr, bins);
    const size_t total = (size_t)N * bins;
    hipLaunchKernelGGL(gance_audio::vec_db_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const double*)d_mag.ptr, N, bins, (const unsigned long long*)d_key.ptr, d_out);
    GANCE_AUDIO_CHECK(hipGetLastError());
    GANCE_AUDIO_CHECK(hipStreamSynchronize(stream));
    return GANCE_OK;
}

int gance_vec_spectrogram_f64(const float* d_audio, uint64_t num_samples, int32_t num_frequency_bins, double* d_out, void* stream_) {
    return gance_vec_spectrogram2_f64(d_audio, num_samples, num_frequency_bins, 1, d_out, stream_);
}

int gance_vec_minmax_scale_f64(double* d_data, uint64_t count, double lo, double hi, void* stream_) {
    if (d_data == nullptr || count < 1) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL or empty argument to gance_vec_minmax_scale_f64");
    if (!(lo < hi)) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "Minimum of desired feature range must be smaller than maximum.");
    hipStream_t stream;
    if (int rc = vec_begin(d_data, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_data));
    GANCE_AUDIO_CHECK(guard.status());
    DeviceBuffer d_keys;
    GANCE_AUDIO_CHECK(d_keys.alloc(2 * sizeof(unsigned long long)));
    const unsigned long long init_keys[2] = {~0ull, 0ull};
    GANCE_AUDIO_CHECK(hipMemcpyAsync(d_keys.ptr, init_keys, sizeof(init_keys), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gance_audio::vec_minmax_reduce_kernel, dim3(256), dim3(256), 0, stream, (const double*)d_data, (size_t)count,
                       (unsigned long long*)d_keys.ptr);
    unsigned long long keys[2];
    GANCE_AUDIO_CHECK(hipMemcpyAsync(keys, d_keys.ptr, sizeof(keys), hipMemcpyDeviceToHost, stream));
    GANCE_AUDIO_CHECK(hipStreamSynchronize(stream));
    {   // sklearn raises on non-finite input ("Input contains infinity ..."): a silent window's log10(0) = -inf
        const unsigned long long k = keys[0];
        const unsigned long long bits = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        double dmin;
        std::memcpy(&dmin, &bits, 8);
        if (!std::isfinite(dmin)) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "Input contains infinity or a value too large for dtype('float64').");
    }
    hipLaunchKernelGGL(gance_audio::vec_minmax_apply_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, d_data,
                       (size_t)count, (const unsigned long long*)d_keys.ptr, lo, hi);
    GANCE_AUDIO_CHECK(hipGetLastError());
    GANCE_AUDIO_CHECK(hipStreamSynchronize(stream));
    return GANCE_OK;
}

int gance_vec_remap_f64(const double* d_in, uint64_t count, double in_lo, double in_hi, double out_lo, double out_hi, double* d_out,
                        void* stream_) {
    if (d_in == nullptr || d_out == nullptr || count < 1) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL or empty argument to gance_vec_remap_f64");
    hipStream_t stream;
    if (int rc = vec_begin(d_in, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));
    GANCE_AUDIO_CHECK(guard.status());
    hipLaunchKernelGGL(gance_audio::vec_remap_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, d_in, (size_t)count,
                       in_lo, in_hi, out_lo, out_hi, d_out);
    GANCE_AUDIO_CHECK(hipGetLastError());
    return GANCE_OK;
}

int gance_vec_rms_rolling_average(const float* d_audio, uint64_t num_samples, int32_t vector_length, int32_t rolling_window,
                                  int32_t savgol_window_length, int32_t savgol_polyorder, float* d_rms, double* d_rolling,
                                  double* d_smoothed, int32_t num_values, void* stream_) {
    if (d_audio == nullptr || d_rms == nullptr || d_rolling == nullptr || d_smoothed == nullptr)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_vec_rms_rolling_average");
    const int L = vector_length;
    if (L < 1 || num_samples < (uint64_t)L) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "fewer samples than one frame");
    const int n = (int)(1 + (num_samples - L) / 512);  // librosa's default hop of 512 whatever the frame length
    if (num_values != n) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "num_values must be 1 + (num_samples - vector_length) // 512 = " + std::to_string(n));
    if (rolling_window < 1) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "rolling window must be >= 1");
    if (savgol_window_length < 1 || savgol_window_length % 2 == 0) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "window_length must be a positive odd integer.");
    if (savgol_polyorder < 0 || savgol_polyorder >= savgol_window_length) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "polyorder must be less than window_length.");
    if (savgol_window_length > n)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "If mode is 'interp', window_length must be less than or equal to the size of x.");
    hipStream_t stream;
    if (int rc = vec_begin(d_audio, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_audio));
    GANCE_AUDIO_CHECK(guard.status());
    const std::vector<double> table = gance_audio::savgol_table(savgol_window_length, savgol_polyorder);
    DeviceBuffer d_table, d_values;
    GANCE_AUDIO_CHECK(d_table.alloc(table.size() * sizeof(double)));
    GANCE_AUDIO_CHECK(d_values.alloc((size_t)n * sizeof(int)));
    GANCE_AUDIO_CHECK(hipMemcpyAsync(d_table.ptr, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gance_audio::rms_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_audio, (size_t)num_samples, L, n, d_rms);
    gance_audio::ChainArgs chain{d_rms, n, rolling_window, (const double*)d_table.ptr, savgol_window_length, 2, 0,
                                 d_rolling, d_smoothed, (int*)d_values.ptr, nullptr};
    gance_audio::ChainArgs none{};
    hipLaunchKernelGGL(gance_audio::reduce_chain_kernel, dim3(1), dim3(gance_audio::kChainThreads), (n <= gance_audio::kChainStaged ? (size_t)n : 0) * sizeof(float), stream, chain, none);
    GANCE_AUDIO_CHECK(hipGetLastError());
    GANCE_AUDIO_CHECK(hipStreamSynchronize(stream));
    return GANCE_OK;
}

int gance_vec_rms_rolling_max(const float* d_audio, uint64_t num_samples, int32_t vector_length, float* d_rms, float* d_out,
                              int32_t num_values, void* stream_) {
    if (d_audio == nullptr || d_rms == nullptr || d_out == nullptr || d_rms == d_out)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "NULL or aliased argument to gance_vec_rms_rolling_max");
    const int L = vector_length;
    if (L < 1 || num_samples < (uint64_t)L) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "fewer samples than one frame");
    const int n = (int)(1 + (num_samples - L) / 512);  // librosa's default hop of 512 whatever the frame length
    if (num_values != n) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "num_values must be 1 + (num_samples - vector_length) // 512 = " + std::to_string(n));
    hipStream_t stream;
    if (int rc = vec_begin(d_audio, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_audio));
    GANCE_AUDIO_CHECK(guard.status());
    hipLaunchKernelGGL(gance_audio::rms_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_audio, (size_t)num_samples, L, n, d_rms);
    const int feature_length = n / 80;  // int(len(raw_rms) / 80), vector_reduction.py:51
    if (feature_length > 0)
        hipLaunchKernelGGL(gance_audio::rolling_max_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)d_rms, n,
                           feature_length, d_out);
    else
        GANCE_AUDIO_CHECK(hipMemcpyAsync(d_out, d_rms, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    GANCE_AUDIO_CHECK(hipGetLastError());
    return GANCE_OK;
}

int gance_vec_quantize_f64(const double* d_in, int32_t count, int32_t num_indices, int64_t* d_out, void* stream_) {
    if (d_in == nullptr || d_out == nullptr || count < 1 || num_indices < 1)
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "bad argument to gance_vec_quantize_f64");
    hipStream_t stream;
    if (int rc = vec_begin(d_in, &stream, stream_)) return rc;
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));
    GANCE_AUDIO_CHECK(guard.status());
    hipLaunchKernelGGL(gance_audio::vec_quantize_kernel, dim3(1), dim3(64), 0, stream, d_in, count, num_indices, (long long*)d_out);
    GANCE_AUDIO_CHECK(hipGetLastError());
    return GANCE_OK;
}

}  // extern "C"
