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
#include <string>
#include <vector>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace gance_audio {

// ------------------------------------------------------------------------------------------
// host: operator tables
// ------------------------------------------------------------------------------------------

// inverse of the (p+1)x(p+1) normal matrix sum_j x_j^(a+b), x_j = j - h, j = 0..w-1
static void normal_inverse(int w, int p, long double inv[4][4]) {
    const int n = p + 1;
    const long double h = (w - 1) / 2.0L;
    long double a[4][8];
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            long double s = 0;
            for (int j = 0; j < w; ++j) s += powl((long double)j - h, r + c);
            a[r][c] = s;
            a[r][n + c] = (r == c) ? 1.0L : 0.0L;
        }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (fabsl(a[r][col]) > fabsl(a[piv][col])) piv = r;
        for (int c = 0; c < 2 * n; ++c) std::swap(a[col][c], a[piv][c]);
        const long double d = a[col][col];
        for (int c = 0; c < 2 * n; ++c) a[col][c] /= d;
        for (int r = 0; r < n; ++r)
            if (r != col) {
                const long double f = a[r][col];
                for (int c = 0; c < 2 * n; ++c) a[r][c] -= f * a[col][c];
            }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) inv[r][c] = a[r][n + c];
}

// Savitzky-Golay (window w, order p, deriv 0): interior taps c[0..w-1] (symmetric) and the edge
// matrix E[i][j], i < w/2: value at position i of the polynomial fitted to the first w samples
// (scipy.signal.savgol_filter mode='interp', _fit_edges_polyfit). Table layout: c[w] then E[h][w].
static std::vector<double> savgol_table(int w, int p) {
    const int h = w / 2;
    long double inv[4][4];
    normal_inverse(w, p, inv);
    std::vector<double> table((size_t)w + (size_t)h * w);
    const long double hc = (w - 1) / 2.0L;
    auto weight = [&](long double xi, int j) {
        long double s = 0;
        const long double xj = (long double)j - hc;
        for (int a = 0; a <= p; ++a)
            for (int b = 0; b <= p; ++b) s += powl(xi, a) * inv[a][b] * powl(xj, b);
        return s;
    };
    for (int j = 0; j < w; ++j) table[j] = (double)weight(0.0L, j);
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) table[(size_t)w + (size_t)i * w + j] = (double)weight((long double)i - hc, j);
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

// a3: windowed DFT magnitude. One block per frame, thread k = frequency bin. |X| max -> *max_key.
__global__ __launch_bounds__(256) void dft_magnitude_kernel(const float* __restrict__ audio, int L, int m,
                                                            const double* __restrict__ window,
                                                            const double* __restrict__ twiddle,  // cos[m], sin[m]
                                                            double* __restrict__ mag, unsigned long long* max_key) {
    extern __shared__ double lds[];
    double* xs = lds;          // [m]
    double* tc = lds + m;      // [m]
    double* ts = lds + 2 * m;  // [m]
    __shared__ unsigned long long block_max;
    const int t = blockIdx.x;
    const int bins = m / 2;
    if (threadIdx.x == 0) block_max = 0ull;
    for (int n = threadIdx.x; n < m; n += blockDim.x) {
        xs[n] = __dmul_rn((double)audio[(size_t)t * L + n], window[n]);
        tc[n] = twiddle[n];
        ts[n] = twiddle[m + n];
    }
    __syncthreads();
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
        atomicMax(&block_max, order_key(a));
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_key, block_max);
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
    if (lo <= hi) {
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

// a7: float32 RMS per frame, numpy pairwise order (librosa hop is 512 whatever L is).
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
    // correctly rounded float32 sqrt: sqrt in float64 then one rounding (53 >= 2*24+2 bits, so the
    // double rounding is innocuous); the mean's division by L = 512 is exact
    rms[t] = (float)sqrt((double)(total / (float)L));
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
__global__ void reduce_chain_kernel(ChainArgs a0, ChainArgs a1) {
    if (threadIdx.x != 0) return;
    const ChainArgs a = blockIdx.x == 0 ? a0 : a1;
    if (a.n <= 0) return;
    const int n = a.n;
    auto r = [&](int i) { return a.rms[i]; };
    const float fill32 = __fdiv_rn(pairwise_sum_f32(r, 0, n), (float)n);
    const double fill = (double)fill32;
    // pandas roll_mean, fixed window, min_periods = window
    int nobs = 0, neg_ct = 0, same = 0;
    double sum_x = 0.0, comp_add = 0.0, comp_remove = 0.0, prev = NAN;
    for (int i = 0; i < n; ++i) {
        if (i >= a.rolling_window) {
            const double val = (double)a.rms[i - a.rolling_window];
            nobs -= 1;
            const double y = __dsub_rn(-val, comp_remove);
            const double t = __dadd_rn(sum_x, y);
            comp_remove = __dsub_rn(__dsub_rn(t, sum_x), y);
            sum_x = t;
            if (signbit(val)) neg_ct -= 1;
        }
        const double val = (double)a.rms[i];
        nobs += 1;
        const double y = __dsub_rn(val, comp_add);
        const double t = __dadd_rn(sum_x, y);
        comp_add = __dsub_rn(__dsub_rn(t, sum_x), y);
        sum_x = t;
        if (signbit(val)) neg_ct += 1;
        same = (val == prev) ? same + 1 : 1;
        prev = val;
        double result = fill;  // NaN head -> fillna(series.mean())
        if (nobs >= a.rolling_window) {
            result = __ddiv_rn(sum_x, (double)nobs);
            if (same >= nobs) result = prev;
            else if (neg_ct == 0 && result < 0) result = 0.0;
            else if (neg_ct == nobs && result > 0) result = 0.0;
        }
        a.rolling[i] = result;
    }
    double lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < n; ++i) {
        auto x = [&](int ii) { return a.rolling[ii]; };
        const double v = savgol_at(x, i, n, a.sg, a.w);
        a.smoothed[i] = v;
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    // scipy interp1d linear: slope * (x - x_lo) + y_lo with y_lo = 0, y_hi = K - 1
    const double slope = __ddiv_rn((double)(a.num_indices - 1), __dsub_rn(hi, lo));
    long long running = 0;
    for (int i = 0; i < n; ++i) {
        const double q = rint(__dadd_rn(__dmul_rn(slope, __dsub_rn(a.smoothed[i], lo)), 0.0));
        const int v = (int)q;
        a.values[i] = v;
        if (a.cumulative != nullptr) {
            running += v;
            a.cumulative[i] = (int)(running % a.cumulative_mod);
        }
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
};

extern "C" {

void gance_blend_destroy(gance_blend* b) {
    if (!b) return;
    void* ptrs[] = {b->window, b->twiddle, b->RT, b->sg_time, b->sg_bins, b->sg_roll_bins, b->sg_chain_roll,
                    b->sg_chain_index, b->mag, b->db, b->resampled, b->scaled, b->time_smoothed, b->spec,
                    b->rolled, b->final_spec, b->blend_row, b->rolling[0], b->rolling[1], b->smoothed[0],
                    b->smoothed[1], b->rms, b->roll_values, b->cumulative, b->net_indices, b->keys};
    for (void* p : ptrs) hipFree(p);
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
         c.index_savgol_polyorder < 0 || c.index_savgol_polyorder > 3 || c.index_savgol_polyorder >= c.index_savgol_window_length))
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "index savgol: window_length odd in [3, 7], polyorder in [0, 3] and < window_length");
    int device_count = 0;
    const hipError_t count_err = hipGetDeviceCount(&device_count);
    if (count_err != hipSuccess || device_count < 1)
        return audio_fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    if (device < 0 || device >= device_count) return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    GANCE_AUDIO_CHECK(hipSetDevice(device));

    gance_blend* b = new gance_blend();
    b->cfg = c;
    b->device = device;
    const int L = c.vector_length, N = c.num_frames;
    const int m = L - 2, bins = m / 2;
    b->m = m;
    b->bins = bins;
    const long double two_pi = 6.283185307179586476925286766559L;

    std::vector<double> window(m), twiddle(2 * (size_t)m), RT((size_t)bins * L);
    for (int n = 0; n < m; ++n) {
        window[n] = (double)(0.5L - 0.5L * cosl(two_pi * n / m));  // np.hanning(m + 1)[:-1]
        twiddle[n] = (double)cosl(two_pi * n / m);
        twiddle[m + n] = (double)sinl(two_pi * n / m);
    }
    // scipy.signal.resample of a real length-`bins` line to L points: keep bins//2+1 rfft terms,
    // y[j] = (1/bins) * sum_n x[n] * D(theta), D = sin((M + 1/2) theta) / sin(theta / 2),
    // M = (bins - 1) / 2 (bins odd), theta = 2 pi (j / L - n / bins)
    const int M = (bins - 1) / 2;
    for (int n = 0; n < bins; ++n)
        for (int j = 0; j < L; ++j) {
            const long double theta = two_pi * ((long double)j / L - (long double)n / bins);
            const long double half = theta / 2;
            const long double s = sinl(half);
            long double d;
            if (fabsl(s) < 1e-18L) d = 2.0L * M + 1.0L;
            else d = sinl((M + 0.5L) * theta) / s;
            RT[(size_t)n * L + j] = (double)(d / bins);
        }
    if (bins % 2 == 0) {
        delete b;
        return audio_fail(GANCE_ERR_INVALID_ARGUMENT, "vector_length with an even bin count is not supported");
    }
    const std::vector<double> sg_time = gance_audio::savgol_table(7, 3);
    const std::vector<double> sg_bins = gance_audio::savgol_table(5, 3);
    const std::vector<double> sg_roll_bins = gance_audio::savgol_table(51, 2);
    const std::vector<double> sg_chain_roll = gance_audio::savgol_table(7, 3);
    const int index_w = c.index_savgol_window_length == 0 ? 3 : c.index_savgol_window_length;
    const int index_p = c.index_savgol_window_length == 0 ? 2 : c.index_savgol_polyorder;
    const std::vector<double> sg_chain_index = gance_audio::savgol_table(index_w, index_p);

#define GANCE_ALLOC(ptr, count, type)                                                        \
    do {                                                                                     \
        hipError_t gance_err_ = hipMalloc((void**)&(ptr), (size_t)(count) * sizeof(type));   \
        if (gance_err_ != hipSuccess) {                                                      \
            gance_blend_destroy(b);                                                          \
            return audio_fail(GANCE_ERR_OUT_OF_MEMORY, "hipMalloc failed in gance_blend_create"); \
        }                                                                                    \
    } while (0)
#define GANCE_UPLOAD(ptr, vec)                                                               \
    do {                                                                                     \
        GANCE_ALLOC(ptr, (vec).size(), double);                                              \
        if (hipMemcpy(ptr, (vec).data(), (vec).size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { \
            gance_blend_destroy(b);                                                          \
            return audio_fail(GANCE_ERR_HIP, "hipMemcpy failed in gance_blend_create");      \
        }                                                                                    \
    } while (0)
    GANCE_UPLOAD(b->window, window);
    GANCE_UPLOAD(b->twiddle, twiddle);
    GANCE_UPLOAD(b->RT, RT);
    GANCE_UPLOAD(b->sg_time, sg_time);
    GANCE_UPLOAD(b->sg_bins, sg_bins);
    GANCE_UPLOAD(b->sg_roll_bins, sg_roll_bins);
    GANCE_UPLOAD(b->sg_chain_roll, sg_chain_roll);
    GANCE_UPLOAD(b->sg_chain_index, sg_chain_index);
    b->index_w = index_w;
    const size_t NL = (size_t)N * L;
    GANCE_ALLOC(b->mag, (size_t)N * bins, double);
    GANCE_ALLOC(b->db, (size_t)N * bins, double);
    GANCE_ALLOC(b->resampled, NL, double);
    GANCE_ALLOC(b->scaled, NL, double);
    GANCE_ALLOC(b->time_smoothed, NL, double);
    GANCE_ALLOC(b->spec, NL, double);
    GANCE_ALLOC(b->rolled, NL, double);
    GANCE_ALLOC(b->final_spec, NL, double);
    GANCE_ALLOC(b->blend_row, NL, double);
    for (int i = 0; i < 2; ++i) {
        GANCE_ALLOC(b->rolling[i], N, double);
        GANCE_ALLOC(b->smoothed[i], N, double);
    }
    GANCE_ALLOC(b->rms, N, float);
    GANCE_ALLOC(b->roll_values, N, int);
    GANCE_ALLOC(b->cumulative, N, int);
    GANCE_ALLOC(b->net_indices, N, int);
    GANCE_ALLOC(b->keys, 3, unsigned long long);
#undef GANCE_ALLOC
#undef GANCE_UPLOAD
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
    GANCE_AUDIO_CHECK(hipSetDevice(b->device));
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned long long init_keys[3] = {0ull, ~0ull, 0ull};
    GANCE_AUDIO_CHECK(hipMemcpyAsync(b->keys, init_keys, sizeof(init_keys), hipMemcpyHostToDevice, stream));

    hipLaunchKernelGGL(gance_audio::dft_magnitude_kernel, dim3(N), dim3(256), 3 * (size_t)m * sizeof(double), stream,
                       d_audio, L, m, b->window, b->twiddle, b->mag, b->keys);
    hipLaunchKernelGGL(gance_audio::db_resample_kernel,
                       dim3((N + gance_audio::kResampleFrames - 1) / gance_audio::kResampleFrames), dim3(512),
                       (size_t)gance_audio::kResampleFrames * bins * sizeof(double), stream, b->mag, N, bins, L,
                       b->keys, b->RT, debug_stages ? b->db : nullptr, b->resampled, b->keys + 1);
    const int has_range = c.has_amplitude_range ? 1 : 0;
    hipLaunchKernelGGL(gance_audio::smooth_kernel, dim3(N), dim3(512), (size_t)L * sizeof(double), stream,
                       b->resampled, N, L, b->keys + 1, c.amplitude_lo, c.amplitude_hi, has_range, b->sg_time, 7,
                       b->sg_bins, 5, debug_stages ? b->scaled : nullptr, debug_stages ? b->time_smoothed : nullptr,
                       b->spec);
    hipLaunchKernelGGL(gance_audio::rms_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, d_audio, (size_t)num_samples,
                       L, N, b->rms);
    gance_audio::ChainArgs roll{b->rms, c.fft_roll_enabled ? N : 0, 3, b->sg_chain_roll, 7, 3, L,
                                b->rolling[0], b->smoothed[0], b->roll_values, b->cumulative};
    gance_audio::ChainArgs index{b->rms, N, 3, b->sg_chain_index, b->index_w, c.num_networks, 0,
                                 b->rolling[1], b->smoothed[1], b->net_indices, nullptr};
    hipLaunchKernelGGL(gance_audio::reduce_chain_kernel, dim3(2), dim3(64), 0, stream, roll, index);
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
            GANCE_AUDIO_CHECK(hipSetDevice(b->device));
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
    GANCE_AUDIO_CHECK(hipSetDevice(b->device));
    GANCE_AUDIO_CHECK(hipDeviceSynchronize());
    GANCE_AUDIO_CHECK(hipMemcpy(h_out, src, bytes, hipMemcpyDeviceToHost));
    return GANCE_OK;
}

}  // extern "C"
