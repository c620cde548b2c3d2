// Time-stretching the audio to one vector per video frame on the GPU: gance_resample_audio_f32 / _f64 of
// include/gance_hip.h. Replaces `resampy.resample(x, sr_orig, sr_new)` (filter "kaiser_best") of
// _scale_wav_to_sample_rate (gance/vector_sources/music.py:212-230).
//
// resampy (pinned 0.2.2, requirements/prod.txt:25) is a third-party dependency that is not under
// /root/reference; its published algorithm is restated here operation for operation:
//
//   filter   `sinc_window(num_zeros=64, precision=9, window=kaiser(beta=14.769656459379492),
//            rolloff=0.9475937167399596)`: the right half of a Kaiser-windowed sinc sampled 512 times per
//            zero crossing, 32 769 float64 entries (resampy ships exactly this array as data/kaiser_best.npz);
//            multiplied by the sample ratio when down-sampling; `interp_delta` = its first difference.
//   core     `resample_f`: for output t, time_register (a RUNNING SUM of 1/ratio, so its rounding errors
//            accumulate the way resampy's do), n = int(time_register), frac = scale * (time_register - n),
//            then the left wing (x[n], x[n-1], ...) followed by the right wing (x[n+1], x[n+2], ...), every tap
//            weight = win[offset + i*step] + eta * delta[offset + i*step] (linear interpolation between table
//            entries), and `y[t] += weight * x[...]` with y in the INPUT's dtype: a float32 signal is
//            accumulated with one rounding to float32 per tap, in that order. A ratio of exactly 1 is still
//            a pass through the low-pass filter (roll-off 0.9476), not a copy.
//
// Pinned by the one numeric fixture the reference holds on this path (test/test_dynamic_model_switching.py:
// 15-39: claps.wav -> 60 fps, L = 1000 -> RMS of the first vector = 0.00298562): tests/test_music_gpu.py.
//
// One thread per output sample; an output reads <= 2 * 64 / scale consecutive input samples and as many
// table entries (a stride of `step` through the 256 KB table): input and table live in L2. The running sum
// of the time register is inherently serial: it is computed on the host (one add per output sample, about a
// millisecond for 30 s of audio) and uploaded with the call.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace gance_resample {

constexpr int kZeroCrossings = 64;
constexpr int kNumTable = 512;  // 2 ** precision, precision = 9
constexpr int kTableSize = kZeroCrossings * kNumTable + 1;
constexpr double kKaiserBeta = 14.769656459379492;
constexpr double kRolloff = 0.9475937167399596;

// Cephes i0 (the routine behind scipy.special.i0 and, coefficient for coefficient, numpy.i0): Chebyshev
// expansions on [0, 8] and (8, inf)
static const double kI0A[30] = {
    -4.4153416464793395e-18, 3.3307945188222384e-17,  -2.431279846547955e-16, 1.715391285555133e-15,  -1.1685332877993451e-14,
    7.676185498604936e-14,   -4.856446783111929e-13,  2.95505266312964e-12,   -1.726826291441556e-11, 9.675809035373237e-11,
    -5.189795601635263e-10,  2.6598237246823866e-09,  -1.300025009986248e-08, 6.046995022541919e-08,  -2.670793853940612e-07,
    1.1173875391201037e-06,  -4.4167383584587505e-06, 1.6448448070728896e-05, -5.754195010082104e-05, 0.00018850288509584165,
    -0.0005763755745385824,  0.0016394756169413357,   -0.004324309995050576,  0.010546460394594998,   -0.02373741480589947,
    0.04930528423967071,     -0.09490109704804764,    0.17162090152220877,    -0.3046826723431984,    0.6767952744094761};
static const double kI0B[25] = {
    -7.233180487874754e-18, -4.830504485944182e-18, 4.46562142029676e-17,    3.461222867697461e-17,  -2.8276239805165836e-16,
    -3.425485619677219e-16, 1.7725601330565263e-15, 3.8116806693526224e-15,  -9.554846698828307e-15, -4.150569347287222e-14,
    1.54008621752141e-14,   3.8527783827421426e-13, 7.180124451383666e-13,   -1.7941785315068062e-12, -1.3215811840447713e-11,
    -3.1499165279632416e-11, 1.1889147107846439e-11, 4.94060238822497e-10,   3.3962320257083865e-09, 2.266668990498178e-08,
    2.0489185894690638e-07, 2.8913705208347567e-06, 6.889758346916825e-05,   0.0033691164782556943,  0.8044904110141088};

static double chbevl(double x, const double* vals, int count) {
    double b0 = vals[0], b1 = 0.0, b2 = 0.0;
    for (int i = 1; i < count; ++i) {
        b2 = b1;
        b1 = b0;
        b0 = x * b1 - b2 + vals[i];
    }
    return 0.5 * (b0 - b2);
}

static double bessel_i0(double x) {
    x = std::fabs(x);
    if (x <= 8.0) return std::exp(x) * chbevl(x / 2.0 - 2.0, kI0A, 30);
    return std::exp(x) * chbevl(32.0 / x - 2.0, kI0B, 25) / std::sqrt(x);
}

// resampy.filters.sinc_window(64, 9, kaiser(beta), rolloff) -> interp_win (32 769 entries)
static const std::vector<double>& half_window() {
    static const std::vector<double> table = [] {
        std::vector<double> win(kTableSize);
        const int n = kTableSize - 1;
        const double step = (double)kZeroCrossings / n;  // np.linspace(0, 64, n + 1): exact multiples of 2^-9
        const double i0_beta = bessel_i0(kKaiserBeta);
        for (int i = 0; i <= n; ++i) {
            const double position = i == n ? (double)kZeroCrossings : i * step;
            const double x = kRolloff * position;
            const double y = M_PI * (x == 0.0 ? 1.0e-20 : x);  // np.sinc
            const double sinc = kRolloff * (std::sin(y) / y);
            const double r = (double)i / n;  // kaiser(2 n + 1)[n + i]: ((n + i) - alpha) / alpha, alpha = n
            const double taper = bessel_i0(kKaiserBeta * std::sqrt(1.0 - r * r)) / i0_beta;
            win[i] = taper * sinc;
        }
        return win;
    }();
    return table;
}

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ x, long long n_orig, T* __restrict__ y, long long n_out,
                                                       const double* __restrict__ time_register,
                                                       const double* __restrict__ win, double gain, double scale, int index_step) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out) return;
    const double tr = time_register[t];
    const long long n = (long long)tr;
    double frac = __dmul_rn(scale, __dsub_rn(tr, (double)n));
    T acc = (T)0;
    // interp_win[idx] (scaled in place by the ratio when down-sampling) and interp_delta[idx] = diff, 0 at the end
    auto tap = [&](int idx, double eta, T sample) {
        const double w0 = __dmul_rn(win[idx], gain);
        const double delta = idx + 1 < kTableSize ? __dsub_rn(__dmul_rn(win[idx + 1], gain), w0) : 0.0;
        const double weight = __dadd_rn(w0, __dmul_rn(eta, delta));
        acc = (T)__dadd_rn((double)acc, __dmul_rn(weight, (double)sample));  // y[t] += weight * x[...] in y's dtype
    };
    {   // left wing
        const double index_frac = __dmul_rn(frac, (double)kNumTable);
        const int offset = (int)index_frac;
        const double eta = __dsub_rn(index_frac, (double)offset);
        long long i_max = (kTableSize - offset) / index_step;
        if (n + 1 < i_max) i_max = n + 1;
        for (long long i = 0; i < i_max; ++i) tap(offset + (int)i * index_step, eta, x[n - i]);
    }
    frac = __dsub_rn(scale, frac);
    {   // right wing
        const double index_frac = __dmul_rn(frac, (double)kNumTable);
        const int offset = (int)index_frac;
        const double eta = __dsub_rn(index_frac, (double)offset);
        long long k_max = (kTableSize - offset) / index_step;
        if (n_orig - n - 1 < k_max) k_max = n_orig - n - 1;
        for (long long k = 0; k < k_max; ++k) tap(offset + (int)k * index_step, eta, x[n + k + 1]);
    }
    y[t] = acc;
}

static int fail(int code, const std::string& message) { return gance::set_last_error(code, message); }

// the filter table, once per device
static std::mutex g_table_mutex;
static double* g_table[gance::kMaxDevices] = {};

static hipError_t device_table(const double** out) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (device < 0 || device >= gance::kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_table_mutex);
    if (g_table[device] == nullptr) {
        double* ptr = nullptr;
        if ((e = hipMalloc((void**)&ptr, kTableSize * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMemcpy(ptr, half_window().data(), kTableSize * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
            hipFree(ptr);
            return e;
        }
        g_table[device] = ptr;
    }
    *out = g_table[device];
    return hipSuccess;
}

template <typename T>
static int resample(const T* d_in, uint64_t num_in, double sr_orig, double sr_new, T* d_out, uint64_t num_out, void* stream_ptr,
                    const char* name) {
    if (d_in == nullptr || d_out == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, std::string("NULL argument to ") + name);
    if (!(sr_orig > 0.0)) return fail(GANCE_ERR_INVALID_ARGUMENT, "Invalid sample rate: sr_orig=" + std::to_string(sr_orig));
    if (!(sr_new > 0.0)) return fail(GANCE_ERR_INVALID_ARGUMENT, "Invalid sample rate: sr_new=" + std::to_string(sr_new));
    const double ratio = sr_new / sr_orig;  // float(sr_new) / sr_orig
    const uint64_t expected = (uint64_t)((double)num_in * ratio);  // int(shape[axis] * sample_ratio)
    if (expected < 1)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "Input signal length=" + std::to_string(num_in) + " is too small to resample from " +
                                                    std::to_string(sr_orig) + "->" + std::to_string(sr_new));
    if (num_out != expected)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "num_out must be int(num_in * sr_new / sr_orig) = " + std::to_string(expected));
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
        return fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));  // launch where the samples live
    if (guard.status() != hipSuccess) return fail(GANCE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.status()));
    hipStream_t stream = (hipStream_t)stream_ptr;
    const double* d_win = nullptr;
    hipError_t err = device_table(&d_win);
    if (err != hipSuccess) return fail(GANCE_ERR_HIP, std::string("filter table: ") + hipGetErrorString(err));

    const double scale = ratio < 1.0 ? ratio : 1.0;   // min(1.0, sample_ratio)
    const double gain = ratio < 1.0 ? ratio : 1.0;    // interp_win *= sample_ratio when down-sampling
    const double time_increment = 1.0 / ratio;
    const int index_step = (int)(scale * kNumTable);
    if (index_step < 1) return fail(GANCE_ERR_INVALID_ARGUMENT, "sample ratio below 1 / 512 is not supported (resampy divides by zero there)");
    // time_register += time_increment, once per output sample, in resampy's order
    double* h_time = nullptr;
    if (hipHostMalloc((void**)&h_time, num_out * sizeof(double), hipHostMallocDefault) != hipSuccess)
        return fail(GANCE_ERR_OUT_OF_MEMORY, "hipHostMalloc failed in the resampler");
    double time_register = 0.0;
    for (uint64_t t = 0; t < num_out; ++t) {
        h_time[t] = time_register;
        time_register += time_increment;
    }
    double* d_time = nullptr;
    if (hipMalloc((void**)&d_time, num_out * sizeof(double)) != hipSuccess) {
        hipHostFree(h_time);
        return fail(GANCE_ERR_OUT_OF_MEMORY, "hipMalloc failed in the resampler");
    }
    err = hipMemcpyAsync(d_time, h_time, num_out * sizeof(double), hipMemcpyHostToDevice, stream);
    if (err == hipSuccess) {
        const unsigned blocks = (unsigned)((num_out + 255) / 256);
        resample_kernel<T><<<blocks, 256, 0, stream>>>(d_in, (long long)num_in, d_out, (long long)num_out, d_time, d_win, gain, scale,
                                                        index_step);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(stream);  // the time register dies here
    hipFree(d_time);
    hipHostFree(h_time);
    if (err != hipSuccess) return fail(GANCE_ERR_HIP, std::string("resample_kernel: ") + hipGetErrorString(err));
    return GANCE_OK;
}

}  // namespace gance_resample

extern "C" int gance_resample_audio_f32(const float* d_in, uint64_t num_in, double sr_orig, double sr_new, float* d_out,
                                        uint64_t num_out, void* stream) {
    return gance_resample::resample<float>(d_in, num_in, sr_orig, sr_new, d_out, num_out, stream, "gance_resample_audio_f32");
}

extern "C" int gance_resample_audio_f64(const double* d_in, uint64_t num_in, double sr_orig, double sr_new, double* d_out,
                                        uint64_t num_out, void* stream) {
    return gance_resample::resample<double>(d_in, num_in, sr_orig, sr_new, d_out, num_out, stream, "gance_resample_audio_f64");
}

extern "C" int gance_debug_resample_filter(double* h_out, uint64_t count) {
    if (h_out == nullptr || count != (uint64_t)gance_resample::kTableSize)
        return gance_resample::fail(GANCE_ERR_INVALID_ARGUMENT, "the kaiser_best half window has 32769 entries");
    std::memcpy(h_out, gance_resample::half_window().data(), count * sizeof(double));
    return GANCE_OK;
}
