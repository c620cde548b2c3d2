// The smoothed-noise vector source of `noise-blend` on the GPU: gance_gaussian_noise of
// include/gance_hip.h. Restates what the reference computes with scipy.ndimage / numpy / sklearn:
//
//   gaussian_data                       gance/vector_sources/primatives.py:49-74
//     (after the MT19937 draw, which stays with the caller: it is numpy's stream by contract)
//   minmax_scale(noise, (-4, 4))        gance/data_into_network_visualization/visualization_inputs.py:135-142
//
// The draws arrive as [N][L] float32. scipy's gaussian_filter runs one 1-D correlation per axis
// with sigma > 0 (axis 0 = across vectors, then axis 2 = within a vector), each accumulating in
// float64 in the order  x[c]*w[0] + sum_{j=-r..-1} (x[c+j] + x[c-j]) * w[j]  (its symmetric-kernel
// branch) and rounding to float32 between passes; that order is kept, so a pass is bit-identical
// given identical weights. "wrap" boundary = indices modulo the axis length. The RMS
// normalisation reduces in float64 in a fixed order (the reference: float32 pairwise), so the
// scale factor may differ in the last float32 bit; the element-wise float32 steps after it
// (divide, multiply, add) are the reference's, unfused.

#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace gance_noise {

constexpr int kStatBlocks = 256;
constexpr int kThreads = 256;

// out[n][l] = float( correlate(in along one axis, symmetric weights, wrap) )
// `along_vectors` != 0: the axis is n (stride L, length N); else the axis is l (stride 1, length L)
__global__ void wrap_filter_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int L,
                                   int along_vectors, const double* __restrict__ weights, int radius) {
    const long long total = (long long)N * L;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / L), l = (int)(idx % L);
    const int length = along_vectors ? N : L;
    const int centre = along_vectors ? n : l;
    const float* line = along_vectors ? in + l : in + (size_t)n * L;
    const size_t stride = along_vectors ? (size_t)L : 1;
    const double* w = weights + radius;  // w[-radius .. radius]
    double acc = (double)line[(size_t)centre * stride] * w[0];
    for (int j = -radius; j < 0; ++j) {
        int lo = (centre + j) % length;
        if (lo < 0) lo += length;
        const int hi = (centre - j) % length;
        acc += ((double)line[(size_t)lo * stride] + (double)line[(size_t)hi * stride]) * w[j];
    }
    out[idx] = (float)acc;
}

// partials[b] = {sum of float32(x*x) in float64, min, max} over block b's grid-stride share
__global__ void noise_stats_kernel(const float* __restrict__ x, long long total, double* __restrict__ partial_sum,
                                   float* __restrict__ partial_min, float* __restrict__ partial_max) {
    __shared__ double s_sum[kThreads];
    __shared__ float s_min[kThreads], s_max[kThreads];
    double sum = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const float v = x[i];
        const float sq = v * v;
        sum += (double)sq;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    s_sum[threadIdx.x] = sum;
    s_min[threadIdx.x] = lo;
    s_max[threadIdx.x] = hi;
    __syncthreads();
    for (int step = kThreads / 2; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + step];
            s_min[threadIdx.x] = fminf(s_min[threadIdx.x], s_min[threadIdx.x + step]);
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + step]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial_sum[blockIdx.x] = s_sum[0];
        partial_min[blockIdx.x] = s_min[0];
        partial_max[blockIdx.x] = s_max[0];
    }
}

// params = {rms, scale, offset}: x <- x / rms; then (with a feature range) x <- x * scale; x <- x + offset
__global__ void noise_params_kernel(const double* __restrict__ partial_sum, const float* __restrict__ partial_min,
                                    const float* __restrict__ partial_max, long long total, int has_range,
                                    float range_lo, float range_hi, float* __restrict__ params) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sum = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < kStatBlocks; ++b) {
        sum += partial_sum[b];
        lo = fminf(lo, partial_min[b]);
        hi = fmaxf(hi, partial_max[b]);
    }
    const float mean = (float)(sum / (double)total);
    const float rms = (float)sqrt((double)mean);  // correctly rounded float32 sqrt
    params[0] = rms;
    params[1] = 1.0f;
    params[2] = 0.0f;
    if (has_range) {
        // sklearn MinMaxScaler on one float32 feature: scale_ = (hi - lo) / range, min_ = lo - data_min * scale_
        const float data_min = lo / rms, data_max = hi / rms;
        float data_range = data_max - data_min;
        if (data_range < 10.0f * 1.1920929e-07f) data_range = 1.0f;  // _handle_zeros_in_scale
        const float scale = (range_hi - range_lo) / data_range;
        params[1] = scale;
        params[2] = range_lo - data_min * scale;
    }
}

__global__ void noise_scale_kernel(float* __restrict__ x, long long total, int has_range, const float* __restrict__ params) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    float v = x[idx] / params[0];
    if (has_range) {
        v = v * params[1];
        v = v + params[2];
    }
    x[idx] = v;
}

// normalised Gaussian taps w[-r..r], r = int(4 sigma + 0.5) (scipy's truncate = 4.0)
static std::vector<double> gaussian_taps(double sigma, int* radius) {
    const int r = (int)(4.0 * sigma + 0.5);
    std::vector<double> w(2 * r + 1);
    const double scale = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int j = -r; j <= r; ++j) sum += (w[j + r] = std::exp(scale * (double)j * (double)j));
    for (double& v : w) v /= sum;
    *radius = r;
    return w;
}

static int fail(int code, const std::string& message) { return gance::set_last_error(code, message); }

#define GANCE_NOISE_CHECK(expr)                                                                              \
    do {                                                                                                     \
        hipError_t gance_err_ = (expr);                                                                      \
        if (gance_err_ != hipSuccess) {                                                                      \
            hipFree(workspace);                                                                              \
            return fail(gance_err_ == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY : GANCE_ERR_HIP,         \
                        std::string(#expr) + ": " + hipGetErrorString(gance_err_));                          \
        }                                                                                                    \
    } while (0)

}  // namespace gance_noise

extern "C" int gance_gaussian_noise(const float* d_randn, int32_t num_vectors, int32_t vector_length, double sigma_across,
                                    double sigma_within, const double* feature_range, float* d_out, void* stream_ptr) {
    using namespace gance_noise;
    if (d_randn == nullptr || d_out == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_gaussian_noise");
    if (d_randn == d_out) return fail(GANCE_ERR_INVALID_ARGUMENT, "gance_gaussian_noise does not run in place");
    if (num_vectors < 1 || vector_length < 1) return fail(GANCE_ERR_INVALID_ARGUMENT, "num_vectors and vector_length must be >= 1");
    if (!(sigma_across >= 0.0) || !(sigma_within >= 0.0) || sigma_across > 1e5 || sigma_within > 1e5)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "sigmas must be in [0, 1e5]");
    if (feature_range != nullptr && !(feature_range[0] < feature_range[1]))
        return fail(GANCE_ERR_INVALID_ARGUMENT, "Minimum of desired feature range must be smaller than maximum.");
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
        return fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");

    gance::DeviceGuard guard(gance::device_of_pointer(d_randn));  // launch where the draws live
    if (guard.status() != hipSuccess) return fail(GANCE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.status()));
    hipStream_t stream = (hipStream_t)stream_ptr;
    const long long total = (long long)num_vectors * vector_length;
    int radius_across = 0, radius_within = 0;
    const std::vector<double> taps_across = sigma_across > 0 ? gaussian_taps(sigma_across, &radius_across) : std::vector<double>();
    const std::vector<double> taps_within = sigma_within > 0 ? gaussian_taps(sigma_within, &radius_within) : std::vector<double>();

    // workspace: taps | partial sums | partial min | partial max | params | one float32 plane (two-pass case)
    const size_t tap_count = taps_across.size() + taps_within.size();
    const bool two_pass = sigma_across > 0 && sigma_within > 0;
    const size_t bytes = (tap_count + kStatBlocks) * sizeof(double) + (2 * kStatBlocks + 4) * sizeof(float) +
                         (two_pass ? (size_t)total * sizeof(float) : 0);
    char* workspace = nullptr;
    GANCE_NOISE_CHECK(hipMalloc((void**)&workspace, bytes));
    double* d_taps = (double*)workspace;
    double* d_partial_sum = d_taps + tap_count;
    float* d_partial_min = (float*)(d_partial_sum + kStatBlocks);
    float* d_partial_max = d_partial_min + kStatBlocks;
    float* d_params = d_partial_max + kStatBlocks;
    float* d_plane = d_params + 4;
    if (!taps_across.empty())
        GANCE_NOISE_CHECK(hipMemcpyAsync(d_taps, taps_across.data(), taps_across.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    if (!taps_within.empty())
        GANCE_NOISE_CHECK(hipMemcpyAsync(d_taps + taps_across.size(), taps_within.data(), taps_within.size() * sizeof(double),
                                         hipMemcpyHostToDevice, stream));

    const unsigned blocks = (unsigned)((total + kThreads - 1) / kThreads);
    const float* source = d_randn;
    if (sigma_across > 0) {
        float* dst = two_pass ? d_plane : d_out;
        wrap_filter_kernel<<<blocks, kThreads, 0, stream>>>(source, dst, num_vectors, vector_length, 1, d_taps, radius_across);
        source = dst;
    }
    if (sigma_within > 0) {
        wrap_filter_kernel<<<blocks, kThreads, 0, stream>>>(source, d_out, num_vectors, vector_length, 0,
                                                            d_taps + taps_across.size(), radius_within);
        source = d_out;
    }
    if (source != d_out)  // both sigmas zero: the filter is the identity
        GANCE_NOISE_CHECK(hipMemcpyAsync(d_out, d_randn, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, stream));
    noise_stats_kernel<<<kStatBlocks, kThreads, 0, stream>>>(d_out, total, d_partial_sum, d_partial_min, d_partial_max);
    noise_params_kernel<<<1, 64, 0, stream>>>(d_partial_sum, d_partial_min, d_partial_max, total, feature_range != nullptr,
                                              feature_range ? (float)feature_range[0] : 0.0f,
                                              feature_range ? (float)feature_range[1] : 0.0f, d_params);
    noise_scale_kernel<<<blocks, kThreads, 0, stream>>>(d_out, total, feature_range != nullptr, d_params);
    GANCE_NOISE_CHECK(hipGetLastError());
    GANCE_NOISE_CHECK(hipStreamSynchronize(stream));  // the workspace (and the host tap vectors) die here
    hipFree(workspace);
    return GANCE_OK;
}
