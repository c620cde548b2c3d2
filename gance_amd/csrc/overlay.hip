// The eye-tracking overlay gate's pixel work on the GPU, on frames that are
// already in HBM: gance_phash_crops_u8 and gance_overlay_boxes_u8 of include/gance_hip.h.
//
//   imagehash.phash(image.crop(box))      gance/overlay/overlay_eye_tracking.py:100-108
//     = PIL convert("L") -> PIL resize((32, 32), LANCZOS) -> scipy.fftpack.dct over both axes
//       -> top-left 8x8 > median   (imagehash.phash, hash_size 8, highfreq_factor 4; the library
//       is a dependency that is not vendored in the reference: its published algorithm is restated)
//   write_boxes_onto_image                gance/overlay/overlay_common.py:104-172
//     = PIL ImageDraw.polygon mask around each bounding box + Image.composite
//
// Both PIL steps are integer algorithms and are followed bit for bit: the ITU-R 601-2 luma
// (19595 R + 38470 G + 7471 B + 0x8000) >> 16, and PIL's two-pass resample (horizontal, then
// vertical, uint8 between the passes) with its 22-bit fixed-point coefficients
// (int)(w / sum * 2^22 +- 0.5), accumulator starting at 2^21, clip8(acc >> 22). The DCT-II is the
// direct O(N^2) sum in float64 (scipy: FFT based; agreement ~1e-12 relative, which only matters
// for a coefficient that ties with the median, e.g. a perfectly flat crop).

#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace gance_overlay {

constexpr int kHashSide = 32;  // hash_size * highfreq_factor
constexpr int kLow = 8;        // hash_size
constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr double kPi = 3.14159265358979323846;  // M_PI, what PIL's sinc_filter multiplies by

__device__ __forceinline__ double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * kPi;
    return sin(x) / x;
}
__device__ __forceinline__ double lanczos_filter(double x) {
    // truncated sinc, support 3
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

// PIL precompute_coeffs + normalize_coeffs_8bpc for ONE output position `out` of an axis that
// maps in_size input pixels to 32 outputs. coeffs[0..ksize) receives the fixed-point taps.
__device__ void resample_coeffs(int in_size, int out, int ksize, int* __restrict__ coeffs, int* xmin_out, int* count_out) {
    const double scale = (double)in_size / kHashSide;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const double center = (out + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += lanczos_filter((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = lanczos_filter((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
        }
        coeffs[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << kPrecisionBits)) : (int)(0.5 + w * (double)(1 << kPrecisionBits));
    }
    *xmin_out = xmin;
    *count_out = xmax;
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One block per crop. Workspace per crop: coefficient tables [2][32][ksize] ints, bounds [2][32][2]
// ints and the horizontally resampled crop [max_h][32] bytes.
__global__ void phash_kernel(const uint8_t* __restrict__ frames, int side, const int* __restrict__ crops /*[n][5]*/,
                             int ksize, int max_h, int* __restrict__ coeff_ws, int* __restrict__ bounds_ws,
                             uint8_t* __restrict__ temp_ws, unsigned long long* __restrict__ hashes) {
    const int crop = blockIdx.x;
    const int frame = crops[crop * 5 + 0];
    const int bx = crops[crop * 5 + 1], by = crops[crop * 5 + 2], bw = crops[crop * 5 + 3], bh = crops[crop * 5 + 4];
    int* const coeffs = coeff_ws + (size_t)crop * 2 * kHashSide * ksize;
    int* const bounds = bounds_ws + (size_t)crop * 2 * kHashSide * 2;
    uint8_t* const temp = temp_ws + (size_t)crop * max_h * kHashSide;
    const uint8_t* const image = frames + (size_t)frame * side * side * 3;
    const int tid = threadIdx.x;

    __shared__ double pixels[kHashSide][kHashSide + 1];
    __shared__ double partial[kLow][kHashSide];
    __shared__ double low[kLow * kLow];
    __shared__ double median;

    // A. coefficient tables: thread t < 32 the horizontal taps of output column t, 32 <= t < 64 the vertical ones
    if (tid < 2 * kHashSide) {
        const int axis = tid / kHashSide, out = tid % kHashSide;
        int xmin, count;
        resample_coeffs(axis == 0 ? bw : bh, out, ksize, coeffs + ((size_t)axis * kHashSide + out) * ksize, &xmin, &count);
        bounds[(axis * kHashSide + out) * 2 + 0] = xmin;
        bounds[(axis * kHashSide + out) * 2 + 1] = count;
    }
    __syncthreads();

    // B. horizontal pass over every crop row: luma is formed on the fly from the RGB frame
    for (int idx = tid; idx < bh * kHashSide; idx += blockDim.x) {
        const int y = idx / kHashSide, xx = idx % kHashSide;
        const int xmin = bounds[xx * 2], count = bounds[xx * 2 + 1];
        const int* k = coeffs + (size_t)xx * ksize;
        const uint8_t* row = image + ((size_t)(by + y) * side + bx + xmin) * 3;
        int acc = 1 << (kPrecisionBits - 1);
        for (int x = 0; x < count; ++x) {
            const int luma = (row[3 * x] * 19595 + row[3 * x + 1] * 38470 + row[3 * x + 2] * 7471 + 0x8000) >> 16;
            acc += luma * k[x];
        }
        temp[(size_t)y * kHashSide + xx] = (uint8_t)clip8(acc);
    }
    __syncthreads();

    // C. vertical pass -> the 32 x 32 image, as float64 for the DCT
    for (int idx = tid; idx < kHashSide * kHashSide; idx += blockDim.x) {
        const int yy = idx / kHashSide, xx = idx % kHashSide;
        const int ymin = bounds[(kHashSide + yy) * 2], count = bounds[(kHashSide + yy) * 2 + 1];
        const int* k = coeffs + ((size_t)kHashSide + yy) * ksize;
        int acc = 1 << (kPrecisionBits - 1);
        for (int y = 0; y < count; ++y) acc += (int)temp[(size_t)(ymin + y) * kHashSide + xx] * k[y];
        pixels[yy][xx] = (double)clip8(acc);
    }
    __syncthreads();

    // D. DCT-II (scipy.fftpack.dct, unnormalised: 2 * sum x[n] cos(pi k (2n+1) / 2N)) along axis 0
    // then axis 1; only the 8 lowest frequencies of each are needed
    for (int idx = tid; idx < kLow * kHashSide; idx += blockDim.x) {
        const int k = idx / kHashSide, x = idx % kHashSide;
        double acc = 0.0;
        for (int n = 0; n < kHashSide; ++n) acc += pixels[n][x] * cos(kPi * k * (2 * n + 1) / (2.0 * kHashSide));
        partial[k][x] = 2.0 * acc;
    }
    __syncthreads();
    if (tid < kLow * kLow) {
        const int k = tid / kLow, l = tid % kLow;
        double acc = 0.0;
        for (int n = 0; n < kHashSide; ++n) acc += partial[k][n] * cos(kPi * l * (2 * n + 1) / (2.0 * kHashSide));
        low[tid] = 2.0 * acc;
    }
    __syncthreads();

    // E. numpy.median of the 64 coefficients = mean of the two middle order statistics
    if (tid == 0) {
        double sorted[kLow * kLow];
        for (int i = 0; i < kLow * kLow; ++i) {
            const double v = low[i];
            int j = i;
            while (j > 0 && sorted[j - 1] > v) {
                sorted[j] = sorted[j - 1];
                --j;
            }
            sorted[j] = v;
        }
        median = (sorted[31] + sorted[32]) / 2.0;
    }
    __syncthreads();
    if (tid < 64) {  // one wave: bit i (row-major) = coefficient i > median; bit 0 is the most significant
        const unsigned long long ballot = __ballot(low[tid] > median);
        if (tid == 0) {
            unsigned long long hash = 0;
            for (int i = 0; i < 64; ++i) hash |= ((ballot >> i) & 1ull) << (63 - i);
            hashes[crop] = hash;
        }
    }
}

// out = background, except inside the padded rectangle of any box of that frame, where it is the
// foreground. boxes: [n][5] = (frame, x, y, w, h); a rectangle spans the INCLUSIVE integer range
// PIL's polygon fill + outline covers: coordinates truncated toward zero, edges drawn.
__global__ void overlay_boxes_kernel(const uint8_t* __restrict__ foreground, const uint8_t* __restrict__ background,
                                     uint8_t* __restrict__ out, int batch, int side, const int* __restrict__ rects /*[n][5]*/,
                                     int num_rects) {
    const long long total = (long long)batch * side * side;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int frame = (int)(idx / ((long long)side * side));
        const int rem = (int)(idx % ((long long)side * side));
        const int y = rem / side, x = rem % side;
        bool inside = false;
        for (int r = 0; r < num_rects; ++r) {
            const int* rect = rects + r * 5;  // frame, x_left, y_upper, x_right, y_lower (inclusive)
            inside |= rect[0] == frame && x >= rect[1] && x <= rect[3] && y >= rect[2] && y <= rect[4];
        }
        const uint8_t* src = (inside ? foreground : background) + idx * 3;
        out[idx * 3 + 0] = src[0];
        out[idx * 3 + 1] = src[1];
        out[idx * 3 + 2] = src[2];
    }
}

static int fail(int code, const std::string& message) { return gance::set_last_error(code, message); }

}  // namespace gance_overlay

#define GANCE_OVERLAY_CHECK(expr)                                                                                  \
    do {                                                                                                           \
        hipError_t gance_err_ = (expr);                                                                            \
        if (gance_err_ != hipSuccess) {                                                                            \
            hipFree(workspace);                                                                                    \
            return gance_overlay::fail(gance_err_ == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY : GANCE_ERR_HIP, \
                                       std::string(#expr) + ": " + hipGetErrorString(gance_err_));                 \
        }                                                                                                          \
    } while (0)

extern "C" {

int gance_phash_crops_u8(const uint8_t* d_frames, int32_t num_frames, int32_t side, const int32_t* h_crops, int32_t num_crops,
                         uint64_t* h_hashes, void* stream_ptr) {
    using namespace gance_overlay;
    if (d_frames == nullptr || h_crops == nullptr || h_hashes == nullptr)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_phash_crops_u8");
    if (num_frames < 1 || side < 1 || num_crops < 0) return fail(GANCE_ERR_INVALID_ARGUMENT, "bad frame or crop count");
    if (num_crops == 0) return GANCE_OK;
    int max_w = 0, max_h = 0;
    for (int i = 0; i < num_crops; ++i) {
        const int32_t* c = h_crops + (size_t)i * 5;
        // PIL would zero-fill a crop that leaves the image; the gate's boxes come from landmarks
        // inside the frame, so anything else is a caller error here
        if (c[0] < 0 || c[0] >= num_frames || c[3] < 1 || c[4] < 1 || c[1] < 0 || c[2] < 0 || c[1] + c[3] > side ||
            c[2] + c[4] > side)
            return fail(GANCE_ERR_INVALID_ARGUMENT, "crop " + std::to_string(i) + " is empty or leaves its frame");
        max_w = std::max(max_w, (int)c[3]);
        max_h = std::max(max_h, (int)c[4]);
    }
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
        return fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    gance::DeviceGuard guard(gance::device_of_pointer(d_frames));  // launch where the frames live
    if (guard.status() != hipSuccess) return fail(GANCE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.status()));
    // PIL: ksize = ceil(support) * 2 + 1 with support = 3 * max(scale, 1)
    const int max_in = std::max(max_w, max_h);
    const double filterscale = std::max(1.0, (double)max_in / kHashSide);
    const int ksize = (int)std::ceil(3.0 * filterscale) * 2 + 1;

    hipStream_t stream = (hipStream_t)stream_ptr;
    const size_t crops_bytes = (size_t)num_crops * 5 * sizeof(int);
    const size_t coeff_bytes = (size_t)num_crops * 2 * kHashSide * ksize * sizeof(int);
    const size_t bounds_bytes = (size_t)num_crops * 2 * kHashSide * 2 * sizeof(int);
    const size_t hash_bytes = (size_t)num_crops * sizeof(unsigned long long);
    const size_t temp_bytes = ((size_t)num_crops * max_h * kHashSide + 7) / 8 * 8;
    char* workspace = nullptr;
    GANCE_OVERLAY_CHECK(hipMalloc((void**)&workspace, crops_bytes + coeff_bytes + bounds_bytes + hash_bytes + temp_bytes));
    int* d_crops = (int*)workspace;
    int* d_coeffs = (int*)(workspace + crops_bytes);
    int* d_bounds = (int*)(workspace + crops_bytes + coeff_bytes);
    unsigned long long* d_hashes = (unsigned long long*)(workspace + crops_bytes + coeff_bytes + bounds_bytes);
    uint8_t* d_temp = (uint8_t*)(workspace + crops_bytes + coeff_bytes + bounds_bytes + hash_bytes);
    GANCE_OVERLAY_CHECK(hipMemcpyAsync(d_crops, h_crops, crops_bytes, hipMemcpyHostToDevice, stream));
    phash_kernel<<<num_crops, 256, 0, stream>>>(d_frames, side, d_crops, ksize, max_h, d_coeffs, d_bounds, d_temp, d_hashes);
    GANCE_OVERLAY_CHECK(hipGetLastError());
    GANCE_OVERLAY_CHECK(hipMemcpyAsync(h_hashes, d_hashes, hash_bytes, hipMemcpyDeviceToHost, stream));
    GANCE_OVERLAY_CHECK(hipStreamSynchronize(stream));
    hipFree(workspace);
    return GANCE_OK;
}

int gance_overlay_boxes_u8(const uint8_t* d_foreground, const uint8_t* d_background, uint8_t* d_out, int32_t num_frames,
                           int32_t side, const int32_t* h_boxes, int32_t num_boxes, void* stream_ptr) {
    using namespace gance_overlay;
    if (d_foreground == nullptr || d_background == nullptr || d_out == nullptr || (num_boxes > 0 && h_boxes == nullptr))
        return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_overlay_boxes_u8");
    if (num_frames < 1 || side < 1 || num_boxes < 0) return fail(GANCE_ERR_INVALID_ARGUMENT, "bad frame or box count");
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
        return fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    gance::DeviceGuard guard(gance::device_of_pointer(d_out));  // launch where the frames live
    if (guard.status() != hipSuccess) return fail(GANCE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.status()));
    // the mask rectangle of overlay_common._draw_mask (:104-137): pads scale with the frame,
    // PIL truncates the float corners toward zero and draws the polygon's edges
    std::vector<int> rects((size_t)std::max(num_boxes, 1) * 5);
    const double y_pad = side * 0.058, x_pad = side * 0.098;
    for (int i = 0; i < num_boxes; ++i) {
        const int32_t* b = h_boxes + (size_t)i * 5;
        if (b[0] < 0 || b[0] >= num_frames) return fail(GANCE_ERR_INVALID_ARGUMENT, "box frame index out of range");
        const double y_center = b[2] + b[4] / 2.0;
        rects[i * 5 + 0] = b[0];
        rects[i * 5 + 1] = (int)(b[1] - x_pad);
        rects[i * 5 + 2] = (int)(y_center - y_pad);
        rects[i * 5 + 3] = (int)(b[1] + (b[3] + x_pad));
        rects[i * 5 + 4] = (int)(y_center + y_pad);
    }
    hipStream_t stream = (hipStream_t)stream_ptr;
    char* workspace = nullptr;
    GANCE_OVERLAY_CHECK(hipMalloc((void**)&workspace, rects.size() * sizeof(int)));
    GANCE_OVERLAY_CHECK(hipMemcpyAsync(workspace, rects.data(), rects.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    const long long total = (long long)num_frames * side * side;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    overlay_boxes_kernel<<<blocks, 256, 0, stream>>>(d_foreground, d_background, d_out, num_frames, side, (const int*)workspace,
                                                     num_boxes);
    GANCE_OVERLAY_CHECK(hipGetLastError());
    GANCE_OVERLAY_CHECK(hipStreamSynchronize(stream));  // `rects` and the workspace die here
    hipFree(workspace);
    return GANCE_OK;
}

}  // extern "C"
