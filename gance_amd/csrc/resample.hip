// Time-stretching the audio to one vector per video frame on the GPU: gance_resample_audio_f32 of
// include/gance_hip.h. Replaces `resampy.resample(..., filter="kaiser_best")` of
// _scale_wav_to_sample_rate (gance/vector_sources/music.py:212-230). resampy is a third-party
// dependency that is absent here and whose sample values no reference test pins (only the output
// length, test/test_vector_source_music.py:13-24), so this is this implementation's own band-limited
// interpolator of the same published design: a Kaiser-windowed sinc with 64 zero crossings,
// beta 14.7697 and roll-off 0.9476, evaluated analytically in float64 per tap (resampy interpolates
// a 512-per-crossing table). Output length int(num_in * sr_new / sr_orig), zero beyond the ends.
//
// One thread per output sample; an output reads <= 2*ceil(64/scale)+1 consecutive input samples
// (129 when up-sampling, ~190 for 44.1 kHz -> 30.72 kHz), neighbouring threads read overlapping
// windows: the input (5 MB for 30 s) lives in L2. ALU-bound on sin + I0: ~1 ms for 30 s of audio.

#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace gance_resample {

constexpr int kZeroCrossings = 64;
constexpr double kKaiserBeta = 14.769656459379492;
constexpr double kRolloff = 0.9475937167399596;
constexpr double kPi = 3.14159265358979323846;

__global__ void resample_kernel(const float* __restrict__ in, long long num_in, float* __restrict__ out, long long num_out,
                                double ratio, double scale, int half_width, double inv_i0_beta) {
    // resampy's published design (`sinc_window` + `resample_f`): with x = (position - index) * min(1, ratio),
    // weight = rolloff * sinc(rolloff * x) * kaiser(x / zero_crossings) * min(1, ratio) for |x| < zero_crossings:
    // the roll-off narrows the sinc only, the Kaiser taper spans the un-scaled 64 crossings
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_out) return;
    const double position = (double)i / ratio;
    const long long centre = (long long)floor(position);
    double acc = 0.0;
    for (int t = -half_width; t <= half_width; ++t) {
        const long long index = centre + t;
        if (index < 0 || index >= num_in) continue;
        const double offset = (position - (double)index) * scale;
        const double window_arg = offset / kZeroCrossings;
        if (fabs(window_arg) >= 1.0) continue;
        const double px = kPi * kRolloff * offset;
        const double sinc = offset == 0.0 ? 1.0 : sin(px) / px;
        const double kaiser = cyl_bessel_i0(kKaiserBeta * sqrt(fmax(1.0 - window_arg * window_arg, 0.0))) * inv_i0_beta;
        acc += (double)in[index] * (sinc * kaiser * (kRolloff * scale));
    }
    out[i] = (float)acc;
}

__global__ void i0_kernel(double x, double* out) { out[0] = cyl_bessel_i0(x); }

static int fail(int code, const std::string& message) { return gance::set_last_error(code, message); }

}  // namespace gance_resample

extern "C" int gance_resample_audio_f32(const float* d_in, uint64_t num_in, double sr_orig, double sr_new, float* d_out,
                                        uint64_t num_out, void* stream_ptr) {
    using namespace gance_resample;
    if (d_in == nullptr || d_out == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_resample_audio_f32");
    if (!(sr_orig > 0.0) || !(sr_new > 0.0)) return fail(GANCE_ERR_INVALID_ARGUMENT, "sample rates must be positive");
    const double ratio = sr_new / sr_orig;
    if (num_in < 1 || num_out != (uint64_t)((double)num_in * ratio))
        return fail(GANCE_ERR_INVALID_ARGUMENT, "num_out must be int(num_in * sr_new / sr_orig) = " +
                                                    std::to_string((uint64_t)((double)num_in * ratio)));
    if (num_out == 0) return GANCE_OK;
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0)
        return fail(GANCE_ERR_NO_DEVICE, "no HIP device visible; libgance_hip has no CPU path");
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));  // launch where the samples live
    if (guard.status() != hipSuccess) return fail(GANCE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.status()));
    hipStream_t stream = (hipStream_t)stream_ptr;
    const double scale = ratio < 1.0 ? ratio : 1.0;  // the filter is stretched to the lower of the two Nyquist rates
    const int half_width = (int)std::ceil(kZeroCrossings / scale);
    // I0(beta) from the same device routine as the taps, so the window is exactly 1 at its centre
    double* d_norm = nullptr;
    double norm = 0.0;
    if (hipMalloc((void**)&d_norm, sizeof(double)) != hipSuccess) return fail(GANCE_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    i0_kernel<<<1, 1, 0, stream>>>(kKaiserBeta, d_norm);
    hipError_t err = hipMemcpyAsync(&norm, d_norm, sizeof(double), hipMemcpyDeviceToHost, stream);
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    hipFree(d_norm);
    if (err != hipSuccess || !(norm > 0.0)) return fail(GANCE_ERR_HIP, std::string("I0 normalisation: ") + hipGetErrorString(err));
    const unsigned blocks = (unsigned)((num_out + 255) / 256);
    resample_kernel<<<blocks, 256, 0, stream>>>(d_in, (long long)num_in, d_out, (long long)num_out, ratio, scale, half_width,
                                                1.0 / norm);
    err = hipGetLastError();
    if (err != hipSuccess) return fail(GANCE_ERR_HIP, std::string("resample_kernel: ") + hipGetErrorString(err));
    return GANCE_OK;
}
