// Microbenchmark (not product code): how many single-issue VALU instructions hide in the shadow of one
// v_mfma_f32_32x32x2_f32 (64-cycle) when ONE wave per SIMD issues both, and when TWO waves share the SIMD?
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_fillers.hip -o mfma_f32_fillers.co && ./mfma_f32_fillers.co
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FILL, int NACC>
__global__ __launch_bounds__(NACC == 16 ? 256 : 512, 1) void loop_kernel(float* out, int iters, float seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < FILL; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][7];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FILL, int NACC>
static void run(int threads, const char* label) {
    float* d_out;
    hipMalloc(&d_out, 256 * 1024 * 4);
    const int iters = 20000 / NACC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        loop_kernel<FILL, NACC><<<256, threads>>>(d_out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfma_per_simd = (double)iters * NACC * (threads / 256);  // waves per SIMD x MFMAs per wave
    std::printf("%s fill %d: %.3f ms, %.1f cycles per MFMA per SIMD (2.4 GHz)\n", label, FILL, best, best * 1e-3 * 2.4e9 / mfma_per_simd);
    hipFree(d_out);
}

int main() {
    run<0, 16>(256, "1 wave/SIMD");
    run<2, 16>(256, "1 wave/SIMD");
    run<4, 16>(256, "1 wave/SIMD");
    run<6, 16>(256, "1 wave/SIMD");
    run<8, 16>(256, "1 wave/SIMD");
    run<0, 8>(512, "2 waves/SIMD");
    run<4, 8>(512, "2 waves/SIMD");
    run<6, 8>(512, "2 waves/SIMD");
    run<8, 8>(512, "2 waves/SIMD");
    return 0;
}
