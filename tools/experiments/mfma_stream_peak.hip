// Microbenchmark (not product code): what a bare stream of v_mfma_f32_16x16x4_f32 reaches on this chip, at the clock it holds --
// the yardstick behind "0.87 of the nominal peak" in DESIGN.md (round-4 verdict, item 6). Independent accumulators (1 ... 32 per
// wave: a dependent accumulator comes round after 4 / 8 / 32 instructions of 32 cycles), one or two waves per SIMD, every CU busy
// for a few milliseconds; the shader clock is read beside the wall clock (s_memtime against s_memrealtime at 100 MHz).
// The same for v_mfma_f32_16x16x32_bf16 (16 cycles) for reference.
//   hipcc -O3 --offload-arch=gfx950 mfma_stream_peak.hip -o mfma_stream_peak && ./mfma_stream_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF16>
__global__ __launch_bounds__(512, 1) void stream_kernel(float* out, long long* clocks, int iters, float seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
    bf16x8 ah, bh;
#pragma unroll
    for (int k = 0; k < 8; ++k) ah[k] = (__bf16)(a + k), bh[k] = (__bf16)(b - k);
    const long long c0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if constexpr (BF16) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][3];
    const long long c1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clocks[2 * blockIdx.x] = c1 - c0;
        clocks[2 * blockIdx.x + 1] = w1 - w0;
    }
}

template <int NACC, bool BF16>
static void run(int threads, const char* label) {
    float* d_out;
    long long* d_clk;
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_clk, 256 * 2 * 8);
    const int iters = (BF16 ? 400000 : 200000) / NACC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    long long clk[512];
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        stream_kernel<NACC, BF16><<<256, threads>>>(d_out, d_clk, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost);
        }
    }
    double cyc = 0, wall = 0;
    for (int i = 0; i < 256; ++i) cyc += clk[2 * i], wall += clk[2 * i + 1];
    const double ghz = cyc / wall * 0.1;  // wall_clock64 ticks at 100 MHz
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    const double flops = mfma_per_simd * 1024 * (BF16 ? 16384.0 : 2048.0);
    const double cycles_per_mfma = cyc / 256 / ((double)iters * NACC * waves_per_simd);
    std::printf("%-34s %2d accumulators: %7.3f ms  %7.1f TFLOP/s  shader clock %.3f GHz  %.2f cycles per MFMA and SIMD  (%.3f of %s at that clock)\n", label, NACC, best,
                flops / (best * 1e-3) / 1e12, ghz, cycles_per_mfma, (BF16 ? 16.0 : 32.0) / cycles_per_mfma, BF16 ? "16" : "32");
    hipFree(d_out);
    hipFree(d_clk);
}

int main() {
    run<4, false>(256, "f32 16x16x4, 1 wave per SIMD");
    run<8, false>(256, "f32 16x16x4, 1 wave per SIMD");
    run<32, false>(256, "f32 16x16x4, 1 wave per SIMD");
    run<4, false>(512, "f32 16x16x4, 2 waves per SIMD");
    run<8, false>(512, "f32 16x16x4, 2 waves per SIMD");
    run<32, false>(512, "f32 16x16x4, 2 waves per SIMD");
    run<1, true>(256, "bf16 16x16x32, 1 wave per SIMD");
    run<2, true>(256, "bf16 16x16x32, 1 wave per SIMD");
    run<4, true>(256, "bf16 16x16x32, 1 wave per SIMD");
    run<6, true>(256, "bf16 16x16x32, 1 wave per SIMD");
    run<8, true>(256, "bf16 16x16x32, 1 wave per SIMD");
    run<32, true>(256, "bf16 16x16x32, 1 wave per SIMD");
    run<32, true>(512, "bf16 16x16x32, 2 waves per SIMD");
    return 0;
}
