// Go / no-go experiment (not product code): can ONE wave per SIMD keep the fp32 matrix pipe busy in a
// Winograd F(2x2,3x3) inner loop -- 16 position accumulators (256 accumulator registers), B operands
// made on the fly from the LDS patch by the input transform (8 ds_read_b64 + 32 adds per k-step),
// A operands read from LDS (16 ds_read_b32) and scaled by the style -- with no global traffic at all?
//   hipcc -O3 --offload-arch=gfx950 winograd_inner_loop.hip -o /tmp/wino && /tmp/wino
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 4, BM = 32, PH = 10, PW = 72;

__global__ __launch_bounds__(256, 1) void wino_loop(const float* __restrict__ init, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float U[16 * KC * BM];
    __shared__ __attribute__((aligned(16))) float P[KC * PH * PW];
    __shared__ float S[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 16 * KC * BM; i += 256) U[i] = init[i];
    for (int i = tid; i < KC * PH * PW; i += 256) P[i] = init[4096 + i];
    if (tid < 64) S[tid] = 1.0f + 0.001f * tid;
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    // tile n = l31 of tile row `wave`: patch rows 2*wave .. 2*wave+3, cols 2n+4 .. 2n+7
    const float* pbase = P + (2 * wave) * PW + 2 * l31 + 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int ci = 2 * kk + lh;
            const float* pc = pbase + ci * (PH * PW);
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 a = *reinterpret_cast<const float2*>(pc + r * PW);
                const float2 b = *reinterpret_cast<const float2*>(pc + r * PW + 2);
                d[r][0] = a.x; d[r][1] = a.y; d[r][2] = b.x; d[r][3] = b.y;
            }
            // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
            float t[4][4], V[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[r * 4 + 0] = t[r][0] - t[r][2];
                V[r * 4 + 1] = t[r][1] + t[r][2];
                V[r * 4 + 2] = t[r][2] - t[r][1];
                V[r * 4 + 3] = t[r][1] - t[r][3];
            }
            const float s = S[ci + (it & 7)];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float a = U[(p * KC + ci) * BM + l31] * s;
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, V[p], acc[p], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[p][r];
    out[(size_t)blockIdx.x * 256 + tid] = sum;
}

int main() {
    const int blocks = 256, iters = 4000;
    std::vector<float> h(4096 + KC * PH * PW);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2654435761u >> 20) & 1023) - 0.5f;
    float *d_init, *d_out;
    hipMalloc(&d_init, h.size() * 4);
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipMemcpy(d_init, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        wino_loop<<<blocks, 256>>>(d_init, d_out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * 4 * iters * (KC / 2) * 16;  // per-wave MFMAs, all waves
        const double flops = mfma * 2.0 * 32 * 32 * 2;
        std::printf("rep %d: %.3f ms, %.1f TFLOP/s executed (%.3f of 157.3), %.1f cycles/MFMA/SIMD at 2.3 GHz\n", rep, ms,
                    flops / ms / 1e9, flops / ms / 1e9 / 157.3, ms * 1e-3 * 2.3e9 / ((double)iters * (KC / 2) * 16));
    }
    return 0;
}
