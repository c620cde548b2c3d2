// Modulated 3x3 convolution of StyleGAN2's synthesis network as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// Replaces, for the reference's synthesis call (gance/network_interface/network_functions.py:168),
// the un-vendored `modulated_conv2d_layer` -> tf.nn.conv2d / conv2d_transpose (cuDNN) chain
// (SURVEY.md §8 a18).
//
// Formulation (not the reference's): weights are shared by the whole batch,
//     out[b,co,p] = d[b,co] * sum_{tap,ci} w[tap,ci,co] * ( s[b,ci] * x[b,ci,p+tap] )
// i.e. GEMM  D[M = co][N = pixel] = A[M][K] * B[K][N],  K = taps x Cin, with the style scale s
// applied while the input patch is staged into LDS and the demodulation d applied in the
// epilogue. M = output channel is on the accumulator ROWS so that one accumulator register of a
// wave is 32 consecutive pixels of one channel plane: every global store is a full 128-B segment.
//
// The same kernel computes the stride-2 transposed convolution of the Conv0_up layers, one launch
// per output-parity class (even/even: 4 taps, even/odd and odd/even: 2 taps, odd/odd: 1 tap): a
// class is a small stride-1 convolution from the H x W input grid to one parity plane of the
// (2H+1) x (2W+1) intermediate, so no multiply touches an inserted zero.
//
// Block = 256 threads = 4 waves. Per K-chunk of KC input channels the block stages
//   Wl[tap][KC][BM]              (float4 global loads, co contiguous)
//   Pl[TB][KC][TH+2][TW+2]       (the haloed input patch, scaled by s, zero outside the image)
// into LDS and every wave runs taps x KC/2 MFMA steps on its MT x NT grid of 32x32 accumulators.
// Operand fetches are conflict-free ds_read_b32: 32 consecutive co for A, 32 consecutive x for B.

#include <hip/hip_runtime.h>

#include "kernels.h"

namespace gance {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN>
struct ConvTile {
    static constexpr int kBN = TB * TH * TW;
    static constexpr int kMT = BM / (32 * WM);
    static constexpr int kNT = kBN / (32 * WN);
    static constexpr int kPH = TH + 2;
    static constexpr int kPW = TW + 2;
    static constexpr int kPlane = kPH * kPW;
    static constexpr int kWlFloats = kMaxTaps * KC * BM;
    static constexpr int kPlFloats = TB * KC * kPlane;
    static constexpr size_t kLdsBytes = (size_t)(kWlFloats + kPlFloats) * sizeof(float);
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BM % (32 * WM) == 0 && kBN % (32 * WN) == 0, "wave tiling");
    static_assert(KC % 2 == 0, "k pairs");
};

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN>
__global__ __launch_bounds__(256) void modconv_mfma_kernel(const ConvArgs p) {
    using T = ConvTile<BM, TB, TH, TW, KC, WM, WN>;
    constexpr int MT = T::kMT, NT = T::kNT, PW = T::kPW, PLANE = T::kPlane;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                 // [tap][KC][BM]
    float* Pl = smem + T::kWlFloats;  // [TB][KC][PH][PW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // ---- block -> (m tile, k split, pixel tile) ----
    int id = blockIdx.x;
    const int m_tile = id % p.m_tiles;
    id /= p.m_tiles;
    const int split = id % p.nsplit;
    id /= p.nsplit;
    const int tile_x = id % p.tiles_x;
    id /= p.tiles_x;
    const int tile_y = id % p.tiles_y;
    const int tile_b = id / p.tiles_y;

    const int m0 = m_tile * BM;
    const int b0 = tile_b * TB;
    const int y0 = tile_y * TH;
    const int x0 = tile_x * TW;

    // ---- per-lane B-operand base offsets (one per N tile of the wave) ----
    int boff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + l31;
        const int tb = n / (TH * TW);
        const int yy = (n / TW) % TH;
        const int xx = n % TW;
        boff[j] = (tb * KC + lh) * PLANE + (yy + 1) * PW + (xx + 1);
    }
    const int aoff = lh * BM + wm * (MT * 32) + l31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int chunk_begin = split * p.chunks_per_split;
    const int chunk_end = chunk_begin + p.chunks_per_split;
    const size_t in_plane = (size_t)p.H * p.W;

    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const int ci0 = chunk * KC;
        __syncthreads();  // the previous chunk's MFMAs have read Wl / Pl

        // ---- stage weights: Wl[t][c][m] = w[tap_w[t]][ci0 + c][m0 + m] ----
        for (int t = 0; t < p.ntaps; ++t) {
            const float* wsrc = p.w + ((size_t)p.tap_w[t] * p.Cin + ci0) * p.Cout + m0;
            for (int v = tid; v < KC * BM / 4; v += 256) {
                const int c = v / (BM / 4);
                const int m4 = v % (BM / 4);
                const float4 val =
                    *reinterpret_cast<const float4*>(wsrc + (size_t)c * p.Cout + m4 * 4);
                *reinterpret_cast<float4*>(&Wl[(t * KC + c) * BM + m4 * 4]) = val;
            }
        }
        // ---- stage the haloed input patch, scaled by the style ----
        for (int e = tid; e < T::kPlFloats; e += 256) {
            const int px = e % PW;
            const int py = (e / PW) % T::kPH;
            const int c = (e / PLANE) % KC;
            const int tb = e / (PLANE * KC);
            const int b = b0 + tb;
            const int gy = y0 + py - 1;
            const int gx = x0 + px - 1;
            float val = 0.f;
            if (b < p.B && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                const int ci = ci0 + c;
                val = p.x[(size_t)b * p.x_b_stride + (size_t)ci * in_plane + (size_t)gy * p.W + gx] *
                      p.s[(size_t)b * p.s_stride + ci];
            }
            Pl[e] = val;
        }
        __syncthreads();

        // ---- taps x KC/2 MFMA steps ----
        for (int t = 0; t < p.ntaps; ++t) {
            const int toff = p.tap_dy[t] * PW + p.tap_dx[t];
            const float* wl_t = Wl + t * (KC * BM) + aoff;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                float a[MT], bv[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = wl_t[(2 * kk) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = Pl[boff[j] + toff + (2 * kk) * PLANE];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: demodulate, (noise, bias, leaky relu), store 32 consecutive pixels per reg ----
    float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (wn * NT + j) * 32 + l31;
        const int tb = n / (TH * TW);
        const int oy = y0 + (n / TW) % TH;
        const int ox = x0 + n % TW;
        const int b = b0 + tb;
        if (b >= p.B || oy >= p.OH || ox >= p.OW) continue;
        float nz = 0.f;
        if (p.epilogue == kEpilogueFull && p.noise != nullptr)
            nz = p.noise[(size_t)oy * p.OW + ox] * p.noise_strength;
        float* out_px = out + (size_t)b * p.out_b_stride + (size_t)oy * p.out_row_stride + ox;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] * p.d[(size_t)b * p.d_stride + co];
                if (p.epilogue == kEpilogueFull) {
                    v += nz + p.bias[co];
                    v = (v < 0.f ? 0.2f * v : v) * 1.4142135623730951f;
                }
                out_px[(size_t)co * p.out_c_stride] = v;
            }
        }
    }
}

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN>
static hipError_t launch_one(const ConvArgs& a, int total_blocks, hipStream_t stream) {
    using T = ConvTile<BM, TB, TH, TW, KC, WM, WN>;
    auto kernel = modconv_mfma_kernel<BM, TB, TH, TW, KC, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)T::kLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(total_blocks), dim3(256), T::kLdsBytes, stream, a);
    return hipGetLastError();
}

const ConvTileInfo kConvTiles[kNumConvTiles] = {
    // BM, TB, TH, TW, KC
    {32, 1, 8, 64, 16},   // 0: Cout = 32  (1024^2)
    {64, 1, 4, 64, 16},   // 1: Cout = 64  (512^2)
    {128, 1, 4, 32, 8},   // 2: Cout >= 128, wide grids
    {128, 1, 8, 16, 8},   // 3: 16-wide grids
    {128, 2, 8, 8, 8},    // 4: 8-wide grids
    {128, 8, 4, 4, 8},    // 5: 4-wide grids
};

hipError_t launch_modconv(int tile_id, const ConvArgs& a, int total_blocks, hipStream_t stream) {
    switch (tile_id) {
        case 0: return launch_one<32, 1, 8, 64, 16, 1, 4>(a, total_blocks, stream);
        case 1: return launch_one<64, 1, 4, 64, 16, 1, 4>(a, total_blocks, stream);
        case 2: return launch_one<128, 1, 4, 32, 8, 2, 2>(a, total_blocks, stream);
        case 3: return launch_one<128, 1, 8, 16, 8, 2, 2>(a, total_blocks, stream);
        case 4: return launch_one<128, 2, 8, 8, 8, 2, 2>(a, total_blocks, stream);
        case 5: return launch_one<128, 8, 4, 4, 8, 2, 2>(a, total_blocks, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gance
