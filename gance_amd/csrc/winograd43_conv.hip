// Winograd F(4x4, 3x3) form of the modulated 3x3 stride-1 convolution on v_mfma_f32_16x16x4_f32 (SURVEY.md §8 a18:
// `modulated_conv2d_layer` -> tf.nn.conv2d), for the Conv1 layers whose K loop is deep (64^2 ... 256^2).
//
// Why. F(2x2,3x3) (winograd64_conv.hip) spends 16 multiplies on 4 outputs, F(4x4,3x3) 36 on 16: 2.25 instead of 4 per
// output, i.e. 0.5625 of the matrix work. In float32 the larger transform costs nothing measurable in accuracy on this
// network (tools/experiments/winograd_f43_error.py: max |image - fp64 oracle| 1.4e-5 with F(4x4,3x3) on the 64^2 ...
// 256^2 layers against 0.8e-5 with F(2x2,3x3), bar 1e-3; interpolation points 0, +-1, +-2, weights transformed in
// float64 on the host).
//
// What it costs on gfx950. The fp32 MFMA runs at the fp32 VECTOR rate and shares the vector ALUs, so the 144 vector
// instructions of a 6x6 input transform are not hidden beside the 36 MFMAs they feed -- they add their issue time. A
// wave alone on its SIMD issues a vector instruction every 4 cycles, two waves together every 2: this kernel therefore
// runs TWO waves per SIMD (8 per block, <= 256 registers each): 36 positions x one 16 x 16 accumulator tile = 144
// accumulators per wave, 36 MFMAs (1152 cycles) per k-step of four input channels against ~290 cycles of transform.
//
// Geometry. Block = 8 waves = 2 channel tiles of 16 x 4 tile rows; a wave owns 16 channels x one row of sixteen 4x4 output
// tiles (4 x 64 pixels): the block 32 channels x 16 x 64 pixels. Lane (n = lane % 16, g = lane / 16) transforms the 6x6
// window of tile n for input channel 4 ks + g; as A operand it holds output channel n of its channel tile for the same
// input channel. The input arrives ALREADY multiplied by this layer's style (ConvArgs::x contract, as for the
// 32-channel geometry of winograd64_conv.hip: the producing up layer folds s[b][ci] into its leaky ReLU; V is linear in d).
//
// Staging. K chunk = 4 input channels = one k-step: transformed weights [channel tile][ci][co % 16][36 positions] (a
// lane's 36 weights are contiguous: six 8-byte reads per window column; (tile, ci) units padded to 578 floats so the four
// g groups of a read fall into different banks) + the haloed patch [4][18][72], together 40 LDS-DMA pieces of 1 KB, five
// per wave, three ring slots (120 KB). One wait + barrier per k-step; chunk q+2 is issued right behind the barrier of
// k-step q into the slot k-step q-1 read.
//
// Epilogue: output transform A^T M A in registers (the 36 positions of a (channel, tile) pair live in ONE lane),
// demodulation, noise, bias, leaky ReLU, optional scale by the next layer's style, 16-byte stores (256 contiguous
// bytes per 16 lanes).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kKC = 4;                         // input channels per chunk = one k-step of the 16x16x4 MFMA
constexpr int kBM = 32;                        // output channels per block (2 channel tiles of 16)
constexpr int kTW = 64, kTH = 16;              // pixels per block
constexpr int kPW = kTW + 8, kPH = kTH + 2;    // haloed patch: 18 rows x 72 columns (16-byte aligned row segments)
constexpr int kPlane = kPH * kPW;              // 1296 floats per input channel
constexpr int kUnit = 16 * 36 + 2;             // a (channel tile, ci) unit of weights: [co % 16][36] + 2 floats of padding
constexpr int kWPieces = 19;                   // 8 units = 4624 floats -> 19 pieces of 256 floats (zero padded in HBM)
constexpr int kWFloats = kWPieces * 256;       // 4864
constexpr int kPatchF4 = kKC * kPlane / 4;     // 1296 float4
constexpr int kPPieces = (kPatchF4 + 63) / 64; // 21 (the last one a quarter full)
constexpr int kPieces = kWPieces + kPPieces;   // 40: five per wave
constexpr int kPiecesPerWave = kPieces / 8;
static_assert(kPieces == 40 && kPiecesPerWave == 5, "five LDS-DMA pieces per wave and chunk");
constexpr int kSlot = kPieces * 256;           // 10240 floats = 40 KB
constexpr int kNBUF = 3;

// B^T of F(4,3), points 0, +-1, +-2 (Lavin & Gray): 12 vector instructions
__device__ __forceinline__ void input_transform6(float d0, float d1, float d2, float d3, float d4, float d5, float (&t)[6]) {
    const float a = fmaf(-4.f, d2, d4);
    const float b = fmaf(-4.f, d1, d3);
    const float c = d4 - d2;
    const float e = d3 - d1;
    t[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    t[1] = a + b;
    t[2] = a - b;
    t[3] = fmaf(2.f, e, c);
    t[4] = fmaf(-2.f, e, c);
    t[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}

// A^T of F(4,3): 10 vector instructions
__device__ __forceinline__ void output_transform6(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y[0] = m0 + s1 + s2;
    y[1] = fmaf(2.f, d2, d1);
    y[2] = fmaf(4.f, s2, s1);
    y[3] = fmaf(8.f, d2, d1) + m5;
}

}  // namespace

__global__ __launch_bounds__(512, 1) void winograd43_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n16 = lane & 15, g = lane >> 4;
    const int cot = wave >> 2, pg = wave & 3;  // channel tile and tile row of this wave
    const int Hp = p.H + 2, Wp = p.W + 8;
    const int n = p.total_chunks;

    // virtual block id -> tile, XCD-aware (consecutive ids run on one XCD: the channel tiles of a pixel tile share its L2)
    int b0, m_tile, y0, x0;
    {
        const int v = (int)blockIdx.x;
        const int nwg = p.total_tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        m_tile = id % p.m_tiles;
        id /= p.m_tiles;
        x0 = (id % p.tiles_x) * kTW;
        id /= p.tiles_x;
        y0 = (id % p.tiles_y) * kTH;
        b0 = id / p.tiles_y;
    }

    // ---- LDS-DMA pieces of this wave: piece wave * 5 + r; < 19: weights (linear), else patch piece (per-lane source offset)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b0 * p.x_b_stride), 0, 0x7fffffff, 0x00020000);
    int piece_voff[kPiecesPerWave];
#pragma unroll
    for (int r = 0; r < kPiecesPerWave; ++r) {
        const int piece = wave * kPiecesPerWave + r;
        if (piece < kWPieces) {
            piece_voff[r] = piece * 1024 + lane * 16;
        } else {
            int f = (piece - kWPieces) * 64 + lane;
            if (f >= kPatchF4) f = (piece - kWPieces) * 64;  // (the last piece is a quarter full: the rest re-copy its first float4 into padding)
            const int q4 = f % (kPW / 4);
            const int row = (f / (kPW / 4)) % kPH;
            const int c = f / (kPW / 4 * kPH);
            piece_voff[r] = ((c * Hp + row) * Wp + 4 * q4) * 4;
        }
    }
    const int w_tile_base = m_tile * n * (kWFloats * 4);   // bytes: [m tile][chunk][kWFloats]
    const int x_tile_base = (y0 * Wp + x0) * 4;            // patch row 0 = image row y0 - 1 = buffer row y0; column x0 - 4 = buffer column x0
    const int x_chunk_step = kKC * Hp * Wp * 4;
    auto issue_chunk = [&](int chunk, int slot) {
        float* const base = smem + slot * kSlot;
#pragma unroll
        for (int r = 0; r < kPiecesPerWave; ++r) {
            const int piece = wave * kPiecesPerWave + r;  // (wave is scalar: the branch is uniform)
            if (piece < kWPieces)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + piece * 256), 16, piece_voff[r], w_tile_base + chunk * (kWFloats * 4), 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(base + kWFloats + (piece - kWPieces) * 256), 16, piece_voff[r],
                                                         x_tile_base + chunk * x_chunk_step, 0, 0);
        }
    };

    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand offsets of this lane inside a slot (floats)
    const int win_off = kWFloats + g * kPlane + (4 * pg) * kPW + 4 * n16;  // window rows 4 pg .. + 5, columns 4 n + 3 .. + 8
    const int a_off = (cot * 4 + g) * kUnit + n16 * 36;

    issue_chunk(0, 0);
    if (n > 1) issue_chunk(1, 1);

    int slot = 0;
    for (int q = 0; q < n; ++q) {
        if (q + 1 < n)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerWave) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (q + 2 < n) issue_chunk(q + 2, slot == 0 ? 2 : slot - 1);  // (q + 2) % 3 == (slot + 2) % 3
        const float* const P = smem + slot * kSlot + win_off;
        const float* const U = smem + slot * kSlot + a_off;
        // row pass: W = d B (along x), six rows of the window
        float wt[6][6];
#pragma unroll
        for (int y = 0; y < 6; ++y) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(P + y * kPW);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(P + y * kPW + 4);
            const float last = P[y * kPW + 8];
            input_transform6(lo[3], hi[0], hi[1], hi[2], hi[3], last, wt[y]);
        }
        // column pass + the six positions of a column: V[.][j] = B^T W[.][j]; position j * 6 + i
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float v[6];
            input_transform6(wt[0][j], wt[1][j], wt[2][j], wt[3][j], wt[4][j], wt[5][j], v);
            const f32x2 a01 = *reinterpret_cast<const f32x2*>(U + j * 6);
            const f32x2 a23 = *reinterpret_cast<const f32x2*>(U + j * 6 + 2);
            const f32x2 a45 = *reinterpret_cast<const f32x2*>(U + j * 6 + 4);
            const float a[6] = {a01[0], a01[1], a23[0], a23[1], a45[0], a45[1]};
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[j * 6 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], v[i], acc[j * 6 + i], 0, 0, 0);
        }
        slot = slot == kNBUF - 1 ? 0 : slot + 1;
    }

    // ---- epilogue: lane (n16, g) holds channels 4 g + r (r = 0 .. 3) of tile n16 for all 36 positions ----
    const int oy0 = y0 + 4 * pg, ox0 = x0 + 4 * n16;
    const int co0 = m_tile * kBM + cot * 16 + 4 * g;
    const f32x4 dm = *reinterpret_cast<const f32x4*>(p.d + (size_t)b0 * p.d_stride + co0);
    const f32x4 bm = *reinterpret_cast<const f32x4*>(p.bias + co0);
    f32x4 sn = f32x4{1.f, 1.f, 1.f, 1.f};
    if (p.s_next != nullptr) sn = *reinterpret_cast<const f32x4*>(p.s_next + (size_t)b0 * p.s_stride + co0);
    f32x4 nz[4];
#pragma unroll
    for (int oy = 0; oy < 4; ++oy) {
        nz[oy] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.noise != nullptr) nz[oy] = *reinterpret_cast<const f32x4*>(p.noise + (size_t)(oy0 + oy) * p.OW + ox0) * p.noise_strength;
    }
    float* const out_base = p.out + (size_t)b0 * p.out_b_stride + (size_t)co0 * p.out_c_stride +
                            (size_t)(oy0 + p.out_y_off) * p.out_row_stride + ox0 + p.out_x_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t[4][6];  // A^T M: [output row][position column]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float col[4];
            output_transform6(acc[j * 6 + 0][r], acc[j * 6 + 1][r], acc[j * 6 + 2][r], acc[j * 6 + 3][r], acc[j * 6 + 4][r], acc[j * 6 + 5][r], col);
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) t[oy][j] = col[oy];
        }
#pragma unroll
        for (int oy = 0; oy < 4; ++oy) {
            float yrow[4];
            output_transform6(t[oy][0], t[oy][1], t[oy][2], t[oy][3], t[oy][4], t[oy][5], yrow);
            f32x4 v = f32x4{yrow[0], yrow[1], yrow[2], yrow[3]} * dm[r] + nz[oy] + bm[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.2f * v[k]) * 1.4142135623730951f;
            v *= sn[r];
            *reinterpret_cast<f32x4*>(out_base + (size_t)r * p.out_c_stride + (size_t)oy * p.out_row_stride) = v;
        }
    }
}

bool winograd43_supported(int cin, int cout, int H, int W) {
    return cin % kKC == 0 && cin / kKC >= 2 && cout % kBM == 0 && H % kTH == 0 && W % kTW == 0;
}

size_t winograd43_weight_floats(int cin, int cout) { return (size_t)(cout / kBM) * (cin / kKC) * kWFloats; }

// w_in: the layer's runtime-scaled weights [tap = ky*3+kx][ci][co]; w_out: [m tile of 32][chunk of 4][channel tile][ci][co % 16][36]
// (units of 578 floats, the chunk zero padded to 4864), position j * 6 + i = (G g G^T)[i][j], i along y
void winograd43_transform_weights(const float* w_in, int cin, int cout, float* w_out) {
    const double G[6][3] = {{1. / 4, 0., 0.},          {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    const int chunks = cin / kKC;
    std::fill(w_out, w_out + winograd43_weight_floats(cin, cout), 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double gk[3][3], tmp[6][3], u[6][6];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) gk[ky][kx] = w_in[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
            for (int i = 0; i < 6; ++i)
                for (int kx = 0; kx < 3; ++kx) tmp[i][kx] = G[i][0] * gk[0][kx] + G[i][1] * gk[1][kx] + G[i][2] * gk[2][kx];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) u[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int mtile = co / kBM, m = co % kBM, ch = ci / kKC, kc = ci % kKC;
            float* dst = w_out + ((size_t)mtile * chunks + ch) * kWFloats + ((m / 16) * 4 + kc) * kUnit + (m % 16) * 36;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) dst[j * 6 + i] = (float)u[i][j];
        }
}

hipError_t launch_winograd43_conv(const ConvArgs& args, hipStream_t stream) {
    if (args.epilogue != kEpilogueFull || args.out == nullptr || !winograd43_supported(args.Cin, args.Cout, args.H, args.W)) return hipErrorInvalidValue;
    static PerDeviceInt configured;
    int unused = 0;
    hipError_t e = configured.get(
        [&](int, int* value) {
            *value = 1;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(winograd43_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(kNBUF * kSlot * sizeof(float)));
        },
        &unused);
    if (e != hipSuccess) return e;
    ConvArgs a = args;
    a.tiles_x = a.W / kTW;
    a.tiles_y = a.H / kTH;
    a.m_tiles = a.Cout / kBM;
    a.total_chunks = a.Cin / kKC;
    a.total_tiles = a.m_tiles * a.tiles_x * a.tiles_y * a.B;
    hipLaunchKernelGGL(winograd43_kernel, dim3(a.total_tiles), dim3(512), kNBUF * kSlot * sizeof(float), stream, a);
    return hipGetLastError();
}

}  // namespace gance
