// Winograd F(2x2, 3x3) form of the modulated 3x3 stride-1 convolution on the fp32 matrix cores: the
// same layer as conv_mfma.hip's UP = false kernel, executing 16 multiply-adds per
// 2x2 output tile and input channel instead of 36.
//
//   out tile Y (2x2) = A^T [ sum_ci  U[ci,co] (.) V[ci] ] A,   U = G w G^T (4x4, precomputed per layer),
//   V = B^T d B of the 4x4 input window d (stride 2), (.) = element-wise over the 16 positions.
// Each of the 16 positions is an independent GEMM  M_p[co][tile] += U_p[co][ci] * V_p[ci][tile]:
// sixteen 32x32 accumulator tiles per wave (256 accumulator registers), hence ONE wave per SIMD and one
// block per CU, with a deep LDS ring (buffer LDS-DMA, counted vmcnt) standing in for the latency
// hiding that co-resident blocks give the direct kernel.
//
// Block = 4 waves = 32 output channels x (8 x 64) pixels = 4 x 32 Winograd tiles; wave w owns tile
// row w. Per K chunk (KC = 4 input channels) LDS holds the transformed weights [16][KC][32] and the
// haloed activation patch [KC][10][72] (the direct kernel's patch image, same LDS-DMA code). A lane
// builds its own B operands: it reads the 4x4 window of its tile for its channel from the patch
// (12 LDS reads), applies B^T . B (32 adds) and feeds the 16 results straight to the 16 MFMAs of the
// k-step; the style scale multiplies the weight fragment. The output transform, demodulation,
// noise, bias and leaky ReLU run on the accumulators in registers; a lane stores its tile's 2x2
// pixels as two 8-byte stores per channel (256 contiguous bytes per half wave).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace gance {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {

constexpr int kWBM = 32;             // output channels per block
constexpr int kWTH = 8, kWTW = 64;   // output pixels per block (4 x 32 tiles)
#ifndef GANCE_WINO_KC
#define GANCE_WINO_KC 8
#endif
#ifndef GANCE_WINO_NBUF
#define GANCE_WINO_NBUF 3
#endif
constexpr int kWKC = GANCE_WINO_KC;  // input channels per chunk
constexpr int kWPH = kWTH + 2, kWPW = kWTW + 8;
constexpr int kWNBUF = GANCE_WINO_NBUF;  // ring depth: NBUF-1 chunks in flight ahead of the one being multiplied
constexpr int kWWlFloats = 16 * kWKC * kWBM;              // 1 KiB DMA pieces
constexpr int kWPlFloats = kWKC * kWPH * kWPW;            // 2880
constexpr int kWPlF4 = kWPlFloats / 4;                    // 720
constexpr int kWPlInstr = (kWPlF4 + 63) / 64;             // 12 (last piece partial)
constexpr int kWBufFloats = kWWlFloats + kWPlInstr * 256;  // 5120 floats = 20 KiB
constexpr int kWPieces = kWWlFloats / 256 + kWPlInstr;     // 20
// the same count for every wave (counted vmcnt); a spare slot re-issues an early piece (same bytes)
constexpr int kWPiecesPerWave = (kWPieces + 3) / 4;

inline size_t winograd_lds_bytes(int cin) { return sizeof(float) * ((size_t)kWNBUF * kWBufFloats + cin + 2 * kWBM); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Operands of one k-step (2 input channels: lane half = k index of the MFMA) as they come out of LDS.
struct WinoRaw {
    float d[4][4];  // the tile's 4x4 input window
    float a[16];    // transformed-weight fragment of the 16 positions
    float s;        // style scale of the lane's channel
};
// ... and as the MFMAs take them
struct WinoOps {
    float a[16], v[16];
};

// LDS reads of k-step `kk` of a chunk. U, P and s are __restrict__: that scoped no-alias against the
// ring slot a later chunk's LDS-DMA is writing keeps hipcc from putting `s_waitcnt vmcnt(0)` in front
// of these reads (it would drain the whole ring every chunk).
__device__ __forceinline__ void wino_load(const float* __restrict__ U, const float* __restrict__ P,
                                          const float* __restrict__ s_chunk, int kk, int poff, int l31, int lh, WinoRaw& r) {
    const int cl = 2 * kk + lh;
    // one opaque offset per k-step: the four window rows are then immediate offsets (72 dwords apart) of
    // the LDS reads from ONE address register instead of eight separately added addresses
    typedef const __attribute__((address_space(3))) float* lds_cfloat;
    unsigned window = (unsigned)(uintptr_t)((lds_cfloat)P + cl * (kWPH * kWPW) + poff);  // 32-bit LDS byte address
    asm volatile("" : "+v"(window));
    lds_cfloat pc = (lds_cfloat)(uintptr_t)window;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        r.d[y][0] = pc[y * kWPW];
        r.d[y][1] = pc[y * kWPW + 1];
        r.d[y][2] = pc[y * kWPW + 2];
        r.d[y][3] = pc[y * kWPW + 3];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) r.a[q] = U[(q * kWKC + cl) * kWBM + l31];
    r.s = s_chunk[cl];
}

// V = B^T (s d) B, B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
__device__ __forceinline__ void wino_transform(const WinoRaw& r, WinoOps& o) {
    // the style scale rides on the activation side (V is linear in d): two multiplies and two fused
    // multiply-adds per window column instead of sixteen multiplies on the weight fragment
    float t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float e1 = r.s * r.d[1][c], e2 = r.s * r.d[2][c];
        t[0][c] = fmaf(r.s, r.d[0][c], -e2);
        t[1][c] = e1 + e2;
        t[2][c] = e2 - e1;
        t[3][c] = fmaf(-r.s, r.d[3][c], e1);
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        o.v[y * 4 + 0] = t[y][0] - t[y][2];
        o.v[y * 4 + 1] = t[y][1] + t[y][2];
        o.v[y * 4 + 2] = t[y][2] - t[y][1];
        o.v[y * 4 + 3] = t[y][1] - t[y][3];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) o.a[q] = r.a[q];
}

}  // namespace

__global__ __launch_bounds__(256, 1) void winograd_conv_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const buf0 = smem;
    float* const s_lds = smem + kWNBUF * kWBufFloats;  // [Cin]
    float* const d_lds = s_lds + p.Cin;                // [BM]
    float* const b_lds = d_lds + kWBM;                 // [BM]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;

    // virtual block id -> tile, XCD-aware (same scheme as the direct kernel)
    int id;
    {
        const int nwg = gridDim.x, v = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    const int m_tile = id % p.m_tiles;
    id /= p.m_tiles;
    const int tile_x = id % p.tiles_x;
    id /= p.tiles_x;
    const int tile_y = id % p.tiles_y;
    const int b0 = id / p.tiles_y;
    const int y0 = tile_y * kWTH, x0 = tile_x * kWTW;
    const int m0 = m_tile * kWBM;
    const int Hp = p.H + 2, Wp = p.W + 8;
    const int nchunks = p.total_chunks;

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + ((size_t)m_tile * p.total_chunks) * kWWlFloats), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)min(b0, p.B - 1) * p.x_b_stride), 0, 0x7fffffff, 0x00020000);

    // LDS-DMA of one chunk = kWPiecesPerWave `buffer_load_dwordx4 ... lds` per wave. A piece's per-lane
    // source offset does not depend on the chunk (only the scalar offset does), so it is computed once:
    // a chunk's staging is then 10 instructions per wave with no address arithmetic in the K loop.
    // Pieces of a wave: r < kWWlPerWave are weight pieces (wave + 4r), the rest patch pieces
    // (wave + 4(r - kWWlPerWave), one spare slot wraps round): the KIND of slot r is a compile-time fact,
    // so a chunk's staging is straight-line code that can be woven between MFMAs.
    static_assert((kWWlFloats / 256) % 4 == 0, "weight pieces split evenly over the four waves");
    constexpr int kWWlPerWave = kWWlFloats / 256 / 4;
    constexpr int kWPlPerWave = (kWPlInstr + 3) / 4;
    static_assert(kWWlPerWave + kWPlPerWave == kWPiecesPerWave, "piece slots per wave");
    int dma_voff[kWPiecesPerWave];
    int dma_lds[kWPiecesPerWave];  // float offset of the slot's 1 KiB piece inside a ring buffer (wave-uniform)
#pragma unroll
    for (int r = 0; r < kWPiecesPerWave; ++r) {
        if (r < kWWlPerWave) {
            const int g = wave + 4 * r;
            dma_voff[r] = (g * 256 + lane * 4) * 4;
            dma_lds[r] = g * 256;
        } else {
            int i = wave + 4 * (r - kWWlPerWave);
            if (i >= kWPlInstr) i -= kWPlInstr;            // spare slot: repeat an early piece (same bytes)
            const int f = min(i * 64 + lane, kWPlF4 - 1);  // tail lanes of the last piece repeat its last float4
            const int q = f % (kWPW / 4);
            int rr = f / (kWPW / 4);
            const int py = rr % kWPH;
            const int c = rr / kWPH;
            const int gy = min(y0 + py, Hp - 1);
            const int gx = min(x0 + 4 * q, Wp - 4);
            dma_voff[r] = ((c * Hp + gy) * Wp + gx) * 4;
            dma_lds[r] = kWWlFloats + i * 256;
        }
    }
    const int x_chunk_bytes = kWKC * Hp * Wp * 4;
    auto stage = [&](int chunk, float* buf) {
#pragma unroll
        for (int r = 0; r < kWPiecesPerWave; ++r) {
            if (r < kWWlPerWave)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(buf + dma_lds[r]), 16, dma_voff[r],
                                                         chunk * (kWWlFloats * 4), 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(buf + dma_lds[r]), 16, dma_voff[r],
                                                         chunk * x_chunk_bytes, 0, 0);
        }
    };

    // ring prologue: every slot in flight (every wave issues exactly kWPiecesPerWave pieces per chunk)
    for (int c = 0; c < kWNBUF; ++c) stage(min(c, nchunks - 1), buf0 + c * kWBufFloats);
    {   // per-tile constants: plain loads issued after the ring prologue; hipcc waits for them with
        // vmcnt(0), i.e. for the whole prologue, before the LDS writes below
        const int b = min(b0, p.B - 1);
        const float* sp = p.s + (size_t)b * p.s_stride;
        const float s0 = tid < p.Cin ? sp[tid] : 0.f;
        const float s1 = tid + 256 < p.Cin ? sp[tid + 256] : 0.f;
        const float dv = tid < kWBM ? p.d[(size_t)b * p.d_stride + m0 + tid] : 0.f;
        const float bv = tid < kWBM ? p.bias[m0 + tid] : 0.f;
        if (tid < p.Cin) s_lds[tid] = s0;
        if (tid + 256 < p.Cin) s_lds[tid + 256] = s1;
        if (tid < kWBM) {
            d_lds[tid] = dv;
            b_lds[tid] = bv;
        }
    }

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    // tile (wave, l31): input window rows 2*wave .. 2*wave+3, columns 2*l31+3 .. 2*l31+6 of the patch
    const int poff = (2 * wave) * kWPW + 2 * l31 + 3;
    constexpr int KS = kWKC / 2;  // k-steps per chunk
    static_assert(KS % 2 == 0 && KS >= 2, "the pipeline's register parity is per chunk");

    // Three-stage software pipeline over k-steps g = c*KS + j, so that ONE wave keeps the matrix pipe busy:
    //   L: LDS reads of k-step g+2   T: input transform + style scale of k-step g+1   M: the 16 MFMAs of k-step g
    // The three stages of an iteration are independent, and a sched_group_barrier pattern weaves them:
    // one MFMA, then a few VALU / LDS instructions that fit in its shadow. L runs two k-steps ahead of M,
    // so it is L that crosses into the next chunk first: there the chunk's landing is awaited (counted
    // vmcnt: the younger chunks stay in flight), the block synchronises, and the ring slot L has just
    // left is refilled by LDS-DMA.
    static_assert(kWNBUF >= 2 && (kWNBUF - 1) * kWPiecesPerWave <= 63, "ring depth vmcnt can express");
    WinoRaw raw[2];
    WinoOps ops[2];
    wait_vmcnt<(kWNBUF - 1) * kWPiecesPerWave>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    wino_load(buf0, buf0 + kWWlFloats, s_lds, 0, poff, l31, lh, raw[0]);
    wino_load(buf0, buf0 + kWWlFloats, s_lds, 1, poff, l31, lh, raw[1]);
    wino_transform(raw[0], ops[0]);

    auto mfma16 = [&](const WinoOps& o) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[q], o.v[q], acc[q], 0, 0, 0);
    };
    // weave: 16 x (1 MFMA, 4 VALU, 2 LDS reads [, 1 LDS-DMA issue])
    auto weave = [&](bool with_dma) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (with_dma) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };

    // every chunk but the last: the k-steps L and T reach for always exist
    for (int c = 0; c + 1 < nchunks; ++c) {
        const float* const Uc = buf0 + (c % kWNBUF) * kWBufFloats;
        const float* const Un = buf0 + ((c + 1) % kWNBUF) * kWBufFloats;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (j + 2 == KS && !(p.debug_flags & 8)) {
                // L moves on to chunk c+1: it must have landed (the NBUF-2 younger chunks stay in flight)
                // and every wave must be here before chunk c's slot may be refilled
                wait_vmcnt<(kWNBUF - 2) * kWPiecesPerWave>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            // one k-step later every wave's last LDS reads of chunk c (issued before the barrier above)
            // have long returned: refill its slot, the DMA issues woven between this k-step's MFMAs.
            // Past the end the last chunk is fetched again (same count of pieces in flight, no tail case).
            if (j + 1 == KS && !(p.debug_flags & 10)) stage(min(c + kWNBUF, nchunks - 1), buf0 + (c % kWNBUF) * kWBufFloats);
            if (j + 2 < KS)
                wino_load(Uc, Uc + kWWlFloats, s_lds + c * kWKC, j + 2, poff, l31, lh, raw[j & 1]);
            else
                wino_load(Un, Un + kWWlFloats, s_lds + (c + 1) * kWKC, j + 2 - KS, poff, l31, lh, raw[j & 1]);
            wino_transform(raw[(j + 1) & 1], ops[(j + 1) & 1]);
            mfma16(ops[j & 1]);
            weave(j + 1 == KS);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vmcnt<0>();  // nothing of the ring may still be landing when the block's LDS is given back
    {   // the last chunk: the pipeline drains
        const int c = nchunks - 1;
        const float* const Uc = buf0 + (c % kWNBUF) * kWBufFloats;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < KS) wino_load(Uc, Uc + kWWlFloats, s_lds + c * kWKC, j + 2, poff, l31, lh, raw[j & 1]);
            if (j + 1 < KS) wino_transform(raw[(j + 1) & 1], ops[(j + 1) & 1]);
            mfma16(ops[j & 1]);
            weave(false);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), epilogue, stores ----
    const int oy = y0 + 2 * wave, ox = x0 + 2 * l31;
    const bool in_batch = b0 < p.B;
    float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (p.noise != nullptr && in_batch) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                if (oy + dy < p.OH && ox + dx < p.OW) nz[dy][dx] = p.noise[(size_t)(oy + dy) * p.OW + ox + dx] * p.noise_strength;
    }
    const bool full = p.epilogue == kEpilogueFull;
    const int c_stride_bytes = (int)p.out_c_stride * 4;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + (size_t)b0 * p.out_b_stride + (size_t)m0 * p.out_c_stride), 0, 0x7fffffff, 0x00020000);
    const int voff0 = ((oy + p.out_y_off) * p.out_row_stride + ox + p.out_x_off) * 4 + 4 * lh * c_stride_bytes;
    const bool ok = in_batch && oy < p.OH && ox < p.OW;  // tiles are whole: H and W are multiples of the tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float u[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u[0][j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
            u[1][j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
        }
        const float dm = d_lds[m], bm = b_lds[m];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float y2[2];
            y2[0] = u[dy][0] + u[dy][1] + u[dy][2];
            y2[1] = u[dy][1] - u[dy][2] - u[dy][3];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v = y2[dx] * dm;
                if (full) {
                    v += nz[dy][dx] + bm;
                    v = fmaxf(v, 0.2f * v) * 1.4142135623730951f;
                }
                y2[dx] = v;
            }
            if (ok) {
                const int soff = ((r & 3) + 8 * (r >> 2)) * c_stride_bytes;
                u32x2 pair;
                pair[0] = __float_as_uint(y2[0]);
                pair[1] = __float_as_uint(y2[1]);
                __builtin_amdgcn_raw_buffer_store_b64(pair, o_rsrc, voff0 + dy * p.out_row_stride * 4, soff, 0);
            }
        }
    }
}

size_t winograd_weight_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

// w_in: the layer's runtime-scaled weights [tap = ky*3+kx][ci][co]; w_out: [m_tile][chunk][pos][KC][BM]
void winograd_transform_weights(const float* w_in, int cin, int cout, float* w_out) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int chunks = cin / kWKC;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double g[3][3], tmp[4][3], u[4][4];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w_in[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) tmp[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) u[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int m_tile = co / kWBM, m = co % kWBM, chunk = ci / kWKC, cl = ci % kWKC;
            float* dst = w_out + ((size_t)m_tile * chunks + chunk) * kWWlFloats;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) dst[((i * 4 + j) * kWKC + cl) * kWBM + m] = (float)u[i][j];
        }
}

bool winograd_supported(int cin, int cout, int H, int W) {
    return cin % kWKC == 0 && cin <= 512 && cout % kWBM == 0 && H % kWTH == 0 && W % kWTW == 0 && cin / kWKC >= kWNBUF;
}

hipError_t launch_winograd_conv(const ConvArgs& args, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_conv_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)winograd_lds_bytes(512));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    ConvArgs a = args;
    a.tiles_x = a.W / kWTW;
    a.tiles_y = a.H / kWTH;
    a.m_tiles = a.Cout / kWBM;
    a.total_chunks = a.Cin / kWKC;
    const int blocks = a.m_tiles * a.tiles_x * a.tiles_y * a.B;
    hipLaunchKernelGGL(winograd_conv_kernel, dim3(blocks), dim3(256), winograd_lds_bytes(a.Cin), stream, a);
    return hipGetLastError();
}

}  // namespace gance
