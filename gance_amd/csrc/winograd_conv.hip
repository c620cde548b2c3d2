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
// output pixels per block: 8 x 64 (4 x 32 tiles, wave = tile row) or, for the 32-pixel-wide layer,
// 16 x 32 (8 x 16 tiles, wave = two tile rows); both have a 720-float patch per channel (10x72 / 18x40)
constexpr int kWTH = 8, kWTW = 64;
constexpr int kWTHn = 16, kWTWn = 32;
#ifndef GANCE_WINO_KC
#define GANCE_WINO_KC 8
#endif
#ifndef GANCE_WINO_NBUF
#define GANCE_WINO_NBUF 3
#endif
constexpr int kWKC = GANCE_WINO_KC;  // input channels per chunk
constexpr int kWPH = kWTH + 2, kWPW = kWTW + 8;
constexpr int kWPHn = kWTHn + 2, kWPWn = kWTWn + 8;
static_assert(kWPH * kWPW == kWPHn * kWPWn, "both geometries share the ring-slot layout");
constexpr int kWNBUF = GANCE_WINO_NBUF;  // ring depth: NBUF-1 chunks in flight ahead of the one being multiplied
constexpr int kWWlFloats = 16 * kWKC * kWBM;              // 1 KiB DMA pieces
constexpr int kWPlFloats = kWKC * kWPH * kWPW;            // 2880
constexpr int kWPlF4 = kWPlFloats / 4;                    // 720
constexpr int kWPlInstr = (kWPlF4 + 63) / 64;             // 12 (last piece partial)
constexpr int kWBufFloats = kWWlFloats + kWPlInstr * 256;  // 5120 floats = 20 KiB
constexpr int kWPieces = kWWlFloats / 256 + kWPlInstr;     // 20
// the same count for every wave (counted vmcnt); a spare slot re-issues an early piece (same bytes)
constexpr int kWPiecesPerWave = (kWPieces + 3) / 4;

// per-tile constants, in 64-dword DMA pieces: style [<= 512] | demod [32 of 64] | bias [32 of 64] | and for the
// fused last layer: ToRGB style [32 of 64] | the tile's window of the half-resolution skip image [3][6][34] in 10 pieces
constexpr int kWSkipRows = kWTH / 2 + 2, kWSkipCols = kWTW / 2 + 2;
constexpr int kWSkipPieces = (3 * kWSkipRows * kWSkipCols + 63) / 64;
constexpr int kWConstFloats = 512 + 64 + 64 + 64 + kWSkipPieces * 64;
inline size_t winograd_lds_bytes() {
    return sizeof(float) * ((size_t)kWNBUF * kWBufFloats + 2 * kWConstFloats + (size_t)kWPiecesPerWave * 256 + 128);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Operands of one k-step (2 input channels: lane half = k index of the MFMA) as they come out of LDS.
struct WinoRaw {
    float d[4][4];  // the tile's 4x4 input window
    float s;        // style scale of the lane's channel
};
// ... and as the MFMAs take them
struct WinoOps {
    float a[16], v[16];
};

// LDS reads of k-step `kk` of a chunk. U, P and s are __restrict__: that scoped no-alias against the
// ring slot a later chunk's LDS-DMA is writing keeps hipcc from putting `s_waitcnt vmcnt(0)` in front
// of these reads (it would drain the whole ring every chunk).
__device__ __forceinline__ void wino_load_a(const float* __restrict__ U, int kk, int l31, int lh, WinoOps& o) {
    const int cl = 2 * kk + lh;
#pragma unroll
    for (int q = 0; q < 16; ++q) o.a[q] = U[(q * kWKC + cl) * kWBM + l31];
}
template <int PW>
__device__ __forceinline__ void wino_load(const float* __restrict__ P, const float* __restrict__ s_chunk, int kk, int poff,
                                          int lh, WinoRaw& r) {
    const int cl = 2 * kk + lh;
    // one opaque offset per k-step: the four window rows are then immediate offsets (72 dwords apart) of
    // the LDS reads from ONE address register instead of eight separately added addresses
    typedef const __attribute__((address_space(3))) float* lds_cfloat;
    unsigned window = (unsigned)(uintptr_t)((lds_cfloat)P + cl * (kWPH * kWPW) + poff);  // 32-bit LDS byte address
    asm volatile("" : "+v"(window));
    lds_cfloat pc = (lds_cfloat)(uintptr_t)window;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        r.d[y][0] = pc[y * PW];
        r.d[y][1] = pc[y * PW + 1];
        r.d[y][2] = pc[y * PW + 2];
        r.d[y][3] = pc[y * PW + 3];
    }
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
}

}  // namespace

// RGB = the network's last layer: the epilogue feeds the activation straight to its ToRGB, adds the
// upsampled skip image and stores uint8 (kEpilogueRgb, as in conv_mfma.hip), no activation is written.
template <bool RGB, bool WIDE>
__global__ __launch_bounds__(256, 1) void winograd_conv_kernel(const ConvArgs p) {
    constexpr int TH = WIDE ? kWTH : kWTHn, TW = WIDE ? kWTW : kWTWn;
    constexpr int PH = TH + 2, PW = TW + 8;
    static_assert(!RGB || WIDE, "the fused last layer uses the 8 x 64 geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const buf0 = smem;
    // two sets of per-tile constants (tile parity), in 64-dword DMA pieces: style [512] | demod [64] | bias [64]
    float* const const0 = smem + kWNBUF * kWBufFloats;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int l31_k = l31, lh_k = lh;
    const int Hp = p.H + 2, Wp = p.W + 8;
    const int n = p.total_chunks;  // chunks per tile

    // A block is persistent: it walks tiles blockIdx.x, + gridDim.x, ... and treats their chunks as ONE
    // stream, so the ring keeps prefetching across tile boundaries (with one block per CU nothing else
    // would hide a tile's first fetch).
    const int my_tiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_q = my_tiles * n;
    struct Tile {
        int m_tile, y0, x0, b0;
    };
    auto decode = [&](int i) {
        // virtual block id -> tile, XCD-aware (a persistent block strides by a multiple of 8)
        const int v = (int)blockIdx.x + min(i, my_tiles - 1) * (int)gridDim.x;
        const int nwg = p.total_tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        Tile t;
        t.m_tile = id % p.m_tiles;
        id /= p.m_tiles;
        t.x0 = (id % p.tiles_x) * TW;
        id /= p.tiles_x;
        t.y0 = (id % p.tiles_y) * TH;
        t.b0 = id / p.tiles_y;
        return t;
    };

    // LDS-DMA of one chunk = kWPiecesPerWave `buffer_load_dwordx4 ... lds` per wave. Tiles are whole
    // (H % 8 == 0, W % 64 == 0), so a piece's per-lane source offset depends neither on the chunk nor
    // on the tile: only scalar offsets do. Slot r < kWWlPerWave of a wave is a weight piece (wave + 4r),
    // the rest are patch pieces (one spare slot repeats an early piece): the KIND of a slot is a
    // compile-time fact, so a chunk's staging is straight-line code that can be woven between MFMAs.
    static_assert((kWWlFloats / 256) % 4 == 0, "weight pieces split evenly over the four waves");
    constexpr int kWWlPerWave = kWWlFloats / 256 / 4;
    constexpr int kWPlPerWave = (kWPlInstr + 3) / 4;
    static_assert(kWWlPerWave + kWPlPerWave == kWPiecesPerWave, "piece slots per wave");
    // The per-lane offsets live in LDS, not in registers: they are needed once per chunk, and a register
    // hipcc decided to spill would come back by a scratch load, i.e. a vector-memory operation whose wait
    // drains the whole ring.
    int* const voff_lds = reinterpret_cast<int*>(const0 + 2 * kWConstFloats);  // [kWPiecesPerWave][256]
    float* const rgbw_lds = reinterpret_cast<float*>(voff_lds + kWPiecesPerWave * 256);  // ToRGB weight [32][3] (+ pad)
    if (RGB && tid < 96) rgbw_lds[tid] = p.rgb_w[tid];
    int dma_lds[kWPiecesPerWave];  // float offset of the slot's 1 KiB piece inside a ring buffer (wave-uniform)
#pragma unroll
    for (int r = 0; r < kWPiecesPerWave; ++r) {
        if (r < kWWlPerWave) {
            const int g = wave + 4 * r;
            voff_lds[r * 256 + tid] = (g * 256 + lane * 4) * 4;
            dma_lds[r] = g * 256;
        } else {
            int i = wave + 4 * (r - kWWlPerWave);
            if (i >= kWPlInstr) i -= kWPlInstr;
            const int f = min(i * 64 + lane, kWPlF4 - 1);  // tail lanes of the last piece repeat its last float4
            const int q = f % (PW / 4);
            int rr = f / (PW / 4);
            const int py = rr % PH;
            const int c = rr / PH;
            voff_lds[r * 256 + tid] = ((c * Hp + py) * Wp + 4 * q) * 4;
            dma_lds[r] = kWWlFloats + i * 256;
        }
    }
    const int x_chunk_bytes = kWKC * Hp * Wp * 4;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);

    // ---- staging side of the stream (scalar state): which tile / chunk / ring slot comes next ----
    int st_tile = 0, st_chunk = 0, st_slot = 0;
    int st_w_soff = 0, st_x_soff = 0;
    __amdgpu_buffer_rsrc_t st_x_rsrc = w_rsrc;
    // first chunk of a tile: its source descriptors, and its constants by dword LDS-DMA (3 per wave, every
    // wave, so the counted waits stay uniform; lanes past an array's end read zeros into padding)
    auto stage_setup = [&](int i) {
        const Tile t = decode(i);
        const int b = min(t.b0, p.B - 1);
        st_x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.x_b_stride), 0, 0x7fffffff, 0x00020000);
        st_w_soff = t.m_tile * n * (kWWlFloats * 4);
        st_x_soff = (t.y0 * Wp + t.x0) * 4;
        float* const set = const0 + (i & 1) * kWConstFloats;
        int ln = lane;
        asm volatile("" : "+v"(ln));  // recompute the three offsets here instead of keeping them live
        const __amdgpu_buffer_rsrc_t s_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.s + (size_t)b * p.s_stride), 0, p.Cin * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_ptr_t)(set + wave * 64), 4, (wave * 64 + ln) * 4, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_ptr_t)(set + (wave + 4) * 64), 4, ((wave + 4) * 64 + ln) * 4, 0, 0, 0);
        const bool demod_piece = wave < 2;
        const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(demod_piece ? p.d + (size_t)b * p.d_stride + t.m_tile * kWBM : p.bias + t.m_tile * kWBM), 0, kWBM * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lds_ptr_t)(set + 512 + (demod_piece ? 0 : 64)), 4, ln * 4, 0, 0, 0);
        if (RGB) {
            // ToRGB style of the sample (every wave fetches the same piece: uniform counts)
            const __amdgpu_buffer_rsrc_t rs_rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.rgb_s + (size_t)b * p.s_stride), 0, kWBM * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rsrc, (lds_ptr_t)(set + 640), 4, ln * 4, 0, 0, 0);
            // the tile's window of the previous (half-resolution) skip image, 3 pieces per wave (two spare
            // slots repeat pieces 0 and 1); addresses are clamped into the image, validity is decided at use
            const int Rh = p.OW >> 1;
            const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(p.rgb_y_prev != nullptr ? p.rgb_y_prev + (size_t)b * 3 * Rh * Rh : p.x), 0, 0x7fffffff, 0x00020000);
            const int row0 = (t.y0 >> 1) - 1, col0 = (t.x0 >> 1) - 1;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int piece = wave + 4 * r;
                if (piece >= kWSkipPieces) piece -= kWSkipPieces;
                const int idx = min(piece * 64 + ln, 3 * kWSkipRows * kWSkipCols - 1);
                const int cc = idx % kWSkipCols;
                const int rr = (idx / kWSkipCols) % kWSkipRows;
                const int k = idx / (kWSkipCols * kWSkipRows);
                const int row = min(max(row0 + rr, 0), Rh - 1), col = min(max(col0 + cc, 0), Rh - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (lds_ptr_t)(set + 704 + piece * 64), 4, ((k * Rh + row) * Rh + col) * 4,
                                                         0, 0, 0);
            }
        }
    };
    // the next chunk of the stream into the next ring slot; past the end the last chunk is fetched again
    // (same number of pieces in flight, no tail case in the waits)
    auto stage_next = [&]() {
        float* const buf = buf0 + st_slot * kWBufFloats;
        const int w_soff = st_w_soff + st_chunk * (kWWlFloats * 4);
        const int x_soff = st_x_soff + st_chunk * x_chunk_bytes;
#pragma unroll
        for (int r = 0; r < kWPiecesPerWave; ++r) {
            const int voff = voff_lds[r * 256 + tid];
            if (r < kWWlPerWave)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(buf + dma_lds[r]), 16, voff, w_soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(st_x_rsrc, (lds_ptr_t)(buf + dma_lds[r]), 16, voff, x_soff, 0, 0);
        }
        st_slot = st_slot + 1 == kWNBUF ? 0 : st_slot + 1;
        const bool more = st_tile * n + st_chunk + 1 < total_q;  // else stay on the last chunk
        const bool wrap = more && st_chunk + 1 == n;
        st_chunk = more ? (wrap ? 0 : st_chunk + 1) : st_chunk;
        st_tile = wrap ? st_tile + 1 : st_tile;
    };
    // does the next stage_next() open a tile whose setup has not been done?
    int st_setup_done = 0;  // tiles set up so far
    auto stage_setup_if_due = [&]() {
        if (st_chunk == 0 && st_tile == st_setup_done && st_tile < my_tiles) {
            stage_setup(st_tile);
            st_setup_done = st_tile + 1;
        }
    };

    // ring prologue: every slot in flight (a tile has at least kWNBUF chunks: only tile 0 is touched)
    stage_setup_if_due();
    for (int c = 0; c < kWNBUF; ++c) stage_next();

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    // a lane's tile (tile row tr, tile column tc): input window rows 2 tr .. 2 tr + 3, columns 2 tc + 3 .. 2 tc + 6
    const int tr = WIDE ? wave : 2 * wave + (l31 >> 4), tc = WIDE ? l31 : (l31 & 15);
    const int poff = (2 * tr) * PW + 2 * tc + 3;
    constexpr int KS = kWKC / 2;  // k-steps per chunk
    static_assert(KS % 2 == 0 && KS >= 2, "the pipeline's register parity is per chunk");
    static_assert(kWNBUF >= 2 && (kWNBUF - 1) * kWPiecesPerWave + 7 <= 63, "ring depth vmcnt can express");

    // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), epilogue, stores; clears acc ----
    auto epilogue = [&](int i) {
        const Tile t = decode(i);
        const float* const d_lds = const0 + (i & 1) * kWConstFloats + 512;
        const float* const b_lds = d_lds + 64;
        // lane ids re-derived here from opaque copies: the epilogue's per-lane address arithmetic is then
        // redone per tile (a dozen instructions) instead of being kept live through the K loop, where
        // hipcc would spill it to scratch (a scratch reload is a vector-memory operation: it drains the ring)
        int l31 = l31_k, lh = lh_k;
        asm volatile("" : "+v"(l31), "+v"(lh));
        const int oy = t.y0 + 2 * (WIDE ? wave : 2 * wave + (l31 >> 4)), ox = t.x0 + 2 * (WIDE ? l31 : (l31 & 15));
        const bool ok = t.b0 < p.B && oy < p.OH && ox < p.OW;
        float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (p.noise != nullptr && ok) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) nz[dy][dx] = p.noise[(size_t)t.b0 * p.noise_b_stride + (size_t)(oy + dy) * p.OW + ox + dx] * p.noise_strength;
        }
        if constexpr (RGB) {
            // ---- last layer: activation -> ToRGB (32 channels x 3, style * weight) -> + bias + upsampled skip
            // image -> uint8. A lane holds its tile's 2x2 pixels for the 16 channels of its lane half.
            const float* const set = const0 + (i & 1) * kWConstFloats;
            const float* const srgb = set + 640;
            const float* const skip = set + 704;
            const int R = p.OW, Rh = R >> 1;
            float rgb[2][2][3] = {};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                __builtin_amdgcn_sched_barrier(0);
                const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float u[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[0][j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
                    u[1][j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
                }
                const float dm = d_lds[m], bm = b_lds[m], sm = srgb[m];
                const float c0 = sm * rgbw_lds[m * 3 + 0], c1 = sm * rgbw_lds[m * 3 + 1], c2 = sm * rgbw_lds[m * 3 + 2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    float y2[2];
                    y2[0] = u[dy][0] + u[dy][1] + u[dy][2];
                    y2[1] = u[dy][1] - u[dy][2] - u[dy][3];
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        float v = y2[dx] * dm;
                        v += nz[dy][dx] + bm;
                        v = fmaxf(v, 0.2f * v) * 1.4142135623730951f;
                        rgb[dy][dx][0] = fmaf(v, c0, rgb[dy][dx][0]);
                        rgb[dy][dx][1] = fmaf(v, c1, rgb[dy][dx][1]);
                        rgb[dy][dx][2] = fmaf(v, c2, rgb[dy][dx][2]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                unsigned short q16[3];  // the row's two pixels = 6 bytes = three 16-bit stores
                unsigned bytes[6];
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    // upsample_2d of the skip image ([1,3,3,1]/4 per axis = two taps per axis): lane half 0
                    // takes the upper source row, half 1 the lower one; the halves join in the exchange below
                    const int py = oy + dy, px = ox + dx;
                    const int ya = (py & 1) ? (py >> 1) : (py >> 1) - 1;
                    const int row = ya + lh;
                    const float wrow = ((py & 1) != 0) == (lh == 0) ? 0.75f : 0.25f;
                    const int xa = (px & 1) ? (px >> 1) : (px >> 1) - 1, xb = xa + 1;
                    const float wxa = (px & 1) ? 0.75f : 0.25f, wxb = 1.0f - wxa;
                    const bool row_ok = p.rgb_y_prev != nullptr && row >= 0 && row < Rh;
                    const float* w0 = skip + (row - ((t.y0 >> 1) - 1)) * kWSkipCols - ((t.x0 >> 1) - 1);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float va = (row_ok && xa >= 0) ? w0[k * kWSkipRows * kWSkipCols + xa] : 0.f;
                        const float vb = (row_ok && xb < Rh) ? w0[k * kWSkipRows * kWSkipCols + xb] : 0.f;
                        rgb[dy][dx][k] += wrow * (wxa * va + wxb * vb);
                    }
                }
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float y = rgb[dy][dx][k] + __shfl_xor(rgb[dy][dx][k], 32) + p.rgb_bias[k];
                        if (p.rgb_y != nullptr && lh == 0 && ok)
                            p.rgb_y[((size_t)t.b0 * 3 + k) * R * R + (size_t)(oy + dy) * R + ox + dx] = y;
                        // tf.saturate_cast(x * 127.5 + 128): two roundings (the barrier keeps them apart)
                        float q = y * 127.5f;
                        asm volatile("" : "+v"(q));
                        q += 128.0f;
                        q = fminf(fmaxf(q, 0.f), 255.f);
                        bytes[dx * 3 + k] = (unsigned)(int)q;
                    }
#pragma unroll
                for (int h = 0; h < 3; ++h) q16[h] = (unsigned short)(bytes[2 * h] | (bytes[2 * h + 1] << 8));
                if (p.rgb_u8 != nullptr && lh == 0 && ok) {
                    unsigned short* dst = reinterpret_cast<unsigned short*>(p.rgb_u8 + (((size_t)t.b0 * R + oy + dy) * R + ox) * 3);
                    dst[0] = q16[0];
                    dst[1] = q16[1];
                    dst[2] = q16[2];
                }
            }
        } else {
        const bool full = p.epilogue == kEpilogueFull;
        const int c_stride_bytes = (int)p.out_c_stride * 4;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.out + (size_t)t.b0 * p.out_b_stride + (size_t)(t.m_tile * kWBM) * p.out_c_stride), 0, 0x7fffffff, 0x00020000);
        const int voff0 = ((oy + p.out_y_off) * p.out_row_stride + ox + p.out_x_off) * 4 + 4 * lh * c_stride_bytes;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // one accumulator row at a time: the pipeline's operand registers stay live across the
            // epilogue, and without this fence hipcc pulls all 256 accumulators into VGPRs at once (spills)
            __builtin_amdgcn_sched_barrier(0);
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
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    };

    // Three-stage software pipeline over the k-steps of the stream, so that ONE wave keeps the matrix pipe busy:
    //   L: LDS reads (input window of k-step g+2, weight fragments of k-step g+1)   T: input transform (with the
    //   style scale) of k-step g+1   M: the 16 MFMAs of k-step g
    // The three stages of an iteration are independent, and a sched_group_barrier pattern weaves them:
    // one MFMA, then a few VALU / LDS / DMA-issue instructions that fit in its shadow. L runs two k-steps
    // ahead of M, so it is L that crosses into the next chunk first: there the chunk's landing is awaited
    // (counted vmcnt: the younger chunks stay in flight) and the block synchronises; one k-step later the
    // ring slot L has left is refilled.
    WinoRaw raw[2];
    WinoOps ops[2];
    wait_vmcnt<(kWNBUF - 1) * kWPiecesPerWave>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    wino_load<PW>(buf0 + kWWlFloats, const0, 0, poff, lh, raw[0]);
    wino_load<PW>(buf0 + kWWlFloats, const0, 1, poff, lh, raw[1]);
    wino_load_a(buf0, 0, l31, lh, ops[0]);
    wino_transform(raw[0], ops[0]);

    auto mfma16 = [&](const WinoOps& o) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[q], o.v[q], acc[q], 0, 0, 0);
    };
    // weave: 16 x (1 MFMA, 4 VALU, 2 LDS reads [, 1 LDS-DMA issue])
    auto weave = [&](bool with_dma) {
#ifndef GANCE_WINO_WEAVE
#define GANCE_WINO_WEAVE 1
#endif
#pragma unroll
        for (int q = 0; q < 16 / GANCE_WINO_WEAVE; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, GANCE_WINO_WEAVE, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * GANCE_WINO_WEAVE, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * GANCE_WINO_WEAVE, 0);
            if (with_dma) __builtin_amdgcn_sched_group_barrier(0x020, GANCE_WINO_WEAVE, 0);
        }
    };

    int tile = 0, slot = 0;  // M's position in the stream
    // `count` chunks of tile `tile` starting at chunk c0, none of them the last chunk of the stream:
    // the k-steps L and T reach for always exist (in this tile, or the first of the next one)
    auto run_chunks = [&](int c0, int count) {
        for (int c = c0; c < c0 + count; ++c) {
            const int next_slot = slot + 1 == kWNBUF ? 0 : slot + 1;
            const float* const Uc = buf0 + slot * kWBufFloats;
            const float* const Un = buf0 + next_slot * kWBufFloats;
            const float* const s_c = const0 + (tile & 1) * kWConstFloats + c * kWKC;
            const float* const s_n = c + 1 < n ? s_c + kWKC : const0 + ((tile + 1) & 1) * kWConstFloats;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j + 2 == KS && !(p.debug_flags & 8)) {
                    // L moves on to the next chunk: it must have landed (the NBUF-2 younger chunks stay
                    // in flight) and every wave must be here before this chunk's slot may be refilled
                    wait_vmcnt<(kWNBUF - 2) * kWPiecesPerWave>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    stage_setup_if_due();
                }
                __builtin_amdgcn_sched_barrier(0);
                // one k-step later every wave's last LDS reads of this chunk (issued before the barrier
                // above) have long returned: refill its slot, the DMA issues woven between the MFMAs
                if (j + 1 == KS && !(p.debug_flags & 10)) stage_next();
                if (j + 2 < KS)
                    wino_load<PW>(Uc + kWWlFloats, s_c, j + 2, poff, lh, raw[j & 1]);
                else
                    wino_load<PW>(Un + kWWlFloats, s_n, j + 2 - KS, poff, lh, raw[j & 1]);
                if (j + 1 < KS)
                    wino_load_a(Uc, j + 1, l31, lh, ops[(j + 1) & 1]);
                else
                    wino_load_a(Un, 0, l31, lh, ops[(j + 1) & 1]);
                wino_transform(raw[(j + 1) & 1], ops[(j + 1) & 1]);
                mfma16(ops[j & 1]);
                weave(j + 1 == KS);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = next_slot;
        }
    };
    // every tile but the block's last: all its chunks, then its epilogue (L and T already hold the next
    // tile's first k-steps in their registers)
    for (; tile + 1 < my_tiles; ++tile) {
        run_chunks(0, n);
        epilogue(tile);
    }
    run_chunks(0, n - 1);
    const int c = n - 1;
    wait_vmcnt<0>();  // nothing of the ring may still be landing when the block's LDS is given back
    {   // the last chunk of the stream: the pipeline drains
        const float* const Uc = buf0 + slot * kWBufFloats;
        const float* const s_c = const0 + (tile & 1) * kWConstFloats + c * kWKC;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < KS) wino_load<PW>(Uc + kWWlFloats, s_c, j + 2, poff, lh, raw[j & 1]);
            if (j + 1 < KS) {
                wino_load_a(Uc, j + 1, l31, lh, ops[(j + 1) & 1]);
                wino_transform(raw[(j + 1) & 1], ops[(j + 1) & 1]);
            }
            mfma16(ops[j & 1]);
            weave(false);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    epilogue(tile);
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
    const bool wide = H % kWTH == 0 && W % kWTW == 0, narrow = H % kWTHn == 0 && W % kWTWn == 0;
    return cin % kWKC == 0 && cin <= 512 && cout % kWBM == 0 && (wide || narrow) && cin / kWKC >= kWNBUF;
}

template <bool RGB, bool WIDE>
static hipError_t launch_winograd(const ConvArgs& args, hipStream_t stream) {
    auto kernel = winograd_conv_kernel<RGB, WIDE>;
    // per device: the dynamic-LDS opt-in and the launch size = one block per CU (512 registers per wave), a multiple of 8 (XCDs)
    static PerDeviceInt resident;
    int resident_blocks = 0;
    hipError_t e = resident.get(
        [&](int device, int* value) {
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)winograd_lds_bytes());
            if (err != hipSuccess) return err;
            int cus = 0;
            if ((err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return err;
            *value = std::max(8, cus / 8 * 8);
            return hipSuccess;
        },
        &resident_blocks);
    if (e != hipSuccess) return e;
    ConvArgs a = args;
    a.tiles_x = a.W / (WIDE ? kWTW : kWTWn);
    a.tiles_y = a.H / (WIDE ? kWTH : kWTHn);
    a.m_tiles = a.Cout / kWBM;
    a.total_chunks = a.Cin / kWKC;
    a.total_tiles = a.m_tiles * a.tiles_x * a.tiles_y * a.B;
    const int blocks = std::min(a.total_tiles, resident_blocks);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), winograd_lds_bytes(), stream, a);
    return hipGetLastError();
}

hipError_t launch_winograd_conv(const ConvArgs& args, hipStream_t stream) {
    // kEpilogueRgb: the fused last layer (Cout = 32 = one block's channels)
    const bool wide = args.H % kWTH == 0 && args.W % kWTW == 0;
    if (args.epilogue == kEpilogueRgb) return args.Cout == kWBM && wide ? launch_winograd<true, true>(args, stream) : hipErrorInvalidValue;
    return wide ? launch_winograd<false, true>(args, stream) : launch_winograd<false, false>(args, stream);
}

}  // namespace gance
