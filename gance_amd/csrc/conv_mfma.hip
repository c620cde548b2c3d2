// Modulated 3x3 convolution of StyleGAN2's synthesis network as an implicit GEMM on the gfx950
// fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// Replaces, for the reference's synthesis call (gance/network_interface/network_functions.py:168),
// the un-vendored `modulated_conv2d_layer` -> tf.nn.conv2d / conv2d_transpose (cuDNN) chain
// (SURVEY.md §8 a18).
//
// Formulation (not the reference's): weights are shared by the whole batch,
//     out[b,co,p] = d[b,co] * sum_{tap,ci} ( w[tap,ci,co] * s[b,ci] ) * x[b,ci,p+tap]
// i.e. GEMM  D[M = co][N = pixel] = A[M][K] * B[K][N],  K = taps x Cin. The style scale s is
// applied to the operand fragment in registers and the demodulation d in the epilogue. M = output
// channel is on the accumulator ROWS so that one accumulator register of a wave is 32 consecutive
// pixels of one channel plane: every global store is a full 128-B segment.
//
// UP = true is the stride-2 transposed convolution of the Conv0_up layers: one block computes, for
// a tile of INPUT-grid positions (y', x'), all four output-parity classes T[2y'+py][2x'+px]
// (even/even 4 taps, even/odd and odd/even 2 taps, odd/odd 1 tap = the 9 filter taps, each used
// exactly once), so no multiply touches an inserted zero and the input patch is staged once.
//
// Data layout in HBM: activations are zero-bordered  [B][C][H+2][W+8]  (interior at [y+1][x+4]),
// so a tile's haloed patch is a set of 16-B aligned row segments that need no bounds logic and
// can be copied by LDS-DMA, AND every 32-pixel output segment is one aligned 128-B line. Weights are pre-arranged per (m tile, K chunk) as the exact LDS image
// [tap][KC][BM], so their staging is a linear copy.
//
// Pipeline: two LDS buffers (a third, with counted `s_waitcnt vmcnt(N)`, is a template parameter that
// measured slower: it costs a resident block per CU); the `buffer_load_dwordx4 ... lds` of chunk k+1
// is in flight while the MFMAs of chunk k run; per chunk one `s_waitcnt vmcnt` + one raw s_barrier.
// All LDS is one dynamic array; no ordinary global load sits inside the K loop.
//
// The stride-1 layers at >= 32x32 normally run in Winograd form instead (winograd_conv.hip); this
// kernel keeps the small layers, every transposed conv, and the last layer fused with its ToRGB.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

// Timing ablations (GANCE_DEBUG_CONV: 1 no stores, 2 no DMA after the first chunk, 4 no MFMA,
// 16 per-block phase stamps, 64 no XCD remap) are compiled in only with -DGANCE_CONV_DEBUG=1:
// their uniform branches cost scalar registers and one branch per store in the product kernel.
#ifndef GANCE_CONV_DEBUG
#define GANCE_CONV_DEBUG 0
#endif
#define GANCE_DBG(flag) (GANCE_CONV_DEBUG && (p.debug_flags & (flag)))
#ifndef GANCE_CONV_PERSIST
#define GANCE_CONV_PERSIST 0
#endif
// resident blocks per CU the transposed-conv tiles with BM <= 64 are compiled for (register cap 168 at 3)
// wave priority (0..3) raised for the epilogue, experiment
#ifndef GANCE_TUNE_EPILOGUE_PRIO
#define GANCE_TUNE_EPILOGUE_PRIO 0
#endif
#ifndef GANCE_UP_BLOCKS
#define GANCE_UP_BLOCKS 3
#endif

namespace gance {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---- static tap tables -------------------------------------------------------------------
// regular conv: tap t = ky*3+kx reads input (y+ky-1, x+kx-1), one class.
// transposed conv: 9 (class, dy, dx) entries in the order the weights are stored (engine.hip
// `kUpTapWeight`): EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO (0,0).
template <bool UP>
__host__ __device__ constexpr int tap_cls(int t) {
    if (!UP) return 0;
    return t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3));
}
template <bool UP>
__host__ __device__ constexpr int tap_dy(int t) {
    if (!UP) return t / 3 - 1;
    return (t == 2 || t == 3 || t == 5) ? -1 : 0;
}
template <bool UP>
__host__ __device__ constexpr int tap_dx(int t) {
    if (!UP) return t % 3 - 1;
    return (t == 1 || t == 3 || t == 7) ? -1 : 0;
}
// index of the distinct (dy,dx) shift a tap reads (B fragments are shared between taps)
template <bool UP>
__host__ __device__ constexpr int tap_shift(int t) {
    if (!UP) return t;
    return (tap_dy<true>(t) == -1 ? 2 : 0) + (tap_dx<true>(t) == -1 ? 1 : 0);
}
template <bool UP>
__host__ __device__ constexpr int shift_dy(int s) {
    return UP ? ((s & 2) ? -1 : 0) : s / 3 - 1;
}
template <bool UP>
__host__ __device__ constexpr int shift_dx(int s) {
    return UP ? ((s & 1) ? -1 : 0) : s % 3 - 1;
}

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN, bool UP, int NBUF, bool RT = false, bool PERSIST = false>
struct ConvTile {
    static constexpr int kBN = TB * TH * TW;
    static constexpr int kMT = BM / (32 * WM);
    static constexpr int kNT = kBN / (32 * WN);
    static constexpr int kCls = UP ? 4 : 1;
    static constexpr int kShifts = UP ? 4 : 9;
    static constexpr int kPH = UP ? TH + 1 : TH + 2;
    static constexpr int kPW = UP ? (TW + 7) / 4 * 4 : TW + 8;  // image cols x0-4 .. (no right halo when UP)
    static constexpr int kPlane = kPH * kPW;
    static constexpr int kWlFloats = 9 * KC * BM;
    // RT: per-block runtime tile geometry (8x8 main tile, 1x64 row strip, 64x1 column strip of a
    // transposed conv); the patch region is sized for the largest, the 65 x 8 column-strip patch
    static constexpr int kPlFloats = RT ? KC * 65 * 8 : TB * KC * kPlane;
    static constexpr int kWlInstr = (kWlFloats + 255) / 256;      // 1 KiB DMA pieces (last may be partial)
    static constexpr int kWlRegion = kWlInstr * 256;              // patch region starts piece-aligned
    static constexpr int kPlF4 = kPlFloats / 4;
    static constexpr int kPlInstr = (kPlF4 + 63) / 64;
    static constexpr int kBufFloats = kWlRegion + kPlInstr * 256;  // patch region padded to pieces
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BM % (32 * WM) == 0 && kBN % (32 * WN) == 0, "wave tiling");
    static_assert(KC % 2 == 0 && kWlFloats % 4 == 0 && kPlFloats % 4 == 0, "DMA pieces");
    // every wave issues the same number of DMA pieces per chunk (the counted vmcnt needs that);
    // when the piece count is not a multiple of 4 the spare slots re-issue piece 0 (same bytes)
    static constexpr int kPieces = RT ? kWlInstr : kWlInstr + kPlInstr;
    static constexpr int kPiecesPerWave = (kPieces + 3) / 4;
    static_assert(NBUF == 2 || NBUF == 3, "ring depth");
    static_assert(!RT || (UP && TB == 1 && kBN == 64 && NBUF == 2), "runtime geometry: 64-position up tiles");
    static_assert(!PERSIST || (TB == 1 && NBUF == 2), "persistent blocks: one sample per tile, ring depth 2");
    // dynamic LDS: NBUF staging buffers + style [TB][Cin] + demod [TB][BM] + bias [BM]
    // (persistent blocks: two sets of the constants, alternating per tile) + ToRGB coefficients [BM][4]
    static constexpr int kConstSets = PERSIST ? 2 : 1;
    // the fused last-layer ToRGB (kEpilogueRgb): its coefficients [BM][4] and the tile's window of the
    // half-resolution skip image [3][TH/2 + 2][TW/2 + 2]
    static constexpr bool kCanFuseRgb = !UP && !RT && TB == 1 && WM == 1 && kMT == 1;
    static constexpr int kSkipRows = TH / 2 + 2, kSkipCols = TW / 2 + 2;
    static constexpr int kSkipFloats = (3 * kSkipRows * kSkipCols + 63) / 64 * 64;
    static constexpr int kRgbFloats = kCanFuseRgb ? 4 * BM + kSkipFloats : 0;
    static size_t lds_bytes(int cin) {
        return sizeof(float) * (NBUF * (size_t)kBufFloats + kConstSets * ((size_t)TB * cin + (size_t)TB * BM + BM) + kRgbFloats);
    }
};

// uniform (scalar) description of one output tile
struct TileGeom {
    int m_tile, split, y0, x0, tile_b;
    int tw_log2, PHr, PWr;  // runtime geometry (RT) only
};

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN, bool UP, int NBUF, bool RT, bool PERSIST>
__global__ __launch_bounds__(256, UP ? ((RT && KC >= 4) || (!RT && BM > 64) ? 2 : GANCE_UP_BLOCKS) : 4) void modconv_mfma_kernel(const ConvArgs p) {
    using T = ConvTile<BM, TB, TH, TW, KC, WM, WN, UP, NBUF, RT, PERSIST>;
    constexpr int MT = T::kMT, NT = T::kNT, PH = T::kPH, PW = T::kPW;
    constexpr int NCLS = T::kCls;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const buf0 = smem;
    // constants of a tile: style [TB][Cin] | demod [TB][BM] | bias [BM]; persistent blocks alternate two sets
    float* const const0 = smem + NBUF * T::kBufFloats;
    const int const_floats = TB * p.Cin + TB * BM + BM;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: branches on it are scalar
    const int wm = wave / WN;
    const int wn = wave % WN;
    // lane ids are laundered once per tile (below) so that per-lane address arithmetic is not
    // hoisted out of a persistent block's tile loop into long-lived registers
    int lane = tid & 63;
    int l31 = lane & 31;
    int lh = lane >> 5;

    unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
    if (GANCE_DBG(16)) stamp0 = __builtin_amdgcn_s_memrealtime();

    const int Hp = p.H + 2, Wp = p.W + 8;

    // ---- virtual block id -> tile. XCD-aware remap: the ids handled by one XCD (v % 8; a
    // persistent block strides by a multiple of 8) form a contiguous range, so the m tiles of one
    // pixel tile and neighbouring pixel tiles hit the same L2 at about the same time ----
    auto decode = [&](int v) {
        TileGeom g;
        int id;
        {
            const int nwg = p.total_tiles;
            const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
            id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
            if (GANCE_DBG(64)) id = v;
        }
        g.m_tile = id % p.m_tiles;
        id /= p.m_tiles;
        g.split = id % p.nsplit;
        id /= p.nsplit;
        g.tw_log2 = 0;
        g.PHr = PH;
        g.PWr = PW;
        if (RT) {
            const int main_tiles = p.tiles_x * p.tiles_y;
            const int tiles_total = main_tiles + p.row_tiles + p.col_tiles;
            const int t = id % tiles_total;
            g.tile_b = id / tiles_total;
            if (t < main_tiles) {  // TH x TW tile of the exactly-tiled H x W position grid
                g.y0 = (t / p.tiles_x) * TH;
                g.x0 = (t % p.tiles_x) * TW;
                g.tw_log2 = 3;
                static_assert(!RT || (TH == 8 && TW == 8), "RT main tile is 8x8");
            } else if (t < main_tiles + p.row_tiles) {  // 1 x 64 on the position row y' = H
                g.y0 = p.H;
                g.x0 = (t - main_tiles) * 64;
                g.tw_log2 = 6;
                g.PHr = 2;
                g.PWr = 68;
            } else {  // 64 x 1 on the position column x' = W
                g.y0 = (t - main_tiles - p.row_tiles) * 64;
                g.x0 = p.W;
                g.tw_log2 = 0;
                g.PHr = 65;
                g.PWr = 8;
            }
        } else {
            const int tile_x = id % p.tiles_x;
            id /= p.tiles_x;
            const int tile_y = id % p.tiles_y;
            g.tile_b = id / p.tiles_y;
            g.y0 = tile_y * TH;
            g.x0 = tile_x * TW;
        }
        return g;
    };

    // ---- LDS-DMA staging: `buffer_load_dwordx4 ... lds` (MUBUF). The FLAT-encoded
    // global_load_lds would make hipcc degrade every LDS wait of the MFMA loop to lgkmcnt(0);
    // with the buffer form it emits counted waits and the fragment prefetch really overlaps. ----
    // Staging context = the tile whose chunks are being fetched (the NEXT tile during the last
    // chunk of a persistent block's current tile).
    __amdgpu_buffer_rsrc_t w_rsrc, x_rsrc;
    int st_y0 = 0, st_x0 = 0, st_b0 = 0, st_chunk_begin = 0;
    // RT: the patch image is [KC][PHr][PWr/4] float4s; a wave's pieces are i = wave, wave+4, ...
    // (at most 3); their per-lane source offsets are computed once per tile, a chunk only adds its plane offset
    int rt_off[3] = {0, 0, 0};
    int rt_pieces = 0;
    auto stage_setup = [&](const TileGeom& g) {
        st_y0 = g.y0;
        st_x0 = g.x0;
        st_b0 = g.tile_b * TB;
        st_chunk_begin = g.split * p.chunks_per_split;
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + ((size_t)g.m_tile * p.total_chunks) * T::kWlFloats), 0,
                                                   0x7fffffff, 0x00020000);
        x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)min(st_b0, p.B - 1) * p.x_b_stride), 0,
                                                   0x7fffffff, 0x00020000);
        if (RT) {
            const int f4_total = KC * g.PHr * g.PWr / 4;
            rt_pieces = (f4_total + 63) / 64;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = wave + 4 * r;
                const int f = i * 64 + lane;
                rt_off[r] = -1;
                if (i < rt_pieces && f < f4_total) {
                    const int q4 = g.PWr / 4;
                    const int q = f % q4;
                    int rr = f / q4;
                    const int py = rr % g.PHr;
                    const int c = rr / g.PHr;
                    const int gy = min(g.y0 + py, Hp - 1);
                    const int gx = min(g.x0 + 4 * q, Wp - 4);
                    rt_off[r] = ((c * Hp + gy) * Wp + gx) * 4;
                }
            }
        }
    };
    // chunk = index within the staging tile's K range
    auto stage = [&](int chunk_in_tile, float* buf) {
        const int chunk = st_chunk_begin + chunk_in_tile;
        const int wbase = chunk * T::kWlFloats;
        const int ci0 = chunk * KC;
        float* pl = buf + T::kWlRegion;
#pragma unroll
        for (int r = 0; r < T::kPiecesPerWave; ++r) {
            int g = wave + 4 * r;
            if (g >= T::kPieces) {
                if (NBUF == 2) continue;  // uncounted waits: waves need not issue equal piece counts
                g -= T::kPieces;          // counted vmcnt (ring of 3): spare slot repeats an early piece
            }
            if (g < T::kWlInstr) {
                if (g * 256 + lane * 4 < T::kWlFloats)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(buf + g * 256), 16,
                                                             (wbase + g * 256 + lane * 4) * 4, 0, 0, 0);
            } else if (!RT) {
                const int i = g - T::kWlInstr;
                const int f = min(i * 64 + lane, T::kPlF4 - 1);  // tail lanes repeat the last float4
                // float4 index f of the patch image [TB][KC][PH][PW/4] -> source address
                const int q = f % (PW / 4);
                int rr = f / (PW / 4);
                const int py = rr % PH;
                rr /= PH;
                const int c = rr % KC;
                const int tb = rr / KC;
                const int bb = min(st_b0 + tb, p.B - 1) - min(st_b0, p.B - 1);
                const int gy = min(st_y0 + py, Hp - 1);
                const int gx = min(st_x0 + 4 * q, Wp - 4);
                const long long off = (long long)bb * p.x_b_stride + ((long long)(ci0 + c) * Hp + gy) * Wp + gx;
                if (i * 64 + lane < T::kPlF4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(pl + i * 256), 16,
                                                             (int)(off * 4), 0, 0, 0);
            }
        }
        if (RT) {
            const int chunk_off = ci0 * Hp * Wp * 4;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = wave + 4 * r;
                if (i < rt_pieces && rt_off[r] >= 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(pl + i * 256), 16,
                                                             rt_off[r] + chunk_off, 0, 0, 0);
            }
        }
    };

    // ---- per-tile constants (style [Cin], demod [BM], bias [BM]) -> LDS set `set`, TB == 1:
    // dword LDS-DMA (64 consecutive floats per wave-instruction), no register staging, counted in
    // vmcnt with the chunk pieces they travel with ----
    auto consts_stage = [&](const TileGeom& g, int set) {
        float* const s_dst = const0 + set * const_floats;
        float* const d_dst = s_dst + p.Cin;
        float* const b_dst = d_dst + BM;
        const int b = min(g.tile_b, p.B - 1);
        const int m0g = g.m_tile * BM;
        const __amdgpu_buffer_rsrc_t s_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.s + (size_t)b * p.s_stride), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.d + (size_t)b * p.d_stride + m0g), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t b_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias + m0g), 0, 0x7fffffff, 0x00020000);
        for (int i = wave; i * 64 < p.Cin; i += 4)
            if (i * 64 + lane < p.Cin)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_ptr_t)(s_dst + i * 64), 4, (i * 64 + lane) * 4, 0, 0, 0);
        constexpr int kMPieces = (BM + 63) / 64;  // 1 or 2
        if (wave < kMPieces) {
            if (wave * 64 + lane < BM)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, (lds_ptr_t)(d_dst + wave * 64), 4, (wave * 64 + lane) * 4, 0, 0, 0);
        } else if (wave - 2 >= 0 && wave - 2 < kMPieces) {
            if ((wave - 2) * 64 + lane < BM)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr_t)(b_dst + (wave - 2) * 64), 4, ((wave - 2) * 64 + lane) * 4, 0, 0, 0);
        }
    };

    const int nchunks = p.chunks_per_split;

    int v = blockIdx.x;
    TileGeom cur = decode(v);
    stage_setup(cur);
    // ring prologue: NBUF-1 chunks in flight before the first MFMA
    stage(0, buf0);
    if (NBUF == 3 && nchunks > 1) stage(1, buf0 + T::kBufFloats);
    float* const rgb_lds = const0 + T::kConstSets * const_floats;  // [BM][4]: style * weight of the fused ToRGB
    if (TB == 1) {
        consts_stage(cur, 0);
        if constexpr (T::kCanFuseRgb) {
            if (p.epilogue == kEpilogueRgb) {
                if (tid < BM) {
                    const float sv = p.rgb_s[(size_t)min(cur.tile_b, p.B - 1) * p.s_stride + tid];
                    const float* wv = p.rgb_w + tid * 3;
                    *reinterpret_cast<float4*>(rgb_lds + tid * 4) = make_float4(sv * wv[0], sv * wv[1], sv * wv[2], 0.f);
                }
                // the tile's window of the previous skip image -> LDS by dword LDS-DMA (lands with the
                // first chunk): the epilogue then has no global load to wait for
                if (p.rgb_y_prev != nullptr) {
                    const int Rh = p.OW >> 1;
                    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.rgb_y_prev + (size_t)min(cur.tile_b, p.B - 1) * 3 * Rh * Rh), 0, 0x7fffffff, 0x00020000);
                    const int row0 = (cur.y0 >> 1) - 1, col0 = (cur.x0 >> 1) - 1;
                    for (int i = wave; i * 64 < 3 * T::kSkipRows * T::kSkipCols; i += 4) {
                        const int idx = i * 64 + lane;
                        const int cc = idx % T::kSkipCols;
                        const int rr = (idx / T::kSkipCols) % T::kSkipRows;
                        const int k = idx / (T::kSkipCols * T::kSkipRows);
                        const int row = row0 + rr, col = col0 + cc;
                        if (k < 3 && row >= 0 && row < Rh && col >= 0 && col < Rh)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (lds_ptr_t)(rgb_lds + 4 * BM + i * 64), 4,
                                                                     ((k * Rh + row) * Rh + col) * 4, 0, 0, 0);
                    }
                }
            }
        }
    } else {
        float* s_lds = const0;
        float* d_lds = s_lds + TB * p.Cin;
        float* b_lds = d_lds + TB * BM;
        const int b0 = cur.tile_b * TB, m0 = cur.m_tile * BM;
        for (int i = tid; i < TB * p.Cin; i += 256) {
            const int tb = i / p.Cin, ci = i - tb * p.Cin;
            const int b = min(b0 + tb, p.B - 1);
            s_lds[i] = p.s[(size_t)b * p.s_stride + ci];
        }
        for (int i = tid; i < TB * BM; i += 256) {
            const int tb = i / BM, m = i - tb * BM;
            const int b = min(b0 + tb, p.B - 1);
            d_lds[i] = p.d[(size_t)b * p.d_stride + m0 + m];
        }
        for (int i = tid; i < BM; i += 256) b_lds[i] = p.bias[m0 + i];
    }

    int ring = 0;      // buffer of the chunk about to be consumed (persistent blocks: runs on across tiles)
    int tile_no = 0;   // tiles this block has started (selects the constants set)
    bool first_chunk_landed = false;  // persistent: the next tile's chunk 0 was waited for before the epilogue
    while (true) {
        if (PERSIST) asm volatile("" : "+v"(lane), "+v"(l31), "+v"(lh));
        const int next_v = v + (int)gridDim.x;
        const bool has_next = PERSIST && next_v < p.total_tiles;
        const int set = PERSIST ? (tile_no & 1) : 0;
        const float* const s_lds = const0 + set * const_floats;
        const float* const d_lds = s_lds + TB * p.Cin;
        const float* const b_lds = d_lds + TB * BM;
        const int m0 = cur.m_tile * BM;
        const int b0 = cur.tile_b * TB;
        const int y0 = cur.y0, x0 = cur.x0;
        const int tw_log2 = cur.tw_log2, PWr = cur.PWr;
        const int PLANEr = cur.PHr * cur.PWr;
        const int chunk_begin = cur.split * p.chunks_per_split;

        // ---- per-lane operand offsets ----
        const int aoff = lh * BM + wm * (MT * 32) + l31;
        int boff[NT];
        int stb[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (wn * NT + j) * 32 + l31;
            const int tb = RT ? 0 : n / (TH * TW);
            const int yy = RT ? (n >> tw_log2) : (n / TW) % TH;
            const int xx = RT ? (n & ((1 << tw_log2) - 1)) : n % TW;
            boff[j] = (tb * KC + lh) * PLANEr + (yy + 1) * PWr + (xx + 4);
            stb[j] = tb * p.Cin + lh;
        }

        f32x16 acc[NCLS][MT][NT];
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

        for (int k = 0; k < nchunks; ++k) {
            // Each wave waits for ITS pieces of chunk k (counted: the pieces of chunk k+1 it issued
            // later may stay in flight), then the raw barrier makes chunk k visible to all waves and
            // proves chunk k-1's buffer is no longer read. No __syncthreads here: its fence would make
            // hipcc drain vmcnt to 0 and serialise the ring.
            if (NBUF == 3 && k + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::kPiecesPerWave) : "memory");
            } else if (!(PERSIST && k == 0 && first_chunk_landed)) {
                // (a persistent block already waited for this chunk before the previous tile's
                // epilogue; its stores, still in flight, must not be waited for here)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (GANCE_DBG(16) && k == 0 && tile_no == 0) stamp1 = __builtin_amdgcn_s_memrealtime();
            float* const cur_buf = buf0 + (NBUF == 3 ? (k % NBUF) : ring) * T::kBufFloats;
            float* const nxt_buf = buf0 + (NBUF == 3 ? ((k + NBUF - 1) % NBUF) : (ring ^ 1)) * T::kBufFloats;
            const bool last = k + NBUF - 1 >= nchunks;
            int stage_chunk = k + NBUF - 1;
            bool do_stage = !last;
            if (last && has_next && k + 1 == nchunks) {
                // cross-tile prefetch: the next tile's first chunk and constants fly under this
                // chunk's MFMAs and the epilogue's stores
                const TileGeom nxt = decode(next_v);
                stage_setup(nxt);
                consts_stage(nxt, set ^ 1);
                stage_chunk = 0;
                do_stage = true;
            }
            if (do_stage && !GANCE_DBG(2)) stage(stage_chunk, nxt_buf);
            if (NBUF == 2) ring ^= 1;
            const float* Wl = cur_buf + aoff;
            const float* Pl = cur_buf + T::kWlRegion;
            const int ci0 = (chunk_begin + k) * KC;

            // Flattened steps u = kk*9 + tap, fully unrolled. The operand fragments of step u+1 are
            // read from LDS BEFORE the MFMAs of step u are issued (sched_barrier pins that order), so
            // the matrix pipe never waits on an LDS round trip: one wave alone keeps it busy.
            // A group = the MT weight fragments of one (kk, tap). B group = the NT patch fragments of
            // one (kk, shift): a stride-1 conv has one shift per tap, the transposed conv re-uses its
            // 4 shifts across the 9 taps of a kk, so its B fragments are read once per kk.
            constexpr int U = 9 * (KC / 2);
            constexpr int BG = UP ? 4 : 1;
            float afrag[2][MT];
            float bfrag[2][BG][NT];
            float sfrag[2];
            auto load_a = [&](int u, float (&dst)[MT], float& sdst) {
                const int kk = u / 9, t = u % 9;
                // the style scale is read here but multiplied in at the USE step, so that nothing
                // between two MFMA groups depends on an LDS read issued in the same step
                sdst = (TB == 1) ? s_lds[lh + ci0 + 2 * kk] : 1.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) dst[i] = Wl[(t * KC + 2 * kk) * BM + i * 32];
            };
            auto load_b = [&](int kk, int shift, float (&dst)[NT]) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    dst[j] = Pl[boff[j] + (2 * kk) * PLANEr + shift_dy<UP>(shift) * PWr + shift_dx<UP>(shift)];
                    if (TB > 1) dst[j] *= s_lds[stb[j] + ci0 + 2 * kk];
                }
            };
            if (GANCE_DBG(4)) continue;
            load_a(0, afrag[0], sfrag[0]);
            if (UP) {
#pragma unroll
                for (int sh = 0; sh < 4; ++sh) load_b(0, sh, bfrag[0][sh]);
            } else {
                load_b(0, 0, bfrag[0][0]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = u / 9, t = u % 9;
                if (u + 1 < U) {
                    load_a(u + 1, afrag[(u + 1) & 1], sfrag[(u + 1) & 1]);
                    if (UP) {
                        if (t == 8) {
#pragma unroll
                            for (int sh = 0; sh < 4; ++sh) load_b(kk + 1, sh, bfrag[(kk + 1) & 1][sh]);
                        }
                    } else {
                        load_b((u + 1) / 9, (u + 1) % 9, bfrag[(u + 1) & 1][0]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const int bsel = UP ? (kk & 1) : (u & 1);
                const int bgrp = UP ? tap_shift<UP>(t) : 0;
                float a[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = (TB == 1) ? afrag[u & 1][i] * sfrag[u & 1] : afrag[u & 1][i];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[tap_cls<UP>(t)][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            a[i], bfrag[bsel][bgrp][j], acc[tap_cls<UP>(t)][i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (has_next) {
            // the next tile's first chunk and constants have had a whole chunk of MFMAs to land:
            // wait for them now, so that the stores below are the only vector-memory operations
            // still in flight when the next tile starts (vmcnt counts in issue order)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            first_chunk_landed = true;
        }

        if (GANCE_DBG(16) && tile_no == 0) stamp2 = __builtin_amdgcn_s_memrealtime();
        if (GANCE_TUNE_EPILOGUE_PRIO) __builtin_amdgcn_s_setprio(GANCE_TUNE_EPILOGUE_PRIO);
        // ---- epilogue: demodulate, (noise, bias, leaky relu), store 32 consecutive pixels per reg ----
        // Per-channel constants come out of LDS in one batch (one exposed LDS round trip, not one per
        // element). Stores are MUBUF with the channel plane in the SCALAR offset (register r of a
        // 32x32 accumulator is channel (r&3) + 8(r>>2) + 4*lane_half): the per-lane byte offset of a
        // pixel is computed once per 32-pixel group and no vector ALU work sits between two stores.
        // The epilogue kind is a uniform branch around the whole store loop, not one per element.
        const int c_stride_bytes = (int)p.out_c_stride * 4;
        float* const out_tile = p.out + (size_t)cur.split * p.slab_stride + (size_t)b0 * p.out_b_stride +
                                (size_t)(m0 + wm * (MT * 32)) * p.out_c_stride;
        auto emit = [&](auto full_tag) {
            constexpr bool kFull = decltype(full_tag)::value;
            // the NT noise values of a lane are fetched together, ahead of the store loop
            float nzv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wn * NT + j) * 32 + l31;
                const int tb = RT ? 0 : n / (TH * TW);
                const int oy = y0 + (RT ? (n >> tw_log2) : (n / TW) % TH);
                const int ox = x0 + (RT ? (n & ((1 << tw_log2) - 1)) : n % TW);
                nzv[j] = 0.f;
                if (kFull && p.noise != nullptr && b0 + tb < p.B && oy < p.OH && ox < p.OW)
                    nzv[j] = p.noise[(size_t)(b0 + tb) * p.noise_b_stride + (size_t)oy * p.OW + ox];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float dreg[16], breg[16];
                auto load_consts = [&](int tb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        dreg[r] = d_lds[tb * BM + m];
                        if (kFull) breg[r] = b_lds[m];
                    }
                };
                if (TB == 1) load_consts(0);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = (wn * NT + j) * 32 + l31;
                    const int tb = RT ? 0 : n / (TH * TW);
                    const int oy = y0 + (RT ? (n >> tw_log2) : (n / TW) % TH);
                    const int ox = x0 + (RT ? (n & ((1 << tw_log2) - 1)) : n % TW);
                    const bool in_batch = b0 + tb < p.B;
                    if (TB > 1) load_consts(tb);
                    const float nz = nzv[j] * p.noise_strength;
                    const int voff = ((oy + p.out_y_off) * p.out_row_stride + ox + p.out_x_off) * 4 +
                                     4 * lh * c_stride_bytes + (TB > 1 ? tb * (int)p.out_b_stride * 4 : 0);
#pragma unroll
                    for (int c = 0; c < NCLS; ++c) {
                        // class c = (py, px): valid positions shrink by one where the parity is odd
                        const bool ok = in_batch && oy < p.OH - (UP ? (c >> 1) : 0) && ox < p.OW - (UP ? (c & 1) : 0);
                        if (!ok || GANCE_DBG(1)) continue;
                        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                            (void*)(out_tile + (size_t)c * p.cls_stride + (size_t)(i * 32) * p.out_c_stride), 0, 0x7fffffff,
                            0x00020000);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v_out = acc[c][i][j][r] * dreg[r];
                            if (kFull) {
                                v_out += nz + breg[r];
                                v_out = fmaxf(v_out, 0.2f * v_out) * 1.4142135623730951f;  // lrelu(0.2) * sqrt(2)
                            }
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v_out), o_rsrc, voff, ((r & 3) + 8 * (r >> 2)) * c_stride_bytes, 0);
                        }
                    }
                }
            }
        };
        // ---- last layer: activation -> ToRGB -> + skip image -> uint8, nothing but the image is stored.
        // A wave holds every output channel of its pixels (WM == 1): lane half lh has channels
        // (r&3) + 8(r>>2) + 4 lh of each 32-channel group, so a pixel's RGB is 16 * MT fused
        // multiply-adds per lane and one exchange between the two lane halves.
        auto emit_rgb = [&]() {
            const int R = p.OW;
            const int Rh = R >> 1;
            // upsample_2d of the previous skip image ([1,3,3,1]/4 per axis = two taps per axis), read
            // from the window the prologue put in LDS: lane half 0 takes the upper source row of its
            // pixels, half 1 the lower one; the weighted halves join the channel sums before the one
            // exchange between the lane halves.
            float up[NT][3];
            float nzv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wn * NT + j) * 32 + l31;
                const int oy = y0 + (n / TW) % TH;
                const int ox = x0 + n % TW;
                const bool ok = b0 < p.B && oy < p.OH && ox < p.OW;
                nzv[j] = (p.noise != nullptr && ok) ? p.noise[(size_t)b0 * p.noise_b_stride + (size_t)oy * p.OW + ox] : 0.f;
                const int ya = (oy & 1) ? (oy >> 1) : (oy >> 1) - 1;
                const int row = ya + lh;
                const float wrow = ((oy & 1) != 0) == (lh == 0) ? 0.75f : 0.25f;  // (odd: .75, .25) (even: .25, .75)
                const int xa = (ox & 1) ? (ox >> 1) : (ox >> 1) - 1, xb = xa + 1;
                const float wxa = (ox & 1) ? 0.75f : 0.25f, wxb = 1.0f - wxa;
                const bool row_ok = p.rgb_y_prev != nullptr && ok && row >= 0 && row < Rh;
                const float* skip = rgb_lds + 4 * BM + (row - ((y0 >> 1) - 1)) * T::kSkipCols - ((x0 >> 1) - 1);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float va = (row_ok && xa >= 0) ? skip[k * T::kSkipRows * T::kSkipCols + xa] : 0.f;
                    const float vb = (row_ok && xb < Rh) ? skip[k * T::kSkipRows * T::kSkipCols + xb] : 0.f;
                    up[j][k] = wrow * (wxa * va + wxb * vb);
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wn * NT + j) * 32 + l31;
                const int oy = y0 + (n / TW) % TH;
                const int ox = x0 + n % TW;
                const bool ok = b0 < p.B && oy < p.OH && ox < p.OW;
                const float nz = nzv[j] * p.noise_strength;
                float rgb[3] = {up[j][0], up[j][1], up[j][2]};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v_out = acc[0][0][j][r] * d_lds[m];
                    v_out += nz + b_lds[m];
                    v_out = fmaxf(v_out, 0.2f * v_out) * 1.4142135623730951f;
                    const float4 coef = *reinterpret_cast<const float4*>(rgb_lds + m * 4);
                    rgb[0] = fmaf(v_out, coef.x, rgb[0]);
                    rgb[1] = fmaf(v_out, coef.y, rgb[1]);
                    rgb[2] = fmaf(v_out, coef.z, rgb[2]);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) rgb[k] += __shfl_xor(rgb[k], 32);
                const size_t pix = (size_t)oy * R + ox;
                unsigned packed = 0;  // this pixel's three bytes
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float y = rgb[k] + p.rgb_bias[k];
                    if (p.rgb_y != nullptr && lh == 0 && ok && !GANCE_DBG(1)) p.rgb_y[((size_t)b0 * 3 + k) * R * R + pix] = y;
                    // tf.saturate_cast(x * 127.5 + 128): two roundings (the barrier keeps them apart)
                    float q = y * 127.5f;
                    asm volatile("" : "+v"(q));
                    q += 128.0f;
                    q = fminf(fmaxf(q, 0.f), 255.f);
                    packed |= (unsigned)(int)q << (8 * k);
                }
                // 32 pixels x 3 bytes = 24 dwords, contiguous in the NHWC frame: lane t < 24 assembles
                // dword t from the packed pixels floor(4t/3) and floor(4t/3)+1 and stores it, so the
                // group leaves as ONE 96-byte store instead of three byte-strided ones
                const int first = (4 * l31) / 3, skew = (4 * l31) % 3;
                const unsigned lo = __shfl(packed, first), hi = __shfl(packed, min(first + 1, 31));
                const unsigned word = skew == 0 ? (lo | (hi << 24)) : (skew == 1 ? ((lo >> 8) | (hi << 16)) : ((lo >> 16) | (hi << 8)));
                // (the tile is 64 pixels wide and R % 64 == 0, so a 32-pixel group is whole or absent)
                const bool group_ok = b0 < p.B && oy < p.OH && (ox - l31) + 31 < p.OW;
                if (p.rgb_u8 != nullptr && lh == 0 && l31 < 24 && group_ok && !GANCE_DBG(1))
                    reinterpret_cast<unsigned*>(p.rgb_u8 + ((size_t)b0 * R * R + (pix - l31)) * 3)[l31] = word;
            }
        };
        if constexpr (!UP && TB == 1 && WM == 1 && MT == 1 && !RT) {
            if (p.epilogue == kEpilogueRgb) {
                emit_rgb();
            } else if (p.epilogue == kEpilogueFull) {
                emit(std::true_type{});
            } else {
                emit(std::false_type{});
            }
        } else {
            if (!UP && p.epilogue == kEpilogueFull)
                emit(std::true_type{});
            else
                emit(std::false_type{});
        }
        if (!has_next) break;
        v = next_v;
        cur = decode(v);
        ++tile_no;
    }
    if (GANCE_DBG(16) && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* dst = p.debug_stamps + (size_t)blockIdx.x * 5;
        dst[0] = stamp0;
        dst[1] = stamp1;
        dst[2] = stamp2;
        dst[3] = __builtin_amdgcn_s_memrealtime();
        dst[4] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
}

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN, bool UP, int NBUF, bool RT = false, bool PERSIST = false>
static hipError_t launch_one(const ConvArgs& args, int total_blocks, hipStream_t stream) {
    using T = ConvTile<BM, TB, TH, TW, KC, WM, WN, UP, NBUF, RT, PERSIST>;
    auto kernel = modconv_mfma_kernel<BM, TB, TH, TW, KC, WM, WN, UP, NBUF, RT, PERSIST>;
    // per device: the dynamic-LDS opt-in of this kernel variant and the persistent launch size (what fits the
    // chip at once, a multiple of 8 = XCDs)
    static PerDeviceInt resident;
    int resident_blocks = 0;
    hipError_t e = resident.get(
        [&](int device, int* value) {
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)T::lds_bytes(512));
            if (err != hipSuccess) return err;
            int cus = 0, per_cu = 0;
            if ((err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return err;
            if ((err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, T::lds_bytes(512))) != hipSuccess) return err;
            *value = std::max(8, cus * std::max(per_cu, 1) / 8 * 8);
            return hipSuccess;
        },
        &resident_blocks);
    if (e != hipSuccess) return e;
    ConvArgs a = args;
    a.total_tiles = total_blocks;
    int grid = total_blocks;
    if (PERSIST) {
        // LDS use shrinks with Cin, so more blocks may fit than at Cin = 512: ask for this launch
        int per_cu = 0, device = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, T::lds_bytes(a.Cin)) == hipSuccess &&
            hipGetDevice(&device) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && per_cu > 0)
            grid = std::min(total_blocks, std::max(8, cus * per_cu / 8 * 8));
        else
            grid = std::min(total_blocks, resident_blocks);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), T::lds_bytes(a.Cin), stream, a);
    return hipGetLastError();
}

const ConvTileInfo kConvTiles[kNumConvTiles] = {
    // BM, TB, TH, TW, KC, up
    {32, 1, 8, 64, 8, 0},   // 0: Cout = 32
    {64, 1, 4, 64, 8, 0},   // 1: Cout = 64
    {128, 1, 4, 32, 4, 0},  // 2: Cout >= 128, wide grids
    {128, 1, 8, 16, 4, 0},  // 3
    {128, 2, 8, 8, 4, 0},   // 4
    {128, 8, 4, 4, 4, 0},   // 5
    {32, 1, 16, 16, 8, 1},  // 6: transposed, Cout = 32
    {64, 1, 8, 16, 8, 1},   // 7: transposed, Cout = 64
    {128, 1, 8, 8, 4, 1},   // 8: transposed, Cout >= 128: runtime geometry (8x8 + edge strips)
    {128, 1, 4, 16, 4, 1},  // 9: unused
    {32, 1, 8, 64, 4, 0},   // 10: as 0 with KC = 4 (more blocks per CU)
    {64, 1, 4, 64, 4, 0},   // 11: as 1 with KC = 4
    {32, 1, 16, 16, 4, 1},  // 12: as 6 with KC = 4
    {64, 1, 8, 16, 4, 1},   // 13: as 7 with KC = 4
    {128, 1, 8, 8, 2, 1},   // 14: as 8 with KC = 2 (three resident blocks per CU)
};

hipError_t launch_modconv(int tile_id, const ConvArgs& a, int total_blocks, hipStream_t stream) {
    // ring depth 2: depth 3 (prefetch distance 2, template parameter NBUF) measured slower, it costs
    // a resident block per CU and DMA latency is already covered.
    // Persistent blocks (PERSIST: a block strides over tiles and fetches the next tile's first chunk
    // under the current tile's last MFMAs and epilogue) were built and measured: 4 % slower over
    // the whole path, +-0 % on the runtime-geometry tile alone; the hardware's own workgroup
    // turnover is cheaper than the scalar state a tile loop keeps alive. The variants are only
    // instantiated with -DGANCE_CONV_PERSIST=1 (GANCE_TUNE_PERSIST = bit mask over tile ids).
#if GANCE_CONV_PERSIST
    static const unsigned persist_mask = [] {
        const char* v = std::getenv("GANCE_TUNE_PERSIST");
        return v ? (unsigned)std::strtoul(v, nullptr, 0) : 0u;
    }();
    const bool persist = (persist_mask >> tile_id) & 1u;
#define GANCE_CASE(id, ...)                                                              \
    case id:                                                                             \
        return persist ? launch_one<__VA_ARGS__, 2, false, true>(a, total_blocks, stream) \
                       : launch_one<__VA_ARGS__, 2>(a, total_blocks, stream);
#else
#define GANCE_CASE(id, ...) \
    case id:               \
        return launch_one<__VA_ARGS__, 2>(a, total_blocks, stream);
#endif
#define GANCE_CASE_TB(id, ...) \
    case id:                  \
        return launch_one<__VA_ARGS__, 2>(a, total_blocks, stream);
    switch (tile_id) {
        GANCE_CASE(0, 32, 1, 8, 64, 8, 1, 4, false)
        GANCE_CASE(1, 64, 1, 4, 64, 8, 1, 4, false)
        GANCE_CASE(2, 128, 1, 4, 32, 4, 2, 2, false)
        GANCE_CASE(3, 128, 1, 8, 16, 4, 2, 2, false)
        GANCE_CASE_TB(4, 128, 2, 8, 8, 4, 2, 2, false)
        GANCE_CASE_TB(5, 128, 8, 4, 4, 4, 2, 2, false)
        GANCE_CASE(6, 32, 1, 16, 16, 8, 1, 4, true)
        GANCE_CASE(7, 64, 1, 8, 16, 8, 2, 2, true)
        case 8:
#if GANCE_CONV_PERSIST
            if (persist) return launch_one<128, 1, 8, 8, 4, 4, 1, true, 2, true, true>(a, total_blocks, stream);
#endif
            return launch_one<128, 1, 8, 8, 4, 4, 1, true, 2, true>(a, total_blocks, stream);
        GANCE_CASE(9, 128, 1, 4, 16, 4, 4, 1, true)
        GANCE_CASE(10, 32, 1, 8, 64, 4, 1, 4, false)
        GANCE_CASE(11, 64, 1, 4, 64, 4, 1, 4, false)
        GANCE_CASE(12, 32, 1, 16, 16, 4, 1, 4, true)
        GANCE_CASE(13, 64, 1, 8, 16, 4, 2, 2, true)
        case 14: return launch_one<128, 1, 8, 8, 2, 4, 1, true, 2, true>(a, total_blocks, stream);
        default: return hipErrorInvalidValue;
    }
#undef GANCE_CASE
#undef GANCE_CASE_TB
}

}  // namespace gance
