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
// Pipeline: a ring of 3 LDS buffers; `buffer_load_dwordx4 ... lds` of chunks k+1 and k+2 are in
// flight while the MFMAs of chunk k run; per chunk one counted `s_waitcnt vmcnt(N)` + one raw
// s_barrier. All LDS is one dynamic array; no ordinary global load sits inside the K loop.

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

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

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN, bool UP, int NBUF, bool RT = false>
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
    // dynamic LDS: NBUF staging buffers + style [TB][Cin] + demod [TB][BM] + bias [BM]
    static size_t lds_bytes(int cin) {
        return sizeof(float) * (NBUF * (size_t)kBufFloats + (size_t)TB * cin + (size_t)TB * BM + BM);
    }
};

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN, bool UP, int NBUF, bool RT>
__global__ __launch_bounds__(256, UP ? 2 : 4) void modconv_mfma_kernel(const ConvArgs p) {
    using T = ConvTile<BM, TB, TH, TW, KC, WM, WN, UP, NBUF, RT>;
    constexpr int MT = T::kMT, NT = T::kNT, PH = T::kPH, PW = T::kPW;
    constexpr int NCLS = T::kCls;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const buf0 = smem;
    float* const s_lds = smem + NBUF * T::kBufFloats;  // [TB][Cin]
    float* const d_lds = s_lds + TB * p.Cin;        // [TB][BM]
    float* const b_lds = d_lds + TB * BM;           // [BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
    if (p.debug_flags & 16) stamp0 = __builtin_amdgcn_s_memrealtime();
    // ---- XCD-aware block remap: blocks that share an XCD (bid % 8) get a contiguous id range,
    // so the m tiles of one pixel tile and neighbouring pixel tiles hit the same L2 ----
    int id;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (p.debug_flags & 64) id = bid;
    }
    if ((p.debug_flags & 8) && blockIdx.x < 2048) {
        // experiment: de-phase the first generation of blocks
        const int reps = (blockIdx.x >> 8) & 7;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int m_tile = id % p.m_tiles;
    id /= p.m_tiles;
    const int split = id % p.nsplit;
    id /= p.nsplit;
    // tile geometry: compile-time, or (RT) chosen per block
    int y0, x0, tile_b;
    int tw_log2 = 0, th_r = TH, PHr = PH, PWr = PW;
    if (RT) {
        const int main_tiles = p.tiles_x * p.tiles_y;
        const int tiles_total = main_tiles + p.row_tiles + p.col_tiles;
        const int t = id % tiles_total;
        tile_b = id / tiles_total;
        if (t < main_tiles) {  // TH x TW tile of the exactly-tiled H x W position grid
            y0 = (t / p.tiles_x) * TH;
            x0 = (t % p.tiles_x) * TW;
            tw_log2 = 3;
            static_assert(!RT || (TH == 8 && TW == 8), "RT main tile is 8x8");
        } else if (t < main_tiles + p.row_tiles) {  // 1 x 64 on the position row y' = H
            y0 = p.H;
            x0 = (t - main_tiles) * 64;
            tw_log2 = 6;
            th_r = 1;
            PHr = 2;
            PWr = 68;
        } else {  // 64 x 1 on the position column x' = W
            y0 = (t - main_tiles - p.row_tiles) * 64;
            x0 = p.W;
            tw_log2 = 0;
            th_r = 64;
            PHr = 65;
            PWr = 8;
        }
    } else {
        const int tile_x = id % p.tiles_x;
        id /= p.tiles_x;
        const int tile_y = id % p.tiles_y;
        tile_b = id / p.tiles_y;
        y0 = tile_y * TH;
        x0 = tile_x * TW;
    }
    const int PLANEr = PHr * PWr;
    (void)th_r;

    const int m0 = m_tile * BM;
    const int b0 = tile_b * TB;
    const int Hp = p.H + 2, Wp = p.W + 8;

    // ---- LDS-DMA staging: `buffer_load_dwordx4 ... lds` (MUBUF). The FLAT-encoded
    // global_load_lds would make hipcc degrade every LDS wait of the MFMA loop to lgkmcnt(0);
    // with the buffer form it emits counted waits and the fragment prefetch really overlaps. ----
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + ((size_t)m_tile * p.total_chunks) * T::kWlFloats), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)min(b0, p.B - 1) * p.x_b_stride), 0, 0x7fffffff, 0x00020000);
    // RT: the patch image is [KC][PHr][PWr/4] float4s; a wave's pieces are i = wave, wave+4, ...
    // (at most 3); their per-lane source offsets are computed once, a chunk only adds its plane offset
    int rt_off[3] = {0, 0, 0};
    int rt_pieces = 0;
    if (RT) {
        const int f4_total = KC * PLANEr / 4;
        rt_pieces = (f4_total + 63) / 64;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = wave + 4 * r;
            const int f = i * 64 + lane;
            rt_off[r] = -1;
            if (i < rt_pieces && f < f4_total) {
                const int q4 = PWr / 4;
                const int q = f % q4;
                int rr = f / q4;
                const int py = rr % PHr;
                const int c = rr / PHr;
                const int gy = min(y0 + py, Hp - 1);
                const int gx = min(x0 + 4 * q, Wp - 4);
                rt_off[r] = ((c * Hp + gy) * Wp + gx) * 4;
            }
        }
    }
    auto stage = [&](int chunk, float* buf) {
        const int wbase = chunk * T::kWlFloats;
        const int ci0 = chunk * KC;
        float* pl = buf + T::kWlRegion;
#pragma unroll
        for (int r = 0; r < T::kPiecesPerWave; ++r) {
            int g = wave + 4 * r;
            if (g >= T::kPieces) g -= T::kPieces;  // spare slot: repeat an early piece
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
                const int bb = min(b0 + tb, p.B - 1) - min(b0, p.B - 1);
                const int gy = min(y0 + py, Hp - 1);
                const int gx = min(x0 + 4 * q, Wp - 4);
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

    // ---- per-lane operand offsets ----
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
    const int aoff = lh * BM + wm * (MT * 32) + l31;

    f32x16 acc[NCLS][MT][NT];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

    const int chunk_begin = split * p.chunks_per_split;
    const int nchunks = p.chunks_per_split;

    // ring prologue: NBUF-1 chunks in flight before the first MFMA
    stage(chunk_begin, buf0);
    if (NBUF == 3 && nchunks > 1) stage(chunk_begin + 1, buf0 + T::kBufFloats);

    // ---- one-time staging of style, demod and bias. All global loads of a thread are issued
    // before the first LDS write, so the block pays one memory round trip, not three. ----
    if (TB == 1) {
        const int b = min(b0, p.B - 1);
        const float* sp = p.s + (size_t)b * p.s_stride;
        const float s0 = tid < p.Cin ? sp[tid] : 0.f;
        const float s1 = tid + 256 < p.Cin ? sp[tid + 256] : 0.f;
        const float dv = tid < BM ? p.d[(size_t)b * p.d_stride + m0 + tid] : 0.f;
        const float bv = tid < BM ? p.bias[m0 + tid] : 0.f;
        if (tid < p.Cin) s_lds[tid] = s0;
        if (tid + 256 < p.Cin) s_lds[tid + 256] = s1;
        if (tid < BM) {
            d_lds[tid] = dv;
            b_lds[tid] = bv;
        }
    } else {
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

    for (int k = 0; k < nchunks; ++k) {
        // Each wave waits for ITS pieces of chunk k (counted: the pieces of chunk k+1 it issued
        // later may stay in flight), then the raw barrier makes chunk k visible to all waves and
        // proves chunk k-1's buffer is no longer read. No __syncthreads here: its fence would make
        // hipcc drain vmcnt to 0 and serialise the ring.
        if (NBUF == 3 && k + 1 < nchunks) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::kPiecesPerWave) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ((p.debug_flags & 16) && k == 0) stamp1 = __builtin_amdgcn_s_memrealtime();
        float* const cur = buf0 + (k % NBUF) * T::kBufFloats;
        if (k + NBUF - 1 < nchunks && !(p.debug_flags & 2))
            stage(chunk_begin + k + NBUF - 1, buf0 + ((k + NBUF - 1) % NBUF) * T::kBufFloats);
        const float* Wl = cur + aoff;
        const float* Pl = cur + T::kWlRegion;
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
        if (p.debug_flags & 4) continue;
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

    if (p.debug_flags & 16) stamp2 = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue: demodulate, (noise, bias, leaky relu), store 32 consecutive pixels per reg ----
    // Per-channel constants come out of LDS in one batch (one exposed LDS round trip, not one per
    // element); channel-plane pointers advance by adds of two precomputed strides (register r of a
    // 32x32 accumulator is channel (r&3) + 8(r>>2) + 4*lane_half: +1,+1,+1,+5,...).
    const bool full = !UP && p.epilogue == kEpilogueFull;
    const long long stride1 = p.out_c_stride, stride5 = 5 * p.out_c_stride;
    float* const out = p.out + (size_t)split * p.slab_stride +
                       (size_t)(m0 + wm * (MT * 32) + 4 * lh) * p.out_c_stride;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float dreg[16], breg[16];
        auto load_consts = [&](int tb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                dreg[r] = d_lds[tb * BM + m];
                breg[r] = b_lds[m];
            }
        };
        if (TB == 1) load_consts(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (wn * NT + j) * 32 + l31;
            const int tb = RT ? 0 : n / (TH * TW);
            const int oy = y0 + (RT ? (n >> tw_log2) : (n / TW) % TH);
            const int ox = x0 + (RT ? (n & ((1 << tw_log2) - 1)) : n % TW);
            const int b = b0 + tb;
            const bool in_batch = b < p.B;
            if (TB > 1) load_consts(tb);
            float nz = 0.f;
            if (full && p.noise != nullptr && in_batch && oy < p.OH && ox < p.OW)
                nz = p.noise[(size_t)oy * p.OW + ox] * p.noise_strength;
            float* const out_px = out + (size_t)b * p.out_b_stride + (size_t)(i * 32) * p.out_c_stride +
                                  (size_t)(oy + p.out_y_off) * p.out_row_stride + ox + p.out_x_off;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                // class c = (py, px): valid positions shrink by one where the parity is odd
                const bool ok = in_batch && oy < p.OH - (UP ? (c >> 1) : 0) && ox < p.OW - (UP ? (c & 1) : 0);
                if (!ok || (p.debug_flags & 1)) continue;
                float* ptr = out_px + (size_t)c * p.cls_stride;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[c][i][j][r] * dreg[r];
                    if (full) {
                        v += nz + breg[r];
                        v = (v < 0.f ? 0.2f * v : v) * 1.4142135623730951f;
                    }
                    if (p.debug_flags & 32) __builtin_nontemporal_store(v, ptr); else *ptr = v;
                    ptr += ((r & 3) == 3) ? stride5 : stride1;
                }
            }
        }
    }
    if ((p.debug_flags & 16) && tid == 0) {
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

template <int BM, int TB, int TH, int TW, int KC, int WM, int WN, bool UP, int NBUF, bool RT = false>
static hipError_t launch_one(const ConvArgs& a, int total_blocks, hipStream_t stream) {
    using T = ConvTile<BM, TB, TH, TW, KC, WM, WN, UP, NBUF, RT>;
    auto kernel = modconv_mfma_kernel<BM, TB, TH, TW, KC, WM, WN, UP, NBUF, RT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)T::lds_bytes(512));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(total_blocks), dim3(256), T::lds_bytes(a.Cin), stream, a);
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
};

hipError_t launch_modconv(int tile_id, const ConvArgs& a, int total_blocks, hipStream_t stream) {
    // ring depth 2: depth 3 (prefetch distance 2, template parameter NBUF) measured slower, it costs
    // a resident block per CU and DMA latency is already covered
#define GANCE_CASE(id, ...) \
    case id:               \
        return launch_one<__VA_ARGS__, 2>(a, total_blocks, stream);
    switch (tile_id) {
        GANCE_CASE(0, 32, 1, 8, 64, 8, 1, 4, false)
        GANCE_CASE(1, 64, 1, 4, 64, 8, 1, 4, false)
        GANCE_CASE(2, 128, 1, 4, 32, 4, 2, 2, false)
        GANCE_CASE(3, 128, 1, 8, 16, 4, 2, 2, false)
        GANCE_CASE(4, 128, 2, 8, 8, 4, 2, 2, false)
        GANCE_CASE(5, 128, 8, 4, 4, 4, 2, 2, false)
        GANCE_CASE(6, 32, 1, 16, 16, 8, 1, 4, true)
        GANCE_CASE(7, 64, 1, 8, 16, 8, 2, 2, true)
        case 8: return launch_one<128, 1, 8, 8, 4, 4, 1, true, 2, true>(a, total_blocks, stream);
        GANCE_CASE(9, 128, 1, 4, 16, 4, 4, 1, true)
        GANCE_CASE(10, 32, 1, 8, 64, 4, 1, 4, false)
        GANCE_CASE(11, 64, 1, 4, 64, 4, 1, 4, false)
        GANCE_CASE(12, 32, 1, 16, 16, 4, 1, 4, true)
        GANCE_CASE(13, 64, 1, 8, 16, 4, 2, 2, true)
        default: return hipErrorInvalidValue;
    }
#undef GANCE_CASE
}

}  // namespace gance
