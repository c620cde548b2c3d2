// Conv0_up in ONE kernel: stride-2 transposed modulated 3x3 convolution on the fp32 matrix cores, its
// [1,3,3,1] x [1,3,3,1] FIR, noise, bias and leaky ReLU, for the layers whose input is >= 64 wide.
//
// Replaces, for the reference's synthesis call (gance/network_interface/network_functions.py:168), the
// un-vendored `upsample_conv_2d` (tf.nn.conv2d_transpose + upfirdn_2d.cu, pad 1/1, gain 4) followed by
// `fused_bias_act.cu`, and replaces this library's own two-pass form
// (conv_mfma.hip UP=true + aux_kernels.hip fir_epilogue_kernel) on those layers: the (2H+1)^2
// intermediate T never goes to HBM. The layer then reads its input once and writes its output once.
//
// Decomposition. T[2y'+py][2x'+px] (four parity classes of a position (y',x') of the INPUT grid, the
// 9 filter taps split 4/2/2/1 over them) is what the matrix cores produce, exactly as in
// conv_mfma.hip. out[oy][ox] = sum_{a,b} k[a] k[b] T[oy+a-1][ox+b-1] needs a halo of 3 T rows / columns
// around an output tile. A block therefore SWEEPS a column strip of 64 positions top to bottom in steps
// of 8 position rows:
//   * horizontally the two halo position columns (x' = X0-1: odd column parity only; x' = X0+64) are
//     recomputed: 16 positions = one extra tile of 16 slots whose four classes are split over the four
//     waves, on v_mfma_f32_16x16x4_f32 (two channel halves, four input channels per MFMA: {4,2,2,1} x 2
//     MFMAs of 32 cycles per two pairs of input channels beside the 72 of 64 cycles on the main tiles);
//   * vertically nothing is recomputed: the last three T rows of a step stay in LDS (32 channels x 3 rows
//     x 131 columns = 50 KB) and are the top of the next step's FIR window.
// One block = 32 output channels x (8 x 64 + 16) positions x 4 classes = 256 + 8 accumulator registers per
// lane: one wave per SIMD, one block per CU, like the Winograd kernel. A step = K loop (LDS-DMA ring of
// two slots, chunks of 8 input channels: weight image [9][8][32] + haloed patch [8][9][72]), then the
// epilogue in four passes of 8 channels: accumulators -> LDS T window [8][16][132], barrier, every
// thread filters a 16-row x 4-column output strip of one channel (two aligned ds_read_b128 per T row,
// horizontal taps carry demod * sqrt 2, vertical taps, + noise + bias, leaky ReLU) and stores float4s.
// The next step's first chunk is fetched under the epilogue; its arrival is waited for BEFORE the first
// store so that the K loop never waits behind the epilogue's stores (vmcnt counts in issue order).
//
// Layout contracts are those of conv_mfma.hip: zero-bordered activations [B][C][H+2][W+8], interior at
// [y+1][x+4]; borders are never written, which is what makes every out-of-image tap and the T cells
// outside [0, 2H]^2 come out as exact zeros without a bounds test.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

// Timing ablations (GANCE_DEBUG_UPFIR: 1 no stores, 2 no epilogue, 4 no MFMA, 8 no DMA after the first chunk, 16 no
// accumulator dump into the T window, 32 no FIR rows; Makefile target upfirdbg)
// exist only in a -DGANCE_UPFIR_DEBUG=1 build: their uniform branches cost scalar registers the product kernel
// does not have (it already spills some to VGPR lanes).
#ifndef GANCE_UPFIR_DEBUG
#define GANCE_UPFIR_DEBUG 0
#endif
#define UPFIR_DBG (GANCE_UPFIR_DEBUG ? p.debug_flags : 0)

namespace gance {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kTH = 8;                    // position rows per step
constexpr int kSW = 64;                   // position columns per strip
constexpr int kBM = 32;                   // output channels per block
constexpr int kKC = 8;                    // input channels per chunk
constexpr int kPH = kTH + 1;              // patch rows: input rows y0-1 .. y0+7
constexpr int kPW = kSW + 8;              // patch columns: input columns X0-4 .. X0+67
constexpr int kPlane = kPH * kPW;         // 648
constexpr int kWlFloats = 9 * kKC * kBM;  // 2304
constexpr int kWlPieces = 9;              // 1 KiB DMA pieces
constexpr int kWlRegion = kWlPieces * 256;
static_assert(kWlRegion == kWlFloats, "the weight image is a whole number of pieces");
constexpr int kPlFloats = kKC * kPlane;   // 5184
constexpr int kPlF4 = kPlFloats / 4;      // 1296
constexpr int kPlPieces = 21;             // the last one partial
constexpr int kPieces = kWlPieces + kPlPieces;        // 30
constexpr int kPiecesPerWave = (kPieces + 3) / 4;     // 8
constexpr int kSlot = kWlRegion + kPlFloats;  // 7488 floats
constexpr int kTW = 132;                      // T window row: T columns 2X0-1 .. 2X0+129 (+ pad)
constexpr int kCarryRows = 3;
constexpr int kCarryFloats = kBM * kCarryRows * kTW;  // 12672
constexpr int kStageCh = 8;
constexpr int kStageRows = 2 * kTH;                        // 16
constexpr int kStageFloats = kStageCh * kStageRows * kTW;  // 16896
constexpr float kSqrt2f = 1.4142135623730951f;

// transposed-conv tap tables, in the order the weights are stored (engine.hip kUpTapWeight):
// EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO (0,0); class = 2*py + px
__host__ __device__ constexpr int tap_cls(int t) { return t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3)); }
__host__ __device__ constexpr int tap_shift(int t) {
    return ((t == 2 || t == 3 || t == 5) ? 2 : 0) + ((t == 1 || t == 3 || t == 7) ? 1 : 0);
}
__host__ __device__ constexpr int shift_off(int s) { return ((s & 2) ? -kPW : 0) + ((s & 1) ? -1 : 0); }

// LDS hand-over between the waves of the block without __syncthreads: its fence would also drain the
// vector-memory counter, i.e. wait for every output store of the previous pass
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace

// LDS (floats): ring slot 0 | ring slot 1 = first part of the T window | rest of the T window | carry |
// style [Cin] | demod [32] | bias [32] | noise tile [16][128]. A step has an even number of chunks and
// starts in slot 0, so slot 1 is the idle one while the epilogue runs (slot 0 receives the next step's
// first chunk) and the T window [8][16][132] can lie over it.
constexpr int kStageOff = kSlot;
constexpr int kCarryOff = kStageOff + (kStageFloats > kSlot ? kStageFloats : kSlot);
constexpr int kConstOff = kCarryOff + kCarryFloats;
size_t upfir_lds_bytes(int cin) { return sizeof(float) * ((size_t)kConstOff + cin + 3 * kBM + kStageRows * 2 * kSW); }

// kPre: the input arrives ALREADY multiplied by this layer's style (the Winograd kernel that produced it folded
// s[b][ci] into its stores): the per-tap `weight fragment x style` multiply and the wait state hipcc puts between it
// and the MFMAs -- nine of each per pair of input channels, sitting between the MFMA groups -- cost 5 ... 7 % of the layer.
template <bool kPre>
__device__ __forceinline__ void upfir_fused_body(const UpFirArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ring0 = smem;
    float* const stage = smem + kStageOff;      // [8 ch][16 rows][132], over ring slot 1
    float* const carry = smem + kCarryOff;      // [32 ch][3 rows][132]
    float* const s_lds = smem + kConstOff;      // style [Cin]
    float* const d_lds = s_lds + p.Cin;         // demod [32]
    float* const b_lds = d_lds + kBM;           // bias [32]
    float* const sn_lds = b_lds + kBM;          // the next layer's style of these 32 channels (or 1): rides on the leaky ReLU
    float* const nz_lds = sn_lds + kBM;         // noise tile of the step [16][128] (raw: strength * sqrt 2 is applied where it is added)

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // ---- block -> (sample, channel tile, strip, row segment); blocks of one XCD take contiguous ids so
    // that the channel tiles of one strip (same input patch) and neighbouring strips share its L2 ----
    int id;
    {
        const int v = blockIdx.x, nwg = p.total_blocks;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    const int m_tile = id % p.m_tiles;
    id /= p.m_tiles;
    const int strip = id % p.strips;
    id /= p.strips;
    const int seg = id % p.segs;
    const int b = id / p.segs;
    const int m0 = m_tile * kBM;
    const int X0 = strip * kSW;
    const int H = p.H, W = p.W;
    const int Hp = H + 2, Wp = W + 8;
    const int y_begin = seg * p.rows_per_seg;
    const int y_end = min(H, y_begin + p.rows_per_seg);
    // steps: one priming step above a segment that does not start at the image top (fills the carried T
    // rows, emits nothing), the segment's own steps, and after the image's last rows one flush step on
    // position row y' = H (T row 2H) that emits output rows 2H-2 and 2H-1
    const int step_first = seg > 0 ? -1 : 0;
    const int step_main = (y_end - y_begin) / kTH;
    const int step_last = step_main + (y_end == H ? 1 : 0);  // exclusive
    const int nchunks = p.Cin / kKC;

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.w + (size_t)m_tile * nchunks * kWlFloats), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.x_b_stride), 0, 0x7fffffff, 0x00020000);

    // ---- LDS-DMA staging: pieces 0..8 = weight image, 9..29 = patch; wave w issues pieces w, w+4, ... ----
    int poff[kPiecesPerWave];  // per-lane source byte offsets of this wave's patch pieces for the step being staged (-1: none)
    auto stage_setup = [&](int y0) {
#pragma unroll
        for (int r = 0; r < kPiecesPerWave; ++r) {
            const int i = wave + 4 * r - kWlPieces;
            const int f = i * 64 + lane;
            poff[r] = -1;
            if (i >= 0 && f < kPlF4) {
                const int q = f % (kPW / 4);
                const int row = (f / (kPW / 4)) % kPH;
                const int c = f / (kPW / 4 * kPH);
                const int gy = min(y0 + row, Hp - 1);  // rows below the image (flush step) read the zero border
                poff[r] = ((c * Hp + gy) * Wp + X0 + 4 * q) * 4;
            }
        }
    };
    // one DMA piece (r = 0..7 of this wave: piece g = wave + 4 r) of `chunk` into ring slot `buf`. Which kind a piece is
    // is known at compile time except for r = 2 (g = 8 is the last weight piece, 9..11 are patch pieces) and r = 7
    // (g = 28, 29 exist, 30, 31 do not); only piece 29 is partial (16 of its 64 lanes).
    auto stage_piece = [&](auto rtag, int chunk, float* buf) {
        constexpr int r = decltype(rtag)::value;
        const int g = wave + 4 * r;
        const bool weights = r < 2 || (r == 2 && wave == 0);
        if (weights) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(buf + g * 256), 16, (g * 256 + lane * 4) * 4,
                                                     chunk * kWlFloats * 4, 0, 0);
        } else if (r < 7 || wave < 2) {
            if (r < 7 || poff[r] >= 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(buf + kWlRegion + (g - kWlPieces) * 256), 16, poff[r],
                                                         chunk * kKC * Hp * Wp * 4, 0, 0);
        }
    };
    // (r is a constant wherever this is called from an unrolled loop: the switch folds away)
    auto stage_piece_n = [&](int r, int chunk, float* buf) {
        switch (r) {
            case 0: stage_piece(std::integral_constant<int, 0>{}, chunk, buf); break;
            case 1: stage_piece(std::integral_constant<int, 1>{}, chunk, buf); break;
            case 2: stage_piece(std::integral_constant<int, 2>{}, chunk, buf); break;
            case 3: stage_piece(std::integral_constant<int, 3>{}, chunk, buf); break;
            case 4: stage_piece(std::integral_constant<int, 4>{}, chunk, buf); break;
            case 5: stage_piece(std::integral_constant<int, 5>{}, chunk, buf); break;
            case 6: stage_piece(std::integral_constant<int, 6>{}, chunk, buf); break;
            default: stage_piece(std::integral_constant<int, 7>{}, chunk, buf); break;
        }
    };
    auto stage_chunk = [&](int chunk, float* buf) {
        stage_piece(std::integral_constant<int, 0>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 1>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 2>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 3>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 4>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 5>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 6>{}, chunk, buf);
        stage_piece(std::integral_constant<int, 7>{}, chunk, buf);
    };

    // ---- staggered start: identical blocks would otherwise all reach their epilogues together and their
    // stores would arrive at HBM as one burst per step, with nothing in between. Phase q of Q waits q/Q of a
    // step (bounded: the realtime counter advances) ----
    if (p.stagger_phases > 1) {
        const unsigned long long wait = (unsigned long long)(((unsigned)blockIdx.x >> 3) % (unsigned)p.stagger_phases) * (unsigned)p.stagger_ticks;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }

    stage_setup(y_begin + kTH * step_first);
    stage_chunk(0, ring0);

    // ---- per-block constants and the zeroed carry ----
    for (int i = tid; i < p.Cin; i += 256) s_lds[i] = p.s[(size_t)b * p.s_stride + i];
    if (tid < kBM) {
        d_lds[tid] = p.d[(size_t)b * p.d_stride + m0 + tid];
        b_lds[tid] = p.bias[m0 + tid];
        sn_lds[tid] = p.s_next != nullptr ? p.s_next[(size_t)b * p.s_stride + m0 + tid] : 1.0f;
    }
    for (int i = tid; i < kCarryFloats; i += 256) carry[i] = 0.f;

    // ---- per-lane operand offsets (floats) ----
    // main tiles jj = 0..3 of wave w: position row ry = 2w + jj/2, columns cx = 32*(jj&1) + l31
    // halo tile: slot l31 & 15: ry = slot & 7, cx = -1 (slot < 8) or 64
    const int aoff = lh * kBM + l31;
    int boff[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
        boff[jj] = lh * kPlane + (2 * wave + (jj >> 1) + 1) * kPW + 32 * (jj & 1) + l31 + 4;
    // halo tile = 32 channels x 16 slots on v_mfma_f32_16x16x4_f32 (two channel halves, FOUR input channels per
    // MFMA): lane (slot = lane % 16, q4 = lane / 16) supplies input channel 4 j + q4 of the pair of channel pairs j
    const int hslot = l31 & 15;
    const int q4 = 2 * lh + (l31 >> 4);
    const int aoffh = q4 * kBM + hslot;  // + (t * kKC + 4 j) * kBM + 16 half
    const int boffh = q4 * kPlane + ((hslot & 7) + 1) * kPW + ((hslot >> 3) ? 64 : -1) + 4;  // + 4 j * kPlane + shift

    // ---- epilogue roles ----
    // dump: lane writes T[cl = rr + 4 lh][trow = 2 ry + py][sc = 2 cx + px + 1] of pass g = r >> 2, rr = r & 3
    const int dump_base = (4 * lh) * (kStageRows * kTW) + (4 * wave) * kTW + 2 * l31 + 1;
    const int hpy = wave >> 1, hpx = wave & 1;  // the halo tile's class held by this wave
    // (halo tile: result register rr of half h = channel 16 h + 4 q4 + rr: pass g takes half g / 2 from the lanes with
    // q4 / 2 == g % 2, i.e. lh == g % 2, as channel 4 (q4 % 2) + rr of the pass)
    const bool halo_writes = (hslot >> 3) == 1 || hpx == 1;
    const int dump_halo = (4 * (l31 >> 4)) * (kStageRows * kTW) + (2 * (hslot & 7) + hpy) * kTW + ((hslot >> 3) ? 129 + hpx : 0);
    // filter: thread = (channel c of the pass, column group cg): output columns 2 X0 + 4 cg .. + 3
    const int fc = 2 * wave + lh;
    const int cg = l31;
    const int OW = 2 * W, OWp = OW + 8;
    const long long oplane = (long long)(2 * H + 2) * OWp;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + ((size_t)b * p.Cout + m0) * oplane), 0, 0x7fffffff, 0x00020000);
    const int o_voff = (int)((fc * oplane + 4 * cg) * 4);
    const bool has_noise = p.noise != nullptr;
    // (the noise plane [2H][2W] as a bounded resource: a row outside it reads as zeros instead of faulting)
    const __amdgpu_buffer_rsrc_t nz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(has_noise ? p.noise + (size_t)b * p.noise_b_stride : nullptr), 0, (2 * H) * (2 * W) * 4, 0x00020000);
    const float ns2 = p.noise_strength * kSqrt2f;

    int ring = 0;
    bool landed = false;  // the chunk about to be consumed was already waited for (before the previous epilogue)
    // One step; the flush form (position row y' = H only) is a separate instantiation so that the two K loops
    // do not meet in one control-flow graph (hipcc then loses track of the 272 accumulator registers).
    auto run_step = [&](auto flush_tag, const int si) {
        constexpr bool kFlush = decltype(flush_tag)::value;
        const int y0 = y_begin + kTH * si;
        const int oy_noise = 2 * y0 - 2;  // output row of the epilogue's window row 0 (= oy0 there)
        f32x16 acc[4][4];
        f32x4 acch[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[c][jj][r] = 0.f;
        }

        for (int k = 0; k < nchunks; ++k) {
            if (!(k == 0 && landed)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (k == 0 && has_noise && si >= 0) {
                // The step's noise tile [16 rows][128 columns] by LDS-DMA, two 4 KB pieces per block: behind this barrier
                // every wave has left the previous epilogue (the tile's last reader), and the data lands under the first
                // chunk's MFMAs. (It used to be loaded into registers in the epilogue, a memory round trip in front of the
                // first store of every step.) Rows above / below the image are outside the resource: they read as zero.
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = oy_noise + 2 * wave + 8 * q + (lane >> 5);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(nz_rsrc, (lds_ptr_t)(nz_lds + 4 * (wave * 64 + 256 * q)), 16,
                                                             (row * (2 * W) + 2 * X0 + 4 * (lane & 31)) * 4, 0, 0, 0);
                }
            }
            float* const cur_buf = ring0 + ring * kSlot;
            float* const nxt_buf = ring0 + (ring ^ 1) * kSlot;
            // the next chunk of the stream (this step's k+1, or the first one of the next step): its eight DMA pieces
            // are issued one at a time BETWEEN the MFMA groups below, where their issue cost hides behind the matrix pipe
            int next_chunk = -1;
            if (UPFIR_DBG & 8) {
            } else if (k + 1 < nchunks) {
                next_chunk = k + 1;
            } else if (si + 1 < step_last) {
                stage_setup(y0 + kTH);
                next_chunk = 0;
            }
            ring ^= 1;
            const float* Wl = cur_buf + aoff;
            const float* Pl = cur_buf + kWlRegion;
            const float* sp = s_lds + k * kKC + lh;

            if (UPFIR_DBG & 4) {
                if (next_chunk >= 0) stage_chunk(next_chunk, nxt_buf);
                continue;
            }
            if constexpr (!kFlush) {
                // Per pair of input channels (kk): 9 taps in four class groups EE (4 taps) EO (2) OE (2) OO (1). A group's
                // taps each issue 4 MFMAs on the wave's main tiles; the wave that holds the group's class of the halo tile
                // then adds that tile's MFMAs for the whole group (one uniform branch per group, not per tap). The weight
                // fragment of the next tap and, at the last tap of a kk, the 20 patch fragments of the next kk are read from
                // LDS before the current tap's MFMAs are issued; the style scale is read once per kk.
                float bfrag[2][4][4];
                float bhalo[4];  // the halo tile's B operands of a pair of channel pairs, times the style scale
                float anext;
                float snext;
                auto load_b = [&](int kk) {
#pragma unroll
                    for (int sh = 0; sh < 4; ++sh) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bfrag[kk & 1][sh][jj] = Pl[boff[jj] + 2 * kk * kPlane + shift_off(sh)];
                    }
                };
                const float* const Wh = cur_buf + aoffh;
                const float* const sph = s_lds + k * kKC + q4;
                anext = Wl[0];
                snext = kPre ? 1.0f : sp[0];
                load_b(0);
                constexpr int kGroupFirst[4] = {0, 4, 6, 8}, kGroupTaps[4] = {4, 2, 2, 1};
#pragma unroll
                for (int kk = 0; kk < kKC / 2; ++kk) {
                    const float sv = snext;
                    if ((kk & 1) == 0) {
                        // the halo operands of channels 4 j .. 4 j + 3 (j = kk / 2), used after the main MFMAs of kk + 1
#pragma unroll
                        for (int sh = 0; sh < 4; ++sh) bhalo[sh] = Pl[boffh + 2 * kk * kPlane + shift_off(sh)];
                        if constexpr (!kPre) {
                            const float sh4 = sph[2 * kk];
#pragma unroll
                            for (int sh = 0; sh < 4; ++sh) bhalo[sh] *= sh4;
                        }
                    }
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        float a[4];
#pragma unroll
                        for (int i = 0; i < kGroupTaps[grp]; ++i) {
                            const int t = kGroupFirst[grp] + i;
                            const float araw = anext;
                            if (t + 1 < 9) {
                                anext = Wl[((t + 1) * kKC + 2 * kk) * kBM];
                            } else if (kk + 1 < kKC / 2) {
                                anext = Wl[(2 * (kk + 1)) * kBM];
                                if constexpr (!kPre) snext = sp[2 * (kk + 1)];
                                load_b(kk + 1);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            a[i] = kPre ? araw : araw * sv;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                acc[grp][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bfrag[kk & 1][tap_shift(t)][jj], acc[grp][jj], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            // one DMA piece of the next chunk behind every other tap's MFMAs (8 pieces over the first 16 taps)
                            if ((kk * 9 + t) % 2 == 1 && (kk * 9 + t) / 2 < kPiecesPerWave) {
                                if (next_chunk >= 0) stage_piece_n((kk * 9 + t) / 2, next_chunk, nxt_buf);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        // The halo tile (one class per wave): hipcc pins builtin MFMA accumulators to the 256 AGPRs, which the
                        // 16 main tiles fill, so it is issued in the VGPR form by hand -- as 16x16x4 MFMAs, once per PAIR of
                        // channel pairs: 16 slots is all the tile needs, and a tap costs 2 x 32 matrix cycles per four input
                        // channels instead of 2 x 64 (the wave with the four-tap class was 10 % behind its siblings).
                        if ((kk & 1) == 1 && wave == grp) {
                            float ah[4][2];
#pragma unroll
                            for (int i = 0; i < kGroupTaps[grp]; ++i)
#pragma unroll
                                for (int h = 0; h < 2; ++h) ah[i][h] = Wh[((kGroupFirst[grp] + i) * kKC + 2 * (kk - 1)) * kBM + 16 * h];
#pragma unroll
                            for (int i = 0; i < kGroupTaps[grp]; ++i)
#pragma unroll
                                for (int h = 0; h < 2; ++h)
                                    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                                                 : "+v"(acch[h])
                                                 : "v"(ah[i][h]), "v"(bhalo[tap_shift(kGroupFirst[grp] + i)]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
              if (next_chunk >= 0) stage_chunk(next_chunk, nxt_buf);
              if (wave < 2) {
                // flush step (si == step_main): only T row 2H exists below the image = position row y' = H, even row parity,
                // and only its taps on input row H-1 are non-zero: EE taps 2, 3 (wave 0: its two tiles of the
                // first position row and the halo tile) and EO tap 5 (wave 1: the halo tile; wave 0: its tiles)
#pragma unroll
                for (int kk = 0; kk < kKC / 2; ++kk) {
                    const float sv = kPre ? 1.0f : sp[2 * kk];
#pragma unroll
                    for (int t = 2; t <= 5; ++t) {
                        if (t == 4) continue;
                        const float a = Wl[(t * kKC + 2 * kk) * kBM] * sv;
                        if (wave == 0) {
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj)
                                acc[tap_cls(t)][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    a, Pl[boff[jj] + 2 * kk * kPlane + shift_off(tap_shift(t))], acc[tap_cls(t)][jj], 0, 0, 0);
                        }
                        if ((kk & 1) == 1 && wave == tap_cls(t)) {  // the halo tile, four input channels at a time
                            const float bh = Pl[boffh + 2 * (kk - 1) * kPlane + shift_off(tap_shift(t))] * (kPre ? 1.0f : s_lds[k * kKC + 2 * (kk - 1) + q4]);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const float ah = cur_buf[aoffh + (t * kKC + 2 * (kk - 1)) * kBM + 16 * h];
                                asm volatile("s_nop 4\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acch[h]) : "v"(ah), "v"(bh));
                            }
                        }
                    }
                }
              }
            }
        }
        // (the T window lies over ring slot 1, which the last chunk was read from)
        lds_barrier();

        // ---- epilogue ----
        if (UPFIR_DBG & 2) return;
        const bool emit = si >= 0;
        const int oy0 = 2 * y0 - 2;                  // output row of window row r = 0
        const int r_lo = max(0, -oy0);               // first image step: rows -2, -1 do not exist
        const int r_hi = min(16, 2 * H - oy0);       // flush step: only rows 2H-2, 2H-1
        // the next step's first chunk must have landed before the first store is issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed = true;

#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // -- dump: accumulator registers 4g .. 4g+3 = channels 8g + rr + 4 lh (g is unrolled: the
            // register indices are static and a pass's accumulators die with its dump) --
            if (!(UPFIR_DBG & 16))
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        stage[dump_base + rr * (kStageRows * kTW) + (2 * (jj >> 1) + (c >> 1)) * kTW + 64 * (jj & 1) + (c & 1)] =
                            acc[c][jj][4 * g + rr];
            if (halo_writes && lh == (g & 1)) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) stage[dump_halo + rr * (kStageRows * kTW)] = acch[g >> 1][rr];
            }
            lds_barrier();

            // -- filter: window row i = T row 2 y0 - 3 + i; rows 0..2 from the carry, 3..18 from this step --
            const int ch = 8 * g + fc;
            const float dsc = d_lds[ch] * kSqrt2f;
            const float kh0 = 0.25f * dsc, kh1 = 0.75f * dsc;
            const float bias2 = b_lds[ch] * kSqrt2f;
            const float lr6 = 0.6f * sn_lds[ch], lr4 = 0.4f * sn_lds[ch];  // leaky ReLU x the next layer's style
            const float* const carry_c = carry + ch * (kCarryRows * kTW) + 4 * cg;
            const float* const stage_c = stage + fc * (kStageRows * kTW) + 4 * cg;
            const int o_soff_base = (int)((8 * g * oplane + (long long)(oy0 + 1) * OWp + 2 * X0 + 4) * 4);
            f32x4 keep_a[3], keep_b[3];  // raw T rows 16..18 of the own columns: the next step's carry
            f32x4 h0, h1, h2, h3;        // ring of horizontally filtered rows
            // one window row: two aligned 16-byte LDS reads (+ one of the noise tile), 4 horizontal outputs (mul + 3 fma
            // each), and when a whole 4-row window ends here the vertical taps (4 fma), leaky ReLU as 0.6 v + 0.4 |v| (2
            // ops, no NaN canonicalisation as fmaxf would add) and one 16-byte store. Scalar on purpose: packed-f32 forms
            // need even-aligned register pairs and cost more moves than they save here. The reads of a trip's four rows
            // are issued together in front of its arithmetic: read where they are used, behind the row's own branch, each
            // row exposed its LDS latency to the only wave of the SIMD (19 times per pass).
            // (the noise read is unconditional: a branch around it would split the block the reads are gathered in)
            auto read_row = [&](const float* rowp, int r, f32x4& ta, f32x4& tb, f32x4& nz) {
                ta = *reinterpret_cast<const f32x4*>(rowp);
                tb = *reinterpret_cast<const f32x4*>(rowp + 4);
                nz = *reinterpret_cast<const f32x4*>(nz_lds + r * (2 * kSW) + 4 * cg);  // (r < 0: inside the constants in front of the tile, unused)
            };
            auto window_row = [&](const f32x4& ta, const f32x4& tb, const f32x4& nz, f32x4& hnew, const f32x4& ha, const f32x4& hb, const f32x4& hc, int r) {
                if (!emit) return;
                const float t[7] = {ta[0], ta[1], ta[2], ta[3], tb[0], tb[1], tb[2]};
#pragma unroll
                for (int o = 0; o < 4; ++o) hnew[o] = fmaf(kh0, t[o + 3], fmaf(kh1, t[o + 2], fmaf(kh1, t[o + 1], kh0 * t[o])));
                if (r >= r_lo && r < r_hi) {
                    f32x4 v;
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = fmaf(0.25f, hnew[o], fmaf(0.75f, hc[o], fmaf(0.75f, hb[o], fmaf(0.25f, ha[o], bias2))));
                    if (has_noise) v += ns2 * nz;
#pragma unroll
                    for (int o = 0; o < 4; ++o) v[o] = fmaf(lr6, v[o], lr4 * __builtin_fabsf(v[o]));
                    if (!(UPFIR_DBG & 1) || v[0] == 12345.f)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, o_voff, o_soff_base + r * OWp * 4, 0);
                }
            };
            // rows 0..15 in four trips of four (rolled: the compiler must not hoist all 38 LDS reads into registers),
            // rows 16..18 peeled (they are also the carry). Window row i closes output row r = i - 3.
#pragma unroll 1
            for (int i4 = (UPFIR_DBG & 32) ? 4 : 0; i4 < 4; ++i4) {
                const float* const st = stage_c + (4 * i4 - 3) * kTW;
                const float* const r0 = i4 == 0 ? carry_c : st;
                f32x4 ta0, tb0, nz0, ta1, tb1, nz1, ta2, tb2, nz2, ta3, tb3, nz3;
                read_row(r0, 4 * i4 - 3, ta0, tb0, nz0);
                read_row(r0 + kTW, 4 * i4 - 2, ta1, tb1, nz1);
                read_row(r0 + 2 * kTW, 4 * i4 - 1, ta2, tb2, nz2);
                read_row(st + 3 * kTW, 4 * i4, ta3, tb3, nz3);
                window_row(ta0, tb0, nz0, h0, h1, h2, h3, 4 * i4 - 3);
                window_row(ta1, tb1, nz1, h1, h2, h3, h0, 4 * i4 - 2);
                window_row(ta2, tb2, nz2, h2, h3, h0, h1, 4 * i4 - 1);
                window_row(ta3, tb3, nz3, h3, h0, h1, h2, 4 * i4);
            }
            {
                f32x4 nz0, nz1, nz2;
                read_row(stage_c + 13 * kTW, 13, keep_a[0], keep_b[0], nz0);
                read_row(stage_c + 14 * kTW, 14, keep_a[1], keep_b[1], nz1);
                read_row(stage_c + 15 * kTW, 15, keep_a[2], keep_b[2], nz2);
                window_row(keep_a[0], keep_b[0], nz0, h0, h1, h2, h3, 13);
                window_row(keep_a[1], keep_b[1], nz1, h1, h2, h3, h0, 14);
                window_row(keep_a[2], keep_b[2], nz2, h2, h3, h0, h1, 15);
            }
            lds_barrier();
            // -- the last three T rows of this step become the carry of these 8 channels --
            float* const carry_w = carry + ch * (kCarryRows * kTW) + 4 * cg;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                *reinterpret_cast<f32x4*>(carry_w + i * kTW) = keep_a[i];
                if (cg == 31) *reinterpret_cast<f32x4*>(carry_w + i * kTW + 4) = keep_b[i];
            }
        }
    };
#pragma unroll 1
    for (int si = step_first; si < step_main; ++si) run_step(std::false_type{}, si);
    if (step_last > step_main) run_step(std::true_type{}, step_main);
}

bool upfir_supported(int cin, int cout, int H, int W) {
    // (an even number of chunks per step: the T window lies over the ring slot that is idle after an even count)
    return H == W && W % kSW == 0 && H % kTH == 0 && cin % (2 * kKC) == 0 && cout % kBM == 0 && cin <= 512;
}

size_t upfir_weight_floats(int cin, int cout) { return (size_t)9 * cin * cout; }

// w_in: scaled filter [tap = wy*3+wx][cin][cout]; w_out: [m tile][chunk][slot][4][32], slot t = filter tap up_tap_weight[t]
void upfir_arrange_weights(const float* w_in, int cin, int cout, const int* up_tap_weight, float* w_out) {
    const int m_tiles = cout / kBM, chunks = cin / kKC;
    for (int mt = 0; mt < m_tiles; ++mt)
        for (int ch = 0; ch < chunks; ++ch)
            for (int t = 0; t < 9; ++t)
                for (int kc = 0; kc < kKC; ++kc)
                    for (int m = 0; m < kBM; ++m)
                        w_out[((((size_t)mt * chunks + ch) * 9 + t) * kKC + kc) * kBM + m] =
                            w_in[((size_t)up_tap_weight[t] * cin + ch * kKC + kc) * cout + mt * kBM + m];
}

// Row segments: as few as give every CU a block (a segment costs one extra priming step).
// (a->Cin must be set: the start stagger depends on the length of a step)
void upfir_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* a) {
    a->m_tiles = cout / kBM;
    a->strips = W / kSW;
    const int base = B * a->m_tiles * a->strips;
    const int steps = H / kTH;
    int segs = 1;
    while (base * segs < num_cus && segs * 2 <= steps && steps % (segs * 2) == 0) segs *= 2;
    a->segs = segs;
    a->rows_per_seg = H / segs;
    a->step_rows = kTH;
    a->total_blocks = base * segs;
    // Staggered starts (see the kernel): a step's K loop takes about 5.5 us per chunk of 8 input channels, its
    // stores (32 channels x 16 rows x 128 columns per block) take blocks x 256 KB / ~5 TB/s when every block
    // stores at once: 13 us of a 50 us step at Cin = 64, 3 % of one at Cin = 512. Four phases where that matters.
    static const int env_phases = [] { const char* v = std::getenv("GANCE_TUNE_UPFIR_PHASES"); return v ? std::atoi(v) : -1; }();
    static const int env_ticks = [] { const char* v = std::getenv("GANCE_TUNE_UPFIR_TICKS"); return v ? std::atoi(v) : -1; }();
    static const int env_debug = [] { const char* v = std::getenv("GANCE_DEBUG_UPFIR"); return v ? std::atoi(v) : 0; }();
    a->debug_flags = env_debug;
    const int cin = a->Cin;
    const double step_us = cin / kKC * 5.5 + 8.0;
    // (measured: no effect at 1024^2, 4.05 / 4.08 / 4.09 ms with 1 / 4 / 8 phases: the epilogue is bound by its own
    // instruction stream, not by the store burst. Off by default; the knob stays for experiments.)
    a->stagger_phases = env_phases >= 0 ? env_phases : 1;
    if (a->total_blocks < num_cus / 2) a->stagger_phases = 1;
    a->stagger_ticks = env_ticks >= 0 ? env_ticks : (int)(step_us * 100.0 / std::max(1, a->stagger_phases));
}

// (plain kernels around the templated body: see winograd64_conv.hip on kernel templates and the host pass)
__global__ __launch_bounds__(256, 1) void upfir_fused_kernel(const UpFirArgs p) { upfir_fused_body<false>(p); }
__global__ __launch_bounds__(256, 1) void upfir_fused_pre_kernel(const UpFirArgs p) { upfir_fused_body<true>(p); }

hipError_t launch_upfir_fused(const UpFirArgs& args, hipStream_t stream) {
    static PerDeviceInt ready;  // the dynamic-LDS opt-in is per device
    int unused = 0;
    const hipError_t e = ready.get(
        [&](int, int* value) {
            *value = 1;
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(upfir_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)upfir_lds_bytes(512));
            if (err != hipSuccess) return err;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(upfir_fused_pre_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)upfir_lds_bytes(512));
        },
        &unused);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(args.input_prescaled ? upfir_fused_pre_kernel : upfir_fused_kernel, dim3(args.total_blocks), dim3(256),
                       upfir_lds_bytes(args.Cin), stream, args);
    return hipGetLastError();
}

}  // namespace gance
