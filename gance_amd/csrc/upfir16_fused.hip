// Conv0_up in ONE kernel, second geometry (round 4): 16 output channels per block, TWO blocks per CU.
//
// Same layer, same decomposition and same layout contracts as upfir_fused.hip (read its header first): stride-2 transposed
// modulated 3x3 convolution as four parity classes on the fp32 matrix cores, [1,3,3,1] x [1,3,3,1] FIR, noise, bias, leaky
// ReLU, one launch, the (2H+1)^2 intermediate T never in HBM; a block sweeps a strip of 64 position columns top to bottom in
// steps of 8 position rows, the two halo position columns are one extra tile, the last three T rows of a step are carried in
// LDS. Replaces, for the reference's synthesis call (gance/network_interface/network_functions.py:168), the un-vendored
// `upsample_conv_2d` + `fused_bias_act` pair.
//
// What is different, and why. upfir_fused.hip holds 32 channels x 8 x 64 positions x 4 classes = 256 accumulator registers per
// lane: one wave per SIMD, one block per CU, and whatever a wave does besides MFMAs -- the FIR epilogue (a fifth of a step at
// 1024^2), the accumulator dump, barriers, LDS latencies -- leaves the matrix pipe idle (0.59 of the roof at 1024^2, flat for
// two rounds). Nothing inside one block can overlap them: the tile's results (262 KB per step) have nowhere to wait while the
// next K loop runs. Two INDEPENDENT blocks per CU can: while one filters and stores, the other multiplies. That needs half the
// registers and half the LDS per block:
//   * 16 channels per block on v_mfma_f32_16x16x4_f32: 8 position tiles of 16 x 4 classes = 32 accumulator tiles of 4
//     registers = 128 per lane; a wave owns position rows w and w + 4 of the step (so that each HALF of the step holds one
//     row of every wave). A = weights (lane: channel slot m = lane % 16, input channel k = lane / 16), B = patch (lane:
//     input channel k, position n = lane % 16). MFMA row m = 4 q + r holds channel 4 r + q of the block (the host permutes
//     the weight image): accumulator REGISTER r of every lane is then channel group r, and the epilogue pass of channel
//     group g dumps register g of all 64 lanes (lane quarter q = channel g * 4 + q) -- full lanes, no register shuffles.
//   * LDS 77 KB: ring of two slots (weights [9][8][16] padded to 5 KB + haloed patch [8][9][72] = 25 KB each), carry
//     [16][3][132], and the T window of an epilogue pass -- 4 channels x 8 T rows x 132 = 17 KB -- lying over the ring slot
//     that is idle after an even number of chunks. A step's epilogue is therefore EIGHT passes: two halves of four position
//     rows x four channel groups; thread (wave = channel of the pass, column group of 4, row group of 4 output rows) filters
//     a 7-row window (rows 0..2 of the upper row group come from the carry).
//   * the noise tile does not fit: the FIR threads load their noise rows from HBM, one pass ahead (the loads of pass p + 1
//     are issued before the stores of pass p: gfx9 has ONE vector-memory counter, a load behind a store waits for it).
//   * no block is special: nothing depends on which waves share a SIMD.
// The halo tile (16 slots x 16 channels) is one more 16x16x4 tile whose four classes are split over the four waves, as before.
//
// Three strip geometries (template Geo16<CT, RW>: CT column tiles of 16 per position row, RW position rows per wave, a step =
// 4 RW rows): <4, 2> = strips of 64 columns, steps of 8 rows (inputs >= 64 wide, the layers upfir_fused.hip also takes);
// <2, 4> = 32 columns x 16 rows (the 32 -> 64 layer); <1, 4> = 16 columns x 16 rows, four tiles per wave (the 16 -> 32
// layer): the layers that ran as two passes (transposed conv into T planes in HBM + FIR pass) until round 4. An epilogue pass
// always covers four position rows (one per wave) x four channels: 4 RW passes per step.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

// Timing ablations (GANCE_DEBUG_UPFIR: 1 no stores, 2 no epilogue, 4 no MFMA, 8 no DMA after the first chunk, 16 no
// accumulator dump into the T window, 32 no FIR rows, 64 no barrier per chunk, 128 no operand reads in the K loop, 256 non-temporal
// output stores) exist only in a -DGANCE_UPFIR16_DEBUG=1 build (Makefile target
// upfir16dbg): wrong results, and their uniform branches cost scalar registers.
#ifndef GANCE_UPFIR16_DEBUG
#define GANCE_UPFIR16_DEBUG 0
#endif
#define UPFIR16_DBG (GANCE_UPFIR16_DEBUG ? p.debug_flags : 0)

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kBM = 16;            // output channels per block
constexpr int kKC = 8;             // input channels per chunk (two k-steps of the 16x16x4 MFMA)
constexpr int kCarryRows = 3;
constexpr int kPassCh = 4;         // channels per epilogue pass
constexpr int kPassRows = 8;       // T rows per pass: four position rows
constexpr float kSqrt2f = 1.4142135623730951f;

// WX: the two-tap classes along x (EE, OE) in Winograd F(2,2) form over PAIRS of position columns (see the header): an N tile of
// the MFMA is 16 pairs = 32 columns, 15 MFMAs per pair tile and k-step instead of 18, 10 accumulator tiles per pair tile instead of 8.
template <int CT_, int RW_, bool WX_ = false>
struct Geo16 {
    static constexpr int CT = CT_, RW = RW_;
    static constexpr bool WX = WX_;
    static constexpr int kWlSlots = WX ? 12 : 9;   // weight fragments per (input channel, channel): the nine taps, or EE (h0, h0+h1, h1) x 2 row taps, EO x 2, OE (h0, h0+h1, h1), OO
    static constexpr int kWlPieces = WX ? 6 : 5;   // weight image [slots][8][16] = 1152 / 1536 floats in 1 KiB DMA pieces (1152 padded to 1280 in HBM and in LDS)
    static constexpr int kWlRegion = kWlPieces * 256;
    static constexpr int kAcc = WX ? 10 * (CT / 2) * RW : 4 * CT * RW;  // accumulator tiles per wave: 40 / 32 / 32 / 16
    static constexpr int kSW = 16 * CT;           // position columns per strip
    static constexpr int kTH = 4 * RW;            // position rows per step
    static constexpr int kTiles = CT * RW;        // accumulator tiles per wave and class: 8, 8, 4
    static constexpr int kGroups = kTiles / 4;    // MFMA groups (4 tiles x 9 taps) per k-step
    static constexpr int kPH = kTH + 1;           // patch rows: input rows y0-1 .. y0+kTH-1
    static constexpr int kPW = kSW + 8;           // patch columns: input columns X0-4 .. X0+kSW+3
    static constexpr int kPlane = kPH * kPW;
    static constexpr int kPlFloats = kKC * kPlane;
    static constexpr int kPlF4 = kPlFloats / 4;
    static constexpr int kPlPieces = (kPlF4 + 63) / 64;   // 21 / 22 / 13 (the last one partial)
    static constexpr int kPieces = kWlPieces + kPlPieces;  // 26 / 27 / 18
    static constexpr int kPiecesPerWave = (kPieces + 3) / 4;  // 7 / 7 / 5
    static constexpr int kSlot = kWlRegion + kPlFloats;
    static constexpr int kTW = 2 * kSW + 4;       // T window row: T columns 2X0-1 .. 2X0+2kSW+1 (+ pad)
    static constexpr int kCarryFloats = kBM * kCarryRows * kTW;
    static constexpr int kStageFloats = kPassCh * kPassRows * kTW;
    static constexpr int kHaloTiles = (2 * kTH + 15) / 16;  // halo slots: kTH rows x 2 sides
    static constexpr int kCG = kSW / 2;           // column groups of 4 output columns
    static constexpr int kRG = 64 / kCG;          // row groups of a wave's 64 filter threads: 2 / 4 / 8
    static constexpr int kFR = kPassRows / kRG;   // output rows per filter thread: 4 / 2 / 1
    static constexpr int kWin = kFR + 3;          // its window rows
    static constexpr int kNzPieces = RW * kPassRows * 2 * kSW / 256;  // noise of a step's output rows in 1 KiB DMA pieces: 8 / 8 / 4
    static_assert(kNzPieces % 4 == 0 && kStageFloats + kNzPieces * 256 <= kSlot, "the step's noise lies behind the T window in ring slot 1");
    static_assert(kTiles % 4 == 0 && kStageFloats <= kSlot && kPiecesPerWave <= 4 * 2 * kGroups && (!WX || CT % 2 == 0), "geometry");
    // LDS (floats): ring slot 0 | ring slot 1 = T window of a pass | carry | style [Cin] | demod [16] | bias [16] | next style [16]
    static constexpr int kStageOff = kSlot, kCarryOff = 2 * kSlot, kConstOff = kCarryOff + kCarryFloats;
    static constexpr size_t lds_bytes(int cin) { return sizeof(float) * ((size_t)kConstOff + cin + 3 * kBM); }
};

// transposed-conv tap tables, in the order the weights are stored (engine.hip kUpTapWeight):
// EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO (0,0); class = 2*py + px
__host__ __device__ constexpr int tap_cls(int t) { return t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3)); }
// shift of a tap: bit 1 = input row above (dy = -1), bit 0 = input column to the left (dx = -1)
__host__ __device__ constexpr int tap_shift(int t) {
    return ((t == 2 || t == 3 || t == 5) ? 2 : 0) + ((t == 1 || t == 3 || t == 7) ? 1 : 0);
}

// LDS hand-over between the waves of the block without __syncthreads: its fence would also drain the
// vector-memory counter, i.e. wait for every output store of the previous pass
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The lane id, computed where it is used: per-lane values derived once at kernel entry would stay live across the K loop, where
// every register is taken (128 accumulators + two sets of operand fragments); hipcc then spills INSIDE the loop, and a scratch
// reload there is a vector-memory load whose wait drains the LDS-DMA ring. The epilogue re-derives its per-lane values from this.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

}  // namespace

// kNoise: the layer adds noise (a compile-time form: as a run-time flag the four selects per output row stayed in the no-noise path)
template <typename G, bool kPre, bool kNoise>
__device__ __forceinline__ void upfir16_body(const UpFirArgs& p) {
    constexpr int CT = G::CT, RW = G::RW, kSW = G::kSW, kTH = G::kTH, kPH = G::kPH, kPW = G::kPW, kPlane = G::kPlane, kPlF4 = G::kPlF4;
    constexpr int kPieces = G::kPieces, kPiecesPerWave = G::kPiecesPerWave, kSlot = G::kSlot, kTW = G::kTW, kCarryFloats = G::kCarryFloats;
    constexpr int kGroups = G::kGroups, kTiles = G::kTiles, kHaloTiles = G::kHaloTiles, kCG = G::kCG, kFR = G::kFR, kWin = G::kWin;
    constexpr int kWlPieces = G::kWlPieces, kWlRegion = G::kWlRegion;
    constexpr bool WX = G::WX;
    constexpr int kPT = CT / 2 > 0 ? CT / 2 : 1;  // pair tiles (16 pairs = 32 position columns) per row of the strip
    static_assert(!WX || kPre, "the pair form takes its input pre-scaled (no room for the style vector beside two blocks' LDS)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ring0 = smem;
    float* const stage = smem + G::kStageOff;   // [4 ch][8 rows][kTW], over ring slot 1
    float* const nz_lds = stage + G::kStageFloats;  // [8 RW output rows][2 kSW]: the step's noise, behind the T window in ring slot 1
    constexpr int kNzPieces = G::kNzPieces;
    float* const carry = smem + G::kCarryOff;   // [16 ch][3 rows][kTW]
    float* const s_lds = smem + G::kConstOff;   // style [Cin]
    float* const d_lds = s_lds + (kPre ? 0 : p.Cin);  // demod [16] (the pre-scaled form keeps no style vector)
    float* const b_lds = d_lds + kBM;           // bias [16]
    float* const sn_lds = b_lds + kBM;          // the next layer's style of these 16 channels (or 1): rides on the leaky ReLU

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n16 = lane & 15;  // MFMA: position of a tile (B operand, accumulator column) / channel slot (A operand)
    const int q4 = lane >> 4;   // MFMA: input channel of a k-step (operands) / channel quarter (accumulator rows 4 q .. 4 q + 3)

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
        (void*)(p.w + (size_t)m_tile * nchunks * kWlRegion), 0, 0x7fffffff, 0x00020000);
    // (bounded by the sample's tensor: the flush step stages patch rows below the buffer's last row -- rows it never reads
    // from LDS; inside the tensor they are the next channel's first rows, behind its end they read as zeros instead of faulting)
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.x_b_stride), 0, p.Cin * Hp * Wp * 4, 0x00020000);

    // ---- LDS-DMA staging: pieces 0..4 = weight image, 5.. = patch; wave w issues pieces w, w+4, ... ----
    // Per-lane source byte offsets of this wave's patch pieces at patch row 0 = buffer row 0 (-1: none): constants of the
    // kernel. The step being staged only moves the SCALAR offset (y_stage buffer rows): recomputing the offsets per step put
    // their integer divisions -- hoisted by hipcc to the top of every chunk, operands reloaded from scratch -- into the K loop.
    int poff[kPiecesPerWave];
#pragma unroll
    for (int r = 0; r < kPiecesPerWave; ++r) {
        const int i = wave + 4 * r - kWlPieces;
        const int f = i * 64 + lane;
        poff[r] = -1;
        if (i >= 0 && f < kPlF4) {
            const int q = f % (kPW / 4);
            const int row = (f / (kPW / 4)) % kPH;
            const int c = f / (kPW / 4 * kPH);
            poff[r] = ((c * Hp + row) * Wp + X0 + 4 * q) * 4;
        }
    }
    int y_stage = 0;  // first patch row of the step being staged, as a buffer row (= image row y0 - 1, + 1 for the border)
    auto stage_setup = [&](int y0) { y_stage = y0; };
    // one DMA piece (r of this wave: piece g = wave + 4 r) of `chunk` into ring slot `buf`. Which kind a piece is is known at
    // compile time except for r = 1 (g = 4 is the last weight piece, 5..7 are patch pieces) and the last r (pieces that do not
    // exist, and the partial last piece: poff = -1 masks its idle lanes)
    auto stage_piece = [&](auto rtag, int chunk, float* buf) {
        constexpr int r = decltype(rtag)::value;
        const int g = wave + 4 * r;
        const bool weights = g < kWlPieces;  // (r = 0: every wave; r = 1: wave 0, and wave 1 too in the pair form's six pieces)
        if (weights) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(buf + g * 256), 16, (g * 256 + lane * 4) * 4,
                                                     chunk * kWlRegion * 4, 0, 0);
        } else if (r < kPiecesPerWave - 1 || g < kPieces) {
            if (r < kPiecesPerWave - 1 || poff[r] >= 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr_t)(buf + kWlRegion + (g - kWlPieces) * 256), 16, poff[r],
                                                         (chunk * kKC * Hp + y_stage) * Wp * 4, 0, 0);
        }
    };
    // (r is a constant wherever this is called from an unrolled loop: the switch folds away)
    auto stage_piece_n = [&](int r, int chunk, float* buf) {
        switch (r) {
            case 0: stage_piece(std::integral_constant<int, 0>{}, chunk, buf); break;
            case 1: stage_piece(std::integral_constant<int, 1>{}, chunk, buf); break;
            case 2: stage_piece(std::integral_constant<int, 2>{}, chunk, buf); break;
            case 3: stage_piece(std::integral_constant<int, 3>{}, chunk, buf); break;
            case 4: stage_piece(std::integral_constant<int, (4 < kPiecesPerWave ? 4 : kPiecesPerWave - 1)>{}, chunk, buf); break;
            case 5: stage_piece(std::integral_constant<int, (5 < kPiecesPerWave ? 5 : kPiecesPerWave - 1)>{}, chunk, buf); break;
            default: stage_piece(std::integral_constant<int, kPiecesPerWave - 1>{}, chunk, buf); break;
        }
    };
    auto stage_chunk = [&](int chunk, float* buf) {
#pragma unroll
        for (int r = 0; r < kPiecesPerWave; ++r) stage_piece_n(r, chunk, buf);
    };

    // Staggered start (experiment, off by default: GANCE_TUNE_UPFIR16_STAGGER_US): the two blocks of a CU start together and do
    // the same work, so they reach their epilogues -- and their flush steps, which are bound by DMA latency -- together. Every
    // second wave of 256 blocks (the second resident block of each CU in the launch's first round) waits `stagger_ticks` x 10 ns.
    if (p.stagger_ticks > 0 && ((blockIdx.x >> 8) & 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
    }
    stage_setup(y_begin + kTH * step_first);
    stage_chunk(0, ring0);

    // ---- per-block constants and the zeroed carry ----
    if constexpr (!kPre)
        for (int i = tid; i < p.Cin; i += 256) s_lds[i] = p.s[(size_t)b * p.s_stride + i];
    if (tid < kBM) {
        d_lds[tid] = p.d[(size_t)b * p.d_stride + m0 + tid];
        b_lds[tid] = p.bias[m0 + tid];
        sn_lds[tid] = p.s_next != nullptr ? p.s_next[(size_t)b * p.s_stride + m0 + tid] : 1.0f;
    }
    for (int i = tid; i < kCarryFloats; i += 256) carry[i] = 0.f;

    // ---- per-lane operand offsets (floats; lds_*: byte addresses in LDS of ring slot 0's fragments, shifted up / left by one patch
    // row and column so that every tap shift is a non-negative immediate) ----
    // A (weights): [tap][ci 0..7][channel slot 0..15]; lane (m = n16, k = q4) reads slot m of input channel 4 j + k
    const int aoff = q4 * kBM + n16;
    // B (patch) of main tile (row rw, column tile ct): position row w + 4 rw, columns 16 ct + n; interior at column + 4
    const int boff = q4 * kPlane + (wave + 1) * kPW + n16 + 4;  // + 4 j kPlane + 4 rw kPW - dy kPW + 16 ct - dx
    // halo tile ht, slot n16: halo slot 16 ht + n16 = (side, position row): slot % kTH, column -1 (side 0) or kSW
    int boffh[kHaloTiles];
#pragma unroll
    for (int ht = 0; ht < kHaloTiles; ++ht) {
        const int slot = 16 * ht + n16;
        boffh[ht] = q4 * kPlane + (slot % kTH + 1) * kPW + ((slot / kTH) ? kSW : -1) + 4;
    }
    const unsigned lds_0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
    const unsigned lds_a = lds_0 + aoff * 4;
    // (pair form: lane n16 is PAIR n16 of its pair tile: its three columns are 2 n16 - 1, 2 n16, 2 n16 + 1 = immediates 0, 1, 2 from here)
    const unsigned lds_b = lds_0 + (kWlRegion + boff + (WX ? n16 : 0) - kPW - 1) * 4;
    unsigned lds_h[kHaloTiles];
#pragma unroll
    for (int ht = 0; ht < kHaloTiles; ++ht) lds_h[ht] = lds_0 + (kWlRegion + boffh[ht] - kPW - 1) * 4;

    const int OW = 2 * W, OWp = OW + 8;
    const long long oplane = (long long)(2 * H + 2) * OWp;
    // (the resource starts TWO ROWS ABOVE the block's first channel plane: a pass's first window row lies two rows above the image
    // in the image's first step, and both parts of a store's offset -- per-lane rows in the vector offset, the row of the unrolled
    // loop in the scalar one -- must stay non-negative: a negative offset is a 4 GB jump, not a subtraction. Rows above the
    // image are never stored.)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + ((size_t)b * p.Cout + m0) * oplane - 2 * OWp), 0, 0x7fffffff, 0x00020000);
    constexpr bool has_noise = kNoise;
    // (the sample's noise plane [2H][2W] as a bounded resource: a row outside it reads as zeros instead of faulting)
    const __amdgpu_buffer_rsrc_t nz_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_noise ? p.noise + (size_t)b * p.noise_b_stride : nullptr), 0, has_noise ? (2 * H) * (2 * W) * 4 : 0, 0x00020000);
    const float ns2 = p.noise_strength * kSqrt2f;

    int ring = 0;
    bool landed = false;  // the chunk about to be consumed was already waited for (before the previous epilogue)
    // One step; the flush form (position row y' = H only) is a separate instantiation so that the two K loops
    // do not meet in one control-flow graph.
    auto run_step = [&](auto flush_tag, const int si) {
        constexpr bool kFlush = decltype(flush_tag)::value;
        const int y0 = y_begin + kTH * si;
        // direct form: [class][tile: row rw * CT + column tile ct]. Pair form: accw[pair tile: row rw * 2 + pt][product]: 0..2 = EE (m1, m2,
        // m3), 3, 4 = EO at x = 2 n, 2 n + 1, 5..7 = OE (m1, m2, m3), 8, 9 = OO at 2 n, 2 n + 1 (acc is then unused and folds away)
        f32x4 acc[4][kTiles];
        f32x4 accw[WX ? kPT * RW : 1][10];
        if constexpr (WX) {
#pragma unroll
            for (int t4 = 0; t4 < kPT * RW; ++t4)
#pragma unroll
                for (int c = 0; c < 10; ++c) accw[t4][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acch[kHaloTiles];
#pragma unroll
        for (int ht = 0; ht < kHaloTiles; ++ht) acch[ht] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!WX) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int tl = 0; tl < kTiles; ++tl) acc[c][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        for (int k = 0; k < nchunks; ++k) {
            if (!(k == 0 && landed)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(UPFIR16_DBG & 64)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float* const cur_buf = ring0 + ring * kSlot;
            float* const nxt_buf = ring0 + (ring ^ 1) * kSlot;
            const unsigned ring_bytes = ring * (kSlot * 4);
            // the next chunk of the stream (this step's k+1, or the first one of the next step): its DMA pieces
            // are issued one at a time BETWEEN the MFMA groups below
            int next_chunk = -1;
            if (UPFIR16_DBG & 8) {
            } else if (k + 1 < nchunks) {
                next_chunk = k + 1;
            } else if (si + 1 < step_last) {
                stage_setup(y0 + kTH);
                next_chunk = 0;
            }
            ring ^= 1;

            if (UPFIR16_DBG & 4) {
                if (next_chunk >= 0) stage_chunk(next_chunk, nxt_buf);
                continue;
            }
            if constexpr (!kFlush && WX) {
                // Pair form. A chunk = two k-steps x two groups of two pair tiles (position row w + 4 rw, pair tiles pt = 0, 1: 64 columns)
                // = 30 MFMAs each. Fragments of a pair tile and row tap dy: e0 - e1, e1, e1 - e2, e2 with (e0, e1, e2) = input columns
                // (2 n - 1, 2 n, 2 n + 1): three reads and two subtractions; weights: twelve fragments per k-step (slot order: Geo16).
                // Products per pair tile: EE m1..m3 += (h0, h0 + h1, h1)[dy] x (e0 - e1, e1, e1 - e2)[dy] for both row taps, EO += t[dy] x
                // (e1, e2)[dy], OE m1..m3 += (h0, h0 + h1, h1) x (...)[0], OO += t8 x (e1, e2)[0]: 15 MFMAs instead of 18.
                constexpr int kNG = 2 * RW;  // groups per chunk: (k-step, row of the wave)
                // (one set of weight fragments: 160 accumulators leave no room for two -- the second k-step's twelve are read behind the
                // first one's last MFMAs and waited for there; the other block's wave covers the gap)
                float a[1][12], raw[2][kPT][2][3], bh[kHaloTiles][4];
                const unsigned a_lb = lds_a + ring_bytes, b_lb = lds_b + ring_bytes;
                auto load_a = [&](int jj, float(&dst)[12]) {
#pragma unroll
                    for (int t = 0; t < 12; ++t) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[t]) : "v"(a_lb), "i"((t * kKC + 4 * jj) * kBM * 4));
                };
                auto load_raw = [&](int grp, float(&dst)[kPT][2][3]) {  // group grp = (k-step grp / RW, row grp % RW): [pair tile][dy][column]
#pragma unroll
                    for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                asm volatile("ds_read_b32 %0, %1 offset:%2"
                                             : "=v"(dst[pt][dy][e])
                                             : "v"(b_lb), "i"((4 * (grp / RW) * kPlane + (4 * (grp % RW) - dy + 1) * kPW + 32 * pt + e) * 4));
                };
                auto load_halo = [&](int jj) {
#pragma unroll
                    for (int ht = 0; ht < kHaloTiles; ++ht) {
                        const unsigned h_lb = lds_h[ht] + ring_bytes;
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bh[ht][2 * dy + dx]) : "v"(h_lb), "i"((4 * jj * kPlane + (1 - dy) * kPW + 1 - dx) * 4));
                    }
                };
                auto land_raw = [&](float(&x)[kPT][2][3]) {
                    if constexpr (kPT == 2)
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(x[0][0][0]), "+v"(x[0][0][1]), "+v"(x[0][0][2]), "+v"(x[0][1][0]), "+v"(x[0][1][1]), "+v"(x[0][1][2]),
                                       "+v"(x[kPT - 1][0][0]), "+v"(x[kPT - 1][0][1]), "+v"(x[kPT - 1][0][2]), "+v"(x[kPT - 1][1][0]), "+v"(x[kPT - 1][1][1]),
                                       "+v"(x[kPT - 1][1][2]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0][0]), "+v"(x[0][0][1]), "+v"(x[0][0][2]), "+v"(x[0][1][0]), "+v"(x[0][1][1]), "+v"(x[0][1][2]));
                };
                auto land_a = [&](float(&x)[12]) {
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
                };
                auto land_halo = [&]() {
#pragma unroll
                    for (int ht = 0; ht < kHaloTiles; ++ht) asm volatile("" : "+v"(bh[ht][0]), "+v"(bh[ht][1]), "+v"(bh[ht][2]), "+v"(bh[ht][3]));
                };
                // direct tap t of the halo tile from the slot order (t0 = s2, t1 = s0, t2 = s5, t3 = s3, t4 = s6, t5 = s7, t6 = s10, t7 = s8, t8 = s11)
                constexpr int kTapSlot[9] = {2, 0, 5, 3, 6, 7, 10, 8, 11};
                load_a(0, a[0]);
                load_raw(0, raw[0]);
                load_halo(0);
                land_raw(raw[0]);
                land_a(a[0]);
                land_halo();
#pragma unroll
                for (int grp = 0; grp < kNG; ++grp) {
                    const int jj = grp / RW, rw = grp % RW;
                    if (grp + 1 < kNG) load_raw(grp + 1, raw[(grp + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // the group's fragments: two subtractions per pair tile and row tap
                    float bw[kPT][2][4];
#pragma unroll
                    for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) {
                            const float e0 = raw[grp & 1][pt][dy][0], e1 = raw[grp & 1][pt][dy][1], e2 = raw[grp & 1][pt][dy][2];
                            bw[pt][dy][0] = e0 - e1;
                            bw[pt][dy][1] = e1;
                            bw[pt][dy][2] = e1 - e2;
                            bw[pt][dy][3] = e2;
                        }
                    // (the MFMAs below are inline assembly: hipcc knows no hazard of theirs. The eight differences are tied to a two-cycle
                    // wait here -- a vector write needs two wait states before an MFMA reads it -- and every MFMA reads them after it.)
                    if constexpr (kPT == 2)
                        asm volatile("s_nop 1"
                                     : "+v"(bw[0][0][0]), "+v"(bw[0][0][2]), "+v"(bw[0][1][0]), "+v"(bw[0][1][2]), "+v"(bw[kPT - 1][0][0]), "+v"(bw[kPT - 1][0][2]),
                                       "+v"(bw[kPT - 1][1][0]), "+v"(bw[kPT - 1][1][2]));
                    else
                        asm volatile("s_nop 1" : "+v"(bw[0][0][0]), "+v"(bw[0][0][2]), "+v"(bw[0][1][0]), "+v"(bw[0][1][2]));
                    // 15 MFMAs per pair tile as (weight slot, fragment, product): ordered so that the two row taps of a product are far apart
                    constexpr int kOps[15][4] = {  // {slot, dy, fragment, product}
                        {0, 0, 0, 0}, {1, 0, 1, 1}, {2, 0, 2, 2}, {6, 0, 1, 3}, {6, 0, 3, 4}, {8, 0, 0, 5}, {9, 0, 1, 6}, {10, 0, 2, 7},
                        {11, 0, 1, 8}, {11, 0, 3, 9}, {3, 1, 0, 0}, {4, 1, 1, 1}, {5, 1, 2, 2}, {7, 1, 1, 3}, {7, 1, 3, 4}};
#pragma unroll
                    for (int op = 0; op < 15; ++op) {
#pragma unroll
                        // (in the accumulate-in-place form by hand: with 160 accumulators hipcc's builtin took the form whose result is
                        // another register tuple than its addend and rotated the accumulators through spare tuples it did not have: spills.
                        // Products of one accumulator are ten MFMAs apart: no dependent pair back to back.)
                        for (int pt = 0; pt < kPT; ++pt)
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                                         : "+v"(accw[kPT * rw + pt][kOps[op][3]])
                                         : "v"(a[0][kOps[op][0]]), "v"(bw[pt][kOps[op][1]][kOps[op][2]]));
                        // the halo tiles (direct form, one class per wave) with the k-step's first group: nine taps beside the first nine products
                        if (rw == 0 && op < 9 && wave == tap_cls(op)) {
#pragma unroll
                            for (int ht = 0; ht < kHaloTiles; ++ht)
                                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acch[ht]) : "v"(a[0][kTapSlot[op]]), "v"(bh[ht][tap_shift(op)]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // one DMA piece of the next chunk behind every other product of the chunk's first groups
                        if ((op & 1) == 1 && grp * 7 + (op >> 1) < kPiecesPerWave) {
                            if (next_chunk >= 0) stage_piece_n(grp * 7 + (op >> 1), next_chunk, nxt_buf);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (grp + 1 < kNG && (grp + 1) % RW == 0) {
                        load_a(1, a[0]);
                        load_halo(1);
                    }
                    if (grp + 1 < kNG) {
                        land_raw(raw[(grp + 1) & 1]);
                        if ((grp + 1) % RW == 0) {
                            land_a(a[0]);
                            land_halo();
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    (void)jj;
                }
                // (the last MFMAs' results are read by vector instructions in the epilogue, or by the next chunk's MFMAs: a 16x16x4 MFMA
                // takes 8 passes; the chunk loop's wait + barrier are far longer, the epilogue's first barrier is not guaranteed to be)
                if (k + 1 == nchunks) asm volatile("s_nop 15\n\ts_nop 15");
            } else if constexpr (!kFlush) {
                // A chunk = two k-steps (input channels 4 j .. 4 j + 3) x kGroups groups of four tiles = 36 MFMAs each. The operands
                // of group i + 1 -- its tiles' patch fragments at the four tap shifts and, per k-step, the nine weight fragments and
                // the halo tiles' fragments -- are read from LDS BEFORE the MFMAs of group i are issued: a wave issues in order, so
                // the reads' latency passes under its own 1 152 matrix cycles (the other block's wave fills what is left; alone on
                // the SIMD, while that one filters, this wave must not stall at every group).
                // The reads are single ds_read_b32 from a few base registers per chunk (weights, patch, halo slots) with the
                // fragment's place as the instruction's 16-bit immediate, in inline assembly: left to hipcc, every pair of
                // neighbouring patch values became one ds_read2_b32 (8-bit offsets) behind its own v_add_u32 for the base. The
                // compiler does not know these loads are in flight: every group ends in an explicit wait that ties the registers.
                constexpr int kNG = 2 * kGroups;  // groups per chunk
                float a[2][9], bf[2][4][4], bh[kHaloTiles][4];
                const unsigned a_lb = lds_a + ring_bytes, b_lb = lds_b + ring_bytes;
                auto load_a = [&](int jj, float(&dst)[9]) {
                    if (UPFIR16_DBG & 128) return;
#pragma unroll
                    for (int t = 0; t < 9; ++t) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[t]) : "v"(a_lb), "i"((t * kKC + 4 * jj) * kBM * 4));
                };
                auto load_b = [&](int grp, float(&dst)[4][4]) {  // group grp = (k-step grp / kGroups, tiles 4 (grp % kGroups) .. + 3)
                    if (UPFIR16_DBG & 128) return;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const int tl = 4 * (grp % kGroups) + i;
                                asm volatile("ds_read_b32 %0, %1 offset:%2"
                                             : "=v"(dst[2 * dy + dx][i])
                                             : "v"(b_lb), "i"((4 * (grp / kGroups) * kPlane + (4 * (tl / CT) - dy + 1) * kPW + 16 * (tl % CT) + 1 - dx) * 4));
                            }
                };
                auto load_halo = [&](int jj) {
#pragma unroll
                    for (int ht = 0; ht < kHaloTiles; ++ht) {
                        const unsigned h_lb = lds_h[ht] + ring_bytes;
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bh[ht][2 * dy + dx]) : "v"(h_lb), "i"((4 * jj * kPlane + (1 - dy) * kPW + 1 - dx) * 4));
                    }
                };
                // wait for every LDS read in flight and tie the registers they fill to this point
                auto land_b = [&](float(&x)[4][4]) {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[0][2]), "+v"(x[0][3]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[1][2]), "+v"(x[1][3]),
                                   "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[2][2]), "+v"(x[2][3]), "+v"(x[3][0]), "+v"(x[3][1]), "+v"(x[3][2]), "+v"(x[3][3]));
                };
                auto land_a = [&](int jj, float(&x)[9]) {
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]));
                    if constexpr (!kPre) {
                        const float sv = s_lds[k * kKC + 4 * jj + q4];
#pragma unroll
                        for (int t = 0; t < 9; ++t) x[t] *= sv;
                    }
                };
                auto land_halo = [&]() {
#pragma unroll
                    for (int ht = 0; ht < kHaloTiles; ++ht) asm volatile("" : "+v"(bh[ht][0]), "+v"(bh[ht][1]), "+v"(bh[ht][2]), "+v"(bh[ht][3]));
                };
                load_a(0, a[0]);
                load_b(0, bf[0]);
                load_halo(0);
                land_b(bf[0]);
                land_a(0, a[0]);
                land_halo();
#pragma unroll
                for (int grp = 0; grp < kNG; ++grp) {
                    const int jj = grp / kGroups, gi = grp % kGroups;
                    // the next group's operands go out before this group's MFMAs
                    if (grp + 1 < kNG) {
                        if ((grp + 1) % kGroups == 0) load_a(1, a[1]);
                        load_b(grp + 1, bf[(grp + 1) & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[tap_cls(t)][4 * gi + i] =
                                __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj][t], bf[grp & 1][tap_shift(t)][i], acc[tap_cls(t)][4 * gi + i], 0, 0, 0);
                        // the halo tiles: one class per wave, with the k-step's first group
                        if (gi == 0 && wave == tap_cls(t)) {
#pragma unroll
                            for (int ht = 0; ht < kHaloTiles; ++ht)
                                acch[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj][t], bh[ht][tap_shift(t)], acch[ht], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // one DMA piece of the next chunk behind every other tap of the chunk's first groups
                        if ((t & 1) == 1 && grp * 4 + (t >> 1) < kPiecesPerWave) {
                            if (next_chunk >= 0) stage_piece_n(grp * 4 + (t >> 1), next_chunk, nxt_buf);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // (the halo fragments of the second k-step: the first one's were consumed with group 0)
                    if (grp + 1 < kNG && (grp + 1) % kGroups == 0) load_halo(1);
                    if (grp + 1 < kNG) {
                        land_b(bf[(grp + 1) & 1]);
                        if ((grp + 1) % kGroups == 0) {
                            land_a(1, a[1]);
                            land_halo();
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (next_chunk >= 0) stage_chunk(next_chunk, nxt_buf);
                // (the flush step's per-lane offsets are derived here, from a fresh lane id: kept from the prologue they would be live
                // across the main K loop, where every register is taken)
                const int fl = fresh_lane();
                const int n16 = fl & 15, q4 = fl >> 4;
                const int boff = q4 * kPlane + (wave + 1) * kPW + n16 + 4;
                int boffh[kHaloTiles];
#pragma unroll
                for (int ht = 0; ht < kHaloTiles; ++ht) {
                    const int slot = 16 * ht + n16;
                    boffh[ht] = q4 * kPlane + (slot % kTH + 1) * kPW + ((slot / kTH) ? kSW : -1) + 4;
                }
                const float* const Wl = cur_buf + q4 * kBM + n16;
                const float* const Pl = cur_buf + kWlRegion;
                // flush step (si == step_main): only T row 2H exists below the image = position row y' = H (local row 0: wave
                // 0's first tile row), even row parity, and only its taps on input row H-1 are non-zero: EE taps 2, 3 and EO
                // tap 5 (wave 0: its tiles of that row and the halo tiles' EE class; wave 1: the halo tiles' EO class)
                if constexpr (WX) {
                    // (pair form: row tap dy = -1 only: EE products with slots 3..5, EO with slot 7; halo: direct taps 2, 3 = slots 5, 3, tap 5 = slot 7)
                    if (wave < 2) {
#pragma unroll
                        for (int j = 0; j < kKC / 4; ++j) {
                            if (wave == 0) {
#pragma unroll
                                for (int pt = 0; pt < kPT; ++pt) {
                                    const float* const src = Pl + boff + n16 + 4 * j * kPlane - kPW + 32 * pt;
                                    const float e0 = src[-1], e1 = src[0], e2 = src[1];
                                    accw[pt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(3 * kKC + 4 * j) * kBM], e0 - e1, accw[pt][0], 0, 0, 0);
                                    accw[pt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(4 * kKC + 4 * j) * kBM], e1, accw[pt][1], 0, 0, 0);
                                    accw[pt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(5 * kKC + 4 * j) * kBM], e1 - e2, accw[pt][2], 0, 0, 0);
                                    accw[pt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(7 * kKC + 4 * j) * kBM], e1, accw[pt][3], 0, 0, 0);
                                    accw[pt][4] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(7 * kKC + 4 * j) * kBM], e2, accw[pt][4], 0, 0, 0);
                                }
                            }
#pragma unroll
                            for (int t = 2; t <= 5; ++t) {
                                if (t == 4) continue;
                                const int slot = t == 2 ? 5 : (t == 3 ? 3 : 7);
                                if (wave == tap_cls(t)) {
#pragma unroll
                                    for (int ht = 0; ht < kHaloTiles; ++ht)
                                        acch[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(slot * kKC + 4 * j) * kBM],
                                                                                        Pl[boffh[ht] + 4 * j * kPlane - kPW - (tap_shift(t) & 1)], acch[ht], 0, 0, 0);
                                }
                            }
                        }
                    }
                } else if (wave < 2) {
#pragma unroll
                    for (int j = 0; j < kKC / 4; ++j) {
                        const float sv = kPre ? 1.0f : s_lds[k * kKC + 4 * j + q4];
#pragma unroll
                        for (int t = 2; t <= 5; ++t) {
                            if (t == 4) continue;
                            const float a = Wl[(t * kKC + 4 * j) * kBM] * sv;
                            const int dx = tap_shift(t) & 1;
                            if (wave == 0) {
#pragma unroll
                                for (int ct = 0; ct < CT; ++ct)
                                    acc[tap_cls(t)][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        a, Pl[boff + 4 * j * kPlane - kPW + 16 * ct - dx], acc[tap_cls(t)][ct], 0, 0, 0);
                            }
                            if (wave == tap_cls(t)) {
#pragma unroll
                                for (int ht = 0; ht < kHaloTiles; ++ht)
                                    acch[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Pl[boffh[ht] + 4 * j * kPlane - kPW - dx], acch[ht], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        // (the T window lies over ring slot 1, which the last chunk was read from)
        lds_barrier();

        // ---- epilogue: 4 RW passes (four position rows of the step x channel group) ----
        if (UPFIR16_DBG & 2) return;
        // per-lane roles (re-derived here: see fresh_lane)
        const int elane = fresh_lane();
        const int en16 = elane & 15, eq4 = elane >> 4;
        // dump: register g of accumulator tile (class, rw, ct) = channel 4 g + q4 of the block at position (w + 4 rw, 16 ct + n16):
        // T window [q4][2 w + py][2 (16 ct + n16) + px + 1]
        const int dump_base = eq4 * (kPassRows * kTW) + (2 * wave) * kTW + 2 * en16 + 1;
        const int hpy = wave >> 1, hpx = wave & 1;  // the halo tiles' class held by this wave
        // filter: thread = (channel fc of the pass, column group cg: output columns 2 X0 + 4 cg .. + 3, row group rg: output rows kFR rg .. + kFR - 1 of the pass)
        const int fc = wave;
        const int cg = elane % kCG;
        const int rg = elane / kCG;
        // (per lane: channel plane, the row group's first row, column group)
        const int o_voff = (int)((fc * oplane + (long long)(kFR * rg) * OWp + 4 * cg) * 4);
        const bool emit = si >= 0 && !(UPFIR16_DBG & 32);
        // Noise of the step's output rows [8 RW][2 kSW] (the same for every channel): by DMA into the part of ring slot 1 the T
        // window leaves free, so that it costs no registers and no vector load sits behind the stores (the single vmcnt counter: a
        // load behind a store waits for it). A lane's 16 bytes: float f = 256 piece + 4 lane of the region; rows above / below the
        // plane are outside the bounded resource and arrive as zeros (they are never stored). The whole offset is in the VECTOR
        // part: the range check does not see the scalar one.
        if (has_noise && emit) {
#pragma unroll
            for (int i = 0; i < kNzPieces / 4; ++i) {
                const int f = (wave + 4 * i) * 256 + 4 * elane;
                const int row = f / (2 * kSW), col = f % (2 * kSW);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nz_rsrc, (lds_ptr_t)(nz_lds + (wave + 4 * i) * 256), 16,
                                                         ((2 * y0 - 2 + row) * OW + 2 * X0 + col) * 4, 0, 0, 0);
            }
        }
        // the next step's first chunk (and the noise) must have landed before the first store is issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed = true;

        constexpr int kRowPasses = kFlush ? 1 : RW;
#pragma unroll
        for (int rw = 0; rw < kRowPasses; ++rw) {
            const int oy0 = 2 * (y0 + 4 * rw) - 2;       // output row of the pass's window row r = 0
            const int r_lo = max(0, -oy0);                // first image step: rows -2, -1 do not exist
            const int r_hi = min(kPassRows, 2 * H - oy0);  // flush step: only rows 2H-2, 2H-1
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // -- dump: accumulator register g = channel 4 g + q4 (g, rw are unrolled: register indices are static) --
                if constexpr (WX) {
                    // pair form: the output transform (y0 = m1 + m2, y1 = m2 - m3) on the way out; a lane holds the FOUR consecutive T
                    // columns of its pair per row parity: T columns 4 n .. 4 n + 3 of the pair tile = (EE y0, EO x0, EE y1, EO x1) / (OE y0, OO x0, OE y1, OO x1)
#pragma unroll
                    for (int pt = 0; pt < kPT; ++pt) {
                        const f32x4 (&m)[10] = accw[kPT * rw + pt];
                        float* const dst = stage + dump_base + 2 * en16 + 64 * pt;  // (dump_base holds 2 n16 + 1: pairs are 4 T columns apart)
                        dst[0] = m[0][g] + m[1][g];
                        dst[1] = m[3][g];
                        dst[2] = m[1][g] - m[2][g];
                        dst[3] = m[4][g];
                        dst[kTW + 0] = m[5][g] + m[6][g];
                        dst[kTW + 1] = m[8][g];
                        dst[kTW + 2] = m[6][g] - m[7][g];
                        dst[kTW + 3] = m[9][g];
                    }
                } else if (!(UPFIR16_DBG & 16)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            stage[dump_base + (c >> 1) * kTW + 32 * ct + (c & 1)] = acc[c][rw * CT + ct][g];
                }
#pragma unroll
                for (int ht = 0; ht < kHaloTiles; ++ht) {
                    const int slot = 16 * ht + en16, row = slot % kTH, side = slot / kTH;
                    if ((side == 1 || hpx == 1) && row / 4 == rw)
                        stage[eq4 * (kPassRows * kTW) + (2 * (row % 4) + hpy) * kTW + (side ? 2 * kSW + 1 + hpx : 0)] = acch[ht][g];
                }
                lds_barrier();

                // -- filter: window row i of row group rg = row R = kFR rg + i of (the channel's three carried T rows, the pass's eight):
                // rows R < 3 come from the carry, the others from the T window (a select per lane for the first three rows only) --
                const int ch = 4 * g + fc;
                const float dsc = d_lds[ch] * kSqrt2f;
                const float kh0 = 0.25f * dsc, kh1 = 0.75f * dsc;
                const float bias2 = b_lds[ch] * kSqrt2f;
                const float lr6 = 0.6f * sn_lds[ch], lr4 = 0.4f * sn_lds[ch];  // leaky ReLU x the next layer's style
                const float* const win_lo = carry + ch * (kCarryRows * kTW) + (kFR * rg) * kTW + 4 * cg;
                const float* const win_hi = stage + fc * (kPassRows * kTW) + (kFR * rg - kCarryRows) * kTW + 4 * cg;
                f32x4 ta[kWin], tb[kWin];
#pragma unroll
                for (int i = 0; i < kWin; ++i) {
                    const float* const rowp = (i < kCarryRows && kFR * rg + i < kCarryRows ? win_lo : win_hi) + i * kTW;
                    ta[i] = *reinterpret_cast<const f32x4*>(rowp);
                    tb[i] = *reinterpret_cast<const f32x4*>(rowp + 4);
                }
                if (emit) {
                    // Vertical taps FIRST, on the raw window: an output row's seven columns are four aligned register pairs of
                    // its rows' two 16-byte reads, so the pass is 16 PACKED operations per output row (v_pk_mul / v_pk_fma_f32:
                    // the fp32 vector instructions are what this epilogue costs -- they do not run beside another wave's fp32
                    // MFMAs, SQ_VALU_MFMA_COEXEC_CYCLES = 0 --, and filtering seven rows horizontally for four output rows
                    // was 28 per row); then the horizontal taps, which carry demod * sqrt 2 and the bias: 16 per row.
                    // (row oy0 + kFR rg + r of the image = row oy0 + 3 + kFR rg + r of the resource: oy0 >= -2)
                    const int o_soff = (int)((4 * g * oplane + (long long)(oy0 + 3) * OWp + 2 * X0 + 4) * 4);
#pragma unroll
                    for (int r = 0; r < kFR; ++r) {
                        const int rr = kFR * rg + r;
                        if (rr >= r_lo && rr < r_hi) {
                            f32x2 tv[4];
#pragma unroll
                            for (int c2 = 0; c2 < 4; ++c2) {
                                auto pair = [&](int i) { return c2 < 2 ? f32x2{ta[i][2 * c2], ta[i][2 * c2 + 1]} : f32x2{tb[i][2 * c2 - 4], tb[i][2 * c2 - 3]}; };
                                tv[c2] = 0.25f * pair(r) + 0.75f * pair(r + 1) + 0.75f * pair(r + 2) + 0.25f * pair(r + 3);
                            }
                            const float t[7] = {tv[0][0], tv[0][1], tv[1][0], tv[1][1], tv[2][0], tv[2][1], tv[3][0]};
                            f32x4 v;
#pragma unroll
                            for (int o = 0; o < 4; ++o) v[o] = fmaf(kh0, t[o + 3], fmaf(kh1, t[o + 2], fmaf(kh1, t[o + 1], fmaf(kh0, t[o], bias2))));
                            if (has_noise) v += ns2 * *reinterpret_cast<const f32x4*>(nz_lds + (kPassRows * rw + rr) * (2 * kSW) + 4 * cg);
#pragma unroll
                            for (int o = 0; o < 4; ++o) v[o] = fmaf(lr6, v[o], lr4 * __builtin_fabsf(v[o]));
                            if (UPFIR16_DBG & 256)  // (experiment: non-temporal stores)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, o_voff, o_soff + r * OWp * 4, 2);
                            else if (!(UPFIR16_DBG & 1) || v[0] == 12345.f)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, o_voff, o_soff + r * OWp * 4, 0);
                        }
                    }
                }
                lds_barrier();
                // -- the last three T rows of the window become the carry of these 4 channels (the last row group read them as
                // the last three rows of its window) --
                if (rg == G::kRG - 1) {
                    float* const carry_w = carry + ch * (kCarryRows * kTW) + 4 * cg;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        *reinterpret_cast<f32x4*>(carry_w + i * kTW) = ta[kWin - 3 + i];
                        if (cg == kCG - 1) *reinterpret_cast<f32x4*>(carry_w + i * kTW + 4) = tb[kWin - 3 + i];
                    }
                }
            }
        }
    };
#pragma unroll 1
    for (int si = step_first; si < step_main; ++si) run_step(std::false_type{}, si);
    if (step_last > step_main) run_step(std::true_type{}, step_main);
}

// the strip geometry of an input W wide: 64-column strips where they tile it, else 32 / 16 columns (the 32- and 16-wide layers)
static int upfir16_strip(int W) { return W % 64 == 0 ? 64 : (W == 32 ? 32 : (W == 16 ? 16 : 0)); }
static int upfir16_step_rows(int W) { return W % 64 == 0 ? 8 : 16; }

bool upfir16_supported(int cin, int cout, int H, int W) {
    // (an even number of chunks per step: the T window lies over the ring slot that is idle after an even count)
    return H == W && upfir16_strip(W) != 0 && H % upfir16_step_rows(W) == 0 && cin % (2 * kKC) == 0 && cout % kBM == 0 && cin <= 512;
}

size_t upfir16_weight_floats(int cin, int cout) { return (size_t)(cout / kBM) * (cin / kKC) * Geo16<4, 2>::kWlRegion; }

// w_in: scaled filter [tap = wy*3+wx][cin][cout]; w_out: [m tile of 16][chunk of 8][1280]: [slot][ci 0..7][row m 0..15] + padding, slot t =
// filter tap up_tap_weight[t], MFMA row m = 4 q + r holds channel 4 r + q of the tile (see the header)
void upfir16_arrange_weights(const float* w_in, int cin, int cout, const int* up_tap_weight, float* w_out) {
    const int m_tiles = cout / kBM, chunks = cin / kKC;
    std::fill(w_out, w_out + upfir16_weight_floats(cin, cout), 0.f);
    for (int mt = 0; mt < m_tiles; ++mt)
        for (int ch = 0; ch < chunks; ++ch)
            for (int t = 0; t < 9; ++t)
                for (int kc = 0; kc < kKC; ++kc)
                    for (int m = 0; m < kBM; ++m) {
                        const int channel = 4 * (m & 3) + (m >> 2);
                        w_out[((size_t)mt * chunks + ch) * Geo16<4, 2>::kWlRegion + (t * kKC + kc) * kBM + m] =
                            w_in[((size_t)up_tap_weight[t] * cin + ch * kKC + kc) * cout + mt * kBM + channel];
                    }
}

// The pair form (Geo16<4, 2, true>: F(2,2) along x on the classes with two taps there): inputs whose width the 64-column strips tile.
bool upfir16x_supported(int cin, int cout, int H, int W) { return upfir16_supported(cin, cout, H, W) && (W % 64 == 0 || W == 32); }
size_t upfir16x_weight_floats(int cin, int cout) { return (size_t)(cout / kBM) * (cin / kKC) * Geo16<4, 2, true>::kWlRegion; }

// w_out: [m tile of 16][chunk of 8][12 slots][ci 0..7][row m 0..15]; with filter taps in the slot order of the direct image (t0..t8 =
// up_tap_weight[0..8]: EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO) the slots are
// EE row tap 0: t1, t0 + t1, t0 | EE row tap -1: t3, t2 + t3, t2 | EO: t4, t5 | OE: t7, t6 + t7, t6 | OO: t8   (h0 = the tap on column x - 1)
void upfir16x_arrange_weights(const float* w_in, int cin, int cout, const int* up_tap_weight, float* w_out) {
    const int m_tiles = cout / kBM, chunks = cin / kKC;
    constexpr int kRegion = Geo16<4, 2, true>::kWlRegion;
    for (int mt = 0; mt < m_tiles; ++mt)
        for (int ch = 0; ch < chunks; ++ch)
            for (int kc = 0; kc < kKC; ++kc)
                for (int m = 0; m < kBM; ++m) {
                    const int channel = 4 * (m & 3) + (m >> 2);
                    float t[9];
                    for (int i = 0; i < 9; ++i) t[i] = w_in[((size_t)up_tap_weight[i] * cin + ch * kKC + kc) * cout + mt * kBM + channel];
                    const float slots[12] = {t[1], (float)((double)t[0] + (double)t[1]), t[0], t[3], (float)((double)t[2] + (double)t[3]), t[2],
                                             t[4], t[5], t[7], (float)((double)t[6] + (double)t[7]), t[6], t[8]};
                    for (int sl = 0; sl < 12; ++sl) w_out[((size_t)mt * chunks + ch) * kRegion + (sl * kKC + kc) * kBM + m] = slots[sl];
                }
}

// Row segments: as few as give every CU a block (a segment costs one extra priming step; the second block per CU comes
// with the batch: 64 frames are two to four rounds of 512 resident blocks on every layer).
void upfir16_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* a) {
    const int strip = upfir16_strip(W), step_rows = upfir16_step_rows(W);
    a->m_tiles = cout / kBM;
    a->strips = W / strip;
    a->step_rows = step_rows;
    const int base = B * a->m_tiles * a->strips;
    const int steps = H / step_rows;
    int segs = 1;
    while (base * segs < num_cus && segs * 2 <= steps && steps % (segs * 2) == 0) segs *= 2;
    a->segs = segs;
    a->rows_per_seg = H / segs;
    a->total_blocks = base * segs;
    static const int env_stagger_us = [] { const char* v = std::getenv("GANCE_TUNE_UPFIR16_STAGGER_US"); return v ? std::atoi(v) : 0; }();
    a->stagger_phases = 1;
    a->stagger_ticks = env_stagger_us * 100;  // (s_memrealtime ticks of 10 ns)
    static const int env_debug = [] { const char* v = std::getenv("GANCE_DEBUG_UPFIR"); return v ? std::atoi(v) : 0; }();
    a->debug_flags = env_debug;
}

// (plain kernels around the templated body: see winograd64_conv.hip on kernel templates and the host pass)
#define GANCE_UPFIR16_KERNELS(SUFFIX, CT, RW)                                                                                              \
    __global__ __launch_bounds__(256, 2) void upfir16_fused##SUFFIX##_kernel(const UpFirArgs p) { upfir16_body<Geo16<CT, RW>, false, false>(p); }        \
    __global__ __launch_bounds__(256, 2) void upfir16_fused##SUFFIX##_pre_kernel(const UpFirArgs p) { upfir16_body<Geo16<CT, RW>, true, false>(p); }     \
    __global__ __launch_bounds__(256, 2) void upfir16_fused##SUFFIX##_noise_kernel(const UpFirArgs p) { upfir16_body<Geo16<CT, RW>, false, true>(p); }   \
    __global__ __launch_bounds__(256, 2) void upfir16_fused##SUFFIX##_pre_noise_kernel(const UpFirArgs p) { upfir16_body<Geo16<CT, RW>, true, true>(p); }
GANCE_UPFIR16_KERNELS(, 4, 2)
GANCE_UPFIR16_KERNELS(_w32, 2, 4)
GANCE_UPFIR16_KERNELS(_w16, 1, 4)
#undef GANCE_UPFIR16_KERNELS
// the pair form: input pre-scaled only
__global__ __launch_bounds__(256, 2) void upfir16x_fused_pre_kernel(const UpFirArgs p) { upfir16_body<Geo16<4, 2, true>, true, false>(p); }
__global__ __launch_bounds__(256, 2) void upfir16x_fused_pre_noise_kernel(const UpFirArgs p) { upfir16_body<Geo16<4, 2, true>, true, true>(p); }
__global__ __launch_bounds__(256, 2) void upfir16x_fused_w32_pre_kernel(const UpFirArgs p) { upfir16_body<Geo16<2, 4, true>, true, false>(p); }
__global__ __launch_bounds__(256, 2) void upfir16x_fused_w32_pre_noise_kernel(const UpFirArgs p) { upfir16_body<Geo16<2, 4, true>, true, true>(p); }

hipError_t launch_upfir16_fused(const UpFirArgs& args, hipStream_t stream) {
    using Kernel = void (*)(const UpFirArgs);
    struct Variant {
        Kernel kernel[4];  // [pre * 2 + noise]
        size_t lds_plain, lds_pre;
    };
    static const Variant variants[3] = {
        {{upfir16_fused_kernel, upfir16_fused_noise_kernel, upfir16_fused_pre_kernel, upfir16_fused_pre_noise_kernel}, Geo16<4, 2>::lds_bytes(512), Geo16<4, 2>::lds_bytes(0)},
        {{upfir16_fused_w32_kernel, upfir16_fused_w32_noise_kernel, upfir16_fused_w32_pre_kernel, upfir16_fused_w32_pre_noise_kernel}, Geo16<2, 4>::lds_bytes(512), Geo16<2, 4>::lds_bytes(0)},
        {{upfir16_fused_w16_kernel, upfir16_fused_w16_noise_kernel, upfir16_fused_w16_pre_kernel, upfir16_fused_w16_pre_noise_kernel}, Geo16<1, 4>::lds_bytes(512), Geo16<1, 4>::lds_bytes(0)},
    };
    static PerDeviceInt ready;  // the dynamic-LDS opt-in is per device
    int unused = 0;
    const hipError_t e = ready.get(
        [&](int, int* value) {
            *value = 1;
            for (const auto kernel : {upfir16x_fused_pre_kernel, upfir16x_fused_pre_noise_kernel}) {
                const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           (int)Geo16<4, 2, true>::lds_bytes(0));
                if (err != hipSuccess) return err;
            }
            for (const auto kernel : {upfir16x_fused_w32_pre_kernel, upfir16x_fused_w32_pre_noise_kernel}) {
                const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           (int)Geo16<2, 4, true>::lds_bytes(0));
                if (err != hipSuccess) return err;
            }
            for (const Variant& v : variants)
                for (int i = 0; i < 4; ++i) {
                    const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(v.kernel[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                               (int)(i >= 2 ? v.lds_pre : v.lds_plain));
                    if (err != hipSuccess) return err;
                }
            return hipSuccess;
        },
        &unused);
    if (e != hipSuccess) return e;
    const int strip = upfir16_strip(args.W);
    if (strip == 0) return hipErrorInvalidValue;
    if (args.pair_form) {  // (args.w is upfir16x_arrange_weights' image)
        if (strip < 32 || !args.input_prescaled) return hipErrorInvalidValue;
        const size_t lds_64 = Geo16<4, 2, true>::lds_bytes(0), lds_32 = Geo16<2, 4, true>::lds_bytes(0);
        const size_t lds_x = strip == 64 ? lds_64 : lds_32;
        const Kernel kernel_x = strip == 64 ? (args.noise != nullptr ? upfir16x_fused_pre_noise_kernel : upfir16x_fused_pre_kernel)
                                            : (args.noise != nullptr ? upfir16x_fused_w32_pre_noise_kernel : upfir16x_fused_w32_pre_kernel);
        hipLaunchKernelGGL(kernel_x, dim3(args.total_blocks), dim3(256), lds_x, stream, args);
        return hipGetLastError();
    }
    const Variant& v = variants[strip == 64 ? 0 : (strip == 32 ? 1 : 2)];
    const int pre = args.input_prescaled ? 1 : 0, noise = args.noise != nullptr ? 1 : 0;
    const size_t lds = strip == 64 ? Geo16<4, 2>::lds_bytes(pre ? 0 : args.Cin) : (strip == 32 ? Geo16<2, 4>::lds_bytes(pre ? 0 : args.Cin) : Geo16<1, 4>::lds_bytes(pre ? 0 : args.Cin));
    hipLaunchKernelGGL(v.kernel[2 * pre + noise], dim3(args.total_blocks), dim3(256), lds, stream, args);
    return hipGetLastError();
}

}  // namespace gance
