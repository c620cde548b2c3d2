// Conv0_up in ONE kernel, fourth form (round 5): split operands on the bf16 matrix cores AND the FIR epilogue beside the K loop.
//
// Same layer, same arithmetic, same weight image and same layout contracts as upfir_split.hip (read its header first): stride-2
// transposed modulated 3x3 convolution as four parity classes from operands split into three bf16 parts (six part products, fp32
// accumulation: fp32 accuracy), [1,3,3,1] x [1,3,3,1] FIR, noise, bias, leaky ReLU, one launch, the (2H+1)^2 intermediate T never in
// HBM. Replaces, for the reference's synthesis call (gance/network_interface/network_functions.py:168), the un-vendored
// `upsample_conv_2d` + `fused_bias_act` pair.
//
// What is different: who does what. In upfir_split.hip one wave per SIMD does everything in turn, and the FIR epilogue (vector work:
// 14 instructions per output value) is SERIAL with the K loop -- at 512 -> 1024, two chunks of 32 input channels per step, the
// epilogue is longer than the products. A bf16 MFMA holds its SIMD's vector issue for 8 of its 16 cycles only, but a wave cannot
// filter its own accumulators while it multiplies into them. So the block has EIGHT waves, two per SIMD, in two roles:
//   * waves 0..3, the MATRIX waves: weight fragments of a chunk in registers (108), the haloed patch row by row from a two-slot ring in
//     LDS, 54 MFMAs per row and wave (one tile column of 16 positions x 4 position rows x 4 classes: 64 accumulators) -- nothing else.
//     After a step's last row they dump the accumulators into the step's T window in LDS ([16 channels][8 T rows][132]).
//   * waves 4..7, the VECTOR waves: everything else. They stage the patch rows (eight coalesced dword loads per lane five rows ahead,
//     the split into three bf16 parts -- 44 vector instructions per lane and row -- and three 16-byte LDS writes), keep the halo side
//     buffer, and run the FIR epilogue of step s - 1 out of the T window WHILE the matrix waves multiply step s: four passes of four
//     channels, spread over the step's row periods.
// The two roles meet at ONE barrier per patch row (and one per dump): a lock-step software pipeline, no flags, no polling. A register
// budget of 256 per wave (two waves per SIMD) holds 4 position rows per step instead of 8: the ring rows at a step's edge are staged
// 5/4 times instead of 9/8; the MFMA count does not change with the step height. Position row y' = H (T row 2H) is one more step
// over the bottom border row (no special instantiation): its window holds T rows 2H, 2H + 1 (zero), its passes store two rows.
//
// Ring and barriers: stream row r (step, chunk, j) is written by the vector waves in period r - 2, its fragments are read by the
// matrix waves in period r - 1 and multiplied in period r; slot = parity of r (five rows per chunk, an even number of chunks per step).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"

// Timing ablations (wrong results; only in builds with -DGANCE_UPFIRR_ABLATE=<flags>, Makefile target ../libgance_hip_upfirrab<flags>.so, used
// through GANCE_HIP_LIBRARY): 1 no split arithmetic, 2 no FIR passes, 4 no global loads of patch rows in the row loop, 8 no MFMAs,
// 16 no LDS writes of staged rows, 32 no weight fragments after the first chunk's, 64 no halo tile, 128 no accumulator dump, 256 no output stores,
// 512 no global loads of the halo columns in the row loop
#ifndef GANCE_UPFIRR_ABLATE
#define GANCE_UPFIRR_ABLATE 0
#endif
// wave priorities (s_setprio): 0 none, 1 the matrix waves above the vector waves, 2 the vector waves above the matrix waves
#ifndef GANCE_UPFIRR_PRIO
#define GANCE_UPFIRR_PRIO 1
#endif

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kBM = 16;             // output channels per block
constexpr int kKC = 32;             // input channels per chunk = one k-step of v_mfma_f32_16x16x32_bf16
constexpr int kSW = 64;             // position columns per strip
constexpr int kTH = 4;              // position rows per step
constexpr int kRows = kTH + 1;      // patch rows of a step: input rows y0 - 1 .. y0 + 3
constexpr int kPlanes = 12;         // 16-byte units per position: part x k-group (a unit = 8 channels of one position: a lane's MFMA fragment)
constexpr int kPlaneStride = 80;    // units per plane row in the ring: a multiple of 16, so the k-groups of a fragment read fall on distinct banks
constexpr int kSlotUnits = kPlanes * kPlaneStride;
constexpr int kRing = 2;
constexpr int kHaloCols = 4;        // input columns X0 - 2, X0 - 1, X0 + 63, X0 + 64
constexpr int kHaloUnits = kRows * kHaloCols * kPlanes;
constexpr int kHaloTasks = kRows * kHaloCols * 4;  // (row, column, k-group): 80
constexpr int kDepth = 2 * kRows;   // patch rows in flight between their global loads and their LDS writes (8 registers each): ten row periods of 0.4 us cover an HBM miss under load
constexpr int kCarryRows = 3;
constexpr int kPassRows = 2 * kTH;  // T rows of a step
constexpr int kTW = 2 * kSW + 4;    // T window row: T columns 2 X0 - 1 .. 2 X0 + 129 (+ pad)
constexpr int kCG = kSW / 2;        // column groups of 4 output columns
constexpr int kRG = 64 / kCG;       // row groups of a wave's 64 filter threads
constexpr int kFR = kPassRows / kRG;  // output rows per filter thread
constexpr int kWin = kFR + 3;
constexpr int kChRows = kCarryRows + kPassRows;  // the window of one channel: the three carried T rows, then the step's eight
constexpr int kChFloats = kChRows * kTW;
constexpr int kWindowFloats = kBM * kChFloats;
constexpr int kNoiseFloats = kPassRows * 2 * kSW;  // the noise of a step's output rows
constexpr int kSlices = 4 * kFR;                 // FIR work of a step per vector wave: (channel group g, output row r of each row group)
constexpr int kMatrixWaves = 4;
constexpr int kThreads = 512;
constexpr float kSqrt2f = 1.4142135623730951f;
static_assert(kRows % 2 == 1 && kDepth == 2 * kRows && kRing == 2, "slot = parity of the stream row; the staging registers rotate with the rows of a chunk pair");

constexpr int kWUnits = 27 * 64;    // 16-byte units of a chunk's weight fragments: [tap][part][lane]

// LDS (bytes): ring | halo side buffers (two chunks) | weight fragments of the next chunk | T window of a step (with the carried rows) | noise | demod | bias | next style
constexpr size_t kRingBytes = (size_t)kRing * kSlotUnits * 16;
constexpr size_t kHaloBytes = (size_t)2 * kHaloUnits * 16;
constexpr size_t kWBytes = (size_t)kWUnits * 16;
constexpr size_t kLdsBytes = kRingBytes + kHaloBytes + kWBytes + sizeof(float) * ((size_t)kWindowFloats + kNoiseFloats + 3 * kBM);
static_assert(kLdsBytes <= 160 * 1024, "one block per CU");

// transposed-conv tap tables, in the order the weights are stored (engine.hip kUpTapWeight):
// EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO (0,0); class = 2*py + px
__host__ __device__ constexpr int tap_cls(int t) { return t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3)); }
__host__ __device__ constexpr int tap_dy(int t) { return (t == 2 || t == 3 || t == 5) ? 1 : 0; }  // 1: the input row above
__host__ __device__ constexpr int tap_dx(int t) { return (t == 1 || t == 3 || t == 7) ? 1 : 0; }  // 1: the input column to the left

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// eight fp32 values (the channels of one k-group at one position) -> the position's three 16-byte units. The four channel pairs are
// independent chains of seven dependent steps (convert, unpack, subtract, convert, unpack, subtract, convert): written stage by stage
// across the pairs, with the scheduler fenced between stages, so that a dependent instruction is at least four instructions behind its
// source -- alone on its SIMD's vector pipe a wave otherwise waits out every step of a chain (measured: the split cost 2.5 x its issue time).
__device__ __forceinline__ void split_unit(const unsigned (&raw)[8], u32x4 (&part)[3]) {
    if (GANCE_UPFIRR_ABLATE & 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) part[q] = u32x4{raw[q], raw[q + 1], raw[q + 2], raw[q + 3]};
        return;
    }
    auto pack = [](float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2)); };
    auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
    float a[4], b[4];
    unsigned w[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = __builtin_bit_cast(float, raw[2 * e]);
        b[e] = __builtin_bit_cast(float, raw[2 * e + 1]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[q][e] = pack(a[e], b[e]);
        fence();
        if (q == 2) break;
        float ha[4], hb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ha[e] = __builtin_bit_cast(float, w[q][e] << 16);
            hb[e] = __builtin_bit_cast(float, w[q][e] & 0xffff0000u);
        }
        fence();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] -= ha[e];
            b[e] -= hb[e];
        }
        fence();
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) part[q] = u32x4{w[q][0], w[q][1], w[q][2], w[q][3]};
}

// part products, smallest first: {x part, w part}
constexpr int kTerms[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};

}  // namespace

template <bool kNoise>
__device__ __forceinline__ void upfirr_body(const UpFirArgs& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const ring = reinterpret_cast<u32x4*>(smem_raw);
    u32x4* const halo = ring + kRing * kSlotUnits;
    u32x4* const w_lds = halo + 2 * kHaloUnits;                             // [tap][part][lane]: the fragments of the chunk after the one being multiplied
    float* const window = reinterpret_cast<float*>(w_lds + kWUnits);       // [16 ch][3 carried T rows + 8 T rows][kTW]
    float* const nz_lds = window + kWindowFloats;                           // [8 output rows][2 kSW]
    float* const d_lds = nz_lds + kNoiseFloats;
    float* const b_lds = d_lds + kBM;
    float* const sn_lds = b_lds + kBM;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // ---- block -> (sample, strip, channel tile); blocks of one XCD take contiguous ids (see upfir_split.hip) ----
    int id;
    {
        const int v = blockIdx.x, nwg = p.total_blocks;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    const int m_tile = id % p.m_tiles;
    id /= p.m_tiles;
    const int strip = id % p.strips;
    const int b = id / p.strips;
    const int m0 = m_tile * kBM;
    const int X0 = strip * kSW;
    const int H = p.H, W = p.W;
    const int Hp = H + 2, Wp = W + 8;
    const int HpWp4 = Hp * Wp * 4;
    const int chunks = p.Cin / kKC;
    const int steps = H / kTH + 1;  // the last one is position row y' = H over the bottom border

    // ---- constants and the zeroed carry (every thread) ----
    if (tid < kBM) {
        d_lds[tid] = p.d[(size_t)b * p.d_stride + m0 + tid];
        b_lds[tid] = p.bias[m0 + tid];
        sn_lds[tid] = p.s_next != nullptr ? p.s_next[(size_t)b * p.s_stride + m0 + tid] : 1.0f;
    }
    for (int i = tid; i < kBM * kCarryRows * kTW; i += kThreads) window[(i / (kCarryRows * kTW)) * kChFloats + i % (kCarryRows * kTW)] = 0.f;  // (T rows -3 .. -1)

    const __amdgpu_buffer_rsrc_t nz_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(kNoise ? p.noise + (size_t)b * p.noise_b_stride : nullptr), 0, kNoise ? (2 * H) * (2 * W) * 4 : 0, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.w) + (size_t)m_tile * chunks * 27 * 1024), 0, chunks * 27 * 1024, 0x00020000);

    if (wave < kMatrixWaves) {
        // =========================================== the matrix waves ===========================================
        if (GANCE_UPFIRR_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        const int n16 = lane & 15, kg = lane >> 4;
        // fragment reads of position 16 wave + n16 (dx = 0) and of the position to its left (dx = -1; left of the strip: the edge column)
        const int m_pos = 16 * wave + n16;
        const u32x4* const ring_r0 = ring + kg * kPlaneStride + m_pos + 1;
        const u32x4* const ring_r1 = ring + kg * kPlaneStride + m_pos;
        // halo tile slot n16 = (side n16 / 8, position row n16 % 8): rows 0 .. 3 exist (the others repeat row 3 and are never dumped)
        const int h_slot_row = min(n16 & 7, kTH - 1);
        const u32x4* const halo_r = halo + (h_slot_row * kHaloCols + 2 * (n16 >> 3)) * kPlanes + kg;

        const int h_first = (wave == 0 ? kHaloCols * 0 + 1 : kHaloCols + 1) * kPlanes;  // tap 2: patch row r, column x'; taps 4, 6, 8: row r + 1, column x'
        u32x4 A[9][3];
        const int a_voff = lane * 16;
        auto load_a3 = [&](int chunk, int t, u32x4(&dst)[3]) {
#pragma unroll
            for (int q = 0; q < 3; ++q) dst[q] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, a_voff, (chunk * 27 + t * 3 + q) * 1024, 0);
        };
        // The fragments of the NEXT chunk come out of LDS: the four matrix waves hold the same 27 KB, and 4 x 27 KB per chunk through the
        // CU's vector-memory path (64 bytes per clock, L1 hit or not) was what bounded the K loop; the vector waves fetch them once.
        const u32x4* const w_r = w_lds + lane;
        auto read_a3 = [&](int t, u32x4(&dst)[3]) {
#pragma unroll
            for (int q = 0; q < 3; ++q) dst[q] = w_r[(t * 3 + q) * 64];
        };
        // The noise of filter step fs (output rows 8 fs - 2 .. 8 fs + 5, columns 2 X0 .. 2 X0 + 127) by LDS-DMA, one piece of two rows per
        // matrix wave: these waves have no other vector-memory traffic, their wait for it (vmcnt(0), a period later) waits for nothing else.
        // (rows above and below the plane are outside the bounded resource and arrive as zeros; they are not stored either)
        auto noise_dma = [&](int fs) {
            int nl;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(nl));
            const int off = ((2 * kTH * fs - 2 + 2 * wave + (nl >> 5)) * (2 * W) + 2 * X0 + 4 * (nl & 31)) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(nz_rsrc, (lds_ptr_t)(nz_lds + wave * 256), 16, off, 0, 0, 0);
        };
        // The next chunk's weight fragments: 27 pieces of 1 KB ([tap][part][lane]) by LDS-DMA, seven (six) per matrix wave, issued in row 0 of a
        // chunk (18 MFMAs: the row with time to spare; the buffer was read in row 4 of the chunk before) and landed before the barrier of row 3.
        // On the vector waves they shared the in-order vmcnt queue with ten rows of staging loads and the output stores: 62 of its 63 slots at
        // 512 -> 1024, and a wait for them was a wait for the stores in front of them.
        auto weights_dma = [&](int chunk) {
            int wl;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(wl));
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < 6 || wave < 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(w_lds + (wave + 4 * k) * 64), 16, wl * 16, (chunk * 27 + wave + 4 * k) * 1024, 0, 0);
        };
        u32x4 Bf[2][2][3];
        auto load_b = [&](int slot, u32x4(&dst)[2][3]) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                dst[0][q] = ring_r0[slot * kSlotUnits + q * 4 * kPlaneStride];
                dst[1][q] = ring_r1[slot * kSlotUnits + q * 4 * kPlaneStride];
            }
        };
#pragma unroll
        for (int t = 0; t < 9; ++t) load_a3(0, t, A[t]);
        lds_barrier();  // B0: constants
        lds_barrier();  // B1: rows 0 and 1, the first halo buffer
        lds_barrier();  // B2: their edge columns
        load_b(0, Bf[0]);
        lds_barrier();  // B3: (slot 0 is rewritten in the first period)

        const int hpy = wave >> 1, hpx = wave & 1;  // the halo tile's class held by this wave

#pragma unroll 1
        for (int si = 0; si < steps; ++si) {
            f32x4 acc[kTH][4];
#pragma unroll
            for (int r = 0; r < kTH; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acch = f32x4{0.f, 0.f, 0.f, 0.f};  // halo tile (this wave's class)

            u32x4 hf[3];  // fragments of this wave's next halo tap
            auto run_chunk = [&](auto parity, const int chunk) {
                constexpr int ab = decltype(parity)::value;
                const int n_chunk = chunk + 1 < chunks ? chunk + 1 : 0;
#pragma unroll
                for (int j = 0; j < kRows; ++j) {
                    if (kNoise && ab == 0 && j == 0 && chunk == 0 && si > 0) noise_dma(si - 1);
                    if (j == 0 && !(GANCE_UPFIRR_ABLATE & 32)) weights_dma(n_chunk);
                    // Row j of this chunk: its fragments are in Bf[cur]; row j + 1 is in the ring (written before the last barrier): read them now
                    const int cur = (j + ab) & 1;
                    if (j + 1 < kRows) load_b(cur ^ 1, Bf[cur ^ 1]);
                    if (j + 1 == kRows && !(GANCE_UPFIRR_ABLATE & 64)) {
                        // the first halo tap of this wave's class in the last row's order (EE: tap 2, dy = -1; the others: their (0, 0) tap).
                        // (In the last row the next row's fragments are read at the END: beside them and the halo fragments the register
                        // file has no room for a third set.)
#pragma unroll
                        for (int q = 0; q < 3; ++q) hf[q] = halo_r[ab * kHaloUnits + h_first + q * 4];
                    }
                    if (j + 1 < kRows) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const int row = tap_dy(t) ? j : j - 1;
                            if (row < 0 || row >= kTH) continue;
#pragma unroll
                            for (int term = 0; term < ((GANCE_UPFIRR_ABLATE & 8) ? 0 : 6); ++term)
                                acc[row][tap_cls(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]), __builtin_bit_cast(bf16x8, Bf[cur][tap_dx(t)][kTerms[term][0]]),
                                    acc[row][tap_cls(t)], 0, 0, 0);
                        }
#pragma unroll
                        for (int i = 0; i < 54; ++i) {
                            if (i >= (j == 0 ? 18 : 54)) break;                 // (row 0 only has the dy = -1 taps)
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x092, 1, 0);  // one of: vector ALU, vector memory, LDS
                        }
                    } else {
                        // The chunk's last row, tap by tap: the row's dy = 0 products, this wave's share of the chunk's halo tile (one class per
                        // wave) -- and then the tap's fragments of the NEXT chunk. The order keeps the taps of one class apart (a halo tap's
                        // fragments are read one tap of that class ahead, under the products in between) and the dy = -1 taps early (the next
                        // chunk's first row needs their fragments first).
                        constexpr int kOrder[9] = {4, 2, 6, 3, 5, 0, 7, 1, 8};
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            const int t = kOrder[i];
                            if (!tap_dy(t)) {
#pragma unroll
                                for (int term = 0; term < ((GANCE_UPFIRR_ABLATE & 8) ? 0 : 6); ++term)
                                    acc[kTH - 1][tap_cls(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]), __builtin_bit_cast(bf16x8, Bf[cur][tap_dx(t)][kTerms[term][0]]),
                                        acc[kTH - 1][tap_cls(t)], 0, 0, 0);
                            }
                            if (wave == tap_cls(t) && !(GANCE_UPFIRR_ABLATE & 64)) {
#pragma unroll
                                for (int term = 0; term < 6; ++term)
                                    acch = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]),
                                                                                   __builtin_bit_cast(bf16x8, hf[kTerms[term][0]]), acch, 0, 0, 0);
                                // the class's next tap in the order
                                int nt = -1;
#pragma unroll
                                for (int k = 8; k > i; --k)
                                    if (tap_cls(kOrder[k]) == tap_cls(t)) nt = kOrder[k];
                                if (nt >= 0) {
#pragma unroll
                                    for (int q = 0; q < 3; ++q) hf[q] = halo_r[ab * kHaloUnits + ((1 - tap_dy(nt)) * kHaloCols + 1 - tap_dx(nt)) * kPlanes + q * 4];
                                }
                            }
                            if (!(GANCE_UPFIRR_ABLATE & 32)) read_a3(t, A[t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        load_b(cur ^ 1, Bf[cur ^ 1]);
                    }
                    if (kNoise && ab == 0 && j == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the noise: the first FIR slice runs in period 2)
                    if (j == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // (the weight fragments: read in row 4)
                    lds_barrier();
                }
            };
#pragma unroll 1
            for (int chunk = 0; chunk < chunks; chunk += 2) {
                run_chunk(std::integral_constant<int, 0>{}, chunk);
                run_chunk(std::integral_constant<int, 1>{}, chunk + 1);
            }

            // ---- dump: accumulator register g = channel 4 g + q4; T row 2 r + py, T column 2 x' + px (window column + 1) ----
            if (!(GANCE_UPFIRR_ABLATE & 128)) {
                // (per-lane addresses of the dump re-derived from a fresh lane id: nothing of them lives through the K loop, whose register
                // budget is full)
                int dl;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(dl));
                const int d16 = dl & 15, q4 = dl >> 4, h_side = d16 >> 3;
                float* const dump_w = window + q4 * kChFloats + kCarryRows * kTW + 32 * wave + 2 * d16 + 1;
                const bool h_dump = (h_side == 1 || hpx == 1) && (d16 & 7) < kTH;
                float* const dump_h = window + q4 * kChFloats + (kCarryRows + 2 * (d16 & 3) + hpy) * kTW + (h_side ? 2 * kSW + 1 + hpx : 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int r4 = 0; r4 < kTH; ++r4)
#pragma unroll
                        for (int c = 0; c < 4; ++c) dump_w[g * 4 * kChFloats + (2 * r4 + (c >> 1)) * kTW + (c & 1)] = acc[r4][c][g];
                    if (h_dump) dump_h[g * 4 * kChFloats] = acch[g];
                }
            } else {
                float sum = acch[0];
#pragma unroll
                for (int r = 0; r < kTH; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) sum += acc[r][c][0] + acc[r][c][3];
                if (sum == 12345.678f) p.out[tid] = sum;
            }
            lds_barrier();
        }
        if (kNoise) {
            noise_dma(steps - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();
        return;
    }

    // =========================================== the vector waves ===========================================
    if (GANCE_UPFIRR_PRIO == 2) __builtin_amdgcn_s_setprio(2);
    const int v = wave - kMatrixWaves;  // k-group staged by this wave, and the channel of a pass's four it filters
    const int vtid = tid - 64 * kMatrixWaves;
    // The input arrives SPLIT (upfirr_split_activation: x times this layer's style, three bf16 parts per value, 16-byte units of 8 channels):
    // [chunk][bordered row][plane = part * 4 + k-group][bordered column] units per sample. A patch row of a chunk is 12 plane rows; a lane
    // copies three units per row (planes 3 v .. 3 v + 2, column X0 + lane): loads ten rows ahead into registers, LDS writes -- no arithmetic.
    const size_t xs_sample = (size_t)chunks * Hp * kPlanes * Wp * 16;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.x_units) + (size_t)b * xs_sample), 0, (unsigned)xs_sample, 0x00020000);
    u32x4* const ring_w = ring + 3 * v * kPlaneStride + lane + 1;  // ring column = x - X0 + 1
    // column X0 - 1 of a ring row (ring column 0): twelve lanes of the last vector wave copy its units from the halo side buffer (lane = plane);
    // every other lane copies the same unit into the padding of its plane row (columns 65 .. 79 are never read): no branch
    const bool edge_copy = v == 3 && lane < kPlanes;
    const int e_unit = lane % kPlanes;
    u32x4* const ring_e = ring + e_unit * kPlaneStride + (edge_copy ? 0 : 66 + v);
    const u32x4* const halo_e = halo + e_unit;  // ... from unit [row][column 1][plane]
    const int st_voff = (X0 + 4 + lane) * 16;
    u32x4 st[kDepth][3];
    auto stage_load = [&](u32x4(&dst)[3], int chunk, int step, int jrow, bool in_loop = false) {
        if ((GANCE_UPFIRR_ABLATE & 4) && in_loop) return;
        const int brow = min(kTH * step + jrow, H + 1);  // (below the bottom border: the border again)
        const int soff = ((chunk * Hp + brow) * kPlanes + 3 * v) * Wp * 16;
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_voff, soff + k * Wp * 16, 0);
    };
    auto stage_store = [&](const u32x4(&src)[3], int slot) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (!(GANCE_UPFIRR_ABLATE & 16)) ring_w[slot * kSlotUnits + k * kPlaneStride] = src[k];
    };
    auto edge_store = [&](int slot, int hbuf, int hrow) { ring_e[slot * kSlotUnits] = halo_e[hbuf * kHaloUnits + (hrow * kHaloCols + 1) * kPlanes]; };
    u32x4 edge_n;  // (the edge unit of the row written in the NEXT period: read one period ahead, so that no period waits for its own LDS round trip)
    // halo side buffer of a chunk: 240 units [row][column][plane], one per lane (the lanes beyond the 240 repeat the first ones)
    const int h_task = vtid % kHaloUnits;
    const int h_plane = h_task % kPlanes, h_col = (h_task / kPlanes) & 3, h_row = h_task / (kPlanes * kHaloCols);
    const int h_colunits = X0 + 2 + (h_col & 1) + (h_col >> 1) * 65;  // bordered columns of X0 - 2, X0 - 1, X0 + 63, X0 + 64
    u32x4* const halo_w = halo + h_task;
    u32x4 hreg;
    auto halo_load = [&](int step, int chunk) {
        const int brow = min(kTH * step + h_row, H + 1);
        const int voff = (((chunk * Hp + brow) * kPlanes + h_plane) * Wp + h_colunits) * 16;
        hreg = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff, 0, 0);
    };
    auto halo_store = [&](int buf) { halo_w[buf * kHaloUnits] = hreg; };

    // ---- the FIR passes: thread = (column group cg of 4 output columns, row group rg of kFR output rows) of channel 4 g + v ----
    const int cg = lane % kCG;
    const int rg = lane / kCG;
    const int OW = 2 * W, OWp = OW + 8;
    const long long oplane = (long long)(2 * H + 2) * OWp;
    // (the resource starts TWO ROWS ABOVE the block's first channel plane: see upfir16_fused.hip)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + ((size_t)b * p.Cout + m0) * oplane - 2 * OWp), 0, 0x7fffffff, 0x00020000);
    const float ns2 = p.noise_strength * kSqrt2f;
    const int o_voff = (int)((v * oplane + (long long)(kFR * rg) * OWp + 4 * cg) * 4);
    // One SLICE of a step's FIR work: channel 4 g + v, output row r of both row groups (lane = (column group cg of 4 output columns, row
    // group rg)): the four T rows R = kFR rg + r .. + 3 of the channel's window (rows 0 .. 2: carried from the step before), four outputs,
    // one 16-byte store. Sixteen slices per step, dealt out over the step's row periods (run_chunk) -- whole passes of four rows made
    // four long periods per step the matrix waves waited in. Slices of a channel run in order on one wave: the last one (r = 3) moves the
    // window's last three rows to rows 0 .. 2 behind every read of them.
    f32x4 wa[4], wb[4];
    auto slice_reads = [&](int g, int r) {
        if (GANCE_UPFIRR_ABLATE & 2) return;
        const float* const rowp = window + (4 * g + v) * kChFloats + (kFR * rg + r) * kTW + 4 * cg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wa[i] = *reinterpret_cast<const f32x4*>(rowp + i * kTW);
            wb[i] = *reinterpret_cast<const f32x4*>(rowp + i * kTW + 4);
        }
    };
    auto slice_compute = [&](int g, int r, int fs) {
        if (GANCE_UPFIRR_ABLATE & 2) return;
        const int oy0 = 2 * kTH * fs - 2;              // output row of the window's row R = 0
        const int r_lo = max(0, -oy0);                 // first step: rows -2, -1 do not exist
        const int r_hi = min(kPassRows, 2 * H - oy0);  // row y' = H: only rows 2H-2, 2H-1
        const int ch = 4 * g + v;
        const float dsc = d_lds[ch] * kSqrt2f;
        const float kh0 = 0.25f * dsc, kh1 = 0.75f * dsc;
        const float bias2 = b_lds[ch] * kSqrt2f;
        const float lr6 = 0.6f * sn_lds[ch], lr4 = 0.4f * sn_lds[ch];  // leaky ReLU x the next layer's style
        const int rr = kFR * rg + r;
        if (rr >= r_lo && rr < r_hi) {
            f32x2 tv[4];
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                auto pair = [&](int i) { return c2 < 2 ? f32x2{wa[i][2 * c2], wa[i][2 * c2 + 1]} : f32x2{wb[i][2 * c2 - 4], wb[i][2 * c2 - 3]}; };
                tv[c2] = 0.25f * pair(0) + 0.75f * pair(1) + 0.75f * pair(2) + 0.25f * pair(3);
            }
            const float t[7] = {tv[0][0], tv[0][1], tv[1][0], tv[1][1], tv[2][0], tv[2][1], tv[3][0]};
            f32x4 o4;
#pragma unroll
            for (int o = 0; o < 4; ++o) o4[o] = fmaf(kh0, t[o + 3], fmaf(kh1, t[o + 2], fmaf(kh1, t[o + 1], fmaf(kh0, t[o], bias2))));
            if constexpr (kNoise) o4 += ns2 * *reinterpret_cast<const f32x4*>(nz_lds + rr * (2 * kSW) + 4 * cg);
#pragma unroll
            for (int o = 0; o < 4; ++o) o4[o] = fmaf(lr6, o4[o], lr4 * __builtin_fabsf(o4[o]));
            const int o_soff = (int)((4 * g * oplane + (long long)(oy0 + 3 + r) * OWp + 2 * X0 + 4) * 4);
            if (!(GANCE_UPFIRR_ABLATE & 256)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), o_rsrc, o_voff, o_soff, 0);
        }
        // the last three T rows of the window (R = 8 .. 10: rows 1 .. 3 of row group 1's last slice) become rows 0 .. 2
        if (r == kFR - 1 && rg == kRG - 1) {
            float* const carry_w = window + ch * kChFloats + 4 * cg;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                *reinterpret_cast<f32x4*>(carry_w + i * kTW) = wa[1 + i];
                if (cg == kCG - 1) *reinterpret_cast<f32x4*>(carry_w + i * kTW + 4) = wb[1 + i];
            }
        }
    };

    // the (step, chunk) after a (step, chunk) in the stream; past the end: the last step's first chunk again (memory that exists)
    auto advance = [&](int& step, int& chunk) {
        if (chunk + 1 < chunks) {
            ++chunk;
        } else {
            chunk = 0;
            if (step + 1 < steps) ++step;
        }
    };

    // ---- prologue: rows 0 and 1 into the ring, the halo buffer of the first chunk, rows 2 .. 11 in flight, the second chunk's halo
    // units in flight ----
    stage_load(st[0], 0, 0, 0);
    stage_load(st[1], 0, 0, 1);
    halo_load(0, 0);
    lds_barrier();  // B0: constants
    stage_store(st[0], 0);
    stage_store(st[1], 1);
    halo_store(0);
    {
        int ps = 0, pc = 0;
#pragma unroll
        for (int r = 2; r < 2 + kDepth; ++r) {  // (row r of the stream waits in st[r % kDepth])
            if (r % kRows == 0) advance(ps, pc);
            stage_load(st[r % kDepth], pc, ps, r % kRows);
        }
    }
    halo_load(0, 1);  // (chunks >= 2)
    lds_barrier();  // B1
    edge_store(0, 0, 0);
    edge_store(1, 0, 1);
    edge_n = halo_e[(2 * kHaloCols + 1) * kPlanes];
    lds_barrier();  // B2
    lds_barrier();  // B3

#pragma unroll 1
    for (int si = 0; si < steps; ++si) {
        // slice s of the step before this one is due in period 2 + s (P - 2) / 16 of this step's P = 5 chunks periods (the step's noise lands
        // in periods 0 and 1; two chunks: two slices per period from period 2 on)
        int slice = si > 0 ? 0 : kSlices;
        const int slice_span = kRows * chunks - 2;

        auto run_chunk = [&](auto parity, const int chunk) {
            constexpr int ab = decltype(parity)::value;
            int s1 = si, c1 = chunk;
            advance(s1, c1);
            int s2 = s1, c2 = c1;
            advance(s2, c2);
            int s3 = s2, c3 = c2;
            advance(s3, c3);
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                // Period (chunk, j): the matrix waves multiply row j and read row j + 1. Row j + 2 is in registers (loaded ten rows ago): write
                // it into the slot row j left; row j + 12: issue its loads into the same registers. A FIR pass that is due reads its window
                // first and computes last, the copies under its LDS latency.
                const int slot = (j + ab) & 1;
                // A slice's window rows are read into registers right behind the slice BEFORE it (the first one's in period 0, behind the dump's
                // barrier): a vector wave is alone on its SIMD's issue port for vector work, an LDS round trip it waits for is lost time.
                const int period = kRows * chunk + j;
                if (ab == 0 && j == 0 && chunk == 0 && slice < kSlices) slice_reads(0, 0);
                if (slice < kSlices && 2 + ((slice * slice_span) >> 4) <= period) {
                    slice_compute(slice >> 2, slice & 3, si - 1);
                    ++slice;
                    if (slice < kSlices) slice_reads(slice >> 2, slice & 3);
                }
                const bool here = j + 2 < kRows;   // row j + 2 is a row of this chunk
                const bool here3 = j + 3 < kRows;  // ... and row j + 3
                stage_store(st[(kRows * ab + j + 2) % kDepth], slot);
                ring_e[slot * kSlotUnits] = edge_n;
                edge_n = halo_e[(here3 ? ab : ab ^ 1) * kHaloUnits + (((j + 3) % kRows) * kHaloCols + 1) * kPlanes];
                if (j == 1) {
                    // the next chunk's halo units go to LDS (loaded five periods ago); the ones of the chunk after it: issue
                    halo_store(ab ^ 1);
                    if (!(GANCE_UPFIRR_ABLATE & 512)) halo_load(s2, c2);
                }
                stage_load(st[(kRows * ab + j + 2) % kDepth], here ? c2 : c3, here ? s2 : s3, (j + 2) % kRows, true);
                if (slice < kSlices && 2 + ((slice * slice_span) >> 4) <= period) {  // (two chunks per step: a second one)
                    slice_compute(slice >> 2, slice & 3, si - 1);
                    ++slice;
                    if (slice < kSlices) slice_reads(slice >> 2, slice & 3);
                }
                lds_barrier();
            }
        };
#pragma unroll 1
        for (int chunk = 0; chunk < chunks; chunk += 2) {
            run_chunk(std::integral_constant<int, 0>{}, chunk);
            run_chunk(std::integral_constant<int, 1>{}, chunk + 1);
        }
        lds_barrier();  // (the matrix waves dump the step)
    }
    // the last step's slices (position row y' = H), beside nothing
    lds_barrier();  // (its noise: the matrix waves' last act)
    slice_reads(0, 0);
#pragma unroll 1
    for (int sl = 0; sl < kSlices; ++sl) {
        slice_compute(sl >> 2, sl & 3, steps - 1);
        if (sl + 1 < kSlices) slice_reads((sl + 1) >> 2, (sl + 1) & 3);
    }
}

bool upfirr_supported(int cin, int cout, int H, int W) {
    return H == W && W % kSW == 0 && H % kTH == 0 && cin % (2 * kKC) == 0 && cout % kBM == 0 && cin <= 512;
}

void upfirr_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* a) {
    (void)num_cus;
    a->m_tiles = cout / kBM;
    a->strips = W / kSW;
    a->step_rows = kTH;
    a->segs = 1;
    a->rows_per_seg = H;
    a->total_blocks = B * a->m_tiles * a->strips;
    a->stagger_phases = 1;
    a->stagger_ticks = 0;
    a->debug_flags = 0;
}

__global__ __launch_bounds__(512, 1) void upfirr_fused_kernel(const UpFirArgs p) { upfirr_body<false>(p); }
__global__ __launch_bounds__(512, 1) void upfirr_fused_noise_kernel(const UpFirArgs p) { upfirr_body<true>(p); }

// The layer's input as the K loop wants it: x (zero-bordered fp32 [B][Cin][H+2][W+8]) times the layer's style (s == nullptr: the
// producer has multiplied it in), every value split into three bf16 parts (x = x0 + x1 + x2 exactly, round to nearest even each time:
// upfir_split.hip), as 16-byte units of 8 channels: out[b][chunk of 32][bordered row][plane = part * 4 + k-group][bordered column].
// One thread per (sample, chunk, row, k-group, column): eight coalesced dword loads, three coalesced 16-byte stores; the borders are split
// with everything else (zeros stay zeros). 4 + 6 bytes of HBM traffic per value -- the price of keeping the split out of the up
// kernel's lock-stepped vector waves until the producing kernel's epilogue writes this image itself.
__global__ __launch_bounds__(256) void upfirr_split_activation_kernel(const float* __restrict__ x, long long x_b_stride, const float* __restrict__ s, int s_stride,
                                                                       u32x4* __restrict__ out, int chunks, int Hp, int Wp, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % Wp);
    long long t = i / Wp;
    const int kg = (int)(t & 3);
    t >>= 2;
    const int row = (int)(t % Hp);
    t /= Hp;
    const int chunk = (int)(t % chunks);
    const int b = (int)(t / chunks);
    const float* const src = x + (size_t)b * x_b_stride + ((size_t)(chunk * kKC + kg * 8) * Hp + row) * Wp + col;
    unsigned raw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float val = src[(size_t)e * Hp * Wp];
        if (s != nullptr) val *= s[(size_t)b * s_stride + chunk * kKC + kg * 8 + e];
        raw[e] = __builtin_bit_cast(unsigned, val);
    }
    u32x4 part[3];
    split_unit(raw, part);
    u32x4* const dst = out + ((((size_t)b * chunks + chunk) * Hp + row) * kPlanes + kg) * Wp + col;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[(size_t)q * 4 * Wp] = part[q];
}

size_t upfirr_units_bytes(int B, int cin, int H, int W) { return (size_t)B * cin * (H + 2) * (W + 8) * 6; }

hipError_t launch_upfirr_split_activation(const float* x, long long x_b_stride, const float* s, int s_stride, void* out, int B, int cin, int H, int W, hipStream_t stream) {
    const int chunks = cin / kKC, Hp = H + 2, Wp = W + 8;
    const long long total = (long long)B * chunks * Hp * 4 * Wp;
    hipLaunchKernelGGL(upfirr_split_activation_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, x_b_stride, s, s_stride, reinterpret_cast<u32x4*>(out),
                       chunks, Hp, Wp, total);
    return hipGetLastError();
}

hipError_t launch_upfir_split_roles(const UpFirArgs& args, hipStream_t stream) {
    using Kernel = void (*)(const UpFirArgs);
    static const Kernel kernels[2] = {upfirr_fused_kernel, upfirr_fused_noise_kernel};
    static PerDeviceInt ready;  // the dynamic-LDS opt-in is per device
    int unused = 0;
    const hipError_t e = ready.get(
        [&](int, int* value) {
            *value = 1;
            for (int i = 0; i < 2; ++i) {
                const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[i]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
                if (err != hipSuccess) return err;
            }
            return hipSuccess;
        },
        &unused);
    if (e != hipSuccess) return e;
    if (!upfirr_supported(args.Cin, args.Cout, args.H, args.W) || args.segs != 1 || args.step_rows != kTH || args.x_units == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kernels[args.noise != nullptr ? 1 : 0], dim3(args.total_blocks), dim3(kThreads), kLdsBytes, stream, args);
    return hipGetLastError();
}

}  // namespace gance
