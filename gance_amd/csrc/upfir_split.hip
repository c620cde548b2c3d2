// Conv0_up in ONE kernel, third form (round 5): the K loop on the bf16 matrix cores from SPLIT operands, fp32 accuracy.
//
// Same layer, same decomposition, same epilogue and same layout contracts as upfir16_fused.hip (read its header first): stride-2
// transposed modulated 3x3 convolution as four parity classes, [1,3,3,1] x [1,3,3,1] FIR, noise, bias, leaky ReLU, one launch, the
// (2H+1)^2 intermediate T never in HBM; a block owns 16 output channels of a strip of 64 position columns of one sample and sweeps it
// top to bottom in steps of 8 position rows; the last three T rows of a step are carried in LDS. Replaces, for the reference's
// synthesis call (gance/network_interface/network_functions.py:168), the un-vendored `upsample_conv_2d` + `fused_bias_act` pair
// (SURVEY.md section 8 a18).
//
// What is different: the products. gfx950's fp32 MFMA runs at the fp32 VECTOR rate on the vector ALUs (157 TFLOP/s; the pair form of
// upfir16_fused.hip executes at 0.56 of it and its FIR epilogue cannot run beside it). v_mfma_f32_16x16x32_bf16 does 8192 MACs in 16
// cycles where v_mfma_f32_16x16x4_f32 does 1024 in 32. Every fp32 operand is split into THREE bf16 numbers that hold its 24 mantissa
// bits (x = x0 + x1 + x2, each part the round-to-nearest-even bf16 of what is left: exact for every finite x whose parts stay normal,
// fp32's own exponent range), and a product is the sum of the six largest of the nine part products (x0 w0, x0 w1, x1 w0, x1 w1,
// x0 w2, x2 w0; the three left out are below 2^-24 |x w|), each exact in the MFMA's fp32 accumulator, smallest first. Measured
// against the fp64 oracle this is as close as the fp32 MFMA's own summation order (tools/experiments/bf16_split_error.py: 4.0e-7 vs
// 1.2e-6 of the result's range at K = 4608; tools/experiments/upconv_bf16x3.hip on the GPU: 3.6e-7). Six terms cost 6/16 of the
// fp32 matrix time.
//
// What that needs, and why the kernel looks as it does:
//   * 6 bytes per value and k-steps of 32 input channels: the haloed patch of a step (9 rows x 66 columns x 32 channels) would be 114
//     KB of LDS. So patch ROWS stream through a three-slot ring (15 KB each): patch row j of a step feeds the dy = 0 taps of position
//     row j - 1 and the dy = -1 taps of position row j, 54 MFMAs per wave and row, one barrier per row. A wave owns one tile column
//     (16 positions) of ALL 8 rows of the step, so every wave works on the row that is resident.
//   * the split runs WHILE STAGING, from the fp32 activations as every other kernel stores them ([C][H+2][W+8], zero border): wave kg
//     loads, for column X0 + lane, the eight channels of k-group kg of the row five rows ahead (eight coalesced dword loads), and
//     three rows later splits them (44 vector instructions: v_cvt_pk_bf16_f32 on channel pairs, shift / mask, subtract) and writes
//     three 16-byte units: a bf16 MFMA holds the vector issue for 8 of its 16 cycles, the split rides in its shadow.
//   * the weight fragments of a chunk (9 taps x 3 parts: 108 registers) live in REGISTERS, loaded from the split image in global
//     memory one chunk ahead (27 x 16 bytes per lane, three per row): 128 accumulators + two sets of them + fragments and staging
//     registers are > 256, so one wave per SIMD, one block per CU -- the FIR epilogue stays serial with the K loop as in the
//     32-channel kernel; what pays for it is a K loop 0.5 x as long.
//   * the two halo position columns (x' = X0 - 1: odd column parity only; x' = X0 + 64) are one extra 16-slot tile as before, but
//     its rows are never resident together: a per-chunk side buffer holds columns X0 - 2, X0 - 1, X0 + 63, X0 + 64 of the chunk's
//     nine rows (144 staging tasks per chunk, written half a chunk ahead), read once per chunk by the 54 halo MFMAs (one class per
//     wave). The same buffer serves the one fragment the ring does not hold: column X0 - 1 for the dx = -1 taps of the first lane.
//   * position row y' = H (T row 2H, the "flush step" of the other kernels) is the dy = -1 taps of the LAST step's ninth patch row:
//     18 more MFMAs in that step's last row and one more epilogue pass, no step of its own.
// A block sweeps the image's whole height, or -- for calls too small to fill the chip with whole images -- a row SEGMENT of it behind one
// "priming" step (the step above the segment, computed for its last three T rows only: upfirs_plan picks the number of segments).
//
// Weight image (upfirs_arrange_weights): [channel tile of 16][chunk of 32][tap][part][k-group][row m][8 channels] bf16: a lane's A
// fragment (row m = lane % 16, k-group = lane / 16) is 16 contiguous bytes, a wave's load 1 KB; MFMA row m = 4 q + r holds channel
// 4 r + q of the tile, so accumulator REGISTER r of every lane is channel group r (the epilogue dumps one register per pass).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"

// Timing ablations (wrong results; only in builds with -DGANCE_UPFIRS_ABLATE=<flags>, Makefile target ../libgance_hip_upfirsab<flags>.so, used
// through GANCE_HIP_LIBRARY): 1 no split arithmetic (the raw values are written), 2 no epilogue, 4 no global loads of patch rows in the row
// loop, 8 no MFMAs, 16 no LDS writes of staged rows, 32 no barrier per row, 64 no halo tile / edge copy, 128 no output stores
#ifndef GANCE_UPFIRS_ABLATE
#define GANCE_UPFIRS_ABLATE 0
#endif

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kBM = 16;             // output channels per block
constexpr int kKC = 32;             // input channels per chunk = one k-step of v_mfma_f32_16x16x32_bf16
constexpr int kSW = 64;             // position columns per strip
constexpr int kTH = 8;              // position rows per step
constexpr int kRows = kTH + 1;      // patch rows of a step: input rows y0 - 1 .. y0 + 7
constexpr int kPlanes = 12;         // 16-byte units per position: part x k-group
constexpr int kPlaneStride = 80;    // units per plane row in the ring: a multiple of 16, so the k-groups of a fragment read fall on distinct banks
constexpr int kSlotUnits = kPlanes * kPlaneStride;
constexpr int kRing = 3;
constexpr int kHaloCols = 4;        // input columns X0 - 2, X0 - 1, X0 + 63, X0 + 64
constexpr int kHaloUnits = kRows * kHaloCols * kPlanes;
constexpr int kHaloTasks = kRows * kHaloCols * 4;  // (row, column, k-group): 144
#ifndef GANCE_UPFIRS_WEAVE
#define GANCE_UPFIRS_WEAVE 2  // other instructions dealt out per MFMA in a row's scheduling region (measured: 1 and 3 are within 1 % of 2)
#endif
#ifndef GANCE_UPFIRS_MIN_SEG_ROWS
#define GANCE_UPFIRS_MIN_SEG_ROWS 16  // shortest row segment of a block (upfirs_plan)
#endif
#ifndef GANCE_UPFIRS_DEPTH
#define GANCE_UPFIRS_DEPTH 6
#endif
constexpr int kDepth = GANCE_UPFIRS_DEPTH;  // patch rows in flight between their global loads and their LDS writes (3 or 6: 8 registers each)
constexpr int kCarryRows = 3;
constexpr int kPassCh = 4;          // channels per epilogue pass
constexpr int kPassRows = 8;        // T rows per pass: four position rows
constexpr int kTW = 2 * kSW + 4;    // T window row: T columns 2 X0 - 1 .. 2 X0 + 129 (+ pad)
constexpr int kCG = kSW / 2;        // column groups of 4 output columns
constexpr int kRG = 64 / kCG;       // row groups of a wave's 64 filter threads
constexpr int kFR = kPassRows / kRG;  // output rows per filter thread
constexpr int kWin = kFR + 3;
constexpr int kCarryFloats = kBM * kCarryRows * kTW;
constexpr int kStageFloats = kPassCh * kPassRows * kTW;
constexpr int kNzPieces = 2 * kPassRows * 2 * kSW / 256;  // noise of a step's 16 output rows in 1 KiB DMA pieces
constexpr float kSqrt2f = 1.4142135623730951f;
static_assert((2 * kRows) % kDepth == 0 && kDepth + 2 < kRows && kRows % kRing == 0, "ring slots and staging registers rotate with the unrolled rows of a chunk pair");

// LDS (bytes): ring | halo side buffers (two chunks) | T window of a pass | the step's noise | carry | style [Cin] | demod | bias | next style
constexpr size_t kRingBytes = (size_t)kRing * kSlotUnits * 16;
constexpr size_t kHaloBytes = (size_t)2 * kHaloUnits * 16;
constexpr int kWUnits = 27 * 64;  // 16-byte units of a chunk's weight fragments: [tap][part][lane]
constexpr size_t kWBytes = (size_t)kWUnits * 16;
constexpr size_t lds_bytes(int cin) {
    return kRingBytes + kHaloBytes + kWBytes + sizeof(float) * ((size_t)kStageFloats + kNzPieces * 256 + kCarryFloats + cin + 3 * kBM);
}

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

__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Ring column of position P = x - X0 (0 .. 63); column 64 holds x = X0 - 1. Staging is two 16-byte loads per lane (4 columns x 2 channels:
// a wave may have 63 vector-memory operations outstanding, and with dword loads -- 256 bytes per instruction, six rows of eight in flight
// beside the 27 weight loads -- the K loop was bound by that window, round 5), split, then a 4 x 4 transpose over the four lane rows
// (v_permlane32_swap, v_permlane16_swap): lane (row r, column group g) ends up with the units of position 4 g + r. Eight neighbouring
// lanes then write positions 4 apart, sixteen lanes of a fragment read 16 neighbouring positions. Bits of the column: c0 = P2, c1 = P3,
// c2 = P4 ^ P0, c3 = P1, c4 = P0, c5 = P5 -- c[2:0] runs through all 8 bank quads over P[4:2] (the 8 lanes of a ds_write_b128 group) and
// c[3:0] through all 16 over P[3:0] (the 16 lanes of a ds_read_b128 group): both conflict-free.
__device__ __forceinline__ int ring_column(int P) {
    return ((P >> 2) & 3) | ((((P >> 4) ^ P) & 1) << 2) | (((P >> 1) & 1) << 3) | ((P & 1) << 4) | (P & 32);
}
constexpr int kEdgeColumn = 64;

// two neighbouring channels of one position -> their three bf16 parts, packed (a in the low half): round to nearest even each
// time, the residuals exact in fp32 (Sterbenz: a part and what it was rounded from agree in their leading bits)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    auto pack = [](float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2)); };
    p0 = pack(a, b);
    const float ra = a - __builtin_bit_cast(float, p0 << 16), rb = b - __builtin_bit_cast(float, p0 & 0xffff0000u);
    p1 = pack(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, p1 << 16), sb = rb - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = pack(sa, sb);
}

// eight fp32 values (the channels of one k-group at one position) -> the position's three 16-byte units
__device__ __forceinline__ void split_unit(const unsigned (&raw)[8], u32x4 (&part)[3]) {
    unsigned w[3][4];
    if (GANCE_UPFIRS_ABLATE & 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) part[q] = u32x4{raw[q], raw[q + 1], raw[q + 2], raw[q + 3]};
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(__builtin_bit_cast(float, raw[2 * e]), __builtin_bit_cast(float, raw[2 * e + 1]), w[0][e], w[1][e], w[2][e]);
#pragma unroll
    for (int q = 0; q < 3; ++q) part[q] = u32x4{w[q][0], w[q][1], w[q][2], w[q][3]};
}

}  // namespace

template <bool kPre, bool kNoise>
__device__ __forceinline__ void upfirs_body(const UpFirArgs& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const ring = reinterpret_cast<u32x4*>(smem_raw);
    u32x4* const halo = ring + kRing * kSlotUnits;
    u32x4* const w_lds = halo + 2 * kHaloUnits;                            // [tap][part][lane]: the weight fragments of the chunk after the one being multiplied
    float* const stage = reinterpret_cast<float*>(w_lds + kWUnits);        // [4 ch][8 rows][kTW]
    float* const nz_lds = stage + kStageFloats;                            // [16 output rows][2 kSW]
    float* const carry = nz_lds + kNzPieces * 256;                         // [16 ch][3 rows][kTW]
    float* const s_lds = carry + kCarryFloats;                             // style [Cin] (absent when the input is pre-scaled)
    float* const d_lds = s_lds + (kPre ? 0 : p.Cin);
    float* const b_lds = d_lds + kBM;
    float* const sn_lds = b_lds + kBM;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n16 = lane & 15, kg = lane >> 4;

    // ---- block -> (sample, strip, channel tile); blocks of one XCD take contiguous ids so that the channel tiles of one strip (same
    // input rows) and neighbouring strips share its L2 ----
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
    const int seg = id % p.segs;  // row segment of the image (calls too small to fill the chip with whole images: upfirs_plan)
    const int b = id / p.segs;
    const int m0 = m_tile * kBM;
    const int X0 = strip * kSW;
    const int H = p.H, W = p.W;
    const int Hp = H + 2, Wp = W + 8;
    const int HpWp4 = Hp * Wp * 4;
    const int chunks = p.Cin / kKC;
    // A block sweeps position rows [seg * rows_per_seg, (seg + 1) * rows_per_seg) in steps of 8. A segment below the image's top runs the
    // step ABOVE it first ("priming": its last three T rows are the carry the segment's first FIR window needs; nothing of it is stored).
    const int steps = H / kTH;  // of the whole image
    const int s_begin = seg * (p.rows_per_seg / kTH), s_end = s_begin + p.rows_per_seg / kTH;
    const int s_first = seg > 0 ? s_begin - 1 : s_begin;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.x_b_stride), 0, p.Cin * Hp * Wp * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.w) + (size_t)m_tile * chunks * 27 * 1024), 0, chunks * 27 * 1024, 0x00020000);

    // ---- the stream of patch rows: (step, chunk, row j) in the order they are consumed; the producer runs 2 + kDepth rows ahead,
    // so the row it loads is row (j + 2 + kDepth) % 9 of the consumer's chunk or of the chunk after it: known where the load is written ----
    // per-lane LDS bases: everything that changes with the row, the chunk parity or the part is a compile-time offset from one of them
    // (the instruction's immediate): per-row address registers would not fit beside 144 accumulators and 108 weight registers
    const int st_pair = lane >> 4, st_g = lane & 15;                              // staging task: channel pair 4 wave + st_pair, columns X0 + 4 st_g .. + 3
    u32x4* const ring_w = ring + wave * kPlaneStride + ring_column(4 * st_g + st_pair);  // ... after the transpose: the units of position 4 st_g + st_pair, k-group = wave
    // fragment reads of position 16 wave + n16 (dx = 0) and of the position to its left (dx = -1; left of the strip: the edge column)
    const int m_pos = 16 * wave + n16;
    const u32x4* const ring_r0 = ring + kg * kPlaneStride + ring_column(m_pos);
    const u32x4* const ring_r1 = ring + kg * kPlaneStride + (m_pos > 0 ? ring_column(m_pos - 1) : kEdgeColumn);
    // column X0 - 1 of a ring row (the dx = -1 fragment of the strip's first lane) is not a main staging task (64 lanes = 64 columns):
    // twelve lanes of wave 3 copy its units (lane = part * 4 + k-group) from the halo side buffer of the row's chunk; every other lane
    // copies the same unit into the padding of its plane row (columns 65 .. 79 are never read): no branch in the row's instruction stream
    const bool edge_copy = wave == 3 && lane < kPlanes;
    const int e_unit = lane % kPlanes;
    u32x4* const ring_e = ring + (e_unit >> 2) * 4 * kPlaneStride + (e_unit & 3) * kPlaneStride + (edge_copy ? kEdgeColumn : 66 + wave);
    const u32x4* const halo_e = halo + e_unit;                                    // ... from unit [row][column 1][part * 4 + k-group]
    const u32x4* const halo_r = halo + ((n16 & 7) * kHaloCols + 2 * (n16 >> 3)) * kPlanes + kg;  // halo tile slot n16 = (side, position row)
    const u32x4* const halo_f = halo + ((kRows - 1) * kHaloCols + 2 * (n16 >> 3)) * kPlanes + kg;  // ... of position row y' = H: the chunk's last row
    // main staging task of this lane: column X0 + lane, k-group = wave: eight dword loads (buffer row `brow` of the bordered tensor)
    const int st_voff = (2 * st_pair * Hp * Wp + X0 + 4 + 4 * st_g) * 4;
    unsigned st[kDepth][8];
    auto stage_load = [&](unsigned(&dst)[8], int chunk, int brow, bool in_loop = false) {
        if ((GANCE_UPFIRS_ABLATE & 4) && in_loop) return;
        const int soff = ((chunk * kKC + wave * 8) * Hp + brow) * Wp * 4;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const u32x4 q4 = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, st_voff, soff + e * HpWp4, 0);  // channel 2 pair + e, four columns
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[2 * c + e] = q4[c];  // (dst[2 c], dst[2 c + 1]: the pair at column c)
        }
    };
    // the main tasks hold two channels at four positions (the halo tasks eight channels of one position: scale8)
    auto scale_pair = [&](unsigned(&raw)[8], int chunk) {
        if constexpr (!kPre) {
            const f32x2 s2 = *reinterpret_cast<const f32x2*>(s_lds + chunk * kKC + 8 * wave + 2 * st_pair);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                raw[2 * c] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, raw[2 * c]) * s2[0]);
                raw[2 * c + 1] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, raw[2 * c + 1]) * s2[1]);
            }
        }
    };
    // 4 x 4 transpose over the four lane rows (16 lanes each): register c of row r (channel pair r, position c of the lane's four) becomes
    // register i of row r = what register r of row i was (pair i, position r)
    auto transpose_rows = [](u32x4& u) {
        auto a02 = __builtin_amdgcn_permlane32_swap(u[0], u[2], false, false);
        auto a13 = __builtin_amdgcn_permlane32_swap(u[1], u[3], false, false);
        auto b01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
        auto b23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
        u = u32x4{b01[0], b01[1], b23[0], b23[1]};
    };
    // (style of the channels this lane stages: the consumer's chunk is known where the write happens)
    auto scale8 = [&](unsigned(&raw)[8], int chunk, int group) {
        if constexpr (!kPre) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(s_lds + chunk * kKC + group * 8);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(s_lds + chunk * kKC + group * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                raw[e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, raw[e]) * s0[e]);
                raw[4 + e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, raw[4 + e]) * s1[e]);
            }
        }
    };
    auto stage_store = [&](unsigned(&raw)[8], int slot, int chunk) {
        scale_pair(raw, chunk);
        u32x4 part[3];
        split_unit(raw, part);
#pragma unroll
        for (int q = 0; q < 3; ++q) transpose_rows(part[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (!(GANCE_UPFIRS_ABLATE & 16)) ring_w[slot * kSlotUnits + q * 4 * kPlaneStride] = part[q];
    };
    auto edge_store = [&](int slot, int hbuf, int hrow) { if (!(GANCE_UPFIRS_ABLATE & 64)) ring_e[slot * kSlotUnits] = halo_e[hbuf * kHaloUnits + (hrow * kHaloCols + 1) * kPlanes]; };
    // halo side buffer of a chunk: task = (row, column, k-group); unit [(row * 4 + column) * 12 + part * 4 + k-group]
    // (every lane has one: the lanes beyond the 144 repeat the first ones -- the same values to the same place, no branch)
    const int h_task = tid < kHaloTasks ? tid : tid - kHaloTasks;
    const int h_col = h_task & 3, h_kg = (h_task >> 2) & 3, h_row = h_task >> 4;
    const int h_voff = ((h_kg * 8 * Hp + h_row) * Wp + X0 + 2 + (h_col & 1) + (h_col >> 1) * 65) * 4;  // bordered columns of X0 - 2, X0 - 1, X0 + 63, X0 + 64
    u32x4* const halo_w = halo + (h_row * kHaloCols + h_col) * kPlanes + h_kg;
    unsigned hraw[8];
    auto halo_load = [&](int step, int chunk) {
        const int soff = (chunk * kKC * Hp + kTH * step) * Wp * 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) hraw[e] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, h_voff, soff + e * HpWp4, 0);
    };
    auto halo_store = [&](int chunk, int buf) {
        scale8(hraw, chunk, h_kg);
        u32x4 part[3];
        split_unit(hraw, part);
#pragma unroll
        for (int q = 0; q < 3; ++q) halo_w[buf * kHaloUnits + q * 4] = part[q];
    };

    // ---- weight fragments: A[tap][part], 16 bytes per lane each, from the split image. ONE set (108 registers; two sets spilled,
    // and a scratch reload inside the row loop is a vector-memory load whose wait drains the rows in flight): the next chunk's
    // fragment of a tap is loaded in the chunk's LAST row right behind the tap's last MFMAs -- the first row of a chunk only
    // needs the dy = -1 taps, which that row handles first, and the image is hot in L2 (every block of the layer reads it) ----
    u32x4 A[9][3];
    const int a_voff = lane * 16;
    // The fragments of the NEXT chunk come through LDS (round 5): the four waves hold the same 27 KB, and four copies per chunk through the
    // CU's vector-memory path were a quarter of its traffic and 27 of the 63 operations a wave may have outstanding. Each wave fetches seven
    // (six) pieces by LDS-DMA in row 0 of a chunk (the buffer was read in the last row of the chunk before); they are older than every
    // staging load that is still in flight when the last row needs them, so the counted wait for them (in front of row 7's barrier: all
    // but the youngest 24 operations = row 0's halo loads and rows 0 .. 7's staging loads) waits for nothing else.
    auto weights_dma = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (k < 6 || wave < 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(w_lds + (wave + 4 * k) * 64), 16, lane * 16, (chunk * 27 + wave + 4 * k) * 1024, 0, 0);
    };
    const u32x4* const w_r = w_lds + lane;
    auto read_a3 = [&](int t, u32x4(&dst)[3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q] = w_r[(t * 3 + q) * 64];
    };
    auto load_a3 = [&](int chunk, int t, u32x4(&dst)[3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, a_voff, (chunk * 27 + t * 3 + q) * 1024, 0);
    };

    // ---- B fragments of a patch row: [dx][part]; the lane's position is column 16 wave + n16 of the strip ----
    u32x4 Bf[2][2][3];
    auto load_b = [&](int slot, u32x4(&dst)[2][3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            dst[0][q] = ring_r0[slot * kSlotUnits + q * 4 * kPlaneStride];
            dst[1][q] = ring_r1[slot * kSlotUnits + q * 4 * kPlaneStride];
        }
    };

    // ---- prologue: constants, the zeroed carry, halo buffer of the first chunk, rows 0 and 1 in the ring, rows 2..4 in flight ----
    stage_load(st[0], 0, kTH * s_first + 0);
    stage_load(st[1], 0, kTH * s_first + 1);
    load_a3(0, 0, A[0]);
    halo_load(s_first, 0);
    if constexpr (!kPre)
        for (int i = tid; i < p.Cin; i += 256) s_lds[i] = p.s[(size_t)b * p.s_stride + i];
    if (tid < kBM) {
        d_lds[tid] = p.d[(size_t)b * p.d_stride + m0 + tid];
        b_lds[tid] = p.bias[m0 + tid];
        sn_lds[tid] = p.s_next != nullptr ? p.s_next[(size_t)b * p.s_stride + m0 + tid] : 1.0f;
    }
    for (int i = tid; i < kCarryFloats; i += 256) carry[i] = 0.f;
#pragma unroll
    for (int t = 1; t < 9; ++t) load_a3(0, t, A[t]);
    if constexpr (!kPre) lds_barrier();  // (the style vector is read by the first writes)
    stage_store(st[0], 0, 0);
    stage_store(st[1], 1, 0);
    halo_store(0, 0);
#pragma unroll
    for (int r = 2; r < 2 + kDepth; ++r) stage_load(st[r % kDepth], 0, kTH * s_first + r);  // (row r of the stream waits in st[r % kDepth])
    lds_barrier();
    edge_store(0, 0, 0);
    edge_store(1, 0, 1);
    lds_barrier();
    load_b(0, Bf[0]);

    const int OW = 2 * W, OWp = OW + 8;
    const long long oplane = (long long)(2 * H + 2) * OWp;
    // (the resource starts TWO ROWS ABOVE the block's first channel plane: see upfir16_fused.hip)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + ((size_t)b * p.Cout + m0) * oplane - 2 * OWp), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t nz_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(kNoise ? p.noise + (size_t)b * p.noise_b_stride : nullptr), 0, kNoise ? (2 * H) * (2 * W) * 4 : 0, 0x00020000);
    const float ns2 = p.noise_strength * kSqrt2f;

    // part products, smallest first: {x part, w part}
    constexpr int kTerms[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
    // the taps in an order whose neighbours accumulate into different tiles: (y'-1 EE) (y'-1 EO) (y'-1 OE) (y'-1 OO) (y' EE) (y' EO) (y'-1 EE) (y'-1 OE) (y' EE)
    constexpr int kTapTurn[9] = {0, 4, 6, 8, 2, 5, 1, 7, 3};

#pragma unroll 1
    for (int si = s_first; si < s_end; ++si) {
        const int y0 = kTH * si;
        const bool last_step = si + 1 == steps;  // (of the image: position row y' = H follows)
        const bool priming = si < s_begin;
        f32x4 acc[kTH][4];
#pragma unroll
        for (int r = 0; r < kTH; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 accf[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // position row y' = H (last step): classes EE, EO
        f32x4 acch = f32x4{0.f, 0.f, 0.f, 0.f}, acchf = f32x4{0.f, 0.f, 0.f, 0.f};  // halo tile (this wave's class) and its row y' = H

        auto run_chunk = [&](auto parity, const int chunk) {
            constexpr int ab = decltype(parity)::value;
            // the chunk after this one in the stream
            const int n_chunk = chunk + 1 < chunks ? chunk + 1 : 0;
            const int n_step = chunk + 1 < chunks ? si : (last_step ? si : si + 1);
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                // Row j of this chunk is in ring slot j % 3 and its fragments are in Bf[cur]. Row j + 1 is in the ring (written before
                // the last barrier): read its fragments now; row j + 2 is in registers (loaded three rows ago): split and write it;
                // row j + 5: issue its loads into the registers that row j + 2 leaves. (9 rows per chunk is odd: the fragment buffers
                // alternate by j + chunk parity; the chunk loop is unrolled by two and a step has an even number of chunks)
                const int cur = (j + ab) & 1;
                if (j == 0) weights_dma(n_chunk);
                if (j == 0) halo_load(n_step, n_chunk);
                if (j == 4) halo_store(n_chunk, ab ^ 1);
                load_b((j + 1) % kRing, Bf[cur ^ 1]);
                edge_store((j + 2) % kRing, j + 2 < kRows ? ab : ab ^ 1, (j + 2) % kRows);
                // (row index in the stream modulo the chunk pair: 9 ab + j; its staging registers: that modulo kDepth)
                stage_store(st[(kRows * ab + j + 2) % kDepth], (j + 2) % kRing, j + 2 < kRows ? chunk : n_chunk);
                stage_load(st[(kRows * ab + j + 2) % kDepth], j + 2 + kDepth < kRows ? chunk : n_chunk, kTH * (j + 2 + kDepth < kRows ? si : n_step) + (j + 2 + kDepth) % kRows, true);
                if (j + 1 < kRows) {
                    // Term by term ACROSS the taps, in an order that takes the row's six accumulators in turn: an MFMA that accumulates
                    // into the result of the one before it waits for it -- v_mfma_f32_16x16x32_bf16 back to back on ONE accumulator runs at
                    // 44 cycles per instruction, on two at 22, on four at 19, on 32 at 16.5 (tools/experiments/mfma_stream_peak.hip,
                    // profiles/r05_mfma_stream_peak.txt). Written tap by tap until round 5's last day; hipcc's scheduler had been pulling
                    // the chains apart by itself (no change in time), the source now says what is meant.
                    // (per accumulator the terms still arrive smallest first)
#pragma unroll
                    for (int term = 0; term < ((GANCE_UPFIRS_ABLATE & 8) ? 0 : 6); ++term)
#pragma unroll
                        for (int k = 0; k < 9; ++k) {
                            const int t = kTapTurn[k];
                            const int row = tap_dy(t) ? j : j - 1;
                            if (row < 0 || row >= kTH) continue;
                            acc[row][tap_cls(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]), __builtin_bit_cast(bf16x8, Bf[cur][tap_dx(t)][kTerms[term][0]]),
                                acc[row][tap_cls(t)], 0, 0, 0);
                        }
                    // The row is ONE branch-free scheduling region: its MFMAs, and for the rows ahead six fragment reads, the edge copy, the
                    // split of a staged row (44 vector instructions, three LDS writes) and eight loads. A bf16 MFMA holds the vector issue
                    // for 8 of its 16 cycles: dealt out two per MFMA the other instructions ride in its shadow; in a clump in front of the
                    // MFMAs (where the dependences alone would put them) they cost their full issue time.
#pragma unroll
                    for (int i = 0; i < 54; ++i) {
                        if (i >= (j == 0 ? 18 : 54)) break;                 // (row 0 only has the dy = -1 taps)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x092, GANCE_UPFIRS_WEAVE, 0);  // two of: vector ALU, vector memory, LDS
                    }
                } else {
                    // The chunk's last row, tap by tap (the dy = -1 taps first: the next chunk's first row needs them first): the row's
                    // dy = 0 products, in the image's last step the dy = -1 products of position row y' = H (EE t2, t3; EO t5), this
                    // wave's share of the chunk's halo tile (slot n16 = (side n16 / 8, position row n16 % 8); one class per wave) --
                    // and then the tap's fragments of the NEXT chunk.
                    constexpr int kOrder[9] = {2, 3, 5, 0, 1, 4, 6, 7, 8};
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const int t = kOrder[i];
                        if (!tap_dy(t)) {
#pragma unroll
                            for (int term = 0; term < 6; ++term)
                                acc[kTH - 1][tap_cls(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]), __builtin_bit_cast(bf16x8, Bf[cur][tap_dx(t)][kTerms[term][0]]),
                                    acc[kTH - 1][tap_cls(t)], 0, 0, 0);
                        } else if (last_step) {
#pragma unroll
                            for (int term = 0; term < 6; ++term)
                                accf[tap_cls(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]), __builtin_bit_cast(bf16x8, Bf[cur][tap_dx(t)][kTerms[term][0]]),
                                    accf[tap_cls(t)], 0, 0, 0);
                        }
                        if (wave == tap_cls(t) && !(GANCE_UPFIRS_ABLATE & 64)) {
                            u32x4 hf[3];
#pragma unroll
                            for (int q = 0; q < 3; ++q) hf[q] = halo_r[ab * kHaloUnits + ((1 - tap_dy(t)) * kHaloCols + 1 - tap_dx(t)) * kPlanes + q * 4];
#pragma unroll
                            for (int term = 0; term < 6; ++term)
                                acch = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]),
                                                                               __builtin_bit_cast(bf16x8, hf[kTerms[term][0]]), acch, 0, 0, 0);
                            if (last_step && tap_dy(t)) {
                                // (its row y' = H: every slot of a side reads the side's column in the chunk's last row)
#pragma unroll
                                for (int q = 0; q < 3; ++q) hf[q] = halo_f[ab * kHaloUnits + (1 - tap_dx(t)) * kPlanes + q * 4];
#pragma unroll
                                for (int term = 0; term < 6; ++term)
                                    acchf = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[t][kTerms[term][1]]),
                                                                                    __builtin_bit_cast(bf16x8, hf[kTerms[term][0]]), acchf, 0, 0, 0);
                            }
                        }
                        read_a3(t, A[t]);
                    }
                }
                if (j == kRows - 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // (the next chunk's weight fragments: see weights_dma)
                if (!(GANCE_UPFIRS_ABLATE & 32)) lds_barrier();
            }
        };
#pragma unroll 1
        for (int chunk = 0; chunk < chunks; chunk += 2) {
            run_chunk(std::integral_constant<int, 0>{}, chunk);
            run_chunk(std::integral_constant<int, 1>{}, chunk + 1);
        }

        // ---- epilogue: two halves of four position rows x four channel groups, then (last step) the pass of position row y' = H.
        // Ported from upfir16_fused.hip: only the dump differs (a wave holds a tile COLUMN of all rows: half rw = its rows 4 rw .. 4 rw + 3).
        const int elane = fresh_lane();
        const int en16 = elane & 15, eq4 = elane >> 4;
        const int dump_base = eq4 * (kPassRows * kTW) + 32 * wave + 2 * en16 + 1;
        const int hpy = wave >> 1, hpx = wave & 1;  // the halo tile's class held by this wave
        const int fc = wave;
        const int cg = elane % kCG;
        const int rg = elane / kCG;
        const int o_voff = (int)((fc * oplane + (long long)(kFR * rg) * OWp + 4 * cg) * 4);

        auto run_passes = [&](auto flush_tag) {
            constexpr bool kFlush = decltype(flush_tag)::value;
            const int ys = kFlush ? H : y0;  // first position row of the passes
            if (kNoise) {
#pragma unroll
                for (int i = 0; i < kNzPieces / 4; ++i) {
                    const int f = (wave + 4 * i) * 256 + 4 * elane;
                    const int row = f / (2 * kSW), col = f % (2 * kSW);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(nz_rsrc, (lds_ptr_t)(nz_lds + (wave + 4 * i) * 256), 16,
                                                             ((2 * ys - 2 + row) * OW + 2 * X0 + col) * 4, 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            constexpr int kRowPasses = kFlush ? 1 : 2;
#pragma unroll
            for (int rw = 0; rw < kRowPasses; ++rw) {
                const int oy0 = 2 * (ys + 4 * rw) - 2;        // output row of the pass's window row r = 0
                const int r_lo = max(0, -oy0);                 // first step: rows -2, -1 do not exist
                const int r_hi = priming ? 0 : min(kPassRows, 2 * H - oy0);  // row y' = H: only rows 2H-2, 2H-1; a priming step stores nothing
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // -- dump: accumulator register g = channel 4 g + q4 --
                    if constexpr (kFlush) {
                        // (only T row 2H exists: even row parity; the odd row of the window reads as zero)
#pragma unroll
                        for (int c = 0; c < 4; ++c) stage[dump_base + (c >> 1) * kTW + (c & 1)] = c < 2 ? accf[c][g] : 0.f;
                        if (en16 == 0 || en16 == 8) {
                            const int side = en16 >> 3;
                            if (hpy == 0 && (side == 1 || hpx == 1)) stage[eq4 * (kPassRows * kTW) + (side ? 2 * kSW + 1 + hpx : 0)] = acchf[g];
                            if (hpy == 1 && (side == 1 || hpx == 1)) stage[eq4 * (kPassRows * kTW) + kTW + (side ? 2 * kSW + 1 + hpx : 0)] = 0.f;
                        }
                    } else {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                            for (int c = 0; c < 4; ++c) stage[dump_base + (2 * r4 + (c >> 1)) * kTW + (c & 1)] = acc[4 * rw + r4][c][g];
                        const int row = en16 & 7, side = en16 >> 3;
                        if ((side == 1 || hpx == 1) && row / 4 == rw)
                            stage[eq4 * (kPassRows * kTW) + (2 * (row % 4) + hpy) * kTW + (side ? 2 * kSW + 1 + hpx : 0)] = acch[g];
                    }
                    lds_barrier();

                    // -- filter (see upfir16_fused.hip): window row i of row group rg = row R = kFR rg + i of (three carried T rows, the pass's eight) --
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
                    {
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
                                if (kNoise) v += ns2 * *reinterpret_cast<const f32x4*>(nz_lds + (kPassRows * rw + rr) * (2 * kSW) + 4 * cg);
#pragma unroll
                                for (int o = 0; o < 4; ++o) v[o] = fmaf(lr6, v[o], lr4 * __builtin_fabsf(v[o]));
                                if (!(GANCE_UPFIRS_ABLATE & 128)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, o_voff, o_soff + r * OWp * 4, 0);
                                else asm volatile("" ::"v"(v));
                            }
                        }
                    }
                    lds_barrier();
                    // -- the last three T rows of the window become the carry of these 4 channels --
                    if (!kFlush && rg == kRG - 1) {
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
        if (GANCE_UPFIRS_ABLATE & 2) {  // (the accumulators stay alive: a store that never happens)
            float sum = acch[0] + acchf[0] + accf[0][0] + accf[1][0];
#pragma unroll
            for (int r = 0; r < kTH; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) sum += acc[r][c][0] + acc[r][c][3];
            if (sum == 12345.678f) p.out[tid] = sum;
            continue;
        }
        run_passes(std::false_type{});
        if (last_step) {
            lds_barrier();  // (the carry of the last pass is read by every row group of the flush pass)
            run_passes(std::true_type{});
        }
    }
}

bool upfirs_supported(int cin, int cout, int H, int W) {
    return H == W && W % kSW == 0 && H % kTH == 0 && cin % (2 * kKC) == 0 && cout % kBM == 0 && cin <= 512;
}

size_t upfirs_weight_floats(int cin, int cout) { return (size_t)(cout / kBM) * (cin / kKC) * 27 * 256; }  // (27 KB per channel tile and chunk)

static unsigned short bf16_rne_host(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_value_host(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// w_in: scaled filter [tap = wy*3+wx][cin][cout]; w_out: [m tile of 16][chunk of 32][slot t][part][k-group][row m][8 channels] bf16,
// slot t = filter tap up_tap_weight[t], MFMA row m = 4 q + r holds channel 4 r + q of the tile
void upfirs_arrange_weights(const float* w_in, int cin, int cout, const int* up_tap_weight, float* w_out) {
    unsigned short* const out = reinterpret_cast<unsigned short*>(w_out);
    const int m_tiles = cout / kBM, chunks = cin / kKC;
    for (int mt = 0; mt < m_tiles; ++mt)
        for (int ch = 0; ch < chunks; ++ch)
            for (int t = 0; t < 9; ++t)
                for (int k = 0; k < kKC; ++k)
                    for (int m = 0; m < kBM; ++m) {
                        const int channel = 4 * (m & 3) + (m >> 2);
                        const float v = w_in[((size_t)up_tap_weight[t] * cin + ch * kKC + k) * cout + mt * kBM + channel];
                        unsigned short part[3];
                        part[0] = bf16_rne_host(v);
                        const float r1 = v - bf16_value_host(part[0]);
                        part[1] = bf16_rne_host(r1);
                        part[2] = bf16_rne_host(r1 - bf16_value_host(part[1]));
                        for (int q = 0; q < 3; ++q)
                            out[((((((size_t)mt * chunks + ch) * 9 + t) * 3 + q) * 4 + k / 8) * 16 + m) * 8 + k % 8] = part[q];
                    }
}

void upfirs_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* a) {
    a->m_tiles = cout / kBM;
    a->strips = W / kSW;
    a->step_rows = kTH;
    // Row segments only where whole images leave CUs idle: each costs a priming step of 8 rows, so segments of at least
    // GANCE_UPFIRS_MIN_SEG_ROWS rows, powers of two, until every CU has a block (GANCE_TUNE_UPFIR_SPLIT_SEGS=0: never)
    static const bool segments = [] { const char* v = std::getenv("GANCE_TUNE_UPFIR_SPLIT_SEGS"); return !(v && std::atoi(v) == 0); }();
    // ... the number that minimises (rounds of blocks over the CUs) x (rows a block sweeps, its priming step included)
    int segs = 1;
    const int base = B * a->m_tiles * a->strips;
    long long best = (long long)((base + num_cus - 1) / num_cus) * H;
    for (int n = 2; segments && H / n >= GANCE_UPFIRS_MIN_SEG_ROWS && (H / n) % kTH == 0 && base * (n / 2) < num_cus; n *= 2) {
        const long long cost = (long long)((base * n + num_cus - 1) / num_cus) * (H / n + kTH);
        if (cost < best) best = cost, segs = n;
    }
    a->segs = segs;
    a->rows_per_seg = H / segs;
    a->total_blocks = base * segs;
    a->stagger_phases = 1;
    a->stagger_ticks = 0;
    a->debug_flags = 0;
}

__global__ __launch_bounds__(256, 1) void upfirs_fused_kernel(const UpFirArgs p) { upfirs_body<false, false>(p); }
__global__ __launch_bounds__(256, 1) void upfirs_fused_noise_kernel(const UpFirArgs p) { upfirs_body<false, true>(p); }
__global__ __launch_bounds__(256, 1) void upfirs_fused_pre_kernel(const UpFirArgs p) { upfirs_body<true, false>(p); }
__global__ __launch_bounds__(256, 1) void upfirs_fused_pre_noise_kernel(const UpFirArgs p) { upfirs_body<true, true>(p); }

hipError_t launch_upfir_split(const UpFirArgs& args, hipStream_t stream) {
    using Kernel = void (*)(const UpFirArgs);
    static const Kernel kernels[4] = {upfirs_fused_kernel, upfirs_fused_noise_kernel, upfirs_fused_pre_kernel, upfirs_fused_pre_noise_kernel};  // [pre * 2 + noise]
    static PerDeviceInt ready;  // the dynamic-LDS opt-in is per device
    int unused = 0;
    const hipError_t e = ready.get(
        [&](int, int* value) {
            *value = 1;
            for (int i = 0; i < 4; ++i) {
                const hipError_t err =
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[i]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(i >= 2 ? 0 : 512));
                if (err != hipSuccess) return err;
            }
            return hipSuccess;
        },
        &unused);
    if (e != hipSuccess) return e;
    if (!upfirs_supported(args.Cin, args.Cout, args.H, args.W) || args.segs < 1 || args.rows_per_seg * args.segs != args.H || args.rows_per_seg % kTH != 0 ||
        args.total_blocks != args.B * args.m_tiles * args.strips * args.segs)
        return hipErrorInvalidValue;
    const int pre = args.input_prescaled ? 1 : 0, noise = args.noise != nullptr ? 1 : 0;
    hipLaunchKernelGGL(kernels[2 * pre + noise], dim3(args.total_blocks), dim3(256), lds_bytes(pre ? 0 : args.Cin), stream, args);
    return hipGetLastError();
}

}  // namespace gance
