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
// operations of a 6x6 input transform (84 instructions here: packed column pass, eight-instruction row pass) are not hidden
// beside the 36 MFMAs they feed -- they add their issue time. This kernel runs TWO waves per SIMD (8 per block, <= 256
// registers each): 36 positions x one 16 x 16 accumulator tile = 144 accumulators per wave, 36 MFMAs (1152 cycles) per
// k-step of four input channels against ~340 cycles of transform; while one wave waits for LDS, a barrier or the issue
// of an LDS-DMA piece the other keeps the matrix pipe busy. (One wave per SIMD with 32 channels x 16 tiles does the
// transform once for two channel tiles, but pays every such wait in full: measured 12 % slower)
//
// Geometry. Block = 8 waves = 2 channel tiles of 16 x 4 tile rows; a wave owns 16 channels x one row of sixteen 4x4 output
// tiles (4 x 64 pixels): the block 32 channels x 16 x 64 pixels. The two waves of a SIMD take the two channel tiles of the
// SAME tile row (tile row = SIMD id, channel tile = the wave's ticket on that SIMD): they run the two halves of the pipeline,
// and the ToRGB sums of the first reach the second through LDS, so a block writes ONE partial ToRGB image. Lane (n = lane % 16, g = lane / 16) transforms the 6x6
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
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

#ifndef GANCE_W43_WINDOW
#define GANCE_W43_WINDOW 0
#endif
#ifndef GANCE_W43_ABLATE
#define GANCE_W43_ABLATE 0  // timing ablations (wrong results; Makefile w43ab%): 1 no DMA inside the stream, 2 no input transform,
                            // 4 no weight reads, 8 no window reads (nor transform), 16 no barrier per k-step
#endif

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kKC = 4;                         // input channels per chunk = one k-step of the 16x16x4 MFMA
// LDS banks of the window reads (experiment of round 5, OFF: GANCE_W43_ODD_SHIFT = 2 and GANCE_W43_WINDOW = 3, Makefile target
// ../libgance_hip_w43banks.so). A lane group of an 8-byte read is 32 lanes = the sixteen tiles of TWO input-channel planes; bank =
// (address / 4) mod 64. Tiles are 4 floats apart, planes and rows multiples of 4: every lane of both planes reads the same two residues
// mod 4, two lanes per bank; the dword reads of columns 3 / 8 (32 banks) put four lanes on a bank. With the odd planes two floats to
// the right (their DMA pieces from 8 bytes further left) and columns 3 / 8 read as halves of 8-byte pairs every window read is
// conflict-free by that model. Measured: SQ_LDS_BANK_CONFLICT 5.0e8 -> 4.0e8 per launch (the rest is not in the window reads), run
// time +2 ... 3 % (3.57 / 3.62 / 3.91 / 4.53 / 5.45 -> 3.68 / 3.73 / 4.01 / 4.63 / 5.47 ms per 64 frames; either half alone: +1 ... 2 %):
// the LDS array is 31 % busy in this kernel; its conflicts are not what the waves wait for.
#ifndef GANCE_W43_ODD_SHIFT
#define GANCE_W43_ODD_SHIFT 0
#endif
constexpr int kOddShift = GANCE_W43_ODD_SHIFT;
constexpr int kBM = 32;                        // output channels per block (2 channel tiles of 16)
constexpr int kUnit = 16 * 36;                 // a (channel tile, ci) unit of weights: [co % 16][36]: a lane's 36 weights are nine aligned float4
constexpr int kWPieces = 18;                   // 8 units = 4608 floats = 18 pieces of 256 floats
constexpr int kWFloats = kWPieces * 256;
constexpr int kPiecesPerWave = 5;
// Pixel geometry of a block: TW x (1024 / TW) pixels = 64 tiles of 4x4; a wave's sixteen tiles are TW / 4 tile columns x
// 16 / (TW / 4) tile rows. <64>: 16 x 64 pixels, a wave = one row of 16 tiles (layers >= 64 wide); <32>: 32 x 32 pixels,
// a wave = two rows of 8 tiles (the 32x32 layer).
template <int TW>
struct Geo43 {
    static constexpr int kTW = TW, kTH = 1024 / TW;
    static constexpr int kTC = TW / 4, kTR = 16 / kTC;           // tile columns / tile rows of a wave
    static constexpr int kPW = kTW + 8, kPH = kTH + 2;            // haloed patch (16-byte aligned row segments): 18 x 72 / 34 x 40
    static constexpr int kPlane = kPH * kPW;                      // 1296 / 1360 floats per input channel
    static constexpr int kPatchF4 = kKC * kPlane / 4;
    static constexpr int kPPieces = (kPatchF4 + 63) / 64;         // 21 / 22 (the last one a quarter full)
    static constexpr int kPieces = kWPieces + kPPieces;           // 39 (+ one repeat of patch piece 0: every wave issues five) / 40
    static_assert(kPieces <= 8 * kPiecesPerWave && kPieces > 7 * kPiecesPerWave, "five LDS-DMA pieces per wave and chunk");
    static constexpr int kSlot = kPieces * 256;                   // 39 / 40 KB
    static constexpr int kNoiseRowsPerPiece = 256 / kTW;          // noise rows one 1 KB piece covers: 4 / 8
};
#ifndef GANCE_W43_NBUF
#define GANCE_W43_NBUF 3  // (Makefile target ../libgance_hip_w43nbuf<N>.so builds the ring-depth variants)
#endif
constexpr int kNBUF = GANCE_W43_NBUF;          // ring slots: chunk G + 2 is issued during k-step G (a fourth slot measured no faster)
// constants of a tile, fetched by LDS-DMA with its first chunk (a global load in the epilogue costs its whole latency once
// per tile, and the wait behind it would drain the ring): demod | bias | next layer's style (32 each, padded to 64) |
// noise [16][64] | (RGB) A operands of the ToRGB product [2 channel tiles][4 steps][64 lanes]
constexpr int kConstD = 0, kConstB = 64, kConstS = 128, kConstNoise = 192, kConstRgb = kConstNoise + 1024, kConstFloats = kConstRgb + 512;
constexpr int kStoresPerEpilogue = 16, kRgbStores = 12;
// (RGB) hand-over of a wave's ToRGB sums to the wave that holds the OTHER sixteen channels of the same pixels:
// [tile row][output row x colour = 12][16 lanes] float4
constexpr int kRgbXchgFloats = 4 * 12 * 16 * 4;

// B^T of F(4,3), points 0, +-1, +-2 (Lavin & Gray): 12 vector instructions, on one window line or on two at once
template <typename T>
__device__ __forceinline__ void input_transform6(T d0, T d1, T d2, T d3, T d4, T d5, T (&t)[6]) {
    const T a = d4 - 4.f * d2;
    const T b = d3 - 4.f * d1;
    const T c = d4 - d2;
    const T e = d3 - d1;
    t[0] = 4.f * d0 + (d4 - 5.f * d2);
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + 2.f * e;
    t[4] = c - 2.f * e;
    t[5] = 4.f * d1 + (d5 - 5.f * d3);
}

// The same B^T along a window ROW whose six values lie in three register pairs (d0, d5), (d1, d2), (d3, d4) -- which is how
// the packed column pass leaves them: eight vector instructions instead of twelve. (b, a), (e, c) are one packed operation
// each; t1 = a + b | t2 = a - b and t3 = c + 2 e | t4 = c - 2 e are one each with both result halves reading BOTH halves of
// the same source pair (op_sel), which hipcc does not form by itself; t0 and t5 mix three pairs and stay scalar.
__device__ __forceinline__ void input_transform6_row(f32x2 r05, f32x2 r12, f32x2 r34, float (&t)[6]) {
    const f32x2 ba = r34 - 4.f * r12;
    const f32x2 ec = r34 - r12;
    f32x2 t12, t34;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t12) : "v"(ba));
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(t34) : "v"(ec));
    t[0] = 4.f * r05[0] + (r34[1] - 5.f * r12[1]);
    t[1] = t12[0];
    t[2] = t12[1];
    t[3] = t34[0];
    t[4] = t34[1];
    t[5] = 4.f * r12[0] + (r05[1] - 5.f * r34[0]);
}

// A^T of F(4,3): 10 vector instructions
__device__ __forceinline__ void output_transform6(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y[0] = m0 + s1 + s2;
    y[1] = fmaf(2.f, d2, d1);
    y[2] = fmaf(4.f, s2, s1);
    y[3] = fmaf(8.f, d2, d1) + m5;
}

struct Tile43 {
    int m_tile, y0, x0, b0;
};

// The lane id, computed where it is used: per-lane values derived once at kernel entry would be live across the epilogue,
// where every register is taken; hipcc spills them, and a scratch reload inside the k-steps is a vector-memory load whose
// wait drains the LDS-DMA ring. Everything per-lane is therefore re-derived from this after every epilogue.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

}  // namespace

template <bool RGB, int TW>
__device__ __forceinline__ void winograd43_body(const ConvArgs& p) {
    using Geo = Geo43<TW>;
    constexpr int kTW = Geo::kTW, kTH = Geo::kTH, kPW = Geo::kPW, kPH = Geo::kPH, kPlane = Geo::kPlane, kPatchF4 = Geo::kPatchF4;
    constexpr int kPieces = Geo::kPieces, kSlot = Geo::kSlot, kTC = Geo::kTC, kTR = Geo::kTR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const const0 = smem + kNBUF * kSlot;  // two sets of tile constants (tile parity)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work of this wave: tile row = the SIMD it runs on, channel tile = its ticket there (two waves of 256 registers fill a
    // SIMD, so each SIMD holds exactly two of the block's eight). The two waves of a SIMD then hold the two channel tiles of
    // the SAME pixels and run the two halves of the pipeline (see the stream loop): the ticket is the wave's role as well, and
    // the hand-over of the ToRGB sums between them (epilogue) needs no atomic: role 0 is always a barrier ahead of role 1.
    // Which waves share a SIMD is the dispatcher's choice, so the wave asks the hardware (HW_ID.simd_id).
    __shared__ int simd_tickets[4];
    if (tid < 4) simd_tickets[tid] = 0;
    __syncthreads();
    const int pg = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;  // hwreg(HW_REG_HW_ID, 4, 2)
    int ticket = 0;
    if ((tid & 63) == 0) ticket = atomicAdd(&simd_tickets[pg], 1) & 1;
    const int cot = __builtin_amdgcn_readfirstlane(ticket);
    const int role = cot;
    // Run-time guard of that assumption: were a SIMD ever given one or three of the block's waves (a build whose register
    // count let a third wave fit, tools/check_w43_isa.py skipped), two waves would share a role and write wrong pixels in
    // silence. Every wave of the block sees the same four counts: the block records the fault where the host reads it
    // before the next call (engine.hip: check_call) and leaves, uniformly, before its first barrier-dependent step.
    __syncthreads();
    if (simd_tickets[0] != 2 || simd_tickets[1] != 2 || simd_tickets[2] != 2 || simd_tickets[3] != 2) {
        if (wave == 0 && fresh_lane() == 0 && p.fault_flag != nullptr) __hip_atomic_store(p.fault_flag, 43, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const int Hp = p.H + 2, Wp = p.W + 8;
    const int n = p.total_chunks;               // chunks per tile
    const int my_tiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * n;             // chunks of this block's stream

    // tile i of this block -> tile, XCD-aware (a persistent block strides by a multiple of 8; consecutive ids run on one XCD:
    // the channel tiles of a pixel tile share its L2)
    auto decode = [&](int i) {
        const int v = (int)blockIdx.x + i * (int)gridDim.x;
        const int nwg = p.total_tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        Tile43 t;
        if (p.m_tiles == 16 && p.xcd_blocking) {
            // An XCD's 32 CUs work on 32 consecutive ids at a time: with the channel tile fastest that is 2 pixel tiles x 16
            // channel tiles -- every XCD streams the layer's whole 37.7 MB of transformed weights per 5.2 MB of patches (x5.6 of the
            // algorithmic bytes at 32^2 / 64^2). 4 pixel tiles x 8 channel tiles move 29 MB per round instead of 43: blocks walk K in
            // step, so the sharing is by timing, not by L2 capacity.
            const int j = id & 63, sg = id >> 6;  // (64 ids = 4 pixel tiles x 16 channel tiles)
            t.m_tile = 8 * (j >> 5) + (j & 7);
            id = 4 * sg + ((j & 31) >> 3);
        } else {
            t.m_tile = id % p.m_tiles;
            id /= p.m_tiles;
        }
        t.x0 = (id % p.tiles_x) * kTW;
        id /= p.tiles_x;
        t.y0 = (id % p.tiles_y) * kTH;
        t.b0 = id / p.tiles_y;
        return t;
    };

    // ---- per-lane values of the k-steps (re-derived after every epilogue: see fresh_lane) ----
    // LDS-DMA pieces of this wave: piece wave * 5 + r; < 18: weights (linear), else patch piece (per-lane source offset);
    // operand offsets of the lane inside a slot (floats): its window (rows 4 pg .. + 5, columns 4 n + 3 .. + 8) and its weights
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);
    int piece_voff[kPiecesPerWave];
    int win_off, a_off;
    auto lane_setup = [&]() {
        const int lane = fresh_lane();
        const int n16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int r = 0; r < kPiecesPerWave; ++r) {
            int piece = wave * kPiecesPerWave + r;
            if (piece >= kPieces) piece = kWPieces;  // (the fortieth slot repeats patch piece 0: same bytes to the same place)
            if (piece < kWPieces) {
                piece_voff[r] = piece * 1024 + lane * 16;
            } else {
                int f = (piece - kWPieces) * 64 + lane;
                if (f >= kPatchF4) f = (piece - kWPieces) * 64;  // (the last piece is a quarter full: the rest re-copy its first float4 into padding)
                const int q4 = f % (kPW / 4);
                const int row = (f / (kPW / 4)) % kPH;
                const int c = f / (kPW / 4 * kPH);
                // (odd planes sit TWO FLOATS to the right in LDS -- kOddShift: their 16-byte pieces come from 8 bytes further left; the two
                // floats in front of a row's first piece are the row above's last ones: never read)
                piece_voff[r] = ((c * Hp + row) * Wp + 4 * q4 - kOddShift * (c & 1)) * 4;
            }
        }
        win_off = kWFloats + g * kPlane + 4 * (pg * kTR + n16 / kTC) * kPW + 4 * (n16 % kTC) + kOddShift * (g & 1);
        a_off = (cot * 4 + g) * kUnit + n16 * 36;
    };
    lane_setup();
    // Staging side of the stream (all scalar): the next chunk to fetch = chunk st_q of this block's tile st_tile, into ring
    // slot st_slot; byte offsets of its weights and its patch; the patch's buffer resource (one per sample).
    int st_tile = 0, st_q = 0, st_slot = 0, st_w = 0, st_x = 0, st_count = 0;
    __amdgpu_buffer_rsrc_t st_x_rsrc = w_rsrc;
    const int x_chunk_step = kKC * Hp * Wp * 4;
    int cur_slot = 0, cur_w = 0, cur_x = 0;  // of the chunk being staged
    bool cur_valid = false;
    // Opens the next chunk of the stream (runs once per k-step, outside the MFMA weave: it branches). A chunk that opens a
    // tile also brings the tile's constants: up to two extra DMA instructions per wave, issued here, i.e. BEFORE the
    // chunk's five pieces (the counted waits allow the five youngest operations to be in flight).
    auto stage_begin = [&]() {
        cur_valid = st_count < total;
        if (!cur_valid) return;
        if (st_q == 0) {
            const Tile43 t = decode(st_tile);
            const int lane = fresh_lane();
            st_x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)t.b0 * p.x_b_stride), 0, 0x7fffffff, 0x00020000);
            st_w = t.m_tile * n * (kWFloats * 4);  // bytes: [m tile][chunk][kWFloats]
            st_x = (t.y0 * Wp + t.x0) * 4;         // patch row 0 = image row y0 - 1 = buffer row y0; column x0 - 4 = buffer column x0
            float* const set = const0 + (st_tile & 1) * kConstFloats;
            const int co = t.m_tile * kBM;
            if (wave < 3) {  // demod / bias / next style of the block's 32 channels: one dword piece each (the resource's bound clips the other lanes)
                const float* src = wave == 0 ? p.d + (size_t)t.b0 * p.d_stride + co
                                             : (wave == 1 ? p.bias + co : (p.s_next != nullptr ? p.s_next + (size_t)t.b0 * p.s_stride + co : nullptr));
                if (src != nullptr) {
                    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, kBM * 4, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lds_ptr_t)(set + wave * 64), 4, lane * 4, 0, 0, 0);
                }
            } else if (wave < 7) {  // a quarter of the tile's noise (256 floats = 4 rows of 64 / 8 rows of 32): one 16-byte piece
                if (p.noise != nullptr) {
                    const __amdgpu_buffer_rsrc_t nz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.noise, 0, 0x7fffffff, 0x00020000);
                    const int row = Geo::kNoiseRowsPerPiece * (wave - 3) + lane / kTC;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(nz_rsrc, (lds_ptr_t)(set + kConstNoise + (wave - 3) * 256), 16,
                                                             ((t.y0 + row) * p.OW + t.x0 + 4 * (lane % kTC)) * 4, t.b0 * p.noise_b_stride * 4, 0, 0);
                }
            } else if (RGB) {  // the A operands of the ToRGB product: [b][Cout / 4 steps][64 lanes], steps 8 m_tile .. + 7: two 16-byte pieces
                const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(p.rgb_coef + ((size_t)t.b0 * (p.Cout / 4) + 8 * t.m_tile) * 64), 0, 512 * 4, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(set + kConstRgb), 16, lane * 16, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(set + kConstRgb + 256), 16, lane * 16, 1024, 0, 0);
            }
        }
        cur_slot = st_slot;
        cur_w = st_w;
        cur_x = st_x;
        st_w += kWFloats * 4;
        st_x += x_chunk_step;
        st_slot = st_slot == kNBUF - 1 ? 0 : st_slot + 1;
        ++st_count;
        if (++st_q == n) {
            st_q = 0;
            ++st_tile;
        }
    };
    // piece r of the chunk stage_begin() opened: one instruction, no control flow beyond the wave-uniform choice of its kind
    auto stage_piece = [&](int r) {
        float* const base = smem + cur_slot * kSlot;
        int piece = wave * kPiecesPerWave + r;  // (wave is scalar: the branches are uniform)
        if (piece >= kPieces) piece = kWPieces;
        if (piece < kWPieces)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(base + piece * 256), 16, piece_voff[r], cur_w, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_x_rsrc, (lds_ptr_t)(base + kWFloats + (piece - kWPieces) * 256), 16, piece_voff[r], cur_x, 0, 0);
    };

    float* const rgb_xchg = const0 + 2 * kConstFloats;

    f32x4 acc[36];  // (every tile's first k-step overwrites them: multiply(first))

    // L + T: this lane's weights of the chunk (position j * 6 + i) and V = B^T d B of its window, both into registers.
    // Column pass first, two window columns per packed instruction (the pairs (4,5), (6,7) are aligned 8-byte reads,
    // (3, 8) one ds_read2_b32: fewest bytes; three aligned 16-byte reads per row -- no bank conflicts, twice the bytes -- and
    // one 16-byte read + two DPP row shifts both measured slower), then the row pass line by line.
    float V[36], A[36];
#if GANCE_W43_ABLATE & 12
    for (int k = 0; k < 36; ++k) V[k] = A[k] = 0.f;
#endif
    auto load_transform = [&](int slot) {
        const float* const P = smem + slot * kSlot + win_off;
        const float* const U = smem + slot * kSlot + a_off;
#if GANCE_W43_ABLATE & 8
#pragma unroll
        for (int k = 0; k < 36; ++k) asm volatile("" : "+v"(V[k]));
#if GANCE_W43_ABLATE & 4
#pragma unroll
        for (int k = 0; k < 36; ++k) asm volatile("" : "+v"(A[k]));
#else
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(U + 4 * k);
            A[4 * k + 0] = t[0], A[4 * k + 1] = t[1], A[4 * k + 2] = t[2], A[4 * k + 3] = t[3];
        }
#endif
        return;
#endif
        f32x2 c45[6], c67[6], c38[6];
#pragma unroll
        for (int y = 0; y < 6; ++y) {
#if GANCE_W43_WINDOW == 1
            // (variant: three aligned 16-byte reads per window row, columns 4 n .. 4 n + 11: bank-conflict free, twice the bytes)
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(P + y * kPW);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(P + y * kPW + 4);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(P + y * kPW + 8);
            c45[y] = f32x2{q1[0], q1[1]};
            c67[y] = f32x2{q1[2], q1[3]};
            c38[y] = f32x2{q0[3], q2[0]};
#else
            c45[y] = *reinterpret_cast<const f32x2*>(P + y * kPW + 4);
            c67[y] = *reinterpret_cast<const f32x2*>(P + y * kPW + 6);
#if GANCE_W43_WINDOW == 3
            // (experiment: columns 3 and 8 as the halves of two 8-byte reads -- see kOddShift)
            c38[y][0] = (*reinterpret_cast<const f32x2*>(P + y * kPW + 2))[1];
            c38[y][1] = (*reinterpret_cast<const f32x2*>(P + y * kPW + 8))[0];
#else
            c38[y][0] = P[y * kPW + 3];
            c38[y][1] = P[y * kPW + 8];
#endif
#endif
        }
#if GANCE_W43_ABLATE & 4
#pragma unroll
        for (int k = 0; k < 36; ++k) asm volatile("" : "+v"(A[k]));
#else
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(U + 4 * k);
            A[4 * k + 0] = t[0];
            A[4 * k + 1] = t[1];
            A[4 * k + 2] = t[2];
            A[4 * k + 3] = t[3];
        }
#endif
#if GANCE_W43_ABLATE & 2
#pragma unroll
        for (int y = 0; y < 6; ++y) {
            V[y * 6 + 0] = c38[y][0], V[y * 6 + 1] = c45[y][0], V[y * 6 + 2] = c45[y][1];
            V[y * 6 + 3] = c67[y][0], V[y * 6 + 4] = c67[y][1], V[y * 6 + 5] = c38[y][1];
        }
        return;
#endif
        f32x2 t45[6], t67[6], t38[6];
        input_transform6<f32x2>(c45[0], c45[1], c45[2], c45[3], c45[4], c45[5], t45);
        input_transform6<f32x2>(c67[0], c67[1], c67[2], c67[3], c67[4], c67[5], t67);
        input_transform6<f32x2>(c38[0], c38[1], c38[2], c38[3], c38[4], c38[5], t38);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            // (kept opaque: where a packed result is only read by element, hipcc splits the packed operation again)
            asm("" : "+v"(t45[i]), "+v"(t67[i]), "+v"(t38[i]));
            float v[6];
            input_transform6_row(t38[i], t45[i], t67[i], v);
#pragma unroll
            for (int j = 0; j < 6; ++j) V[j * 6 + i] = v[j];
        }
    };
    // The 36 MFMAs of a k-step, and woven between them (one per seven MFMAs: an LDS-DMA instruction takes the wave about
    // 120 cycles to issue, behind the barrier that sat on every wave's critical path) the five DMA pieces of the chunk
    // stage_begin() opened, if there is one.
    // `first_tag`: the tile's first k-step takes C = 0 instead of the accumulators (144 clears per tile and wave saved).
    auto multiply = [&](auto first_tag) {
        constexpr bool kFirst = decltype(first_tag)::value;
#pragma unroll
        for (int pos = 0; pos < 36; ++pos) {
            acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[pos], V[pos], kFirst ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pos], 0, 0, 0);
            if (pos % 7 == 3 && cur_valid && !(GANCE_W43_ABLATE & 1)) stage_piece(pos / 7);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMAs, then 4 x (one LDS-DMA issue, 7 MFMAs), one issue, 4 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    };

    // ---- epilogue of this block's tile i: lane (n16, g) holds channels 4 g + r (r = 0 .. 3) of tile n16 for all 36
    // positions. Output transform, demodulation, noise, bias, leaky ReLU, (RGB) ToRGB product, stores. ----
    auto epilogue = [&](int i) {
        const Tile43 t = decode(i);
        const float* const set = const0 + (i & 1) * kConstFloats;
        const int lane = fresh_lane();
        const int n16 = lane & 15, g = lane >> 4;
        const int ty4 = 4 * (pg * kTR + n16 / kTC), tx4 = 4 * (n16 % kTC);  // the lane's tile inside the block's pixel tile
        const int oy0 = t.y0 + ty4, ox0 = t.x0 + tx4;
        const int ct = t.m_tile * 2 + cot;  // 16-channel tile of the layer
        const int co0 = ct * 16 + 4 * g;
        const f32x4 dm = *reinterpret_cast<const f32x4*>(set + kConstD + cot * 16 + 4 * g);
        const f32x4 bm = *reinterpret_cast<const f32x4*>(set + kConstB + cot * 16 + 4 * g);
        f32x4 sn = f32x4{1.f, 1.f, 1.f, 1.f};
        if (p.s_next != nullptr) sn = *reinterpret_cast<const f32x4*>(set + kConstS + cot * 16 + 4 * g);
        // (RGB) the layer's ToRGB channel sum over this wave's 16 channels rides on the matrix pipe: k-step r of the product
        // takes B[k = g][n] = this lane's activation of channel 4 g + r and A[m][k] = style x weight of colour m < 3 (else
        // 0) for channel 16 ct + 4 k + r, from the table launch_winograd64_rgb_coef prepares
        f32x4 rgbacc[4][4];
        if constexpr (RGB) {
#pragma unroll
            for (int oy = 0; oy < 4; ++oy)
#pragma unroll
                for (int ox = 0; ox < 4; ++ox) rgbacc[oy][ox] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float* const out_base = p.out == nullptr ? nullptr
                                                 : p.out + (size_t)t.b0 * p.out_b_stride + (size_t)co0 * p.out_c_stride +
                                                       (size_t)(oy0 + p.out_y_off) * p.out_row_stride + ox0 + p.out_x_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_barrier(0);  // one channel's 36 accumulators at a time (unfenced, hipcc hoists all four and spills)
            float a_rgb = 0.f;
            if constexpr (RGB) a_rgb = set[kConstRgb + (cot * 4 + r) * 64 + lane] * 1.4142135623730951f;  // (the leaky ReLU's gain rides on the coefficient)
            const float s2 = sn[r] * 1.4142135623730951f;
            float tr[4][6];  // A^T M: [output row][position column]
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float col[4];
                output_transform6(acc[j * 6 + 0][r], acc[j * 6 + 1][r], acc[j * 6 + 2][r], acc[j * 6 + 3][r], acc[j * 6 + 4][r], acc[j * 6 + 5][r], col);
#pragma unroll
                for (int oy = 0; oy < 4; ++oy) tr[oy][j] = col[oy];
            }
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) {
                float yrow[4];
                output_transform6(tr[oy][0], tr[oy][1], tr[oy][2], tr[oy][3], tr[oy][4], tr[oy][5], yrow);
                f32x4 nz = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.noise != nullptr) nz = *reinterpret_cast<const f32x4*>(set + kConstNoise + (ty4 + oy) * kTW + tx4) * p.noise_strength;
                f32x4 v = f32x4{yrow[0], yrow[1], yrow[2], yrow[3]} * dm[r] + nz + bm[r];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.2f * v[k]);  // (x sqrt(2): in a_rgb and s2)
                if constexpr (RGB) {
#pragma unroll
                    for (int ox = 0; ox < 4; ++ox) rgbacc[oy][ox] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_rgb, v[ox], rgbacc[oy][ox], 0, 0, 0);
                }
                // (the stored activation carries the next layer's style when that layer wants it so; the ToRGB product took the plain one)
                if (out_base != nullptr) *reinterpret_cast<f32x4*>(out_base + (size_t)r * p.out_c_stride + (size_t)oy * p.out_row_stride) = v * s2;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RGB) {
            // Partial image of the block's 32 channels: [Cout / 32][B][3][OH][OW]. Lanes 0 .. 15 of a wave hold (R, G, B, 0)
            // of their tile's pixels for the wave's 16 channels; the other wave of the SIMD holds the other 16 channels of the
            // same pixels. The role-0 wave (channel tile 0) leaves its sums in LDS: its epilogue ends an interval; the role-1
            // wave (channel tile 1) runs its epilogue of the same tile behind the next barrier, adds them to its own and stores.
            f32x4* const xchg = reinterpret_cast<f32x4*>(rgb_xchg) + (pg * 12) * 16 + n16;
            if (cot == 0) {
                if (g == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int oy = 0; oy < 4; ++oy) xchg[(c * 4 + oy) * 16] = f32x4{rgbacc[oy][0][c], rgbacc[oy][1][c], rgbacc[oy][2][c], rgbacc[oy][3][c]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (in LDS before this wave's next barrier)
            } else if (g == 0) {
                float* const y_base = p.rgb_y + (((size_t)t.m_tile * p.B + t.b0) * 3) * p.OH * p.OW + (size_t)oy0 * p.OW + ox0;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int oy = 0; oy < 4; ++oy)
                        *reinterpret_cast<f32x4*>(y_base + ((size_t)c * p.OH + oy) * p.OW) =
                            f32x4{rgbacc[oy][0][c], rgbacc[oy][1][c], rgbacc[oy][2][c], rgbacc[oy][3][c]} + xchg[(c * 4 + oy) * 16];
            }
        }
        lane_setup();
    };

    // ---- ring prologue: the first two chunks of the stream, all pieces at once ----
    for (int c = 0; c < kNBUF - 1; ++c) {
        stage_begin();
        if (cur_valid) {
#pragma unroll
            for (int r = 0; r < kPiecesPerWave; ++r) stage_piece(r);
        }
    }

    // ---- the stream: k-step G of the block multiplies chunk G (chunk G % n of tile G / n), ring slot G % 3 ----
    // The two waves of a SIMD run half a k-step apart: in the interval between two barriers the role-0 wave loads and
    // transforms chunk G and then multiplies it, the role-1 wave first multiplies chunk G - 1 out of its registers and then
    // loads and transforms chunk G. While one of them waits for LDS or issues vector instructions the other keeps the
    // matrix pipe busy. A tile's epilogue runs right behind its last multiply -- for a role-1 wave that is in the first
    // interval of the NEXT tile -- while the ring keeps fetching: the stream never drains between tiles.
    // Shape of the loops: tiles outside, their chunks inside, the epilogue UNCONDITIONALLY behind the inner loop, one nest per
    // role: accumulators that flow through a conditional (an epilogue under `if` inside one loop) cost 244 spilled
    // registers here; do-while because the guard path of a `for` (all accumulators zero) meets the real path in front of the
    // epilogue and costs accumulator copies.
    int slot = 0, G = 0;
    // opens interval G: chunk G must have landed; chunk G + 1 (five pieces) may stay in flight, and so may the stores of an
    // epilogue this wave ran in the previous interval (vmcnt counts in issue order: they are younger than chunk G's pieces)
    auto open_interval = [&](bool stores_behind, bool rgb_stores = false) {
        if (G + 1 >= total)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (stores_behind && rgb_stores)  // (only the role-1 wave stores the partial ToRGB image)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerWave + kStoresPerEpilogue + kRgbStores) : "memory");
        else if (stores_behind)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerWave + kStoresPerEpilogue) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerWave) : "memory");
        if (!(GANCE_W43_ABLATE & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage_begin();  // chunk G + 2 -> the slot k-step G - 1 read (every wave is past the barrier: nobody reads it any more)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto close_interval = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == kNBUF - 1 ? 0 : slot + 1;
        ++G;
    };
    if (role == 0) {
        int tile = 0;
        do {
            open_interval(tile > 0);
            load_transform(slot);
            __builtin_amdgcn_sched_barrier(0);
            multiply(std::true_type{});
            close_interval();
            int q = 1;
            do {
                open_interval(false);
                load_transform(slot);
                __builtin_amdgcn_sched_barrier(0);
                multiply(std::false_type{});
                close_interval();
            } while (++q < n);
            epilogue(tile);
        } while (++tile < my_tiles);
        if constexpr (RGB) __builtin_amdgcn_s_barrier();  // the block's last ToRGB sums are in LDS (the role-1 waves wait for this in front of their last epilogue)
    } else {
        open_interval(false);  // interval 0: nothing to multiply yet; this interval's DMA pieces go out in one burst
        if (cur_valid) {
#pragma unroll
            for (int r = 0; r < kPiecesPerWave; ++r) stage_piece(r);
        }
        load_transform(slot);
        close_interval();
        int tile = 0;
        do {
            open_interval(tile > 0, RGB);
            multiply(std::true_type{});  // chunk 0 of the tile
            __builtin_amdgcn_sched_barrier(0);
            load_transform(slot);
            close_interval();
            int q = 2;
            do {
                open_interval(false);
                multiply(std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
                load_transform(slot);
                close_interval();
            } while (++q < n);
            // the tile's last chunk: multiplied in the first interval of the next tile (if there is one), then the epilogue,
            // then that interval's own load + transform
            const bool more = tile + 1 < my_tiles;
            if (more) {
                open_interval(false);
            } else {
                cur_valid = false;
                if constexpr (RGB) __builtin_amdgcn_s_barrier();
            }
            multiply(std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            epilogue(tile);
            if (more) {
                load_transform(slot);
                close_interval();
            }
        } while (++tile < my_tiles);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring may still be landing when the block's LDS is given back
}

// (plain functions around the templated body: as a kernel TEMPLATE the host pass of hipcc drops the instantiation, see winograd64_conv.hip)
__global__ __launch_bounds__(512, 1) void winograd43_kernel(const ConvArgs p) { winograd43_body<false, 64>(p); }
__global__ __launch_bounds__(512, 1) void winograd43_rgb_kernel(const ConvArgs p) { winograd43_body<true, 64>(p); }
__global__ __launch_bounds__(512, 1) void winograd43_w32_kernel(const ConvArgs p) { winograd43_body<false, 32>(p); }
__global__ __launch_bounds__(512, 1) void winograd43_w32_rgb_kernel(const ConvArgs p) { winograd43_body<true, 32>(p); }

bool winograd43_supported(int cin, int cout, int H, int W) {
    // (>= 4 chunks per tile: two constant sets). Pixel tiles of 16 x 64, or 32 x 32 on the 32-pixel-wide layer.
    return cin % kKC == 0 && cin / kKC >= 4 && cout % kBM == 0 && ((W % 64 == 0 && H % 16 == 0) || (W == 32 && H % 32 == 0));
}

size_t winograd43_weight_floats(int cin, int cout) { return (size_t)(cout / kBM) * (cin / kKC) * kWFloats; }

bool winograd43_rgb_supported(int cout) { return winograd64_rgb_supported(cout); }  // (shares launch_winograd64_rgb_coef's table: Cout = 32 or a multiple of 64)
int winograd43_rgb_partials(int cout) { return cout / kBM; }

// w_in: the layer's runtime-scaled weights [tap = ky*3+kx][ci][co]; w_out: [m tile of 32][chunk of 4][channel tile][ci][co % 16][36],
// position j * 6 + i = (G g G^T)[i][j], i along y
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
    const bool rgb = args.epilogue == kEpilogueFullRgbPart;
    if (!winograd43_supported(args.Cin, args.Cout, args.H, args.W)) return hipErrorInvalidValue;
    if (rgb ? (!winograd43_rgb_supported(args.Cout) || args.rgb_coef == nullptr || args.rgb_y == nullptr) : (args.epilogue != kEpilogueFull || args.out == nullptr))
        return hipErrorInvalidValue;
    const bool narrow = args.W % 64 != 0;  // the 32 x 32 pixel geometry
    void (*const kernel)(const ConvArgs) = narrow ? (rgb ? winograd43_w32_rgb_kernel : winograd43_w32_kernel) : (rgb ? winograd43_rgb_kernel : winograd43_kernel);
    const int tw = narrow ? 32 : 64, th = 1024 / tw;
    const size_t lds_bytes = sizeof(float) * ((size_t)kNBUF * (narrow ? Geo43<32>::kSlot : Geo43<64>::kSlot) + 2 * kConstFloats + (rgb ? kRgbXchgFloats : 0));
    static PerDeviceInt resident[4];  // per device: the dynamic-LDS opt-in and the launch size = one block per CU, a multiple of 8 (XCDs)
    int resident_blocks = 0;
    hipError_t e = resident[(narrow ? 2 : 0) + (rgb ? 1 : 0)].get(
        [&](int device, int* value) {
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (err != hipSuccess) return err;
            int cus = 0;
            if ((err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return err;
            *value = std::max(8, cus / 8 * 8);
            return hipSuccess;
        },
        &resident_blocks);
    if (e != hipSuccess) return e;
    ConvArgs a = args;
    a.tiles_x = a.W / tw;
    a.tiles_y = a.H / th;
    a.m_tiles = a.Cout / kBM;
    a.total_chunks = a.Cin / kKC;
    a.total_tiles = a.m_tiles * a.tiles_x * a.tiles_y * a.B;
    if ((a.tiles_x * a.tiles_y * a.B) % 4 != 0) a.xcd_blocking = 0;  // (the 4 x 8 blocking walks pixel tiles four at a time)
    // blocks per CU (ConvArgs::grid_rounds), as long as a block's stream stays long beside its ring prologue (about two k-steps):
    // at least 64 k-steps per block, else fewer, larger blocks (one frame per call: one block per CU as before)
    int rounds = std::max(1, a.grid_rounds);
    const int min_tiles = std::max(1, 64 / a.total_chunks);
    while (rounds > 1 && a.total_tiles < resident_blocks * rounds * min_tiles) --rounds;
    hipLaunchKernelGGL(kernel, dim3(std::min(a.total_tiles, resident_blocks * rounds)), dim3(512), lds_bytes, stream, a);
    return hipGetLastError();
}

}  // namespace gance
