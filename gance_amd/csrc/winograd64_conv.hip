// Winograd F(2x2, 3x3) form of the modulated 3x3 stride-1 convolution on v_mfma_f32_16x16x4_f32: the same arithmetic as
// winograd_conv.hip (`modulated_conv2d_layer` -> tf.nn.conv2d), every Conv1 layer from 32^2 to 1024^2.
//
// Why a second kernel. In winograd_conv.hip a wave transforms a 4x4 input window per (tile, channel) -- 40 vector
// instructions -- and feeds the 16 results to 16 MFMAs (one 32x32x2 tile of 32 output channels per Winograd
// position): 2.5 vector instructions per MFMA, and on gfx950 the fp32 MFMA does not hide vector work of its own
// wave, so that kernel executes at 0.58 of the matrix peak. Here the MFMA is v_mfma_f32_16x16x4_f32
// (16 channels x 16 tiles x 4 input channels, 32 cycles) and a wave holds FOUR accumulator tiles per position:
// 16 positions x 4 x 4 registers = 256 accumulators, the same register budget, but a transformed window now feeds
// 64 MFMAs (2048 matrix-pipe cycles) instead of 16 (1024). The weight fragments of a position come out of LDS in ONE
// read (the weight image is laid out [pos][ci][m % 16][m / 16]).
//
// Geometries (template <MT channel tiles, TG tile rows per wave>): <4, 1> = 64 output channels x (8 x 32) pixels per
// block; <2, 2> = 32 channels x (16 x 32) pixels (the 32-channel layer at 1024^2). Block = 4 waves; wave w owns tile
// row(s) w TG .. of the block's 16 tile columns, lane (n = lane % 16, kq = lane / 16) transforms the window of tile n for
// input channel 4 ks + kq of k-step ks.
// K chunk = 4 input channels = ONE k-step: transformed weights + haloed patch per ring slot, SIX slots. Blocks are
// persistent over ONE continuous stream of chunks (tiles back to back). The window loads and the transform of k-step
// q+1 run under the MFMAs of k-step q. In front of every EVEN k-step q the wave waits for chunks q+1 and q+2 -- issued
// four and three k-steps earlier, counted vmcnt: the two younger chunks stay in flight -- and the block synchronises;
// the DMA pieces a wave contributes to chunks q+5 / q+6 are woven between the MFMAs of the pair. (A first version with
// 8-channel chunks in a three-slot ring issued a chunk's last pieces right before the wait that needed them and ran at
// 0.59 of the matrix peak -- DMA latency, not vector work, was the limit.)
// The epilogue (output transform A^T M A, demodulation, noise, bias, leaky ReLU, 8-byte stores: 128 contiguous bytes
// per 16 lanes; RGB variants: the layer's ToRGB channel sum as 64 more MFMAs) runs between tiles while the pipeline
// registers already hold the next tile's first k-step. Nothing in the tile loop may spill: a scratch reload is a
// vector-memory load and the wait behind it drains the ring (see fresh_lane / lane_offsets).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

// Timing ablations (GANCE_DEBUG_W64: 2 no DMA after the prologue, 4 no MFMA, 32 no input transform, 64 no per-k-step wait
// and barrier) exist only in a -DGANCE_W64_DEBUG=1 build.
#ifndef GANCE_W64_DEBUG
#define GANCE_W64_DEBUG 0
#endif
#define W64_DBG (GANCE_W64_DEBUG ? p.debug_flags : 0)
// Compile-time epilogue ablation (results are wrong): 1 no stores (make w64ablate).
#ifndef GANCE_W64_ABLATE
#define GANCE_W64_ABLATE 0
#endif

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_lds __attribute__((ext_vector_type(2), aligned(4)));  // a window's column pair: ds_read2_b32
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Geometry of a variant: MT channel tiles of 16 per wave and position, TG tile rows per wave.
//   <4, 1>: 64 output channels x  8 x 32 pixels per block (layers with >= 64 output channels)
//   <2, 2>: 32 output channels x 16 x 32 pixels per block (the 32-channel layers: 1024^2); a wave then transforms TWO
//           windows per k-step, i.e. the vector work per matrix cycle of winograd_conv.hip, but keeps this kernel's
//           ring, weave and one-instruction weight fragments
// Either way 16 positions x TG x MT x 4 registers = 256 accumulators and 64 MFMAs per k-step.
constexpr int kKC = 4;                  // input channels per chunk = one k-step of the 16x16x4 MFMA
constexpr int kTW = 32, kPW = kTW + 8;
constexpr int kNBUF = 6;                // ring depth: a chunk is issued four k-steps before its first use
constexpr int kConstHead = 512 + 64 + 64 + 64;  // style | demod | bias | the next layer's style of a tile, then its noise [pixel row][32]
template <int MT, int TG>
struct Geo {
    static_assert(MT * TG == 4, "256 accumulators");
    static constexpr int kBM = 16 * MT;                       // output channels per block
    static constexpr int kTH = 8 * TG;                        // pixel rows per block
    static constexpr int kPH = kTH + 2;
    static constexpr int kPlane = kPH * kPW;                  // 400 / 720
    static constexpr int kUFloats = 16 * kKC * kBM;           // 4096 / 2048: [pos][ci][m % 16][m / 16]
    static constexpr int kUPieces = kUFloats / 256;           // 16 / 8
    static constexpr int kUSlots = kUPieces / 4;              // weight pieces per wave: 4 / 2
    static constexpr int kPlF4 = kKC * kPlane / 4;            // 400 / 720
    static constexpr int kPlPieces = (kPlF4 + 63) / 64;       // 7 / 12 (the last one a quarter full)
    static constexpr int kPlSlots = (kPlPieces + 3) / 4;      // patch pieces per wave: 2 / 3
    static constexpr int kSlot = kUFloats + kPlPieces * 256;  // 5888 / 5120 floats
    static constexpr int kPiecesPerWave = kUSlots + kPlSlots; // 6 / 5: every wave issues as many, the waits count them
    static_assert((kNBUF - 3) * kPiecesPerWave <= 63, "counted vmcnt");
    static constexpr int kNoiseFloats = kTH * kTW;                      // 256 / 512
    static constexpr int kRgbSteps = 4 * MT;                            // k-steps of the ToRGB product: 16 / 8
    // constants of a tile: head | noise | (RGB) the A operand of every ToRGB k-step [step][lane]
    static constexpr int const_floats(bool rgb) { return kConstHead + kNoiseFloats + (rgb ? kRgbSteps * 64 : 0); }  // 896 (+1024) / 1152 (+512)
    static size_t lds_bytes(bool rgb) { return sizeof(float) * ((size_t)kNBUF * kSlot + 2 * const_floats(rgb)); }
};

struct Tile {
    int m_tile, y0, x0, b0;
};

// The lane id, computed where it is used. The once-per-tile code (tile constants, epilogue) derives its per-lane offsets
// from this instead of from values computed at kernel entry: those would be live across the k-steps, where every
// register is taken, hipcc would spill them, and a scratch reload is a vector-memory load -- the s_waitcnt vmcnt(0)
// behind it drains the whole LDS-DMA ring once per tile.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

}  // namespace

template <int MT, int TG, bool RGB>
__device__ __forceinline__ void winograd64_body(const ConvArgs& p) {
    using G = Geo<MT, TG>;
    constexpr int kBM = G::kBM, kTH = G::kTH, kPH = G::kPH, kPlane = G::kPlane, kUFloats = G::kUFloats, kPlF4 = G::kPlF4;
    constexpr int kConstFloats = G::const_floats(RGB), kRgbOff = kConstHead + G::kNoiseFloats;
    // The <2, 2> geometry takes its input ALREADY multiplied by this layer's style (ConvArgs::x contract; the producing up
    // layer folds s[b][ci] into its leaky ReLU for free: V is linear in d): 4 of the 24 packed transform instructions per
    // window and the style read leave the k-step (-3 %). In <4, 1> the same change (8 of 40 instructions) makes hipcc pick
    // the untied form of 30 MFMAs (destination != accumulator input) although every AGPR is taken: it then stages those
    // accumulators through VGPRs and one spare tuple, 120 copies in and out per loop trip (+20 %); that geometry keeps the
    // scale in its transform. tools/check_w64_isa.py counts the accumulator moves of a build: run it after ANY edit here.
    constexpr bool kPrescaled = TG == 2;
    constexpr int kPlPieces = G::kPlPieces, kSlot = G::kSlot, kPiecesPerWave = G::kPiecesPerWave, kUSlots = G::kUSlots, kPlSlots = G::kPlSlots;
    typedef float afrag_t __attribute__((ext_vector_type(MT)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const buf0 = smem;
    float* const const0 = smem + kNBUF * kSlot;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = p.H + 2, Wp = p.W + 8;
    const int n = p.total_chunks;  // chunks per tile

    const int my_tiles = (p.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto decode = [&](int i) {
        // virtual block id -> tile, XCD-aware (a persistent block strides by a multiple of 8): the channel tiles of a
        // pixel tile and neighbouring pixel tiles land on one XCD's L2 at about the same time
        const int v = (int)blockIdx.x + i * (int)gridDim.x;  // i < my_tiles
        const int nwg = p.total_tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        Tile t;
        t.m_tile = id % p.m_tiles;
        id /= p.m_tiles;
        t.x0 = (id % p.tiles_x) * kTW;
        id /= p.tiles_x;
        t.y0 = (id % p.tiles_y) * kTH;
        t.b0 = id / p.tiles_y;
        return t;
    };

    // ---- LDS-DMA: a chunk = kUPieces weight pieces + kPlPieces patch pieces of 256 floats (64 lanes x 16 bytes) ----
    // (tiles are whole: H % 8 == 0, W % 32 == 0, so a patch piece's per-lane source offset depends on nothing but the lane)
    // A wave's slots: first its weight pieces kUSlots wave + r, then its patch pieces wave + 4 r; a slot beyond the last patch
    // piece repeats patch piece 0 (same bytes: every wave issues as many, the waits count them). The last patch piece is
    // a quarter full: its other lanes copy float4 0 of the piece into the slot's padding. No slot needs a branch or a mask.
    //
    // Every per-lane offset of the k-steps lives in these few registers, and they are recomputed from the lane id at the
    // top of every tile (lane_offsets): values computed once at kernel entry are live across the epilogue, where hipcc
    // spills them, and a scratch reload is a vector-memory load -- the s_waitcnt vmcnt(0) behind it drains the DMA ring.
    int kq, lane16, poff, aoff;
    int p_voff[kPlSlots], p_lds[kPlSlots];
    auto lane_offsets = [&]() {
        const int lane = fresh_lane();
        const int n16 = lane & 15;
        kq = lane >> 4;
        lane16 = lane * 16;
        // operand offsets (floats): window of tile (row wave * TG + tg, column n16): patch rows 2 row .. + 3, columns
        // 2 n16 + 3 .. + 6; weight fragment of (input channel kq, output channels n16 + 16 mt)
        poff = kq * kPlane + (2 * wave * TG) * kPW + 2 * n16 + 3;  // + tg * 2 * kPW
        aoff = (kq * 16 + n16) * MT;                               // + pos * kKC * kBM
#pragma unroll
        for (int r = 0; r < kPlSlots; ++r) {
            int i = wave + 4 * r;
            if (i >= kPlPieces) i = 0;
            int f = i * 64 + lane;
            if (f >= kPlF4) f = i * 64;
            const int q = f % (kPW / 4);
            const int row = (f / (kPW / 4)) % kPH;
            const int c = f / (kPW / 4 * kPH);
            p_voff[r] = ((c * Hp + row) * Wp + 4 * q) * 4;
            p_lds[r] = kUFloats + i * 256;
        }
    };
    lane_offsets();
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);
    // Staging side of the stream, kept as running offsets (every scalar instruction of a k-step costs its issue time
    // beside the MFMAs, so the per-k-step bookkeeping is a handful of adds): the next tile to open, the chunks of the open
    // tile still to fetch, the ring slot to fetch into (float offset in smem) and the global byte offsets of the next
    // chunk's weights (this wave's pieces) and patch. Past the end of the stream the last chunk is fetched again into
    // the free slot (nobody reads it): every k-step issues the same number of pieces and the counted waits need no tail.
    int st_tile = 0, st_left = 0, st_off = 0;
    int st_w = 0, st_x = 0;
    const int x_step = kKC * Hp * Wp * 4;
    int cur_off = 0, cur_w_soff = 0, cur_x_soff = 0;  // of the chunk being staged
    __amdgpu_buffer_rsrc_t st_x_rsrc = w_rsrc;
    const __amdgpu_buffer_rsrc_t nz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.noise, 0, 0x7fffffff, 0x00020000);
    // Before a chunk's pieces go out: tile descriptors and constants if it opens a tile, then the chunk's scalar offsets.
    // Runs at the top of a k-step, OUTSIDE the woven region (it branches).
    auto stage_begin = [&]() {
        if (st_left == 0) {
            if (st_tile < my_tiles) {
                const Tile t = decode(st_tile);
                const int b = t.b0;
                const int l4 = fresh_lane() * 4;  // every constant below: this one register + scalar offsets
                st_x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.x_b_stride), 0, 0x7fffffff, 0x00020000);
                st_w = t.m_tile * n * (kUFloats * 4) + wave * (kUSlots * 1024);  // (of the wave's first weight piece)
                st_x = (t.y0 * Wp + t.x0) * 4;
                // the tile's constants (style of the sample, demod and bias of the channel tile) by dword LDS-DMA
                float* const set = const0 + (st_tile & 1) * kConstFloats;
                if constexpr (!kPrescaled) {
                    const __amdgpu_buffer_rsrc_t s_rsrc =
                        __builtin_amdgcn_make_buffer_rsrc((void*)(p.s + (size_t)b * p.s_stride), 0, p.Cin * 4, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_ptr_t)(set + wave * 64), 4, l4 + wave * 256, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_ptr_t)(set + (wave + 4) * 64), 4, l4 + (wave + 4) * 256, 0, 0, 0);  // (in the vector offset: the resource's bound clips Cin < 512)
                }
                if (wave < 2) {
                    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(wave == 0 ? p.d + (size_t)b * p.d_stride + t.m_tile * kBM : p.bias + t.m_tile * kBM), 0, kBM * 4, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lds_ptr_t)(set + 512 + wave * 64), 4, l4, 0, 0, 0);
                } else if (TG == 1 && wave == 2 && p.s_next != nullptr) {  // (the 32-channel geometry only runs the last layer)
                    // the style of the layer that reads this one's activation (an up layer taking its input pre-scaled)
                    const __amdgpu_buffer_rsrc_t c_rsrc =
                        __builtin_amdgcn_make_buffer_rsrc((void*)(p.s_next + (size_t)b * p.s_stride + t.m_tile * kBM), 0, kBM * 4, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lds_ptr_t)(set + 640), 4, l4, 0, 0, 0);
                }
                // ... and its noise: the 2 TG pixel rows x 32 columns this wave's epilogue adds, two rows per instruction
                // (fetched here, n k-steps ahead: a global load inside the epilogue cost its whole latency once per tile)
                if (p.noise != nullptr) {
                    const int nz_voff = ((l4 & 128) >> 5) * p.OW + (l4 & 127);  // ((lane >> 5) * OW + lane % 32) * 4
#pragma unroll
                    for (int tg = 0; tg < TG; ++tg) {
                        const int row = 2 * (wave * TG + tg);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(nz_rsrc, (lds_ptr_t)(set + kConstHead + row * kTW), 4, nz_voff,
                                                                 (b * p.noise_b_stride + (t.y0 + row) * p.OW + t.x0) * 4, 0, 0);
                    }
                }
                // ... and (RGB) the A operands of the tile's ToRGB product: [step][lane] for its sample, a row per instruction
                if constexpr (RGB) {
                    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.rgb_coef + ((size_t)b * p.m_tiles + t.m_tile) * (G::kRgbSteps * 64)), 0, G::kRgbSteps * 256, 0x00020000);
#pragma unroll
                    for (int i = 0; i < G::kRgbSteps / 4; ++i) {
                        const int step = wave + 4 * i;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(set + kRgbOff + step * 64), 4, l4, step * 256, 0, 0);
                    }
                }
                st_left = n;
                ++st_tile;
            } else {  // past the end: the last chunk again
                st_w -= kUFloats * 4;
                st_x -= x_step;
                st_left = 1;
            }
        }
        cur_off = st_off;
        cur_w_soff = st_w;
        cur_x_soff = st_x;
        st_w += kUFloats * 4;
        st_x += x_step;
        --st_left;
        st_off = st_off + kSlot == kNBUF * kSlot ? 0 : st_off + kSlot;
    };
    // slot r of the chunk set up by stage_begin(): one instruction, no control flow
    auto stage_piece = [&](int r) {
        if (r < kUSlots) {
            // weight piece g = kUSlots wave + r: ONE vector register (lane * 16) and one scalar offset for all of a wave's
            // pieces, r in the instruction's immediate (a register per piece did not survive the epilogue: spilled, and
            // its reload put a vmcnt wait into the k-step)
            // (the immediate offset of an LDS-DMA instruction moves BOTH addresses, global and LDS: one LDS base per wave)
            lds_ptr_t const dst = (lds_ptr_t)(smem + cur_off + kUSlots * wave * 256);
            switch (r) {  // (the immediate must be a literal; r is one after unrolling)
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16, cur_w_soff, 0, 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16, cur_w_soff, 1024, 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16, cur_w_soff, 2048, 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst, 16, lane16, cur_w_soff, 3072, 0); break;
            }
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_x_rsrc, (lds_ptr_t)(smem + cur_off + p_lds[r - kUSlots]), 16, p_voff[r - kUSlots], cur_x_soff, 0, 0);
        }
    };
    // NOTE: st_x_rsrc changes when stage_begin opens a tile; the pieces of that chunk are the first to use it.

    // ring prologue: the first NBUF - 1 chunks of the stream
    for (int c = 0; c < kNBUF - 1; ++c) {
        stage_begin();
#pragma unroll
        for (int r = 0; r < kPiecesPerWave; ++r) stage_piece(r);
    }

    f32x4 acc[16][TG][MT];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int tg = 0; tg < TG; ++tg)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[q][tg][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- pipeline registers ----
    f32x2 win[TG][4][2];  // raw windows of the k-step being transformed: [row][column pair]
    float sval = 0.f;
    float V[2][TG][16];
    // (slot_off, style_off: float offsets in smem. The window base is made opaque to the compiler: folded into the 16
    // reads as one big constant each, it cost a vector add per ds_read2 -- from one base the rows are immediates)
    auto load_window = [&](int slot_off, int style_off) {
        int po = slot_off + kUFloats + poff;
        asm volatile("" : "+v"(po));
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            const float* pc = smem + po + tg * 2 * kPW;
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) win[tg][y][cp] = *reinterpret_cast<const f32x2_lds*>(pc + y * kPW + 2 * cp);
        }
        if constexpr (!kPrescaled) sval = smem[style_off + kq];
    };
    // V = B^T (s d) B, B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]; the style scale rides on the window (V is linear in d).
    // Packed fp32 (v_pk_*_f32, two columns per instruction): the row pass works on column pairs; in the column pass
    // (V0, V3) of a row is one packed subtraction of its two pairs, (V1, V2) two scalar operations: 24 vector
    // instructions per window instead of 40 -- beside the fp32 MFMAs of its own wave every one of them costs its issue time.
    // Only where a wave transforms two windows per k-step: in the <4, 1> geometry the aligned register pairs made hipcc
    // park ten accumulator tiles in VGPRs and copy them in and out of every loop trip (+13 %); there the scalar form stays.
    auto transform = [&](float (&outs)[TG][16]) {
        if constexpr (TG == 1) {
            float tcol[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float w0 = win[0][0][c >> 1][c & 1], w1 = win[0][1][c >> 1][c & 1], w2 = win[0][2][c >> 1][c & 1], w3 = win[0][3][c >> 1][c & 1];
                if constexpr (kPrescaled) {
                    tcol[0][c] = w0 - w2;
                    tcol[1][c] = w1 + w2;
                    tcol[2][c] = w2 - w1;
                    tcol[3][c] = w1 - w3;
                } else {
                    const float e1 = sval * w1, e2 = sval * w2;
                    tcol[0][c] = fmaf(sval, w0, -e2);
                    tcol[1][c] = e1 + e2;
                    tcol[2][c] = e2 - e1;
                    tcol[3][c] = fmaf(-sval, w3, e1);
                }
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                outs[0][y * 4 + 0] = tcol[y][0] - tcol[y][2];
                outs[0][y * 4 + 1] = tcol[y][1] + tcol[y][2];
                outs[0][y * 4 + 2] = tcol[y][2] - tcol[y][1];
                outs[0][y * 4 + 3] = tcol[y][1] - tcol[y][3];
            }
            return;
        }
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            f32x2 t[4][2];
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                static_assert(TG == 1 || kPrescaled, "the packed transform is the <2, 2> geometry's: no style scale in it");
                t[0][cp] = win[tg][0][cp] - win[tg][2][cp];
                t[1][cp] = win[tg][1][cp] + win[tg][2][cp];
                t[2][cp] = win[tg][2][cp] - win[tg][1][cp];
                t[3][cp] = win[tg][1][cp] - win[tg][3][cp];
                // (kept opaque: where a packed result is only read by element, hipcc splits the operation again)
#pragma unroll
                for (int y = 0; y < 4; ++y) asm("" : "+v"(t[y][cp]));
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                f32x2 a = t[y][0] - t[y][1];  // (c0 - c2, c1 - c3)
                asm("" : "+v"(a));
                const float c1 = t[y][0][1], c2 = t[y][1][0];
                outs[tg][y * 4 + 0] = a[0];
                outs[tg][y * 4 + 1] = c1 + c2;
                outs[tg][y * 4 + 2] = c2 - c1;
                outs[tg][y * 4 + 3] = a[1];
            }
        }
    };

    // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), epilogue, stores; clears acc ----
    auto epilogue = [&](int i) {
        const Tile t = decode(i);
        const float* const d_lds = const0 + (i & 1) * kConstFloats + 512;
        const float* const b_lds = d_lds + 64;
        const float* const sn_lds = b_lds + 64;
        const float* const nz_lds = const0 + (i & 1) * kConstFloats + kConstHead;
        const int le = fresh_lane();
        const int n16 = le & 15, kq = le >> 4;  // (shadow the kernel's: see fresh_lane)
        const int c_stride_bytes = (int)p.out_c_stride * 4;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.out + (size_t)t.b0 * p.out_b_stride + (size_t)(t.m_tile * kBM) * p.out_c_stride), 0, 0x7fffffff, 0x00020000);
        // (RGB) The layer's ToRGB product rides on the matrix pipe as well: rgb[colour][pixel] = sum over the block's
        // channels of coefficient x activation is one more small GEMM, and the activations are already where its B operand
        // wants them -- v_mfma_f32_16x16x4 takes B[k][n] from lane (n = lane % 16, k = lane / 16), this lane holds channels
        // 16 mt + 4 kq + r of the pixels in column n16, so k-step (mt, r) of the product reads each lane's OWN channel. The
        // A operand (colour m = lane % 16 < 3, else 0; channel 16 mt + 4 (lane / 16) + r) comes from the table the tile's
        // constants brought in. Accumulators in VGPRs (every AGPR holds a conv accumulator), hence the inline assembly.
        // Lanes 0 .. 15 end up with (R, G, B, 0) of their column in the four result registers.
        const float* const rgb_lds = const0 + (i & 1) * kConstFloats + kRgbOff;
        f32x4 rgbacc[TG][2][2];
        if constexpr (RGB) {
#pragma unroll
            for (int tg = 0; tg < TG; ++tg)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) rgbacc[tg][dy][dx] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            const int oy = t.y0 + 2 * (wave * TG + tg), ox = t.x0 + 2 * n16;
            float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            if (p.noise != nullptr) {
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) nz[dy][dx] = nz_lds[(2 * (wave * TG + tg) + dy) * kTW + 2 * n16 + dx] * p.noise_strength;
            }
            const int voff0 = ((oy + p.out_y_off) * p.out_row_stride + ox + p.out_x_off) * 4 + 4 * kq * c_stride_bytes;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // One accumulator tile set at a time, fenced: the 16 positions of four channels (whole MFMA result
                // tuples: consuming single components of all 64 tuples channel by channel made hipcc shuffle the
                // accumulator file and spill), cleared as soon as they are read. Without the fence hipcc pulls all 256
                // accumulators into VGPRs at once.
                __builtin_amdgcn_sched_barrier(0);
                // demodulation and bias of this lane's channels mt * 16 + 4 kq + (0 .. 3): one 16-byte read each
                const f32x4 dm = *reinterpret_cast<const f32x4*>(d_lds + mt * 16 + 4 * kq);
                const f32x4 bm = *reinterpret_cast<const f32x4*>(b_lds + mt * 16 + 4 * kq);
                // (RGB) the A operands of the group's four ToRGB k-steps, read here with the other constants: read one by one
                // in front of their MFMAs, each cost a full LDS round trip with nothing else to issue
                float a_rgb[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (RGB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a_rgb[r] = rgb_lds[(mt * 4 + r) * 64 + le];
                    // (pinned here: left alone hipcc sinks the reads back down to their first use)
                    asm volatile("" : "+v"(a_rgb[0]), "+v"(a_rgb[1]), "+v"(a_rgb[2]), "+v"(a_rgb[3]));
                }
                f32x4 m[16];
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) {
                    m[pos] = acc[pos][tg][mt];
                    acc[pos][tg][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                f32x4 u[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[0][j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
                    u[1][j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
                }
                f32x4 y[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    y[dy][0] = u[dy][0] + u[dy][1] + u[dy][2];
                    y[dy][1] = u[dy][1] - u[dy][2] - u[dy][3];
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        f32x4 v = y[dy][dx] * dm;
                        v += nz[dy][dx] + bm;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.2f * v[r]) * 1.4142135623730951f;
                        y[dy][dx] = v;
                    }
                }
                if constexpr (RGB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a_r = a_rgb[r];
                        // (the four pixels of a k-step back to back: independent accumulators between dependent ones)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx)
                                // (s_nop: an MFMA may not read a register a vector instruction wrote in the cycle before it; hipcc
                                // places that wait state for the MFMAs it knows, not in front of this one: without it exactly the
                                // pixel whose activation was computed last came out wrong)
                                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(rgbacc[tg][dy][dx]) : "v"(a_r), "v"(y[dy][dx][r]));
                    }
                }
                if (GANCE_W64_ABLATE & 1) {
                    asm volatile("" ::"v"(y[0][0]), "v"(y[0][1]), "v"(y[1][0]), "v"(y[1][1]));
                    continue;
                }
                // (RGB: the network's last layer has no other reader: out == nullptr, nothing but the image leaves the chip)
                if (!RGB || p.out != nullptr) {
                    // (the stored activation carries the next layer's style when that layer wants it so: ConvArgs::s_next;
                    // the ToRGB product above took the plain one)
                    if (TG == 1 && p.s_next != nullptr) {
                        const f32x4 sn = *reinterpret_cast<const f32x4*>(sn_lds + mt * 16 + 4 * kq);
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) y[dy][dx] *= sn;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) {
                            u32x2 pair;
                            pair[0] = __float_as_uint(y[dy][0][r]);
                            pair[1] = __float_as_uint(y[dy][1][r]);
                            __builtin_amdgcn_raw_buffer_store_b64(pair, o_rsrc, voff0 + dy * p.out_row_stride * 4, (mt * 16 + r) * c_stride_bytes, 0);
                        }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RGB) {
            // the partial image (no bias, no skip image yet: torgb_kernel finishes it in place): [B][3][OH][OW], the two
            // pixels of a row as one 8-byte store per colour. The compiler does not know the assembly above is an MFMA:
            // the wait states between an MFMA and a read of its result are spelled out.
#pragma unroll
            for (int tg = 0; tg < TG; ++tg)
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(rgbacc[tg][0][0]), "+v"(rgbacc[tg][0][1]), "+v"(rgbacc[tg][1][0]), "+v"(rgbacc[tg][1][1]));
            const __amdgpu_buffer_rsrc_t y_rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.rgb_y + ((size_t)t.m_tile * p.B + t.b0) * 3 * p.OH * p.OW), 0, 0x7fffffff, 0x00020000);
            if (kq == 0) {
#pragma unroll
                for (int tg = 0; tg < TG; ++tg)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int voff = ((t.y0 + 2 * (wave * TG + tg) + dy) * p.OW + t.x0 + 2 * n16) * 4;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            u32x2 pair;
                            pair[0] = __float_as_uint(rgbacc[tg][dy][0][k]);
                            pair[1] = __float_as_uint(rgbacc[tg][dy][1][k]);
                            __builtin_amdgcn_raw_buffer_store_b64(pair, y_rsrc, voff, k * p.OH * p.OW * 4, 0);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue of the pipeline: first chunk visible, V of its k-step, its weight fragments ----
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kNBUF - 2) * G::kPiecesPerWave) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_window(0, kNBUF * kSlot);
    transform(V[0]);
    // weight fragments of a k-step are read during the k-step before it (2 x 16 x MT registers)
    afrag_t afrag[2][16];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) afrag[0][pos] = *reinterpret_cast<const afrag_t*>(buf0 + aoff + pos * kKC * kBM);

    // ---- the stream: k-step q of the block multiplies chunk q, ring slot q % NBUF ----
    // A k-step is ONE branch-free scheduling region: the 64 MFMAs of k-step q, and for k-step q+1 its 16 weight-fragment
    // reads (ds_read_b128), its window reads and its 40-instruction transform; six DMA pieces of chunk q+5. A wave alone on its SIMD
    // keeps the matrix pipe busy only if the other instructions sit in the 32-cycle shadows BETWEEN the MFMAs (in clumps
    // between groups of MFMAs they cost their full issue time: 0.60 of the peak); the sched_group_barrier pattern below
    // deals them out one MFMA at a time.
    int mul_off = 0;  // ring slot (float offset in smem) of the chunk being multiplied
    // tiles in the outer loop, their chunks in the inner one, the epilogue unconditionally after it: the accumulators
    // must not flow through a conditional (hipcc then moves all 256 of them through VGPRs and scratch)
    // (do-while, both: every block has a tile and a tile has chunks; the guard path of a `for` that hipcc adds -- all
    // accumulators zero, merged with the real path in front of the epilogue -- cost accumulator copies and spills there)
    int tile = 0;
    do {
        lane_offsets();
        // style of the NEXT k-step (float offset in smem): this tile's constant set, then chunk 0 of the next tile's
        const int next_set = kNBUF * kSlot + ((tile + 1) & 1) * kConstFloats;
        int style_n = kNBUF * kSlot + (tile & 1) * kConstFloats + kKC;
        int chunk = 0;
        do {
            // (two k-steps per trip so that V[0] / V[1] alternate with compile-time indices; n is even)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // the chunk after this one: the next ring slot (the stream's last k-step reads a slot nobody filled and
                // transforms it for nothing: no tail case)
                const int next_off = mul_off + kSlot == kNBUF * kSlot ? 0 : mul_off + kSlot;
                const float* const Un = smem + next_off;
                const int style_here = half == 1 && chunk + 2 == n ? next_set : style_n;
                // The MFMAs of k-step q have their operands in registers: the first sixteen go out BEFORE the wait and the
                // barrier, so the matrix pipe works while the wave does the staging bookkeeping (scalar, branchy) and waits
                // for its siblings.
                auto mfma_positions = [&](int first, int last) {
                    if (W64_DBG & 4) return;
#pragma unroll
                    for (int pos = first; pos < last; ++pos)
#pragma unroll
                        for (int tg = 0; tg < TG; ++tg)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[pos][tg][mt] =
                                    __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[half][pos][mt], V[half][tg][pos], acc[pos][tg][mt], 0, 0, 0);
                };
                __builtin_amdgcn_sched_barrier(0);
                mfma_positions(0, 4);
                stage_begin();
                __builtin_amdgcn_sched_barrier(0);
                // ONE wait and barrier per PAIR of k-steps (7 % of the kernel's time went into one per k-step): in front of
                // the even k-step q the wave waits for chunks q+1 and q+2 -- issued four and three k-steps ago, the two
                // younger chunks stay in flight -- and the block synchronises. Behind the barrier both are visible to every
                // wave, and the slots the pair's DMA pieces go to (chunks q+5 and q+6 into the slots of q-1 and q) were last
                // read in k-steps q-2 and q-1, which every wave has left.
                if (half == 0 && !(W64_DBG & 64)) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kNBUF - 4) * G::kPiecesPerWave) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                load_window(next_off, style_here);
#pragma unroll
                for (int pos = 0; pos < 16; ++pos) afrag[(half + 1) & 1][pos] = *reinterpret_cast<const afrag_t*>(Un + aoff + pos * kKC * kBM);
                if (!(W64_DBG & 32)) transform(V[(half + 1) & 1]);
                if (!(W64_DBG & 2)) {
#pragma unroll
                    for (int r = 0; r < kPiecesPerWave; ++r) stage_piece(r);
                }
                mfma_positions(4, 16);
                // weave: 48 x (1 MFMA, then what fits in its shadow): the LDS reads first (16 weight fragments + 9 per window:
                // they feed everything else), the 40 transform instructions per window and the DMA issues spread over the rest
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // 1 MFMA
                    if (i < 17 + (kPrescaled ? 8 : 9) * TG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read
                    if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x002, TG, 0);                  // 1 VALU per window
                    if (i % 8 == 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // 1 LDS-DMA issue
                }
                __builtin_amdgcn_sched_barrier(0);
                mul_off = next_off;
                style_n += kKC;
            }
            chunk += 2;
        } while (chunk < n);
        epilogue(tile);
    } while (++tile < my_tiles);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring may still be landing when the block's LDS is given back
}

bool winograd64_supported(int cin, int cout, int H, int W) {
    // (an even number of chunks per tile, and at least NBUF of them). 32-channel layers take the <2, 2> geometry.
    const bool geometry = cout % 64 == 0 ? (H % 8 == 0 && W % kTW == 0) : (cout % 32 == 0 && H % 16 == 0 && W % kTW == 0);
    return cin % (2 * kKC) == 0 && cin <= 512 && cin / kKC >= kNBUF && geometry;
}

bool winograd64_input_prescaled(int cout) { return cout % 64 != 0; }  // (the <2, 2> geometry: kPrescaled)

size_t winograd64_weight_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

// w_in: the layer's runtime-scaled weights [tap = ky*3+kx][ci][co]; w_out: [m tile of BM][chunk of 4][pos][ci][co % 16][co / 16]
// with BM = 64 channels per block when cout is a multiple of 64, else 32
void winograd64_transform_weights(const float* w_in, int cin, int cout, float* w_out) {
    const double G[4][3] = {{1., 0., 0.}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0., 0., 1.}};
    const int bm = cout % 64 == 0 ? 64 : 32, mt_count = bm / 16;
    const int chunks = cin / kKC, u_floats = 16 * kKC * bm;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double g[3][3], tmp[4][3], u[4][4];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w_in[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) tmp[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) u[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int mtile = co / bm, m = co % bm, ch = ci / kKC, kc = ci % kKC;
            float* dst = w_out + ((size_t)mtile * chunks + ch) * u_floats;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) dst[(((i * 4 + j) * kKC + kc) * 16 + m % 16) * mt_count + m / 16] = (float)u[i][j];
        }
}


// The kernels are plain functions around the templated body: as a kernel TEMPLATE the host pass of hipcc drops the
// instantiation without a diagnostic (the 16-byte LDS-DMA builtin does not pass the host's feature check) and the
// library is left with an undefined stub.
__global__ __launch_bounds__(256, 1) void winograd64_kernel(const ConvArgs p) { winograd64_body<4, 1, false>(p); }
__global__ __launch_bounds__(256, 1) void winograd64_c32_kernel(const ConvArgs p) { winograd64_body<2, 2, false>(p); }
__global__ __launch_bounds__(256, 1) void winograd64_rgb_kernel(const ConvArgs p) { winograd64_body<4, 1, true>(p); }
__global__ __launch_bounds__(256, 1) void winograd64_c32_rgb_kernel(const ConvArgs p) { winograd64_body<2, 2, true>(p); }

template <int MT, int TG, bool RGB>
static hipError_t launch_variant(void (*kernel)(const ConvArgs), const ConvArgs& args, hipStream_t stream) {
    using G = Geo<MT, TG>;
    static PerDeviceInt resident;  // per device: the dynamic-LDS opt-in and the launch size = one block per CU, a multiple of 8 (XCDs)
    int resident_blocks = 0;
    hipError_t e = resident.get(
        [&](int device, int* value) {
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes(RGB));
            if (err != hipSuccess) return err;
            int cus = 0;
            if ((err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return err;
            *value = std::max(8, cus / 8 * 8);
            return hipSuccess;
        },
        &resident_blocks);
    if (e != hipSuccess) return e;
    ConvArgs a = args;
    static const int env_debug = [] { const char* v = std::getenv("GANCE_DEBUG_W64"); return v ? std::atoi(v) : 0; }();
    a.debug_flags = env_debug;
    a.tiles_x = a.W / kTW;
    a.tiles_y = a.H / G::kTH;
    a.m_tiles = a.Cout / G::kBM;
    a.total_chunks = a.Cin / kKC;
    a.total_tiles = a.m_tiles * a.tiles_x * a.tiles_y * a.B;
    const int blocks = std::min(a.total_tiles, resident_blocks);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), G::lds_bytes(RGB), stream, a);
    return hipGetLastError();
}

// A block's channels are its whole share of the ToRGB product: with several channel tiles per pixel (Cout > 64) every
// tile's block writes its OWN partial image and torgb_kernel adds them in a fixed order (no atomics: results stay
// reproducible bit for bit).
bool winograd64_rgb_supported(int cout) { return cout == 32 || (cout % 64 == 0 && cout <= 512); }
int winograd64_rgb_partials(int cout) { return cout == 32 ? 1 : cout / 64; }

// rgb_coef[b][m tile][step = 4 mt + r][lane] = style[b][ch] * w[ch][lane % 16] for lane % 16 < 3, else 0,
// ch = BM m_tile + 16 mt + 4 (lane / 16) + r: the A operand of the fused ToRGB product, lane by lane (see the epilogue)
__global__ void winograd64_rgb_coef_kernel(const float* __restrict__ w, const float* __restrict__ s, int s_stride, int steps, int m_tiles,
                                           float* __restrict__ out) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < m_tiles * steps * 64; i += blockDim.x) {
        const int lane = i & 63, step = (i >> 6) % steps, m_tile = i / (64 * steps);
        const int ch = 4 * steps * m_tile + 16 * (step >> 2) + 4 * (lane >> 4) + (step & 3), colour = lane & 15;
        out[(size_t)b * m_tiles * steps * 64 + i] = colour < 3 ? s[(size_t)b * s_stride + ch] * w[ch * 3 + colour] : 0.f;
    }
}

hipError_t launch_winograd64_rgb_coef(const float* rgb_w, const float* rgb_s, int s_stride, int B, int cout, float* coef, hipStream_t stream) {
    if (!winograd64_rgb_supported(cout)) return hipErrorInvalidValue;
    const int m_tiles = winograd64_rgb_partials(cout);
    hipLaunchKernelGGL(winograd64_rgb_coef_kernel, dim3(B), dim3(256), 0, stream, rgb_w, rgb_s, s_stride, cout / m_tiles / 4, m_tiles, coef);
    return hipGetLastError();
}

hipError_t launch_winograd64_conv(const ConvArgs& args, hipStream_t stream) {
    if (args.epilogue == kEpilogueFullRgbPart) {
        // the partial image(s) [Cout / 64][B][3][OH][OW] and the coefficient table are the caller's
        if (!winograd64_rgb_supported(args.Cout) || args.rgb_coef == nullptr || args.rgb_y == nullptr) return hipErrorInvalidValue;
        return args.Cout % 64 == 0 ? launch_variant<4, 1, true>(winograd64_rgb_kernel, args, stream)
                                   : launch_variant<2, 2, true>(winograd64_c32_rgb_kernel, args, stream);
    }
    if (args.epilogue != kEpilogueFull || args.out == nullptr) return hipErrorInvalidValue;
    return args.Cout % 64 == 0 ? launch_variant<4, 1, false>(winograd64_kernel, args, stream)
                               : launch_variant<2, 2, false>(winograd64_c32_kernel, args, stream);
}

}  // namespace gance
