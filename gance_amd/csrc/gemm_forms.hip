// The layers at 4x4 ... 16x16 as dense fp32 GEMMs: position grids too small to tile, turned into matrix columns.
//
// (1) SCATTER FORM of Conv0_up at 4x4 -> 8x8 and 8x8 -> 16x16. The stride-2 transposed 3x3 convolution of `upsample_conv_2d`
// (called from the reference's synthesis, gance/network_interface/network_functions.py:168) is, per
// tap t, a 1x1 convolution P_t = W_t^T x of the (modulated) input followed by a scatter of P_t to output positions 2 y + wy,
// 2 x + wx. The gather forms of this layer (conv_mfma.hip's transposed tiles, upfir*_fused.hip) tile the (H+1) x (W+1) position
// grid of the four parity classes; on a 5 x 5 or 9 x 9 grid an 8 x 8 tile geometry keeps 81 of 256 tile slots busy (measured: 31
// TFLOP/s at 8x8 -> 16x16, batch 64). The scatter form has no grid to tile:
//   P[t * Cout + co][b * H W + y W + x] = sum_ci  W_t[ci][co] * (s[b][ci] * x[b][ci][y][x])
// is a plain GEMM, M = 9 Cout = 4608, K = Cin = 512, N = B H W (1024 / 4096 at batch 64), every MFMA slot useful. Three
// launches replace the transposed-conv launch (the FIR pass after them is unchanged):
//   1. upgemm_pack_kernel: the zero-bordered activation x style -> the GEMM's B image [n tile of 128][chunk of 16][16 columns
//      tile][16 k][16 n] (an LDS-DMA piece = one 16 x 16 operand tile in MFMA read order, conflict-free),
//   2. tile_gemm_kernel: 128 x 128 block tiles, four waves of 64 x 64 (16 accumulator tiles of v_mfma_f32_16x16x4_f32), K chunks
//      of 16 through a two-slot LDS-DMA ring, three blocks per CU; MFMA rows = positions, columns = channels, so that a lane's
//      four accumulator registers are four consecutive positions of one channel: 16-byte stores,
//   3. upgemm_gather_kernel: T_class[y'][x'] = d[b][co] * sum of the class's taps P_t[y' + dy][x' + dx] (zero outside the
//      image) into the parity planes the FIR pass (aux_kernels.hip fir_epilogue_kernel) reads.
// P costs 9 Cout N floats of HBM traffic each way (75 MB at 8x8 -> 16x16, batch 64: about 40 us) -- the price of no tile waste.
// Larger layers keep the gather forms: their position grids tile well and P would be 4x larger per level.
//
// (2) WINOGRAD F(4x4, 3x3) IN GEMM FORM for the stride-1 layers at 8x8 and 16x16 (`modulated_conv2d_layer` + `fused_bias_act`,
// a18). The fused F(4x4,3x3) kernel (winograd43_conv.hip) needs images at least 32 wide; below that the direct form ran (125
// TFLOP/s at 16x16: all 9 taps' flops). Here the three stages are three launches, the middle one the same GEMM kernel run as 36
// independent products (one per Winograd position g; the rows of group g read B image g):
//   1. winogemm_pack_kernel: V_g = (B^T (s x) B)_g of every 6 x 6 input window -> B image g, column n = sample * tiles + tile,
//   2. tile_gemm_kernel: M_g[co][n] = sum_ci U_g[ci][co] V_g[ci][n], U = G w G^T (host, fp64, the matrices of winograd43_conv.hip),
//   3. winogemm_finish_kernel: A^T M A per (tile, channel), x demod, + noise + bias, leaky ReLU x sqrt 2 -> the bordered activation.
// A quarter of the direct form's flops, every MFMA slot useful; 36 Cin N + 36 Cout N floats through HBM (150 MB at 16x16, batch 64).
//
// (3) EXPERIMENT, off by default (GANCE_TUNE_GEMM_BF16X6=1 at engine creation): the same GEMMs on the bf16 matrix cores with fp32
// accuracy. Each fp32 operand is split into three bf16 numbers that hold its 24 mantissa bits exactly (x = x0 + x1 + x2); a product is
// the sum of the six largest of the nine bf16 x bf16 products (x0 w0, x0 w1, x1 w0, x1 w1, x0 w2, x2 w0: the three left out are below
// 2^-25 |x w|), each exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16, which does 8192 MACs in 16 cycles where the fp32 MFMA
// does 1024 in 32: six terms cost 0.375 of the fp32 matrix time (tools/experiments/bf16_split_error.py: the error of a K = 4608 sum
// against fp64 is 4e-7 of the largest output, fp32 MFMAs in their order 1.2e-6). The pack kernels write the three parts of the B image,
// the host the three of the weights; operand tile = [part][8-channel group][16][8] bf16, so a lane's fragment is one ds_read_b128; block
// tile 256 x 128, eight waves of 64 x 64, K chunks of 32 through a two-slot ring of 72 KB each. It is the first hardware data point of the
// lever DESIGN.md section 9 prices for the convolutions; the contract line does not use it.

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kGM = 128, kGN = 128, kGK = 16;  // block tile and K chunk
constexpr int kTileFloats = kGK * 128;         // an operand tile of a chunk: [8 tiles of 16][16 k][16] = 8 KB = 8 DMA pieces

// fp32 -> three bf16 (round to nearest even each time): x = p[0] + p[1] + p[2] exactly for every finite x whose parts stay normal
__host__ __device__ inline unsigned short bf16_rne(float x) {
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_value(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float x;
    __builtin_memcpy(&x, &u, 4);
    return x;
}
__host__ __device__ inline unsigned short f16_bits(float x) {
    const _Float16 h = (_Float16)x;  // round to nearest even, subnormals kept
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__host__ __device__ inline float f16_value(unsigned short u) {
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
// split mode 1: three bf16 parts (24 mantissa bits, the fp32 exponent range); mode 2: two fp16 parts (22 bits; 4 bytes per value like
// fp32, three product terms instead of six -- but fp16's exponent range: residuals below 6e-5 lose bits, values above 65504 overflow)
__host__ __device__ constexpr int split_parts(int mode) { return mode == 2 ? 2 : 3; }
// fp16 x 2 only: the weight images are stored times a power of two (exact) that lifts them into fp16's normal range -- runtime-scaled
// weights are ~ 1 / sqrt(9 Cin) = 0.015, their Winograd transforms down to 1 / 576 of that: residuals of such values are fp16 subnormals
// and lose bits --, and the gather / finish kernels multiply the products by its inverse.
__host__ __device__ constexpr float split_weight_scale(int mode, bool winograd) { return mode == 2 ? (winograd ? 4096.f : 64.f) : 1.f; }
__host__ __device__ inline void split_value(float x, int mode, unsigned short (&part)[3]) {
    if (mode == 2) {
        part[0] = f16_bits(x);
        part[1] = f16_bits(x - f16_value(part[0]));
        part[2] = 0;
        return;
    }
    part[0] = bf16_rne(x);
    const float r1 = x - bf16_value(part[0]);
    part[1] = bf16_rne(r1);
    part[2] = bf16_rne(r1 - bf16_value(part[1]));
}
constexpr int kSK = 32;  // split forms: k per chunk = one k-step of v_mfma_f32_16x16x32_{bf16,f16}
// index (in 16-bit values) of element (row r of the operand, channel k), part 0, in an image
// [row tile of `rows`][chunk][16-row tile][part][k / 8][16][8]: a (tile, part) is 512 values = 1 KB = one LDS-DMA piece
__host__ __device__ inline size_t split_index(int r, int k, int rows, int chunks, int parts) {
    return ((((size_t)(r / rows) * chunks + k / kSK) * (rows / 16) + (r % rows) / 16) * (parts * 512)) + ((k % kSK) / 8) * 128 + (r % 16) * 8 + k % 8;
}

__host__ __device__ constexpr int up_tap_cls(int t) { return t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3)); }

// ---- 1. pack: thread = one float of the B image ----
__global__ __launch_bounds__(256) void upgemm_pack_kernel(const UpGemmArgs p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)p.n_tiles * p.Cin * kGN;
    if (i >= total) return;
    // i = (((n_tile * chunks + chunk) * 8 + nt) * 16 + k) * 16 + n16
    const int n16 = (int)(i & 15), k = (int)((i >> 4) & 15), nt = (int)((i >> 8) & 7);
    const size_t tc = i >> 11;
    const int chunks = p.Cin / kGK;
    const int chunk = (int)(tc % chunks), n_tile = (int)(tc / chunks);
    const int n = n_tile * kGN + nt * 16 + n16, ci = chunk * kGK + k;
    const int hw = p.H * p.W;
    const int b = n / hw, pos = n - b * hw;
    float v = 0.f;
    if (b < p.B) {
        const int y = pos / p.W, x = pos - y * p.W;
        v = p.x[(size_t)b * p.x_b_stride + ((size_t)ci * (p.H + 2) + y + 1) * (p.W + 8) + x + 4] * p.s[(size_t)b * p.s_stride + ci];
    }
    if (p.bf16_split) {
        // (the same thread -> element map; consecutive threads are consecutive columns: 2-byte stores 16 bytes apart -- small images)
        unsigned short part[3];
        split_value(v, p.bf16_split, part);
        const int parts = split_parts(p.bf16_split);
        unsigned short* const dst = reinterpret_cast<unsigned short*>(p.packed) + split_index(n, ci, kGN, p.Cin / kSK, parts);
        for (int q = 0; q < parts; ++q) dst[q * 512] = part[q];
    } else {
        p.packed[i] = v;
    }
}

// ---- the GEMM: C[m][n] = sum_k A[k][m] B[k][n]; rows in groups of m_tiles_per_group tiles, group g reads B image g ----
struct TileGemmArgs {
    const float* w;       // A image [m tile][chunk][8][16][16]
    const float* packed;  // B images [group][n tile][chunk][8][16][16]
    float* prod;          // C [m][n_tiles * 128]
    int n_tiles, Cin, m_tiles_per_group;
};

#ifndef GANCE_TILE_GEMM_BLOCKS
#define GANCE_TILE_GEMM_BLOCKS 3
#endif
__global__ __launch_bounds__(256, GANCE_TILE_GEMM_BLOCKS) void tile_gemm_kernel(const TileGemmArgs p) {
    __shared__ float smem[2 * 2 * kTileFloats];  // ring of two slots: A tile | B tile
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n16 = lane & 15, q4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;  // the wave's 64 channels / 64 positions of the block tile
    // (consecutive blocks share the weight tile and walk the position tiles)
    const int n_tile = blockIdx.x % p.n_tiles, m_tile = blockIdx.x / p.n_tiles;
    const int chunks = p.Cin / kGK;
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)m_tile * chunks * kTileFloats), 0, chunks * kTileFloats * 4, 0x00020000);
    const int group = m_tile / p.m_tiles_per_group;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.packed + ((size_t)group * p.n_tiles + n_tile) * chunks * kTileFloats), 0, chunks * kTileFloats * 4, 0x00020000);
    // a wave stages pieces 2 w, 2 w + 1 of both tiles of a chunk
    auto stage = [&](int chunk, float* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = 2 * wave + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(buf + piece * 256), 16, (piece * 256 + lane * 4) * 4, chunk * kTileFloats * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr_t)(buf + kTileFloats + piece * 256), 16, (piece * 256 + lane * 4) * 4,
                                                     chunk * kTileFloats * 4, 0, 0);
        }
    };
    f32x4 acc[4][4];  // [position tile][channel tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, smem);
    for (int k = 0; k < chunks; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // this chunk has landed for every wave, and every wave has left the other slot
        asm volatile("" ::: "memory");
        const float* const cur = smem + (k & 1) * (2 * kTileFloats);
        if (k + 1 < chunks) stage(k + 1, smem + ((k + 1) & 1) * (2 * kTileFloats));
        // operand tile [tile][k][16]: lane (n16, q4) of k-step j reads [tile][4 j + q4][n16] -- 64 consecutive floats per read
        const float* const wl = cur + (4 * wm) * 256 + q4 * 16 + n16;
        const float* const xl = cur + kTileFloats + (4 * wn) * 256 + q4 * 16 + n16;
#pragma unroll
        for (int j = 0; j < kGK / 4; ++j) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = wl[i * 256 + 64 * j];
                b[i] = xl[i * 256 + 64 * j];
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[pt], a[ct], acc[pt][ct], 0, 0, 0);
        }
    }
    // accumulator tile (pt, ct), lane (n16, q4): channel row m = .. + n16, positions n = .. + 4 q4 .. + 3
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const size_t m = (size_t)m_tile * kGM + (4 * wm + ct) * 16 + n16;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int n = n_tile * kGN + (4 * wn + pt) * 16 + 4 * q4;
            *reinterpret_cast<f32x4*>(p.prod + m * ((size_t)p.n_tiles * kGN) + n) = acc[pt][ct];
        }
    }
}

// ---- the same product from split operands on the 16-bit matrix cores (experiment, see the header): 256 x 128 block tiles, eight waves ----
// PARTS = 3: bf16, six product terms; PARTS = 2: fp16, three terms
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kSM = 256;  // rows (channels) per block

template <int PARTS>
__global__ __launch_bounds__(512, 1) void tile_gemm_split_kernel(const TileGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_split[];  // ring of two slots
    constexpr int kTileBytes = PARTS * 1024;  // a 16-row tile of a chunk: [part][k group of 8][16][8]
    constexpr int kABytes = (kSM / 16) * kTileBytes, kBBytes = (kGN / 16) * kTileBytes, kSlotBytes = kABytes + kBBytes;
    constexpr int kPieces = kSlotBytes / 1024;  // 72 (48 of weights, 24 of columns) or 48
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n16 = lane & 15, q4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;  // the wave's 64 channels of the block's 256 / 64 positions of its 128
    const int n_tile = blockIdx.x % p.n_tiles, m_tile = blockIdx.x / p.n_tiles;
    const int chunks = p.Cin / kSK;
    const int group = m_tile / p.m_tiles_per_group;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(p.w) + (size_t)m_tile * chunks * kABytes), 0, chunks * kABytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(p.packed) + ((size_t)group * p.n_tiles + n_tile) * chunks * kBBytes), 0, chunks * kBBytes, 0x00020000);
    // wave w stages pieces w, w + 8, ...
    auto stage = [&](int chunk, char* buf) {
#pragma unroll
        for (int i = 0; i < kPieces / 8; ++i) {
            const int piece = wave + 8 * i;
            if (piece < kABytes / 1024)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(buf + piece * 1024), 16, piece * 1024 + lane * 16, chunk * kABytes, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr_t)(buf + piece * 1024), 16, (piece - kABytes / 1024) * 1024 + lane * 16,
                                                         chunk * kBBytes, 0, 0);
        }
    };
    f32x4 acc[4][4];  // [position tile][channel tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, smem_split);
    for (int k = 0; k < chunks; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* const cur = smem_split + (k & 1) * kSlotBytes;
        if (k + 1 < chunks) stage(k + 1, smem_split + ((k + 1) & 1) * kSlotBytes);
        // a lane's fragment of (tile, part): 8 consecutive channels k = 8 q4 .. + 7 of row n16 = bytes [q4][n16][8] of the part: lane x 16
        f32x4 a[4][PARTS], b[4][PARTS];  // (16 bytes = eight 16-bit values)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < PARTS; ++q) {
                a[t][q] = *reinterpret_cast<const f32x4*>(cur + (4 * wm + t) * kTileBytes + q * 1024 + lane * 16);
                b[t][q] = *reinterpret_cast<const f32x4*>(cur + kABytes + (4 * wn + t) * kTileBytes + q * 1024 + lane * 16);
            }
        // the part products, smallest first. bf16 x 3: six of the nine (x2 w0) (x0 w2) (x1 w1) (x1 w0) (x0 w1) (x0 w0); fp16 x 2: (x1 w0) (x0 w1) (x0 w0)
        constexpr int kNumTerms = PARTS == 3 ? 6 : 3;
        constexpr int kTerms[6][2] = {{PARTS == 3 ? 2 : 1, 0}, {0, PARTS == 3 ? 2 : 1}, {PARTS == 3 ? 1 : 0, PARTS == 3 ? 1 : 0}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int term = 0; term < kNumTerms; ++term)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    if constexpr (PARTS == 3)
                        acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[pt][kTerms[term][0]]),
                                                                              __builtin_bit_cast(bf16x8, a[ct][kTerms[term][1]]), acc[pt][ct], 0, 0, 0);
                    else
                        acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[pt][kTerms[term][0]]),
                                                                             __builtin_bit_cast(f16x8, a[ct][kTerms[term][1]]), acc[pt][ct], 0, 0, 0);
                }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const size_t m = (size_t)m_tile * kSM + (4 * wm + ct) * 16 + n16;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int n = n_tile * kGN + (4 * wn + pt) * 16 + 4 * q4;
            *reinterpret_cast<f32x4*>(p.prod + m * ((size_t)p.n_tiles * kGN) + n) = acc[pt][ct];
        }
    }
}

hipError_t launch_tile_gemm_split(const TileGemmArgs& g, int m_rows, int mode, hipStream_t stream) {
    static PerDeviceInt ready;
    int unused = 0;
    constexpr int kSlot3 = (kSM / 16 + kGN / 16) * 3 * 1024, kSlot2 = (kSM / 16 + kGN / 16) * 2 * 1024;
    const hipError_t e = ready.get(
        [&](int, int* value) {
            *value = 1;
            const hipError_t err =
                hipFuncSetAttribute(reinterpret_cast<const void*>(tile_gemm_split_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlot3);
            if (err != hipSuccess) return err;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(tile_gemm_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlot2);
        },
        &unused);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(g.n_tiles * (m_rows / kSM)));
    if (mode == 2) hipLaunchKernelGGL(tile_gemm_split_kernel<2>, grid, dim3(512), 2 * kSlot2, stream, g);
    else hipLaunchKernelGGL(tile_gemm_split_kernel<3>, grid, dim3(512), 2 * kSlot3, stream, g);
    return hipGetLastError();
}

// ---- 3. gather: thread = one position (y', x') of the (H+1) x (W+1) grid of one (sample, channel): its four classes ----
__global__ __launch_bounds__(256) void upgemm_gather_kernel(const UpGemmArgs p) {
    const int PH = p.H + 1, PW = p.W + 1;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)p.B * p.Cout * PH * PW;
    if (i >= total) return;
    const int xq = (int)(i % PW), yq = (int)((i / PW) % PH);
    const size_t bc = i / ((size_t)PW * PH);
    const int co = (int)(bc % p.Cout), b = (int)(bc / p.Cout);
    const size_t N = (size_t)p.n_tiles * kGN;
    const float* const src = p.prod + (size_t)co * N + (size_t)b * p.H * p.W;
    const size_t tap_stride = (size_t)p.Cout * N;
    auto at = [&](int t, int dy, int dx) {
        const int y = yq + dy, x = xq + dx;
        return (y >= 0 && y < p.H && x >= 0 && x < p.W) ? src[t * tap_stride + y * p.W + x] : 0.f;
    };
    // tap slots (engine.hip kUpTapWeight): EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO (0,0)
    const float d = p.d[(size_t)b * p.d_stride + co] * (1.0f / split_weight_scale(p.bf16_split, false));
    const float cls[4] = {(at(0, 0, 0) + at(1, 0, -1)) + (at(2, -1, 0) + at(3, -1, -1)), at(4, 0, 0) + at(5, -1, 0), at(6, 0, 0) + at(7, 0, -1), at(8, 0, 0)};
    float* const dst = p.t + (size_t)b * p.unit_stride + (size_t)co * (p.H + 3) * (p.W + 8) + (size_t)(yq + 1) * (p.W + 8) + xq + 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[(size_t)c * p.cls_stride] = cls[c] * d;
}

// ---- Winograd F(4x4, 3x3) stage 1: thread = one (column n = sample * tiles + tile, input channel) ----
__global__ __launch_bounds__(256) void winogemm_pack_kernel(const WinoGemmArgs p) {
    const int N = p.n_tiles * kGN;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * p.Cin) return;
    // (bf16 form: eight consecutive threads = the eight channels of one 16-byte group of the image, then consecutive columns: a wave stores 128
    // contiguous bytes per part and position instead of 64 two-byte pieces 16 bytes apart)
    const int n = p.bf16_split ? (int)((i >> 3) % N) : (int)(i % N);
    const int ci = p.bf16_split ? (int)((i >> 3) / N) * 8 + (int)(i & 7) : (int)(i / N);
    const int tiles_x = p.W / 4, tiles = tiles_x * (p.H / 4);
    const int b = n / tiles, tile = n - b * tiles;
    float v[6][6];
    if (b < p.B) {
        // window rows 4 ty - 1 .. 4 ty + 4 of the image = rows 4 ty .. 4 ty + 5 of the bordered plane (interior at [y + 1][x + 4])
        const float* src = p.x + (size_t)b * p.x_b_stride + ((size_t)ci * (p.H + 2) + 4 * (tile / tiles_x)) * (p.W + 8) + 4 * (tile % tiles_x) + 3;
        const float sv = p.s[(size_t)b * p.s_stride + ci];
        float d[6][6], t[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) d[r][c] = src[r * (p.W + 8) + c] * sv;
        // B^T d: rows (4 d0 - 5 d2 + d4 | -4 d1 - 4 d2 + d3 + d4 | 4 d1 - 4 d2 - d3 + d4 | -2 d1 - d2 + 2 d3 + d4 | 2 d1 - d2 - 2 d3 + d4 | 4 d1 - 5 d3 + d5)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            t[0][c] = 4.f * d[0][c] - 5.f * d[2][c] + d[4][c];
            t[1][c] = -4.f * (d[1][c] + d[2][c]) + d[3][c] + d[4][c];
            t[2][c] = 4.f * (d[1][c] - d[2][c]) - d[3][c] + d[4][c];
            t[3][c] = -2.f * d[1][c] - d[2][c] + 2.f * d[3][c] + d[4][c];
            t[4][c] = 2.f * d[1][c] - d[2][c] - 2.f * d[3][c] + d[4][c];
            t[5][c] = 4.f * d[1][c] - 5.f * d[3][c] + d[5][c];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            v[r][0] = 4.f * t[r][0] - 5.f * t[r][2] + t[r][4];
            v[r][1] = -4.f * (t[r][1] + t[r][2]) + t[r][3] + t[r][4];
            v[r][2] = 4.f * (t[r][1] - t[r][2]) - t[r][3] + t[r][4];
            v[r][3] = -2.f * t[r][1] - t[r][2] + 2.f * t[r][3] + t[r][4];
            v[r][4] = 2.f * t[r][1] - t[r][2] - 2.f * t[r][3] + t[r][4];
            v[r][5] = 4.f * t[r][1] - 5.f * t[r][3] + t[r][5];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) v[r][c] = 0.f;
    }
    if (p.bf16_split) {
        const int chunks = p.Cin / kSK, parts = split_parts(p.bf16_split);
        const size_t image = (size_t)p.n_tiles * chunks * 8 * (parts * 512);
        unsigned short* const dst = reinterpret_cast<unsigned short*>(p.packed) + split_index(n, ci, kGN, chunks, parts);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                unsigned short part[3];
                split_value(v[r][c], p.bf16_split, part);
                for (int q = 0; q < parts; ++q) dst[(size_t)(r * 6 + c) * image + q * 512] = part[q];
            }
        return;
    }
    // B image g: [n tile][chunk][column tile][k][16]
    const int chunks = p.Cin / kGK;
    const size_t in_image = (((size_t)(n / kGN) * chunks + ci / kGK) * 8 + (n % kGN) / 16) * 256 + (ci % kGK) * 16 + n % 16;
    const size_t image = (size_t)p.n_tiles * chunks * kTileFloats;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) p.packed[(size_t)(r * 6 + c) * image + in_image] = v[r][c];
}

// ---- stage 3: thread = one (column n, output channel): A^T M A, demod, noise, bias, leaky ReLU ----
__global__ __launch_bounds__(256) void winogemm_finish_kernel(const WinoGemmArgs p) {
    const int N = p.n_tiles * kGN;
    const int tiles_x = p.W / 4, tiles = tiles_x * (p.H / 4);
    const int columns = p.B * tiles;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)columns * p.Cout) return;
    const int n = (int)(i % columns), co = (int)(i / columns);
    const int b = n / tiles, tile = n - b * tiles;
    const float* src = p.prod + (size_t)co * N + n;
    const size_t group = (size_t)p.Cout * N;
    float m[6][6], t[4][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) m[r][c] = src[(size_t)(r * 6 + c) * group];
    // A^T m: rows (m0 + m1 + m2 + m3 + m4 | m1 - m2 + 2 (m3 - m4) | m1 + m2 + 4 (m3 + m4) | m1 - m2 + 8 (m3 - m4) + m5)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float s12 = m[1][c] + m[2][c], d12 = m[1][c] - m[2][c], s34 = m[3][c] + m[4][c], d34 = m[3][c] - m[4][c];
        t[0][c] = m[0][c] + s12 + s34;
        t[1][c] = d12 + 2.f * d34;
        t[2][c] = s12 + 4.f * s34;
        t[3][c] = d12 + 8.f * d34 + m[5][c];
    }
    const float d = p.d[(size_t)b * p.d_stride + co] * (1.0f / split_weight_scale(p.bf16_split, true));
    const float bias = p.bias[co];
    const int oy0 = 4 * (tile / tiles_x), ox0 = 4 * (tile % tiles_x);
    const float* const nz = p.noise != nullptr ? p.noise + (size_t)b * p.noise_b_stride + (size_t)oy0 * p.W + ox0 : nullptr;
    float* const dst = p.out + (size_t)b * p.out_b_stride + ((size_t)co * (p.H + 2) + oy0 + 1) * (p.W + 8) + ox0 + 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float s12 = t[r][1] + t[r][2], d12 = t[r][1] - t[r][2], s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
        f32x4 y = {t[r][0] + s12 + s34, d12 + 2.f * d34, s12 + 4.f * s34, d12 + 8.f * d34 + t[r][5]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = y[c] * d + bias;
            if (nz != nullptr) v += nz[r * p.W + c] * p.noise_strength;
            y[c] = fmaxf(v, 0.2f * v) * 1.4142135623730951f;  // lrelu(0.2) * sqrt(2)
        }
        *reinterpret_cast<f32x4*>(dst + r * (p.W + 8)) = y;
    }
}

}  // namespace

// (4x4 and 8x8 inputs at every batch; 32x32 and 64x64 inputs for the calls too small for the fused up kernel, where the transposed-conv
// tiles ran at half the GEMM's rate: the engine takes the form up to upgemm_max_columns(cout, ...) columns, which is what its buffers are sized for)
bool upgemm_supported(int cin, int cout, int H, int W) {
    return H == W && (H == 4 || H == 8 || H == 32 || H == 64 || H == 128) && cin % kSK == 0 && cout % kGM == 0;
}
size_t upgemm_weight_floats(int cin, int cout) { return (size_t)9 * cin * cout; }
int upgemm_n_tiles(int B, int H, int W) { return (B * H * W + kGN - 1) / kGN; }
size_t upgemm_packed_floats(int B, int cin, int H, int W) { return (size_t)upgemm_n_tiles(B, H, W) * kGN * cin; }
size_t upgemm_prod_floats(int B, int cout, int H, int W) { return (size_t)upgemm_n_tiles(B, H, W) * kGN * 9 * cout; }

// [m tile of 128][chunk of 16][channel tile of 16][k][16 channels]; GEMM row m = tap slot * cout + channel
void upgemm_arrange_weights(const float* w_in, int cin, int cout, const int* up_tap_weight, float* w_out) {
    const int chunks = cin / kGK, m_tiles = 9 * cout / kGM;
    for (int mt = 0; mt < m_tiles; ++mt)
        for (int ch = 0; ch < chunks; ++ch)
            for (int tile = 0; tile < 8; ++tile)
                for (int k = 0; k < kGK; ++k)
                    for (int r = 0; r < 16; ++r) {
                        const int m = mt * kGM + tile * 16 + r;
                        const int slot = m / cout, co = m % cout, ci = ch * kGK + k;
                        w_out[((((size_t)mt * chunks + ch) * 8 + tile) * kGK + k) * 16 + r] = w_in[((size_t)up_tap_weight[slot] * cin + ci) * cout + co];
                    }
}

// the same rows as three bf16 parts: [row tile of 256][chunk of 32][16-row tile][part][k / 8][16][8] (1.5 x the fp32 image's bytes)
void upgemm_arrange_weights_split(const float* w_in, int cin, int cout, const int* up_tap_weight, int mode, void* w_out) {
    unsigned short* const out = static_cast<unsigned short*>(w_out);
    const int parts = split_parts(mode);
    for (int m = 0; m < 9 * cout; ++m)
        for (int ci = 0; ci < cin; ++ci) {
            unsigned short part[3];
            split_value(w_in[((size_t)up_tap_weight[m / cout] * cin + ci) * cout + m % cout] * split_weight_scale(mode, false), mode, part);
            for (int q = 0; q < parts; ++q) out[split_index(m, ci, kSM, cin / kSK, parts) + q * 512] = part[q];
        }
}

hipError_t launch_upgemm(const UpGemmArgs& args, hipStream_t stream) {
    if (!upgemm_supported(args.Cin, args.Cout, args.H, args.W) || args.n_tiles != upgemm_n_tiles(args.B, args.H, args.W)) return hipErrorInvalidValue;
    const size_t packed = (size_t)args.n_tiles * kGN * args.Cin;
    hipLaunchKernelGGL(upgemm_pack_kernel, dim3((unsigned)((packed + 255) / 256)), dim3(256), 0, stream, args);
    if (args.bf16_split) {
        if (args.Cin % kSK != 0 || (9 * args.Cout) % kSM != 0) return hipErrorInvalidValue;
        const TileGemmArgs g{args.w, args.packed, args.prod, args.n_tiles, args.Cin, 9 * args.Cout / kSM};
        const hipError_t e = launch_tile_gemm_split(g, 9 * args.Cout, args.bf16_split, stream);
        if (e != hipSuccess) return e;
    } else {
        const int m_tiles = 9 * args.Cout / kGM;
        const TileGemmArgs g{args.w, args.packed, args.prod, args.n_tiles, args.Cin, m_tiles};
        hipLaunchKernelGGL(tile_gemm_kernel, dim3((unsigned)(args.n_tiles * m_tiles)), dim3(256), 0, stream, g);
    }
    const size_t outs = (size_t)args.B * args.Cout * (args.H + 1) * (args.W + 1);
    hipLaunchKernelGGL(upgemm_gather_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, stream, args);
    return hipGetLastError();
}

// (8x8 and 16x16 at every batch; 32x32 ... 128x128 for the calls too small for the fused F(4x4,3x3) kernel to fill the chip: the engine takes
// the form up to kWinoGemmMaxColumns columns, which is what its buffers are sized for)
bool winogemm_supported(int cin, int cout, int H, int W) {
    return H == W && H >= 8 && H <= 128 && (H & (H - 1)) == 0 && cin % kSK == 0 && cout % kGM == 0;
}
size_t winogemm_weight_floats(int cin, int cout) { return (size_t)36 * cin * cout; }
int winogemm_n_tiles(int B, int H, int W) { return (B * (H / 4) * (W / 4) + kGN - 1) / kGN; }
size_t winogemm_packed_floats(int B, int cin, int H, int W) { return (size_t)winogemm_n_tiles(B, H, W) * kGN * 36 * cin; }
size_t winogemm_prod_floats(int B, int cout, int H, int W) { return (size_t)winogemm_n_tiles(B, H, W) * kGN * 36 * cout; }

// A image of the 36 products: GEMM row m = position g * cout + channel, g = 6 r + c of U = G w G^T (r along y)
void winogemm_arrange_weights(const float* w_in, int cin, int cout, float* w_out) {
    const double G[6][3] = {{1. / 4, 0., 0.},          {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    const int chunks = cin / kGK;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double gk[3][3], tmp[6][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) gk[ky][kx] = w_in[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
            for (int r = 0; r < 6; ++r)
                for (int kx = 0; kx < 3; ++kx) tmp[r][kx] = G[r][0] * gk[0][kx] + G[r][1] * gk[1][kx] + G[r][2] * gk[2][kx];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    const double u = tmp[r][0] * G[c][0] + tmp[r][1] * G[c][1] + tmp[r][2] * G[c][2];
                    const int m = (r * 6 + c) * cout + co;
                    const int mt = m / kGM, tile = (m % kGM) / 16, row = m % 16;
                    w_out[((((size_t)mt * chunks + ci / kGK) * 8 + tile) * kGK + ci % kGK) * 16 + row] = (float)u;
                }
        }
}

void winogemm_arrange_weights_split(const float* w_in, int cin, int cout, int mode, void* w_out) {
    std::vector<float> plain(winogemm_weight_floats(cin, cout));
    winogemm_arrange_weights(w_in, cin, cout, plain.data());
    unsigned short* const out = static_cast<unsigned short*>(w_out);
    const int chunks = cin / kGK, parts = split_parts(mode);
    for (int m = 0; m < 36 * cout; ++m)
        for (int ci = 0; ci < cin; ++ci) {
            unsigned short part[3];
            split_value(plain[((((size_t)(m / kGM) * chunks + ci / kGK) * 8 + (m % kGM) / 16) * kGK + ci % kGK) * 16 + m % 16] * split_weight_scale(mode, true), mode,
                        part);
            for (int q = 0; q < parts; ++q) out[split_index(m, ci, kSM, cin / kSK, parts) + q * 512] = part[q];
        }
}

hipError_t launch_winogemm(const WinoGemmArgs& args, hipStream_t stream) {
    if (!winogemm_supported(args.Cin, args.Cout, args.H, args.W) || args.n_tiles != winogemm_n_tiles(args.B, args.H, args.W)) return hipErrorInvalidValue;
    const size_t columns = (size_t)args.n_tiles * kGN;
    hipLaunchKernelGGL(winogemm_pack_kernel, dim3((unsigned)((columns * args.Cin + 255) / 256)), dim3(256), 0, stream, args);
    if (args.bf16_split) {
        if (args.Cin % kSK != 0 || args.Cout % kSM != 0) return hipErrorInvalidValue;
        const TileGemmArgs g{args.w, args.packed, args.prod, args.n_tiles, args.Cin, args.Cout / kSM};
        const hipError_t e = launch_tile_gemm_split(g, 36 * args.Cout, args.bf16_split, stream);
        if (e != hipSuccess) return e;
    } else {
        const int per_group = args.Cout / kGM;
        const TileGemmArgs g{args.w, args.packed, args.prod, args.n_tiles, args.Cin, per_group};
        hipLaunchKernelGGL(tile_gemm_kernel, dim3((unsigned)(args.n_tiles * 36 * per_group)), dim3(256), 0, stream, g);
    }
    const size_t outs = (size_t)args.B * (args.H / 4) * (args.W / 4) * args.Cout;
    hipLaunchKernelGGL(winogemm_finish_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace gance
