// Conv0_up of the two smallest up layers (4x4 -> 8x8, 8x8 -> 16x16) in scatter form: ONE dense fp32 GEMM per layer.
//
// The stride-2 transposed 3x3 convolution of `upsample_conv_2d` (called from the reference's
// synthesis, gance/network_interface/network_functions.py:168) is, per tap t, a 1x1 convolution P_t = W_t^T x of the
// (modulated) input followed by a scatter of P_t to output positions 2 y + wy, 2 x + wx. The gather forms of this layer
// (conv_mfma.hip's transposed tiles, upfir*_fused.hip) tile the (H+1) x (W+1) position grid of the four parity classes; on
// a 5 x 5 or 9 x 9 grid an 8 x 8 tile geometry keeps 81 of 256 tile slots busy (measured: 31 TFLOP/s at 8x8 -> 16x16, batch
// 64). The scatter form has no grid to tile:
//   P[t * Cout + co][b * H W + y W + x] = sum_ci  W_t[ci][co] * (s[b][ci] * x[b][ci][y][x])
// is a plain GEMM, M = 9 Cout = 4608, K = Cin = 512, N = B H W (1024 / 4096 at batch 64), every MFMA slot useful. Three
// launches replace the transposed-conv launch (the FIR pass after them is unchanged):
//   1. upgemm_pack_kernel: the zero-bordered activation x style -> the GEMM's B image [n tile of 128][chunk of 16][16 columns
//      tile][16 k][16 n] (an LDS-DMA piece = one 16 x 16 operand tile in MFMA read order, conflict-free),
//   2. upgemm_kernel: 128 x 128 block tiles, four waves of 64 x 64 (16 accumulator tiles of v_mfma_f32_16x16x4_f32), K chunks
//      of 16 through a two-slot LDS-DMA ring, three blocks per CU; MFMA rows = positions, columns = channels, so that a lane's
//      four accumulator registers are four consecutive positions of one channel: 16-byte stores,
//   3. upgemm_gather_kernel: T_class[y'][x'] = d[b][co] * sum of the class's taps P_t[y' + dy][x' + dx] (zero outside the
//      image) into the parity planes the FIR pass (aux_kernels.hip fir_epilogue_kernel) reads.
// P costs 9 Cout N floats of HBM traffic each way (75 MB at 8x8 -> 16x16, batch 64: about 40 us) -- the price of no tile waste.
// Larger layers keep the gather forms: their position grids tile well and P would be 4x larger per level.

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace gance {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kGM = 128, kGN = 128, kGK = 16;  // block tile and K chunk
constexpr int kTileFloats = kGK * 128;         // an operand tile of a chunk: [8 tiles of 16][16 k][16] = 8 KB = 8 DMA pieces

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
    p.packed[i] = v;
}

// ---- 2. GEMM ----
__global__ __launch_bounds__(256, 3) void upgemm_kernel(const UpGemmArgs p) {
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
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.packed + (size_t)n_tile * chunks * kTileFloats), 0, chunks * kTileFloats * 4, 0x00020000);
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
    const float d = p.d[(size_t)b * p.d_stride + co];
    const float cls[4] = {(at(0, 0, 0) + at(1, 0, -1)) + (at(2, -1, 0) + at(3, -1, -1)), at(4, 0, 0) + at(5, -1, 0), at(6, 0, 0) + at(7, 0, -1), at(8, 0, 0)};
    float* const dst = p.t + (size_t)b * p.unit_stride + (size_t)co * (p.H + 3) * (p.W + 8) + (size_t)(yq + 1) * (p.W + 8) + xq + 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[(size_t)c * p.cls_stride] = cls[c] * d;
}

}  // namespace

bool upgemm_supported(int cin, int cout, int H, int W) { return H == W && (H == 4 || H == 8) && cin % kGK == 0 && cout % kGM == 0; }
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

hipError_t launch_upgemm(const UpGemmArgs& args, hipStream_t stream) {
    if (!upgemm_supported(args.Cin, args.Cout, args.H, args.W) || args.n_tiles != upgemm_n_tiles(args.B, args.H, args.W)) return hipErrorInvalidValue;
    const size_t packed = (size_t)args.n_tiles * kGN * args.Cin;
    hipLaunchKernelGGL(upgemm_pack_kernel, dim3((unsigned)((packed + 255) / 256)), dim3(256), 0, stream, args);
    hipLaunchKernelGGL(upgemm_kernel, dim3((unsigned)(args.n_tiles * (9 * args.Cout / kGM))), dim3(256), 0, stream, args);
    const size_t outs = (size_t)args.B * args.Cout * (args.H + 1) * (args.W + 1);
    hipLaunchKernelGGL(upgemm_gather_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace gance
