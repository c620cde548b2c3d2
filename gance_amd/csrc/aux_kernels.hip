// The non-GEMM kernels of the synthesis path (all HBM- or latency-bound):
// mapping MLP, truncation, style affines, demodulation coefficients, the FIR half of Conv0_up
// with the fused noise/bias/leaky-ReLU epilogue, the split-K finish, and ToRGB + skip upsample +
// uint8 conversion. They restate, MI355X-first, what the reference gets from the un-vendored
// StyleGAN2 TF graph: upfirdn_2d.cu, fused_bias_act.cu, dense layers and
// tflib.convert_images_to_uint8.

#include <hip/hip_runtime.h>

#include <cmath>

#include <cstdlib>

#include "kernels.h"

namespace gance {

static constexpr float kSqrt2 = 1.4142135623730951f;

__device__ __forceinline__ float lrelu_gain(float v) { return (v < 0.f ? 0.2f * v : v) * kSqrt2; }

// ------------------------------------------------------------------------------------------
// Mapping network
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void mapping_dense_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out,
                                                            int normalize) {
    __shared__ float xs[512];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * 64;
    const float v0 = in[(size_t)b * 512 + tid];
    const float v1 = in[(size_t)b * 512 + 256 + tid];
    float scale = 1.f;
    if (normalize) {
        red[tid] = v0 * v0 + v1 * v1;
        __syncthreads();
        for (int stride = 128; stride > 0; stride >>= 1) {
            if (tid < stride) red[tid] += red[tid + stride];
            __syncthreads();
        }
        scale = 1.0f / sqrtf(red[0] * (1.f / 512.f) + 1e-8f);
        __syncthreads();
    }
    xs[tid] = v0 * scale;
    xs[tid + 256] = v1 * scale;
    __syncthreads();
    const int col = tid & 63;
    const int ks = tid >> 6;
    float acc = 0.f;
    const float* wp = w + (size_t)(ks * 128) * 512 + j0 + col;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) acc = fmaf(xs[ks * 128 + k], wp[(size_t)k * 512], acc);
    red[tid] = acc;
    __syncthreads();
    if (tid < 64) {
        const float r = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192] + bias[j0 + tid];
        out[(size_t)b * 512 + j0 + tid] = lrelu_gain(r);
    }
}

hipError_t launch_mapping_dense(const float* in, const float* w, const float* bias, float* out,
                                int B, int normalize, hipStream_t stream) {
    hipLaunchKernelGGL(mapping_dense_kernel, dim3(8, B), dim3(256), 0, stream, in, w, bias, out,
                       normalize);
    return hipGetLastError();
}

__global__ void broadcast_truncate_kernel(const float* __restrict__ w,
                                          const float* __restrict__ avg, float psi,
                                          float* __restrict__ dlat, int num_rows, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % 512);
    const size_t b = i / ((size_t)512 * num_rows);
    const float a = avg[k];
    dlat[i] = a + (w[b * 512 + k] - a) * psi;
}

hipError_t launch_broadcast_truncate(const float* w, const float* avg, float psi, float* dlat,
                                     int B, int num_rows, hipStream_t stream) {
    const size_t total = (size_t)B * num_rows * 512;
    hipLaunchKernelGGL(broadcast_truncate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                       0, stream, w, avg, psi, dlat, num_rows, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Styles and demodulation: skinny GEMMs  out[b][col] = post(sum_k pre(in[b][k]) * M[k][col])
// One block = 32 columns x 8 k-slices; up to 8 samples per pass kept in registers.
// ------------------------------------------------------------------------------------------

template <bool SQUARE_INPUT>
__device__ __forceinline__ void skinny_block(const float* __restrict__ in, size_t in_b_stride,
                                             int K, const float* __restrict__ M, int ld, int nb,
                                             float* lds_in /*[8][512]*/,
                                             float* lds_red /*[8][8][32]*/, float (&result)[1],
                                             int& result_b, bool& has_result) {
    const int tid = threadIdx.x;
    const int c = tid & 31;
    const int ks = tid >> 5;
    __syncthreads();
    for (int i = tid; i < nb * K; i += 256) {
        const int bb = i / K, k = i % K;
        float v = in[(size_t)bb * in_b_stride + k];
        lds_in[bb * 512 + k] = SQUARE_INPUT ? v * v : v;
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) acc[bb] = 0.f;
    const int kslice = K / 8;
    for (int k = ks * kslice; k < (ks + 1) * kslice; ++k) {
        const float m = M[(size_t)k * ld + c];
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) acc[bb] = fmaf(lds_in[bb * 512 + k], m, acc[bb]);
    }
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) lds_red[(ks * 8 + bb) * 32 + c] = acc[bb];
    __syncthreads();
    const int bb = tid >> 5;  // 8 samples x 32 columns = 256 threads
    has_result = bb < nb;
    result_b = bb;
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) sum += lds_red[(s * 8 + bb) * 32 + c];
    result[0] = sum;
}

__global__ __launch_bounds__(256) void styles_kernel(const float* __restrict__ dlat,
                                                     const float* __restrict__ A,
                                                     const float* __restrict__ bias1,
                                                     const int* __restrict__ blk_row,
                                                     float* __restrict__ s, int B, int num_rows,
                                                     int ctot) {
    __shared__ float lds_in[8 * 512];
    __shared__ float lds_red[8 * 8 * 32];
    const int cb = blockIdx.x;
    const int row = blk_row[cb];
    const int col = cb * 32 + (threadIdx.x & 31);
    {  // (a block per group of eight samples, like demod_kernel)
        const int b0 = blockIdx.y * 8;
        const int nb = min(8, B - b0);
        float r[1];
        int rb;
        bool ok;
        skinny_block<false>(dlat + ((size_t)b0 * num_rows + row) * 512, (size_t)num_rows * 512,
                            512, A + cb * 32, ctot, nb, lds_in, lds_red, r, rb, ok);
        if (ok) s[(size_t)(b0 + rb) * ctot + col] = r[0] + bias1[col];
    }
}

hipError_t launch_styles(const float* dlat, const float* A, const float* bias1, const int* blk_row,
                         float* s, int B, int num_rows, int ctot, hipStream_t stream) {
    hipLaunchKernelGGL(styles_kernel, dim3(ctot / 32, (B + 7) / 8), dim3(256), 0, stream, dlat, A, bias1,
                       blk_row, s, B, num_rows, ctot);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ s,
                                                    const float* __restrict__ w2_pool,
                                                    const DemodLayer* __restrict__ layers,
                                                    float* __restrict__ d, int B, int ctot,
                                                    int dtot) {
    __shared__ float lds_in[8 * 512];
    __shared__ float lds_red[8 * 8 * 32];
    const DemodLayer L = layers[blockIdx.y];
    const int co0 = blockIdx.x * 32;
    if (co0 >= L.cout) return;
    const int col = co0 + (threadIdx.x & 31);
    // (a block per group of eight samples: walked in one block, the eight passes of a 64-frame batch were 97 us of latency)
    {
        const int b0 = blockIdx.z * 8;
        const int nb = min(8, B - b0);
        float r[1];
        int rb;
        bool ok;
        skinny_block<true>(s + (size_t)b0 * ctot + L.s_off, (size_t)ctot, L.cin,
                           w2_pool + L.w2_off + co0, L.cout, nb, lds_in, lds_red, r, rb, ok);
        if (ok) d[(size_t)(b0 + rb) * dtot + L.d_off + col] = 1.0f / sqrtf(r[0] + 1e-8f);
    }
}

hipError_t launch_demod(const float* s, const float* w2_pool, const DemodLayer* layers,
                        int num_layers, float* d, int B, int ctot, int dtot, hipStream_t stream) {
    hipLaunchKernelGGL(demod_kernel, dim3(16, num_layers, (B + 7) / 8), dim3(256), 0, stream, s, w2_pool,
                       layers, d, B, ctot, dtot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// FIR half of Conv0_up + noise + bias + leaky ReLU
// ------------------------------------------------------------------------------------------
// T (the (2H+1)x(2W+1) transposed-conv result) lives as four parity planes, each zero-bordered
// [unit = split*B + b][C][H+3][W+8] with T[2y'+py][2x'+px] at [y'+1][x'+4]; cells a class does
// not own are never written and stay zero, so no load needs a bounds test.
//   out[oy][ox] = sum_{a,b} k[a] k[b] T[oy+a-1][ox+b-1],  k = [1,3,3,1]/4, T = 0 outside.
// One thread makes a 2 x 8 output strip (rows 2Y, 2Y+1; columns 2X .. 2X+7, X % 4 == 0) from
// 5 T rows x (5 even + 6 odd columns): 10 aligned float4 loads + 15 dword loads per 16 outputs.

struct __attribute__((packed, aligned(4))) float4u {
    float x, y, z, w;
};

// RQ = position rows per thread: a thread makes 2*RQ output rows x 8 columns from 2*RQ+3 T rows
// (the vertical FIR re-uses each horizontally filtered T row up to four times in registers).
template <int RQ>
__global__ __launch_bounds__(256) void fir_epilogue_kernel(const FirArgs p) {
    const int W4 = p.W >> 2;
    const int HQ = p.H / RQ;
    // XCD-aware block order: a thread re-reads 3 of its 2*RQ+3 T rows that the strip below it also
    // reads. Consecutive block ids land on different XCDs (separate L2s), so with the plain order
    // those re-reads went to HBM (measured 1.6x the algorithmic read bytes); giving every XCD a
    // contiguous range of strips keeps them in its L2.
    size_t vb;
    {
        const unsigned nwg = gridDim.x, bid = blockIdx.x;
        const unsigned per = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        vb = (size_t)(xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (bid >> 3);
    }
    const size_t q = vb * 256 + threadIdx.x;
    const size_t total = (size_t)p.B * p.C * HQ * W4;
    if (q >= total) return;
    const int X = (int)(q % W4) * 4;
    const int Y = (int)((q / W4) % HQ) * RQ;
    const size_t bc = q / ((size_t)W4 * HQ);
    const int c = (int)(bc % p.C);
    const int b = (int)(bc / p.C);
    const int TPW = p.W + 8;
    const size_t plane = (size_t)(p.H + 3) * TPW;
    constexpr int NR = 2 * RQ + 3;  // T rows O[Y-1], E[Y], O[Y], E[Y+1], ..., O[Y+RQ]

    // horizontal pass of one T row held as even columns e[0..4] = Te[X..X+4] and odd columns
    // o[0..5] = To[X-1..X+4]: h[2i] = out col 2(X+i), h[2i+1] = out col 2(X+i)+1
    float hrow[NR][8];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool odd_row = (r & 1) == 0;  // r = 0, 2, 4, ... are odd T rows
        // padded plane row: O[Y-1+k] -> Y+k, E[Y+k] -> Y+k+1
        const int yy = Y + (r >> 1) + (odd_row ? 0 : 1);
        const float* te = (odd_row ? p.t + 2 * p.cls_stride : p.t) + c * plane + (size_t)yy * TPW + X + 4;
        const float* to = te + p.cls_stride;  // class + 1 = odd columns of the same row parity
        float e[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // (four slabs' loads in flight at a time, added in slab order)
        int sp = 0;
        for (; sp + 4 <= p.nsplit; sp += 4) {
            float4 e4[4], o4[4];
            float e5[4], o0[4], o5[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t u = (size_t)((sp + j) * p.B + b) * p.unit_stride;
                e4[j] = *reinterpret_cast<const float4*>(te + u);
                o4[j] = *reinterpret_cast<const float4*>(to + u);
                e5[j] = te[u + 4];
                o0[j] = to[u - 1];
                o5[j] = to[u + 4];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[0] += e4[j].x; e[1] += e4[j].y; e[2] += e4[j].z; e[3] += e4[j].w;
                e[4] += e5[j];
                o[0] += o0[j];
                o[1] += o4[j].x; o[2] += o4[j].y; o[3] += o4[j].z; o[4] += o4[j].w;
                o[5] += o5[j];
            }
        }
        for (; sp < p.nsplit; ++sp) {
            const size_t u = (size_t)(sp * p.B + b) * p.unit_stride;
            const float4 e4 = *reinterpret_cast<const float4*>(te + u);
            const float4 o4 = *reinterpret_cast<const float4*>(to + u);
            e[0] += e4.x; e[1] += e4.y; e[2] += e4.z; e[3] += e4.w;
            e[4] += te[u + 4];
            o[0] += to[u - 1];
            o[1] += o4.x; o[2] += o4.y; o[3] += o4.z; o[4] += o4.w;
            o[5] += to[u + 4];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hrow[r][2 * i] = 0.25f * o[i] + 0.75f * e[i] + 0.75f * o[i + 1] + 0.25f * e[i + 1];
            hrow[r][2 * i + 1] = 0.25f * e[i] + 0.75f * o[i + 1] + 0.75f * e[i + 1] + 0.25f * o[i + 2];
        }
    }
    const int OW = 2 * p.W;
    const int OWp = OW + 8;  // zero-bordered activation [B][C][2H+2][2W+8], interior at [y+1][x+4]
    const float bs = p.bias[c];
    const float sn = p.s_next != nullptr ? p.s_next[(size_t)b * p.s_next_stride + c] : 1.0f;  // the next layer's style
    float* op = p.out + (bc * (size_t)(2 * p.H + 2) + (2 * Y + 1)) * OWp + 2 * X + 4;
#pragma unroll
    for (int k = 0; k < 2 * RQ; ++k) {  // output row 2Y + k uses T rows hrow[k .. k+3]
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            r[i] = 0.25f * hrow[k][i] + 0.75f * hrow[k + 1][i] + 0.75f * hrow[k + 2][i] + 0.25f * hrow[k + 3][i];
        if (p.noise != nullptr) {
            const float* nz = p.noise + (size_t)b * p.noise_b_stride + (size_t)(2 * Y + k) * OW + 2 * X;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] += nz[i] * p.noise_strength;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = lrelu_gain(r[i] + bs) * sn;
        *reinterpret_cast<float4*>(op + (size_t)k * OWp) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(op + (size_t)k * OWp + 4) = make_float4(r[4], r[5], r[6], r[7]);
    }
}

hipError_t launch_fir_epilogue(const FirArgs& args, hipStream_t stream) {
    // 2 position rows per thread (measured best: +7 % over 1, same as 4) when the grid still fills the chip
    static const int forced = [] { const char* v = std::getenv("GANCE_TUNE_FIR_RQ"); return v ? std::atoi(v) : 0; }();
    int rq = (args.H % 2 == 0 && (size_t)args.B * args.C * (args.H / 2) * (args.W / 4) >= 256 * 256 * 4) ? 2 : 1;
    if (forced == 1 || forced == 2 || forced == 4) rq = (args.H % forced == 0) ? forced : 1;
    const size_t total = (size_t)args.B * args.C * (args.H / rq) * (args.W / 4);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (rq == 4) hipLaunchKernelGGL(fir_epilogue_kernel<4>, grid, dim3(256), 0, stream, args);
    else if (rq == 2) hipLaunchKernelGGL(fir_epilogue_kernel<2>, grid, dim3(256), 0, stream, args);
    else hipLaunchKernelGGL(fir_epilogue_kernel<1>, grid, dim3(256), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Split-K finish (small layers only): slabs [nsplit][B][C][H][W] -> zero-bordered activation [..][H+2][W+8]
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs,
                                                            long long slab_stride, int nsplit,
                                                            const float* __restrict__ noise,
                                                            float noise_strength, int noise_b_stride,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int C, int H,
                                                            int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // (eight slabs' loads in flight at a time, added in slab order: one load per add made a one-frame call wait
    // nsplit = up to 128 memory round trips here)
    float v = slabs[i];
    int sp = 1;
    for (; sp + 8 <= nsplit; sp += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = slabs[(size_t)(sp + j) * slab_stride + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    for (; sp < nsplit; ++sp) v += slabs[(size_t)sp * slab_stride + i];
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const size_t bc = i / ((size_t)W * H);
    const int c = (int)(bc % C);
    if (noise != nullptr) v += noise[(bc / C) * (size_t)noise_b_stride + (size_t)y * W + x] * noise_strength;
    out[(bc * (H + 2) + y + 1) * (size_t)(W + 8) + x + 4] = lrelu_gain(v + bias[c]);
}

hipError_t launch_splitk_finish(const float* slabs, long long slab_stride, int nsplit,
                                const float* noise, float noise_strength, int noise_b_stride, const float* bias,
                                float* out, int B, int C, int H, int W, hipStream_t stream) {
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, slabs, slab_stride, nsplit, noise, noise_strength, noise_b_stride, bias, out, C, H,
                       W, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// ToRGB + skip upsample (+ uint8 NHWC)
// ------------------------------------------------------------------------------------------
// y[b,c,p] = sum_ci x[b,ci,p] * (s[b,ci] * w[ci,c]) + bias[c] + upsample_2d(y_prev)[b,c,p]
// x is the zero-bordered activation [B][Cin][R+2][R+8]; y is dense [B][3][R][R].
// upsample_2d = zero-insert x2, pad (2,1), FIR [1,3,3,1]x[1,3,3,1]/16: per axis
//   even o=2Y:  1/4 y[Y-1] + 3/4 y[Y] ;  odd o=2Y+1:  3/4 y[Y] + 1/4 y[Y+1]   (zero outside).
// uint8: tf.saturate_cast(x * 127.5 + 128): separate multiply and add, clamp, truncate.

__global__ __launch_bounds__(256) void torgb_kernel(const ToRgbArgs p) {
    __shared__ float coef[512 * 3];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    if (p.partial == nullptr) {
        for (int i = tid; i < p.Cin * 3; i += 256)
            coef[i] = p.s[(size_t)b * p.s_stride + i / 3] * p.w[i];
    }
    __syncthreads();
    const int R = p.R;
    const size_t npix = (size_t)R * R;
    const size_t p4 = ((size_t)blockIdx.x * 256 + tid) * 4;
    if (p4 >= npix) return;
    const int oy = (int)(p4 / R);
    const int ox0 = (int)(p4 % R);
    const size_t xplane = (size_t)(R + 2) * (R + 8);

    float acc[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
    if (p.partial != nullptr) {  // the conv kernel's epilogue did the channel sum (kEpilogueFullRgbPart)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // (fixed order: reproducible; eight loads in flight: rolled, one load per trip waited for its own latency -- 32
            // partial images at 32^2 made this pass 50 us of latency)
#pragma unroll 8
            for (int m = 0; m < p.partials; ++m) {
                const float4 v = *reinterpret_cast<const float4*>(p.partial + (((size_t)m * p.B + b) * 3 + c) * npix + p4);
                acc[c][0] += v.x; acc[c][1] += v.y; acc[c][2] += v.z; acc[c][3] += v.w;
            }
        }
    }
    const float* xp = p.x + (size_t)b * p.Cin * xplane + (size_t)(oy + 1) * (R + 8) + ox0 + 4;
#pragma unroll 4
    for (int ci = 0; ci < (p.partial != nullptr ? 0 : p.Cin); ++ci) {
        const float4 v = *reinterpret_cast<const float4*>(xp + (size_t)ci * xplane);
        const float c0 = coef[ci * 3 + 0], c1 = coef[ci * 3 + 1], c2 = coef[ci * 3 + 2];
        acc[0][0] = fmaf(v.x, c0, acc[0][0]); acc[0][1] = fmaf(v.y, c0, acc[0][1]);
        acc[0][2] = fmaf(v.z, c0, acc[0][2]); acc[0][3] = fmaf(v.w, c0, acc[0][3]);
        acc[1][0] = fmaf(v.x, c1, acc[1][0]); acc[1][1] = fmaf(v.y, c1, acc[1][1]);
        acc[1][2] = fmaf(v.z, c1, acc[1][2]); acc[1][3] = fmaf(v.w, c1, acc[1][3]);
        acc[2][0] = fmaf(v.x, c2, acc[2][0]); acc[2][1] = fmaf(v.y, c2, acc[2][1]);
        acc[2][2] = fmaf(v.z, c2, acc[2][2]); acc[2][3] = fmaf(v.w, c2, acc[2][3]);
    }

#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float bs = p.bias[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[c][k] += bs;
    }
    if (p.y_prev != nullptr) {
        const int Rh = R >> 1;
        const int Y = oy >> 1;
        // two source rows and their weights
        const int ya = (oy & 1) ? Y : Y - 1;
        const int yb = ya + 1;
        const float wya = (oy & 1) ? 0.75f : 0.25f;
        const float wyb = 1.0f - wya;
        // the thread's four pixels ox0 .. ox0 + 3 (ox0 a multiple of 4) read source columns X0 - 1 .. X0 + 2, X0 = ox0 / 2:
        // pixel 0: (.25, .75) of columns (X0-1, X0); 1: (.75, .25) of (X0, X0+1); 2: (.25, .75) of (X0, X0+1);
        // 3: (.75, .25) of (X0+1, X0+2) -- four loads per row and colour instead of eight
        const int X0 = ox0 >> 1;
        const bool ra = ya >= 0, rb = yb < Rh;  // (ya < Rh and yb >= 0 always)
        const bool c0 = X0 >= 1, c3 = X0 + 2 < Rh;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* yp = p.y_prev + ((size_t)b * 3 + c) * Rh * Rh;
            float ta[4], tb[4];  // source rows ya, yb, columns X0 - 1 .. X0 + 2 (zero outside the image)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool cok = j == 0 ? c0 : (j == 3 ? c3 : true);
                ta[j] = (ra && cok) ? yp[(size_t)ya * Rh + X0 - 1 + j] : 0.f;
                tb[j] = (rb && cok) ? yp[(size_t)yb * Rh + X0 - 1 + j] : 0.f;
            }
            // (the same order of operations as one pixel at a time: along the row first, then the two rows)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = (k + 1) >> 1;  // first source column of pixel k: 0, 1, 1, 2
                const float wxa = (k & 1) ? 0.75f : 0.25f, wxb = 1.0f - wxa;
                const float top = wxa * ta[j] + wxb * ta[j + 1];
                const float bot = wxa * tb[j] + wxb * tb[j + 1];
                acc[c][k] += wya * top + wyb * bot;
            }
        }
    }
    if (!p.skip_y_store) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float4*>(p.y + ((size_t)b * 3 + c) * npix + p4) =
                make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    }

    if (p.u8 != nullptr) {
        uint8_t q[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = __fadd_rn(__fmul_rn(acc[c][k], 127.5f), 128.0f);
                v = fminf(fmaxf(v, 0.f), 255.f);
                q[k * 3 + c] = (uint8_t)(int)v;
            }
        uint32_t* dst = reinterpret_cast<uint32_t*>(p.u8 + ((size_t)b * npix + p4) * 3);
        dst[0] = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
        dst[1] = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
        dst[2] = q[8] | (q[9] << 8) | (q[10] << 16) | ((uint32_t)q[11] << 24);
    }
}

// Small resolutions (R <= 128): the per-pixel loop over Cin = 256..512 channels is a long serial
// chain and there are too few pixels to fill the chip, so the channel sum is split 8 ways across
// the threads of a block (32 pixels x 8 channel slices) and reduced through LDS.
__global__ __launch_bounds__(256) void torgb_small_kernel(const ToRgbArgs p) {
    __shared__ float coef[512 * 3];
    __shared__ float red[8][3][32];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < p.Cin * 3; i += 256)
        coef[i] = p.s[(size_t)b * p.s_stride + i / 3] * p.w[i];
    __syncthreads();
    const int R = p.R;
    const int npix = R * R;
    const int pix = blockIdx.x * 32 + (tid & 31);
    const int slice = tid >> 5;
    const bool live = pix < npix;
    const int oy = live ? pix / R : 0, ox = live ? pix % R : 0;
    const size_t xplane = (size_t)(R + 2) * (R + 8);
    const float* xp = p.x + (size_t)b * p.Cin * xplane + (size_t)(oy + 1) * (R + 8) + ox + 4;
    const int per_slice = p.Cin / 8;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (live) {
#pragma unroll 4
        for (int k = 0; k < per_slice; ++k) {
            const int ci = slice * per_slice + k;
            const float v = xp[(size_t)ci * xplane];
            a0 = fmaf(v, coef[ci * 3 + 0], a0);
            a1 = fmaf(v, coef[ci * 3 + 1], a1);
            a2 = fmaf(v, coef[ci * 3 + 2], a2);
        }
    }
    red[slice][0][tid & 31] = a0;
    red[slice][1][tid & 31] = a1;
    red[slice][2][tid & 31] = a2;
    __syncthreads();
    if (slice != 0 || !live) return;
    float acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) sum += red[sl][c][tid];
        acc[c] = sum + p.bias[c];
    }
    if (p.y_prev != nullptr) {
        const int Rh = R >> 1;
        const int ya = (oy & 1) ? (oy >> 1) : (oy >> 1) - 1, yb = ya + 1;
        const int xa = (ox & 1) ? (ox >> 1) : (ox >> 1) - 1, xb = xa + 1;
        const float wya = (oy & 1) ? 0.75f : 0.25f, wyb = 1.0f - wya;
        const float wxa = (ox & 1) ? 0.75f : 0.25f, wxb = 1.0f - wxa;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* yp = p.y_prev + ((size_t)b * 3 + c) * Rh * Rh;
            auto at = [&](int yy, int xx) -> float {
                return (yy >= 0 && yy < Rh && xx >= 0 && xx < Rh) ? yp[(size_t)yy * Rh + xx] : 0.f;
            };
            const float top = wxa * at(ya, xa) + wxb * at(ya, xb);
            const float bot = wxa * at(yb, xa) + wxb * at(yb, xb);
            acc[c] += wya * top + wyb * bot;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p.y[((size_t)b * 3 + c) * npix + pix] = acc[c];
    if (p.u8 != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = __fadd_rn(__fmul_rn(acc[c], 127.5f), 128.0f);
            v = fminf(fmaxf(v, 0.f), 255.f);
            p.u8[((size_t)b * npix + pix) * 3 + c] = (uint8_t)(int)v;
        }
    }
}

hipError_t launch_torgb(const ToRgbArgs& args, hipStream_t stream) {
    const size_t npix = (size_t)args.R * args.R;
    if (args.partial != nullptr && (args.partials < 1 || args.R % 4 != 0)) return hipErrorInvalidValue;
    if (args.partial == nullptr && args.R <= 128 && args.Cin % 8 == 0) {  // (the finishing pass has no channel loop to split)
        hipLaunchKernelGGL(torgb_small_kernel, dim3((unsigned)((npix + 31) / 32), args.B), dim3(256), 0,
                           stream, args);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(torgb_kernel, dim3((unsigned)((npix / 4 + 255) / 256), args.B), dim3(256),
                       0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Post-synthesis resize: bicubic, a = -0.75, uint8 RGB NHWC
// ------------------------------------------------------------------------------------------
// Replaces cv2.resize(image, (side, side), interpolation=cv2.INTER_CUBIC) of
// gance/image_sources/video_common.py:399-429. OpenCV's uint8 path uses 11-bit fixed-point
// coefficients; this implementation DEFINES the float form of the same filter: source coordinate
// (d + 0.5) * src/dst - 0.5, four taps per axis with the Keys kernel a = -0.75, replicated
// border, float32 accumulation, round half up, saturate. Parity with OpenCV is unpinned (cv2 is
// not available); the oracle is oracle/resize_ref.py.

__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {
    const float a = -0.75f;
    w[0] = ((a * (t + 1.f) - 5.f * a) * (t + 1.f) + 8.f * a) * (t + 1.f) - 4.f * a;
    w[1] = ((a + 2.f) * t - (a + 3.f)) * t * t + 1.f;
    w[2] = ((a + 2.f) * (1.f - t) - (a + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

// One block = a 64 x 4 tile of output pixels of one frame. The source pixels the tile touches are
// copied into LDS once (consecutive threads read consecutive bytes), every thread then takes its
// 16 taps from LDS, and the 768 result bytes leave as 192 coalesced dword stores. Arithmetic and
// its order are those of the per-pixel formulation (row sums over x, then the sum over rows).
constexpr int kResizeTW = 64, kResizeTH = 4;

__global__ __launch_bounds__(256) void resize_bicubic_u8_kernel(const uint8_t* __restrict__ in, int src,
                                                                uint8_t* __restrict__ out, int dst, int tiles_x,
                                                                int tiles_y, int max_cols, int max_rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t resize_lds[];
    uint8_t* const tile = resize_lds;                                 // [rows][cols * 3]
    uint8_t* const result = resize_lds + ((size_t)max_rows * max_cols * 3 + 15) / 16 * 16;  // [kResizeTH][kResizeTW * 3]
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x;
    const int ty = (blockIdx.x / tiles_x) % tiles_y;
    const size_t b = blockIdx.x / ((size_t)tiles_x * tiles_y);
    const int ox0 = tx * kResizeTW, oy0 = ty * kResizeTH;
    const float scale = (float)src / (float)dst;
    // source window of the tile (before clamping): taps iy-1 .. iy+2 of its first and last pixel
    auto first_tap = [&](int o) { return (int)floorf(((float)o + 0.5f) * scale - 0.5f) - 1; };
    const int x_lo = first_tap(ox0), x_hi = first_tap(min(ox0 + kResizeTW, dst) - 1) + 3;
    const int y_lo = first_tap(oy0), y_hi = first_tap(min(oy0 + kResizeTH, dst) - 1) + 3;
    const int cols = x_hi - x_lo + 1, rows = y_hi - y_lo + 1;  // <= max_cols, max_rows
    const uint8_t* img = in + b * (size_t)src * src * 3;
    const int row_bytes = cols * 3;
    for (int i = tid; i < rows * row_bytes; i += 256) {
        const int r = i / row_bytes, cb = i % row_bytes;
        const int yy = min(max(y_lo + r, 0), src - 1);            // replicated border
        const int xx = min(max(x_lo + cb / 3, 0), src - 1);
        tile[i] = img[((size_t)yy * src + xx) * 3 + cb % 3];
    }
    __syncthreads();
    const int lx = tid % kResizeTW, ly = tid / kResizeTW;
    const int ox = ox0 + lx, oy = oy0 + ly;
    if (ox < dst && oy < dst) {
        const float fy = ((float)oy + 0.5f) * scale - 0.5f;
        const float fx = ((float)ox + 0.5f) * scale - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_weights(fy - (float)iy, wy);
        cubic_weights(fx - (float)ix, wx);
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint8_t* px = tile + (size_t)(iy - 1 + r - y_lo) * row_bytes + (ix - 1 - x_lo) * 3;
            float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                row[0] += wx[c] * (float)px[3 * c + 0];
                row[1] += wx[c] * (float)px[3 * c + 1];
                row[2] += wx[c] * (float)px[3 * c + 2];
            }
            acc[0] += wy[r] * row[0];
            acc[1] += wy[r] * row[1];
            acc[2] += wy[r] * row[2];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            result[(ly * kResizeTW + lx) * 3 + ch] = (uint8_t)(int)fminf(fmaxf(floorf(acc[ch] + 0.5f), 0.f), 255.f);
    }
    __syncthreads();
    // a tile row is kResizeTW * 3 = 192 bytes = 48 dwords, 4-byte aligned in the frame when dst % 4 == 0
    const int valid_w = min(kResizeTW, dst - ox0);
    if ((dst & 3) == 0 && valid_w == kResizeTW) {
        if (tid < kResizeTH * 48) {
            const int r = tid / 48, d = tid % 48;
            if (oy0 + r < dst)
                reinterpret_cast<uint32_t*>(out + ((b * dst + oy0 + r) * (size_t)dst + ox0) * 3)[d] =
                    reinterpret_cast<const uint32_t*>(result + r * kResizeTW * 3)[d];
        }
    } else {
        for (int i = tid; i < kResizeTH * valid_w * 3; i += 256) {
            const int r = i / (valid_w * 3), cb = i % (valid_w * 3);
            if (oy0 + r < dst) out[((b * dst + oy0 + r) * (size_t)dst + ox0) * 3 + cb] = result[r * kResizeTW * 3 + cb];
        }
    }
}

// Fresh standard-normal noise (the upstream generator's randomize_noise=True draws tf.random_normal per call): counter-based,
// value i of a buffer = Box-Muller of two uniforms hashed from (seed, stream id, i / 2) by splitmix64. Not TF's stream
// (nothing could reproduce that), only its distribution.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// One plane of `plane` floats per sample: sample b of the launch has the id sample_ids[b] (or first_sample + b), and its
// draws are a function of (seed, layer, id, index) only: a frame's noise does not depend on how the frames were batched.
__global__ void normal_noise_kernel(float* __restrict__ out, size_t plane, unsigned long long seed, unsigned long long layer,
                                    unsigned long long first_sample, const long long* __restrict__ sample_ids) {
    const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * pair >= plane) return;
    const unsigned long long id = sample_ids != nullptr ? (unsigned long long)sample_ids[blockIdx.y] : first_sample + blockIdx.y;
    const unsigned long long key = splitmix64(splitmix64(seed ^ (layer * 0xD1B54A32D192ED03ull)) ^ (id * 0x9FB21C651E98DF25ull));
    const unsigned long long bits = splitmix64(key + pair);
    const float u1 = ((float)(unsigned)(bits >> 40) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(unsigned)((bits >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);  // [0, 1)
    const float radius = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    float* const dst = out + (size_t)blockIdx.y * plane;
    dst[2 * pair] = radius * cs;
    if (2 * pair + 1 < plane) dst[2 * pair + 1] = radius * sn;
}

hipError_t launch_normal_noise(float* out, size_t plane, int samples, unsigned long long seed, unsigned long long layer,
                               unsigned long long first_sample, const long long* sample_ids, hipStream_t stream) {
    if (plane == 0 || samples <= 0) return hipSuccess;
    const size_t pairs = (plane + 1) / 2;
    hipLaunchKernelGGL(normal_noise_kernel, dim3((unsigned)((pairs + 255) / 256), (unsigned)samples), dim3(256), 0, stream, out, plane, seed, layer,
                       first_sample, sample_ids);
    return hipGetLastError();
}

hipError_t launch_resize_bicubic_u8(const uint8_t* in, int batch, int src, uint8_t* out, int dst,
                                    hipStream_t stream) {
    const int tiles_x = (dst + kResizeTW - 1) / kResizeTW, tiles_y = (dst + kResizeTH - 1) / kResizeTH;
    const double scale = (double)src / dst;
    const int max_cols = (int)std::ceil(kResizeTW * scale) + 6, max_rows = (int)std::ceil(kResizeTH * scale) + 6;
    const size_t lds = ((size_t)max_rows * max_cols * 3 + 15) / 16 * 16 + (size_t)kResizeTH * kResizeTW * 3;
    if (lds > 64 * 1024) return hipErrorInvalidValue;  // down-scaling by more than ~10x: not a use of this path
    hipLaunchKernelGGL(resize_bicubic_u8_kernel, dim3((unsigned)((size_t)batch * tiles_x * tiles_y)), dim3(256), lds,
                       stream, in, src, out, dst, tiles_x, tiles_y, max_cols, max_rows);
    return hipGetLastError();
}

}  // namespace gance
