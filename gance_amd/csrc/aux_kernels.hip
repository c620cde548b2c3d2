// The non-GEMM kernels of the synthesis path (all HBM- or latency-bound):
// mapping MLP, truncation, style affines, demodulation coefficients, the FIR half of Conv0_up
// with the fused noise/bias/leaky-ReLU epilogue, the split-K finish, and ToRGB + skip upsample +
// uint8 conversion. They restate, MI355X-first, what the reference gets from the un-vendored
// StyleGAN2 TF graph: upfirdn_2d.cu, fused_bias_act.cu, dense layers and
// tflib.convert_images_to_uint8.

#include <hip/hip_runtime.h>

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
    for (int b0 = 0; b0 < B; b0 += 8) {
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
    hipLaunchKernelGGL(styles_kernel, dim3(ctot / 32), dim3(256), 0, stream, dlat, A, bias1,
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
    for (int b0 = 0; b0 < B; b0 += 8) {
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
    hipLaunchKernelGGL(demod_kernel, dim3(16, num_layers), dim3(256), 0, stream, s, w2_pool,
                       layers, d, B, ctot, dtot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// FIR half of Conv0_up + noise + bias + leaky ReLU
// ------------------------------------------------------------------------------------------
// T (the (2H+1)x(2W+1) transposed-conv result) lives as four parity planes. One thread makes the
// 2x2 output quad (2Y..2Y+1, 2X..2X+1):
//   out[oy][ox] = sum_{a,b} k[a] k[b] T[oy+a-1][ox+b-1],  k = [1,3,3,1]/4, T = 0 outside.

__global__ __launch_bounds__(256) void fir_epilogue_kernel(const FirArgs p) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)p.B * p.C * p.H * p.W;
    if (q >= total) return;
    const int X = (int)(q % p.W);
    const int Y = (int)((q / p.W) % p.H);
    const size_t bc = q / ((size_t)p.W * p.H);
    const int c = (int)(bc % p.C);
    const int H = p.H, W = p.W;

    // summed-over-slabs loads with zero outside each plane
    auto ld = [&](const float* base, long long slab, int rows, int cols, int yy, int xx) -> float {
        if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) return 0.f;
        const float* ptr = base + (bc * rows + yy) * (size_t)cols + xx;
        float v = 0.f;
        for (int sp = 0; sp < p.nsplit; ++sp) v += ptr[(size_t)sp * slab];
        return v;
    };
    // horizontal pass of one T row: te = even columns X, X+1 ; to = odd columns X-1, X, X+1
    auto hpass = [&](float te0, float te1, float to_m1, float to0, float to1, float& h0,
                     float& h1) {
        h0 = 0.25f * to_m1 + 0.75f * te0 + 0.75f * to0 + 0.25f * te1;
        h1 = 0.25f * te0 + 0.75f * to0 + 0.75f * te1 + 0.25f * to1;
    };
    float e0h0, e0h1, e1h0, e1h1;          // even T rows Y, Y+1
    float om1h0, om1h1, o0h0, o0h1, o1h0, o1h1;  // odd T rows Y-1, Y, Y+1
    {
        const float a0 = ld(p.t_ee, p.slab_ee, H + 1, W + 1, Y, X);
        const float a1 = ld(p.t_ee, p.slab_ee, H + 1, W + 1, Y, X + 1);
        const float bm = ld(p.t_eo, p.slab_eo, H + 1, W, Y, X - 1);
        const float b0 = ld(p.t_eo, p.slab_eo, H + 1, W, Y, X);
        const float b1 = ld(p.t_eo, p.slab_eo, H + 1, W, Y, X + 1);
        hpass(a0, a1, bm, b0, b1, e0h0, e0h1);
    }
    {
        const float a0 = ld(p.t_ee, p.slab_ee, H + 1, W + 1, Y + 1, X);
        const float a1 = ld(p.t_ee, p.slab_ee, H + 1, W + 1, Y + 1, X + 1);
        const float bm = ld(p.t_eo, p.slab_eo, H + 1, W, Y + 1, X - 1);
        const float b0 = ld(p.t_eo, p.slab_eo, H + 1, W, Y + 1, X);
        const float b1 = ld(p.t_eo, p.slab_eo, H + 1, W, Y + 1, X + 1);
        hpass(a0, a1, bm, b0, b1, e1h0, e1h1);
    }
#define GANCE_ODD_ROW(yy, h0, h1)                                            \
    {                                                                        \
        const float a0 = ld(p.t_oe, p.slab_oe, H, W + 1, (yy), X);           \
        const float a1 = ld(p.t_oe, p.slab_oe, H, W + 1, (yy), X + 1);       \
        const float bm = ld(p.t_oo, p.slab_oo, H, W, (yy), X - 1);           \
        const float b0 = ld(p.t_oo, p.slab_oo, H, W, (yy), X);               \
        const float b1 = ld(p.t_oo, p.slab_oo, H, W, (yy), X + 1);           \
        hpass(a0, a1, bm, b0, b1, h0, h1);                                   \
    }
    GANCE_ODD_ROW(Y - 1, om1h0, om1h1)
    GANCE_ODD_ROW(Y, o0h0, o0h1)
    GANCE_ODD_ROW(Y + 1, o1h0, o1h1)
#undef GANCE_ODD_ROW

    float r00 = 0.25f * om1h0 + 0.75f * e0h0 + 0.75f * o0h0 + 0.25f * e1h0;  // (2Y,   2X)
    float r01 = 0.25f * om1h1 + 0.75f * e0h1 + 0.75f * o0h1 + 0.25f * e1h1;  // (2Y,   2X+1)
    float r10 = 0.25f * e0h0 + 0.75f * o0h0 + 0.75f * e1h0 + 0.25f * o1h0;   // (2Y+1, 2X)
    float r11 = 0.25f * e0h1 + 0.75f * o0h1 + 0.75f * e1h1 + 0.25f * o1h1;   // (2Y+1, 2X+1)

    const int OW = 2 * W;
    const size_t o0 = (size_t)(2 * Y) * OW + 2 * X;
    if (p.noise != nullptr) {
        const float ns = p.noise_strength;
        r00 += p.noise[o0] * ns;
        r01 += p.noise[o0 + 1] * ns;
        r10 += p.noise[o0 + OW] * ns;
        r11 += p.noise[o0 + OW + 1] * ns;
    }
    const float bs = p.bias[c];
    float* op = p.out + bc * (size_t)(4 * H * W) + o0;
    *reinterpret_cast<float2*>(op) = make_float2(lrelu_gain(r00 + bs), lrelu_gain(r01 + bs));
    *reinterpret_cast<float2*>(op + OW) = make_float2(lrelu_gain(r10 + bs), lrelu_gain(r11 + bs));
}

hipError_t launch_fir_epilogue(const FirArgs& args, hipStream_t stream) {
    const size_t total = (size_t)args.B * args.C * args.H * args.W;
    hipLaunchKernelGGL(fir_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Split-K finish
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs,
                                                            long long slab_stride, int nsplit,
                                                            const float* __restrict__ noise,
                                                            float noise_strength,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int C,
                                                            int HW, size_t total4) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const size_t i = i4 * 4;
    float4 v = *reinterpret_cast<const float4*>(slabs + i);
    for (int sp = 1; sp < nsplit; ++sp) {
        const float4 u = *reinterpret_cast<const float4*>(slabs + (size_t)sp * slab_stride + i);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    if (noise != nullptr) {
        const float4 nz = *reinterpret_cast<const float4*>(noise + pix);
        v.x += nz.x * noise_strength; v.y += nz.y * noise_strength;
        v.z += nz.z * noise_strength; v.w += nz.w * noise_strength;
    }
    const float bs = bias[c];
    *reinterpret_cast<float4*>(out + i) = make_float4(lrelu_gain(v.x + bs), lrelu_gain(v.y + bs),
                                                      lrelu_gain(v.z + bs), lrelu_gain(v.w + bs));
}

hipError_t launch_splitk_finish(const float* slabs, long long slab_stride, int nsplit,
                                const float* noise, float noise_strength, const float* bias,
                                float* out, int B, int C, int H, int W, hipStream_t stream) {
    const size_t total4 = (size_t)B * C * H * W / 4;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       stream, slabs, slab_stride, nsplit, noise, noise_strength, bias, out, C,
                       H * W, total4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// ToRGB + skip upsample (+ uint8 NHWC)
// ------------------------------------------------------------------------------------------
// y[b,c,p] = sum_ci x[b,ci,p] * (s[b,ci] * w[ci,c]) + bias[c] + upsample_2d(y_prev)[b,c,p]
// upsample_2d = zero-insert x2, pad (2,1), FIR [1,3,3,1]x[1,3,3,1]/16: per axis
//   even o=2Y:  1/4 y[Y-1] + 3/4 y[Y] ;  odd o=2Y+1:  3/4 y[Y] + 1/4 y[Y+1]   (zero outside).
// uint8: tf.saturate_cast(x * 127.5 + 128): separate multiply and add, clamp, truncate.

__global__ __launch_bounds__(256) void torgb_kernel(const ToRgbArgs p) {
    __shared__ float coef[512 * 3];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < p.Cin * 3; i += 256)
        coef[i] = p.s[(size_t)b * p.s_stride + i / 3] * p.w[i];
    __syncthreads();
    const int R = p.R;
    const size_t npix = (size_t)R * R;
    const size_t p4 = ((size_t)blockIdx.x * 256 + tid) * 4;
    if (p4 >= npix) return;

    float acc[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
    const float* xp = p.x + (size_t)b * p.Cin * npix + p4;
#pragma unroll 4
    for (int ci = 0; ci < p.Cin; ++ci) {
        const float4 v = *reinterpret_cast<const float4*>(xp + (size_t)ci * npix);
        const float c0 = coef[ci * 3 + 0], c1 = coef[ci * 3 + 1], c2 = coef[ci * 3 + 2];
        acc[0][0] = fmaf(v.x, c0, acc[0][0]); acc[0][1] = fmaf(v.y, c0, acc[0][1]);
        acc[0][2] = fmaf(v.z, c0, acc[0][2]); acc[0][3] = fmaf(v.w, c0, acc[0][3]);
        acc[1][0] = fmaf(v.x, c1, acc[1][0]); acc[1][1] = fmaf(v.y, c1, acc[1][1]);
        acc[1][2] = fmaf(v.z, c1, acc[1][2]); acc[1][3] = fmaf(v.w, c1, acc[1][3]);
        acc[2][0] = fmaf(v.x, c2, acc[2][0]); acc[2][1] = fmaf(v.y, c2, acc[2][1]);
        acc[2][2] = fmaf(v.z, c2, acc[2][2]); acc[2][3] = fmaf(v.w, c2, acc[2][3]);
    }

    const int oy = (int)(p4 / R);
    const int ox0 = (int)(p4 % R);
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
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* yp = p.y_prev + ((size_t)b * 3 + c) * Rh * Rh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ox = ox0 + k;
                const int X = ox >> 1;
                const int xa = (ox & 1) ? X : X - 1;
                const int xb = xa + 1;
                const float wxa = (ox & 1) ? 0.75f : 0.25f;
                const float wxb = 1.0f - wxa;
                auto at = [&](int yy, int xx) -> float {
                    return (yy >= 0 && yy < Rh && xx >= 0 && xx < Rh) ? yp[(size_t)yy * Rh + xx]
                                                                      : 0.f;
                };
                const float top = wxa * at(ya, xa) + wxb * at(ya, xb);
                const float bot = wxa * at(yb, xa) + wxb * at(yb, xb);
                acc[c][k] += wya * top + wyb * bot;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        *reinterpret_cast<float4*>(p.y + ((size_t)b * 3 + c) * npix + p4) =
            make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);

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

hipError_t launch_torgb(const ToRgbArgs& args, hipStream_t stream) {
    const size_t npix = (size_t)args.R * args.R;
    hipLaunchKernelGGL(torgb_kernel, dim3((unsigned)((npix / 4 + 255) / 256), args.B), dim3(256),
                       0, stream, args);
    return hipGetLastError();
}

}  // namespace gance
