// Microbenchmark (not product code): LDS-array cycles per wave instruction for the access patterns of the two kernels whose
// SQ_LDS_BANK_CONFLICT the access-pattern arithmetic does not explain: upfir_split.hip's ring (16-byte reads of a
// fragment, 16-byte writes of the staging, swizzled and not) and winograd43_conv.hip's window / weight reads. One block of four waves per
// CU, every wave the same pattern, eight instructions per wait (inline assembly); cycles per instruction and CU from the shader clock.
//   hipcc -O3 --offload-arch=gfx950 lds_pattern_cycles.hip -o lds_pattern_cycles.co && ./lds_pattern_cycles.co
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int ring_column(int P) {
    return ((P >> 2) & 3) | ((((P >> 4) ^ P) & 1) << 2) | (((P >> 1) & 1) << 3) | ((P & 1) << 4) | (P & 32);
}

// byte address of a lane for pattern `pat` (wave w of the block)
__device__ int pattern_address(int pat, int lane, int w) {
    const int n16 = lane & 15, kg = lane >> 4;
    switch (pat) {
        case 0: return lane * 16;                                              // 16 bytes per lane, consecutive
        case 1: return (kg * 80 + ring_column(16 * w + n16)) * 16;              // ring fragment read, swizzled columns
        case 2: return (kg * 80 + 16 * w + n16 + 1) * 16;                       // ring fragment read, plain columns (+ 1: the edge column in front)
        case 3: return (w * 80 + ring_column(4 * n16 + kg)) * 16;               // staging write after the transpose, swizzled: position 4 g + r
        case 4: return (w * 80 + 4 * n16 + kg + 1) * 16;                        // ... plain columns: lanes 4 units apart
        case 5: return (w * 80 + lane + 1) * 16;                                // staging write of the dword-load version: column = lane
        case 6: return (kg * 1296 + 4 * n16 + 4) * 4;                           // F(4x4,3x3) window, 8-byte read of columns 4, 5
        case 7: return (kg * 1296 + 4 * n16 + 4 + 2 * (kg & 1)) * 4;            // ... odd planes two floats to the right
        case 8: return (kg * 1296 + 4 * n16 + 3) * 4;                           // ... dword read of column 3
        case 9: return (kg * 576 + n16 * 36) * 4;                               // F(4x4,3x3) weights, 16-byte read
        default: return lane * 4;                                              // dword per lane, consecutive
    }
}

template <int BYTES, bool WRITE>
__global__ __launch_bounds__(256, 1) void lds_kernel(long long* clocks, unsigned* sink, int pat, int iters) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    // (the instructions in inline assembly, eight per wait, each into its own registers: nothing but the LDS array between two waits;
    // eight copies of the pattern 8 KB apart: the same banks)
    const unsigned addr = (unsigned)(size_t)(smem) + pattern_address(pat, lane, w);
    u32x4 r[8];
    u32x4 data = u32x4{(unsigned)lane, 1u, 2u, 3u};
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = data;
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define LDS_OP(K)                                                                                                              \
    if constexpr (WRITE) {                                                                                                     \
        if constexpr (BYTES == 16) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(r[K]), "n"(K * 8192) : "memory");          \
        else asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(r[K][0]), "n"(K * 8192) : "memory");                            \
    } else {                                                                                                                   \
        if constexpr (BYTES == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[K]) : "v"(addr), "n"(K * 8192) : "memory");          \
        else if constexpr (BYTES == 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*reinterpret_cast<u32x2*>(&r[K])) : "v"(addr), "n"(K * 8192) : "memory"); \
        else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[K][0]) : "v"(addr), "n"(K * 8192) : "memory");                             \
    }
        LDS_OP(0) LDS_OP(1) LDS_OP(2) LDS_OP(3) LDS_OP(4) LDS_OP(5) LDS_OP(6) LDS_OP(7)
#undef LDS_OP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) clocks[blockIdx.x] = c1 - c0;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += r[k][0] ^ r[k][3];
    sink[blockIdx.x * 256 + threadIdx.x] = acc + reinterpret_cast<unsigned*>(smem)[threadIdx.x];
}

template <int BYTES, bool WRITE>
static void run(int pat, const char* label) {
    long long* d_clk;
    unsigned* d_sink;
    hipMalloc(&d_clk, 256 * 8);
    hipMalloc(&d_sink, 256 * 256 * 4);
    const int iters = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<BYTES, WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    lds_kernel<BYTES, WRITE><<<256, 256, 163840>>>(d_clk, d_sink, pat, iters);
    hipDeviceSynchronize();
    long long clk[256];
    hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 256; ++i) sum += clk[i];
    std::printf("%-78s %6.2f cycles per wave instruction and CU (4 waves issuing)\n", label, sum / 256 / ((double)iters * 8 * 4));
    hipFree(d_clk);
    hipFree(d_sink);
}

int main() {
    run<16, false>(0, "ds_read_b128, 16 bytes per lane consecutive");
    run<16, false>(1, "ds_read_b128, split kernel's fragment read, swizzled ring columns");
    run<16, false>(2, "ds_read_b128, split kernel's fragment read, plain ring columns");
    run<16, false>(9, "ds_read_b128, F(4x4,3x3) weight fragments (unit stride 576 floats, lane stride 36)");
    run<8, false>(6, "ds_read_b64, F(4x4,3x3) window columns 4, 5 (planes 1296 floats apart)");
    run<8, false>(7, "ds_read_b64, the same with the odd planes two floats to the right");
    run<4, false>(8, "ds_read_b32, F(4x4,3x3) window column 3");
    run<4, false>(10, "ds_read_b32, consecutive dwords");
    run<16, true>(5, "ds_write_b128, 16 bytes per lane consecutive (dword-load staging)");
    run<16, true>(3, "ds_write_b128, staging write behind the transpose, swizzled ring columns");
    run<16, true>(4, "ds_write_b128, staging write behind the transpose, plain columns (lanes 4 units apart)");
    return 0;
}
