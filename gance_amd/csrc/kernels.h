// Internal declarations shared by the HIP translation units of libgance_hip.so.
#ifndef GANCE_KERNELS_H
#define GANCE_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <string>

namespace gance {

// Entry points run on the device their engine / buffers live on and hand the caller's current device back
// (a torch process may have another one current).
class DeviceGuard {
  public:
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&previous_) != hipSuccess) previous_ = -1;
        status_ = (device >= 0 && device != previous_) ? hipSetDevice(device) : hipSuccess;
        switched_ = status_ == hipSuccess && device >= 0 && device != previous_;
    }
    ~DeviceGuard() {
        if (switched_ && previous_ >= 0) (void)hipSetDevice(previous_);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    hipError_t status() const { return status_; }

  private:
    int previous_ = -1;
    bool switched_ = false;
    hipError_t status_ = hipSuccess;
};

// device ordinal a device pointer belongs to (-1 if it cannot be told)
inline int device_of_pointer(const void* ptr) {
    hipPointerAttribute_t attr;
    if (ptr == nullptr || hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return attr.device;
}

// One lazily computed int per device (kernel attributes such as the dynamic-LDS opt-in are per device):
// `init(device, &value)` runs once per device, under a lock.
constexpr int kMaxDevices = 64;
struct PerDeviceInt {
    std::mutex mutex;
    bool ready[kMaxDevices] = {};
    int value[kMaxDevices] = {};
    template <typename Init>
    hipError_t get(Init init, int* out) {
        int device = 0;
        hipError_t e = hipGetDevice(&device);
        if (e != hipSuccess) return e;
        if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lock(mutex);
        if (!ready[device]) {
            int v = 0;
            if ((e = init(device, &v)) != hipSuccess) return e;
            value[device] = v;
            ready[device] = true;
        }
        *out = value[device];
        return hipSuccess;
    }
};

// records the message gance_last_error() returns (thread local) and hands `code` back
int set_last_error(int code, const std::string& message);

constexpr int kMaxTaps = 9;
constexpr int kEpilogueRaw = 0;   // out = acc * d            (split-K slabs, transposed-conv planes)
constexpr int kEpilogueFull = 1;  // out = lrelu(acc * d + noise * strength + bias) * sqrt(2)
// kEpilogueFull's activation is not stored but fed straight to the layer's ToRGB (1x1 modulated conv
// to 3 channels, + bias + upsampled skip image) and converted to uint8 NHWC: the LAST layer of the
// network, when one wave holds every output channel of its pixels (Cout = BM = 32: the 1024^2 generator)
constexpr int kEpilogueRgb = 2;
// kEpilogueFull, and the layer's ToRGB product (no bias, no skip image) goes to rgb_y as [B][3][OH][OW]: the 64-channel
// Winograd kernel when one block holds every channel of its pixels (Cout = 64 or 32); out may be nullptr (last layer).
// launch_torgb(partial = that image) finishes it.
constexpr int kEpilogueFullRgbPart = 3;

// One launch of the implicit-GEMM modulated convolution (conv_mfma.hip).
// Activations are zero-bordered: x is [B][Cin][H+2][W+8] with the interior at [y+1][x+4].
struct ConvArgs {
    const float* x;
    const float* w;      // pre-arranged [m tile][K chunk][tap slot 0..8][KC][BM], runtime-scaled
    const float* s;      // style: s[b * s_stride + ci]
    const float* d;      // demodulation: d[b * d_stride + co]
    const float* noise;  // [OH][OW] or nullptr (full epilogue only); sample b reads noise + b * noise_b_stride
    const float* bias;   // [Cout]
    // out[split*slab_stride + cls*cls_stride + b*out_b_stride + co*out_c_stride
    //     + (oy+out_y_off)*out_row_stride + ox + out_x_off]
    float* out;
    int B, Cin, Cout, H, W;  // input tensor (interior size)
    int OH, OW;              // output grid: H x W for a conv, (H+1) x (W+1) positions when up
    int s_stride, d_stride;
    float noise_strength;
    int noise_b_stride;  // floats between the noise planes of two samples: 0 = one stored plane for the whole batch
                         // (randomize_noise=False), OH * OW = a plane per sample (gance_engine_randomize_noise: upstream
                         // draws tf.random_normal([N, 1, H, W]))
    int tiles_x, tiles_y, m_tiles;
    int total_tiles;  // virtual blocks of the launch (set by launch_modconv; a persistent grid is smaller)
    // runtime-geometry launches (transposed conv, BM = 128): after the tiles_x*tiles_y main tiles
    // come row_tiles 1x64 tiles on the position row y' = H and col_tiles 64x1 tiles on x' = W
    int row_tiles, col_tiles;
    int nsplit, chunks_per_split, total_chunks;
    int epilogue;
    int out_row_stride, out_y_off, out_x_off;
    long long out_b_stride, out_c_stride, slab_stride, cls_stride;
    long long x_b_stride;  // Cin*(H+2)*(W+8), or 0 when every sample reads the same tensor
    // kEpilogueRgb only: ToRGB weight [Cout][3] (runtime-scaled), its style rgb_s[b*s_stride + c], bias [3],
    // previous skip image [B][3][H/2][W/2] (or nullptr), outputs [B][3][H][W] float (or nullptr) and
    // [B][H][W][3] uint8 (or nullptr)
    const float* rgb_w;
    const float* rgb_s;
    const float* rgb_bias;
    const float* rgb_y_prev;
    float* rgb_y;
    unsigned char* rgb_u8;
    // winograd64 kernels: nullptr, or the style of the layer that reads `out`, s_next[b * s_stride + co]: the stored
    // activation is multiplied by it (the fused up kernel then skips the style scale in its K loop)
    const float* s_next;
    const float* rgb_coef;  // kEpilogueFullRgbPart: [B][Cout / 4][64], launch_winograd64_rgb_coef; rgb_y: [Cout / 64 or 1][B][3][OH][OW]
    // winograd43 kernels: blocks queued per CU (0 or 1: one persistent block per CU; k: k blocks per CU, each with 1/k of
    // the tiles, handed to the CUs by the hardware dispatcher as they come free -- the hardware is the tile queue: a CU that
    // something else holds for a while (a copy kernel of a collective) delays 1/k of its share, not the launch's tail)
    int grid_rounds;
    int xcd_blocking;  // winograd43 kernels, layers with 16 channel tiles: 1 = an XCD's 32 concurrent tiles are 4 pixel tiles x 8 channel tiles (else 2 x 16)
    // winograd43 kernels: host-visible word a block sets (to 43) when its SIMDs did not each get exactly two of its waves
    // (the pairing the kernel's roles rest on); the launch's output is then invalid and the engine says so on its next call
    int* fault_flag;
    unsigned long long* debug_stamps;  // [blocks][4] s_memrealtime stamps when debug_flags & 16
    int debug_flags;       // timing ablations only (GANCE_DEBUG_CONV): 1 no stores, 2 no DMA after chunk 0, 4 no MFMA
};

struct ConvTileInfo {
    int BM, TB, TH, TW, KC, up;
};
constexpr int kNumConvTiles = 15;
extern const ConvTileInfo kConvTiles[kNumConvTiles];

hipError_t launch_modconv(int tile_id, const ConvArgs& args, int total_blocks, hipStream_t stream);

// Winograd F(2x2, 3x3) form of the stride-1 conv (winograd_conv.hip): 32 output channels x 8x64 pixels
// per block; args.w points at the layer's transformed weights [m_tile][chunk][16][4][32].
bool winograd_supported(int cin, int cout, int H, int W);
size_t winograd_weight_floats(int cin, int cout);
void winograd_transform_weights(const float* w_in /*[9][cin][cout]*/, int cin, int cout, float* w_out);
hipError_t launch_winograd_conv(const ConvArgs& args, hipStream_t stream);

// The same Winograd form for layers with >= 64 output channels (winograd64_conv.hip): 64 channels x 8x32 pixels per
// block on v_mfma_f32_16x16x4_f32; args.w points at [m tile of 64][chunk of 8][16][8][16][4].
bool winograd64_supported(int cin, int cout, int H, int W);
size_t winograd64_weight_floats(int cin, int cout);
void winograd64_transform_weights(const float* w_in /*[9][cin][cout]*/, int cin, int cout, float* w_out);
hipError_t launch_winograd64_conv(const ConvArgs& args, hipStream_t stream);
bool winograd64_input_prescaled(int cout);  // the geometry this Cout runs in expects x already multiplied by the layer's style
bool winograd64_rgb_supported(int cout);
int winograd64_rgb_partials(int cout);  // partial images the launch writes: [partials][B][3][OH][OW]
hipError_t launch_winograd64_rgb_coef(const float* rgb_w, const float* rgb_s, int s_stride, int B, int cout, float* coef, hipStream_t stream);

// Winograd F(4x4, 3x3) form (winograd43_conv.hip): 36 multiplies per 16 outputs instead of 16 per 4; 32 channels x
// 16x64 pixels per block, two waves per SIMD; the input arrives multiplied by the layer's style (prescaled, like the
// 32-channel geometry above); args.w points at [m tile of 32][chunk of 4][4864 floats]. kEpilogueFull only.
bool winograd43_supported(int cin, int cout, int H, int W);
size_t winograd43_weight_floats(int cin, int cout);
void winograd43_transform_weights(const float* w_in /*[9][cin][cout]*/, int cin, int cout, float* w_out);
hipError_t launch_winograd43_conv(const ConvArgs& args, hipStream_t stream);
// kEpilogueFullRgbPart: every block (32 channels: the two channel-tile waves of a SIMD add their sums through LDS) writes its
// own partial ToRGB image, rgb_y [Cout / 32][B][3][OH][OW]; rgb_coef is launch_winograd64_rgb_coef's table
bool winograd43_rgb_supported(int cout);
int winograd43_rgb_partials(int cout);

// Conv0_up as ONE kernel (upfir_fused.hip): transposed conv on the matrix cores + [1,3,3,1]^2 FIR + noise +
// bias + leaky ReLU, for inputs >= 64 wide. Blocks sweep 64-column strips in steps of 8 position rows.
struct UpFirArgs {
    const float* x;      // zero-bordered [B][Cin][H+2][W+8]
    const float* w;      // [m tile of 32][chunk of 4][tap slot 0..8][4][32], runtime-scaled
    const float* s;      // style: s[b * s_stride + ci]
    const float* d;      // demodulation: d[b * d_stride + co]
    const float* noise;  // [2H][2W] or nullptr; sample b reads noise + b * noise_b_stride
    const float* bias;   // [Cout]
    float* out;          // zero-bordered [B][Cout][2H+2][2W+8]
    int B, Cin, Cout, H, W;
    int s_stride, d_stride;
    float noise_strength;
    int noise_b_stride;  // 0 (one plane for the batch) or 4 H W (a plane per sample), as in ConvArgs
    int m_tiles, strips, segs, rows_per_seg, total_blocks;  // set by upfir_plan
    int step_rows;                                           // position rows per step (set by the plan: 8, or 16 in the narrow strip geometries of upfir16_fused.hip)
    int stagger_phases, stagger_ticks;                       // set by upfir_plan: start delay (phase * ticks of 10 ns)
    int debug_flags;  // timing ablations (GANCE_DEBUG_UPFIR): 1 no stores, 2 no epilogue at all, 4 no MFMA, 8 no DMA after the first chunk
    long long x_b_stride;
    // nullptr, or the style of the NEXT layer, s_next[b * s_stride + co]: folded into the leaky ReLU, i.e. the stored
    // activation is multiplied by it (the Winograd kernel on 16x16x4 MFMAs takes its input pre-scaled)
    const float* s_next;
    int input_prescaled;  // x arrives multiplied by this layer's own style (its producer was given s_next): no style scale in the K loop
    int pair_form;        // upfir16 only: w is the pair-form image (upfir16x_arrange_weights): F(2,2) along x, 15 MFMAs per pair of columns instead of 18
    const void* x_units;  // upfir_split_roles only: x times the layer's style, split into bf16 parts (launch_upfirr_split_activation's image)
};
bool upfir_supported(int cin, int cout, int H, int W);
size_t upfir_weight_floats(int cin, int cout);
void upfir_arrange_weights(const float* w_in /*[9][cin][cout] scaled*/, int cin, int cout, const int* up_tap_weight, float* w_out);
void upfir_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* args);
hipError_t launch_upfir_fused(const UpFirArgs& args, hipStream_t stream);

// The same layer with 16 output channels per block and two blocks per CU (upfir16_fused.hip): w points at
// [m tile of 16][chunk of 8][1280 floats]; the plan gives every CU two blocks before it cuts row segments.
bool upfir16_supported(int cin, int cout, int H, int W);
size_t upfir16_weight_floats(int cin, int cout);
void upfir16_arrange_weights(const float* w_in /*[9][cin][cout] scaled*/, int cin, int cout, const int* up_tap_weight, float* w_out);
void upfir16_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* args);
hipError_t launch_upfir16_fused(const UpFirArgs& args, hipStream_t stream);
bool upfir16x_supported(int cin, int cout, int H, int W);
size_t upfir16x_weight_floats(int cin, int cout);
void upfir16x_arrange_weights(const float* w_in /*[9][cin][cout] scaled*/, int cin, int cout, const int* up_tap_weight, float* w_out);

// The same layer with its K loop on the bf16 matrix cores from split operands (upfir_split.hip: three bf16 parts per fp32 value, six
// product terms, fp32 accumulation -- fp32 accuracy): 16 channels per block, one block per CU, whole image height per block (no row
// segments), inputs whose width the 64-column strips tile. w points at upfirs_arrange_weights' image (bf16 parts).
bool upfirs_supported(int cin, int cout, int H, int W);
size_t upfirs_weight_floats(int cin, int cout);
void upfirs_arrange_weights(const float* w_in /*[9][cin][cout] scaled*/, int cin, int cout, const int* up_tap_weight, float* w_out);
void upfirs_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* args);
hipError_t launch_upfir_split(const UpFirArgs& args, hipStream_t stream);

// The split-operand form with the work of a block in two roles (upfir_split_roles.hip): four matrix waves (MFMAs only) and four
// vector waves (staging, split, FIR epilogue of the step before) per block, two waves per SIMD, 4 position rows per step. Same
// weight image as upfir_split.hip (upfirs_arrange_weights); the plan sets step_rows = 4.
bool upfirr_supported(int cin, int cout, int H, int W);
void upfirr_plan(int B, int cout, int H, int W, int num_cus, UpFirArgs* args);
hipError_t launch_upfir_split_roles(const UpFirArgs& args, hipStream_t stream);
// ... and its input image: x (times the style s, or s == nullptr) as three bf16 parts per value in 16-byte units of 8 channels,
// [B][Cin / 32][H + 2][part * 4 + k-group][W + 8]: upfirr_units_bytes(B, Cin, H, W) bytes
size_t upfirr_units_bytes(int B, int cin, int H, int W);
hipError_t launch_upfirr_split_activation(const float* x, long long x_b_stride, const float* s, int s_stride, void* out, int B, int cin, int H, int W, hipStream_t stream);

// The two smallest up layers (4x4 -> 8x8, 8x8 -> 16x16) in scatter form (gemm_forms.hip): pack (x * style -> the GEMM's B image),
// ONE dense GEMM P[tap slot * Cout + co][b H W + position] (M = 9 Cout, K = Cin), gather (taps of a class, x demod) into the parity
// planes the FIR pass reads. Replaces the transposed-conv launch where the position grid tiles badly (81 of 256 tile slots).
struct UpGemmArgs {
    const float* x;  // zero-bordered [B][Cin][H+2][W+8]
    const float* w;  // upgemm_arrange_weights' image
    const float* s;  // style: s[b * s_stride + ci]
    const float* d;  // demodulation: d[b * d_stride + co]
    float* packed;   // workspace: upgemm_packed_floats
    float* prod;     // workspace: upgemm_prod_floats: P [9 Cout][n_tiles * 128]
    float* t;        // parity planes: t + cls * cls_stride + b * unit_stride + co * (H+3) * (W+8) + (y'+1) * (W+8) + x'+4
    long long x_b_stride, cls_stride, unit_stride;
    int B, Cin, Cout, H, W, s_stride, d_stride;
    int n_tiles;  // upgemm_n_tiles(B, H, W)
    int bf16_split;  // experiment (0 off): 1 = three bf16 parts per value, six product terms; 2 = two fp16 parts, three terms. w is upgemm_arrange_weights_split's image
};
// GEMM columns (samples x input positions) of a layer with `cout` channels that the engine's product buffer holds: 9 x 512 x `columns_512`
// floats (engine.hip upgemm_buffer_columns(): 16384 = 302 MB = 16 frames at 32x32 (512 channels), 8 at 64x64 (256 channels), 4 at 128x128 (128))
constexpr int upgemm_max_columns(int cout, int columns_512) { return (int)((long long)columns_512 * 512 / cout); }
bool upgemm_supported(int cin, int cout, int H, int W);
size_t upgemm_weight_floats(int cin, int cout);
int upgemm_n_tiles(int B, int H, int W);
size_t upgemm_packed_floats(int B, int cin, int H, int W);
size_t upgemm_prod_floats(int B, int cout, int H, int W);
void upgemm_arrange_weights(const float* w_in /*[9][cin][cout] scaled*/, int cin, int cout, const int* up_tap_weight, float* w_out);
void upgemm_arrange_weights_split(const float* w_in, int cin, int cout, const int* up_tap_weight, int mode, void* w_out /* 1.5 x upgemm_weight_floats floats */);
hipError_t launch_upgemm(const UpGemmArgs& args, hipStream_t stream);

// The stride-1 layers at 8x8 and 16x16 in Winograd F(4x4, 3x3) GEMM form (gemm_forms.hip): input transform into 36 B images, the
// same GEMM kernel as 36 independent products, output transform + demod + noise + bias + leaky ReLU.
struct WinoGemmArgs {
    const float* x;      // zero-bordered [B][Cin][H+2][W+8]
    const float* w;      // winogemm_arrange_weights' image
    const float* s;      // style: s[b * s_stride + ci]
    const float* d;      // demodulation: d[b * d_stride + co]
    const float* noise;  // [H][W] or nullptr; sample b reads noise + b * noise_b_stride
    const float* bias;   // [Cout]
    float* packed;       // workspace: winogemm_packed_floats
    float* prod;         // workspace: winogemm_prod_floats
    float* out;          // zero-bordered [B][Cout][H+2][W+8]
    long long x_b_stride, out_b_stride;
    float noise_strength;
    int noise_b_stride;
    int B, Cin, Cout, H, W, s_stride, d_stride;
    int n_tiles;  // winogemm_n_tiles(B, H, W)
    int bf16_split;  // experiment (0 off, 1 bf16 x 3, 2 fp16 x 2): w is winogemm_arrange_weights_split's image
};
constexpr int kWinoGemmMaxColumns = 1024;  // GEMM columns (samples x 4x4 tiles) the engine's buffers hold: 64 frames at 16x16, 4 at 64x64, 1 at 128x128
bool winogemm_supported(int cin, int cout, int H, int W);
size_t winogemm_weight_floats(int cin, int cout);
int winogemm_n_tiles(int B, int H, int W);
size_t winogemm_packed_floats(int B, int cin, int H, int W);
size_t winogemm_prod_floats(int B, int cout, int H, int W);
void winogemm_arrange_weights(const float* w_in /*[9][cin][cout] scaled*/, int cin, int cout, float* w_out);
void winogemm_arrange_weights_split(const float* w_in, int cin, int cout, int mode, void* w_out /* 1.5 x winogemm_weight_floats floats */);
hipError_t launch_winogemm(const WinoGemmArgs& args, hipStream_t stream);

// ---- aux_kernels.hip ----

// Mapping network: one dense 512->512 layer with lrelu*sqrt2 (G_mapping DenseN).
//   out[b][j] = lrelu(sum_k in[b][k] * w[k][j] + bias[j]) * sqrt2 ; w, bias already runtime-scaled.
// If `normalize` the input row is first multiplied by rsqrt(mean(in^2) + 1e-8) (first layer).
hipError_t launch_mapping_dense(const float* in, const float* w, const float* bias, float* out,
                                int B, int normalize, hipStream_t stream);
// dlat[b][row][k] = avg[k] + psi * (w[b][k] - avg[k]) for row < num_rows (broadcast + truncation).
hipError_t launch_broadcast_truncate(const float* w, const float* avg, float psi, float* dlat,
                                     int B, int num_rows, hipStream_t stream);

// Styles of every modulated conv at once. Column block `cb` (32 columns) reads dlatent row
// blk_row[cb]:  s[b][col] = sum_k dlat[b][blk_row][k] * A[k][col] + mod_bias_plus_one[col].
hipError_t launch_styles(const float* dlat, const float* A, const float* bias1, const int* blk_row,
                         float* s, int B, int num_rows, int ctot, hipStream_t stream);

struct DemodLayer {
    long long w2_off;  // offset into w2 pool: W2[ci][co] = sum_tap w^2
    int s_off;         // column offset of this layer's styles
    int d_off;         // column offset of this layer's demod coefficients
    int cin, cout;
};
// d[b][d_off + co] = rsqrt(sum_ci s[b][s_off+ci]^2 * W2[ci][co] + 1e-8)
hipError_t launch_demod(const float* s, const float* w2_pool, const DemodLayer* layers,
                        int num_layers, float* d, int B, int ctot, int dtot, hipStream_t stream);

// Conv0_up second half: 4x4 FIR ([1,3,3,1] x [1,3,3,1] / 16, pad 1/1) over the (2H+1)^2
// intermediate held as four zero-bordered parity planes, + noise, bias, lrelu*sqrt2.
//   t + cls*cls_stride + (split*B + b)*unit_stride + c*(H+3)*(W+8) + (y'+1)*(W+8) + x'+4
// cls = 2*py + px. Writes the zero-bordered activation [B][C][2H+2][2W+8].
struct FirArgs {
    const float* t;
    long long cls_stride, unit_stride;
    const float* noise;  // [2H][2W] or nullptr; sample b reads noise + b * noise_b_stride
    const float* bias;   // [C]
    float* out;
    float noise_strength;
    int noise_b_stride;  // 0 or 4 H W, as in ConvArgs
    int B, C, H, W, nsplit;
    // nullptr, or the style of the NEXT layer, s_next[b * s_next_stride + c]: the stored activation is multiplied by it
    // (the Winograd kernel on 16x16x4 MFMAs takes its input pre-scaled: winograd64_conv.hip)
    const float* s_next;
    int s_next_stride;
};
hipError_t launch_fir_epilogue(const FirArgs& args, hipStream_t stream);

// Split-K finish for small stride-1 convs: dense slabs [nsplit][B][C][H][W] ->
// zero-bordered activation, out = lrelu(sum_slabs + noise*strength + bias) * sqrt2.
hipError_t launch_splitk_finish(const float* slabs, long long slab_stride, int nsplit,
                                const float* noise, float noise_strength, int noise_b_stride, const float* bias,
                                float* out, int B, int C, int H, int W, hipStream_t stream);

// ToRGB (modulated 1x1, no demod) + bias + FIR-upsampled skip image; optional uint8 NHWC output.
struct ToRgbArgs {
    const float* x;       // zero-bordered [B][Cin][R+2][R+8]
    const float* w;       // [Cin][3], runtime-scaled
    const float* s;       // s[b*s_stride + ci]
    const float* bias;    // [3]
    const float* y_prev;  // [B][3][R/2][R/2] or nullptr
    float* y;             // [B][3][R][R]
    uint8_t* u8;          // [B][R][R][3] or nullptr
    const float* partial; // nullptr, or [partials][B][3][R][R]: the channel sum is already there, in `partials` pieces added
                          // in order (one piece: may be y itself): x, w, s unused
    int partials;
    bool skip_y_store;    // the last layer when nobody reads the fp32 image: only u8 leaves (torgb_kernel only)
    int B, Cin, R, s_stride;
};
hipError_t launch_torgb(const ToRgbArgs& args, hipStream_t stream);

// out[b][0 .. plane) = standard-normal draws for b < samples, a function of (seed, layer, sample id, index) only, sample id =
// sample_ids[b] (device memory) or first_sample + b (randomize_noise of the vector path)
hipError_t launch_normal_noise(float* out, size_t plane, int samples, unsigned long long seed, unsigned long long layer,
                               unsigned long long first_sample, const long long* sample_ids, hipStream_t stream);

// Bicubic (a = -0.75) resize of uint8 NHWC RGB frames [batch][src][src][3] -> [batch][dst][dst][3].
hipError_t launch_resize_bicubic_u8(const uint8_t* in, int batch, int src, uint8_t* out, int dst,
                                    hipStream_t stream);

}  // namespace gance
#endif
