// libgance_hip.so: the StyleGAN2 config-f engine behind include/gance_hip.h.
//
// Owns one network's weights (re-laid-out for the kernels) and a workspace sized for max_batch
// frames in HBM, and turns a batch of dlatents (or z vectors) into uint8 NHWC frames with a fixed
// sequence of kernel launches on the caller's stream. Replaces the TF1 session behind
// gance/network_interface/network_functions.py:114-192 (wrap_loaded_network) for the reference's
// two call forms (`network.run`, `network.components.synthesis.run`).
//
// There is NO CPU fallback: without a HIP device every entry point fails with an error code.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gance_hip.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& message) {
    g_last_error = message;
    return code;
}
}  // namespace
namespace gance {
int set_last_error(int code, const std::string& message) { return fail(code, message); }
}  // namespace gance
namespace {

#define GANCE_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t gance_err_ = (expr);                                                    \
        if (gance_err_ != hipSuccess) {                                                    \
            return fail(gance_err_ == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY        \
                                                          : GANCE_ERR_HIP,                 \
                        std::string(#expr) + ": " + hipGetErrorString(gance_err_));        \
        }                                                                                  \
    } while (0)

constexpr int kDlatent = 512;
constexpr int kMappingLayers = 8;
constexpr float kMappingLrmul = 0.01f;

int nf(int stage) {
    const int v = (16 << 10) >> stage;
    return std::min(std::max(v, 1), 512);
}

struct ConvLayerHost {
    int layer_idx, res_log2, cin, cout;
    bool up;
};
struct RgbLayerHost {
    int res_log2, cin, row;
};

void build_spec(int res_log2, std::vector<ConvLayerHost>* convs, std::vector<RgbLayerHost>* rgbs) {
    convs->push_back({0, 2, nf(1), nf(1), false});
    rgbs->push_back({2, nf(1), 1});
    for (int res = 3; res <= res_log2; ++res) {
        convs->push_back({res * 2 - 5, res, nf(res - 2), nf(res - 1), true});
        convs->push_back({res * 2 - 4, res, nf(res - 1), nf(res - 1), false});
        rgbs->push_back({res, nf(res - 1), res * 2 - 3});
    }
}

uint64_t blob_floats(int res_log2) {
    std::vector<ConvLayerHost> convs;
    std::vector<RgbLayerHost> rgbs;
    build_spec(res_log2, &convs, &rgbs);
    uint64_t n = 0;
    n += (uint64_t)kMappingLayers * (kDlatent * kDlatent + kDlatent);
    n += kDlatent;
    n += (uint64_t)nf(1) * 16;
    for (const auto& c : convs)
        n += (uint64_t)9 * c.cin * c.cout + (uint64_t)kDlatent * c.cin + c.cin + 1 + c.cout;
    for (const auto& r : rgbs) n += (uint64_t)r.cin * 3 + (uint64_t)kDlatent * r.cin + r.cin + 3;
    for (const auto& c : convs) n += (uint64_t)1 << (2 * c.res_log2);
    return n;
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// One launch of the conv kernel: a stride-1 conv, or a transposed conv with its four parity classes
// fused. Wide transposed convs (BM = 128) tile the H x W position grid exactly with 8x8 tiles and
// cover the extra position row y' = H / column x' = W with 1x64 / 64x1 strip tiles in the SAME
// launch (runtime tile geometry); the narrow ones tile the (H+1) x (W+1) grid directly.
struct LayerPlan {
    int tile_id;
    int OH, OW;  // output bound for masking: H x W, or (H+1) x (W+1) positions when up
    int tiles_x, tiles_y, tiles_b, row_tiles, col_tiles;
    int m_tiles, nsplit, chunks_per_split, total_chunks, total_blocks;
};

int ceil_div(int a, int b) { return (a + b - 1) / b; }

int layer_bm(int cout) { return cout == 32 ? 32 : (cout == 64 ? 64 : 128); }

// K chunk (input channels per LDS stage) of the narrow layers. Tunable for experiments:
// GANCE_TUNE_KC_CONV = 4 or 8 (read once). Transposed convs use 8.
int tuned_kc(bool up) {
    static const int conv_kc = [] { const char* v = std::getenv("GANCE_TUNE_KC_CONV"); return v && std::atoi(v) == 8 ? 8 : 4; }();
    static const int up_kc = [] { const char* v = std::getenv("GANCE_TUNE_KC_UP"); return v && std::atoi(v) == 4 ? 4 : 8; }();
    return up ? up_kc : conv_kc;
}
// GANCE_TUNE_KC_UP128 = 2: the wide transposed convs stage 2 input channels per chunk (tile 14)
int tuned_kc_up128() {
    static const int kc = [] { const char* v = std::getenv("GANCE_TUNE_KC_UP128"); return v && std::atoi(v) == 2 ? 2 : 4; }();
    return kc;
}
int layer_kc(int cout, bool up) { return layer_bm(cout) == 128 ? (up ? tuned_kc_up128() : 4) : tuned_kc(up); }

int choose_tile(int cout, bool up, int OH, int OW, int B) {
    const bool kc4 = tuned_kc(up) == 4;
    if (cout == 32) return up ? (kc4 ? 12 : 6) : (kc4 ? 10 : 0);
    if (cout == 64) return up ? (kc4 ? 13 : 7) : (kc4 ? 11 : 1);
    if (up) return tuned_kc_up128() == 2 ? 14 : 8;
    const int first = 2, last = 5;
    int best = first;
    long best_tiles = -1;
    for (int id = first; id <= last; ++id) {
        const auto& t = gance::kConvTiles[id];
        const long tiles = (long)ceil_div(B, t.TB) * ceil_div(OH, t.TH) * ceil_div(OW, t.TW);
        if (best_tiles < 0 || tiles < best_tiles) {
            best_tiles = tiles;
            best = id;
        }
    }
    return best;
}

int choose_nsplit(int base_blocks, int chunks) {
    if (base_blocks >= 384) return 1;
    const int want = ceil_div(768, base_blocks);
    for (int d = 1; d <= chunks; ++d)
        if (chunks % d == 0 && d >= want) return d;
    return chunks;
}

LayerPlan plan_layer(const ConvLayerHost& c, int B) {
    LayerPlan p{};
    const int res = 1 << c.res_log2;
    // strips pay once the position grid has several tiles per side; below that the launch is
    // latency-bound and extra blocks only hurt
    // (GANCE_TUNE_STRIPS_MIN, read once: smallest input side that takes the strips; default 16)
    static const int strips_min = [] { const char* v = std::getenv("GANCE_TUNE_STRIPS_MIN"); return v ? std::atoi(v) : 16; }();
    const bool strips = c.up && layer_bm(c.cout) == 128 && res / 2 >= strips_min;
    const int grid = c.up ? (strips ? res / 2 : res / 2 + 1) : res;  // the tiled grid
    p.OH = p.OW = c.up ? res / 2 + 1 : res;
    p.tile_id = choose_tile(c.cout, c.up, grid, grid, B);
    const auto& t = gance::kConvTiles[p.tile_id];
    p.tiles_x = ceil_div(grid, t.TW);
    p.tiles_y = ceil_div(grid, t.TH);
    p.tiles_b = ceil_div(B, t.TB);
    p.row_tiles = strips ? ceil_div(res / 2 + 1, 64) : 0;
    p.col_tiles = strips ? ceil_div(res / 2, 64) : 0;
    p.m_tiles = c.cout / t.BM;
    p.total_chunks = c.cin / t.KC;
    const int base = p.m_tiles * (p.tiles_x * p.tiles_y + p.row_tiles + p.col_tiles) * p.tiles_b;
    p.nsplit = choose_nsplit(base, p.total_chunks);
    if (c.up) p.nsplit = std::min(p.nsplit, 8);  // the FIR pass re-reads every slab
    while (p.total_chunks % p.nsplit) --p.nsplit;
    p.chunks_per_split = p.total_chunks / p.nsplit;
    p.total_blocks = base * p.nsplit;
    return p;
}

// zero-bordered geometry
size_t act_plane(int res) { return (size_t)(res + 2) * (res + 8); }       // one channel, interior at [y+1][x+4]
size_t t_plane(int h) { return (size_t)(h + 3) * (h + 8); }               // one channel, one class

// weight slot t of the transposed conv reads filter tap kUpTapWeight[t] (= wy*3+wx); order must
// match conv_mfma.hip's tap tables: EE (0,0) (0,-1) (-1,0) (-1,-1) | EO (0,0) (-1,0) | OE (0,0) (0,-1) | OO
const int kUpTapWeight[9] = {8, 6, 2, 0, 7, 1, 5, 3, 4};

struct StepRecord {
    char name[64];
    double flops, bytes;
    hipEvent_t start, stop;
};

}  // namespace

// Per-call scratch (activations, parity planes, split-K slabs, skip images, styles ...): about 0.8 GB per frame
// of batch capacity at 1024^2 against 0.55 GB of weights per network (135 MB as trained, the rest the same weights as each kernel form's LDS image). It holds nothing between calls except its
// zero borders, which depend only on the resolution, so every engine of one (device, resolution, max_batch)
// shares ONE workspace: 20 resident networks cost 20 x weights + 1 x workspace. Calls that share it are ordered
// by an event (a call waits for the previous user's last kernel, on whatever stream that ran).
struct gance_workspace {
    int device = 0, resolution = 0, max_batch = 0;
    float *dlat = nullptr, *map_a = nullptr, *map_b_buf = nullptr, *z_in = nullptr;
    float *styles = nullptr, *demod = nullptr;
    std::vector<float*> act;      // per conv layer: zero-bordered output [Bmax][cout][res+2][res+8]
    std::vector<float*> tplanes;  // per up layer: [4 cls][max_units][cout][H+3][W+8] (else nullptr)
    float* slabs = nullptr;       // split-K scratch of the small stride-1 convs (dense)
    void* x_units = nullptr;  // an up layer's input split into bf16 parts (upfir_split_roles.hip: launch_upfirr_split_activation)
    size_t x_units_bytes = 0;
    float *up_packed = nullptr, *up_prod = nullptr;  // the GEMM forms' operand images and products (gemm_forms.hip: scatter-form up layers, Winograd at 8x8 / 16x16)
    float* ybuf[2] = {nullptr, nullptr};
    float* rgb_coef = nullptr;  // [Bmax][8 m tiles][16][64]: A operands of a ToRGB product fused into a Winograd conv epilogue
    float* rgb_part = nullptr;  // [m tiles][Bmax][3][R][R]: its partial images where a pixel's channels span several blocks
    uint8_t* u8buf = nullptr;  // staging for the host-buffer entry points
    hipEvent_t last_use = nullptr;
    bool used = false;
    size_t bytes = 0;
    // host-buffer entry points: their own stream and pinned staging (latents in, uint8 frames out)
    hipStream_t host_stream = nullptr;
    int* fault_flag = nullptr;  // host-mapped word: kernels whose wave-placement assumption failed at run time set it (ConvArgs::fault_flag)
    float* pinned_in = nullptr;
    uint8_t* pinned_out = nullptr;
    ~gance_workspace() {
        gance::DeviceGuard guard(device);
        if (host_stream) hipStreamDestroy(host_stream);
        if (pinned_in) hipHostFree(pinned_in);
        if (fault_flag) hipHostFree(fault_flag);
        if (pinned_out) hipHostFree(pinned_out);
        hipFree(dlat);
        hipFree(map_a);
        hipFree(map_b_buf);
        hipFree(z_in);
        hipFree(styles);
        hipFree(demod);
        for (float* ptr : act) hipFree(ptr);
        for (float* ptr : tplanes) hipFree(ptr);
        hipFree(slabs);
        hipFree(x_units);
        hipFree(up_packed);
        hipFree(up_prod);
        hipFree(ybuf[0]);
        hipFree(ybuf[1]);
        hipFree(rgb_coef);
        hipFree(rgb_part);
        hipFree(u8buf);
        if (last_use) hipEventDestroy(last_use);
    }
};

constexpr int kWino43DefaultMaxRes = 1024;  // every Conv1 from 32x32 up (measured faster than the F(2x2,3x3) kernels on all six: DESIGN.md §3)

struct GraphEntry {
    hipGraphExec_t exec = nullptr;
    bool warmed = false;
    bool disabled = false;  // capture failed once: this combination stays on eager launches
};

struct gance_engine {
    gance_engine_config cfg{};
    int res_log2 = 0;
    int num_rows = 0;  // W
    std::vector<ConvLayerHost> convs;
    std::vector<RgbLayerHost> rgbs;

    // device weight pool
    float* pool = nullptr;
    size_t pool_floats = 0;
    // offsets into pool
    size_t map_w[kMappingLayers]{}, map_b[kMappingLayers]{};
    size_t avg_off = 0, const_off = 0, A_off = 0, bias1_off = 0, w2_off = 0;
    std::vector<size_t> conv_w, conv_bias, conv_noise;
    std::vector<size_t> wino_w;  // Winograd-domain weights of the stride-1 layers that support them (else SIZE_MAX)
    std::vector<size_t> wino64_w;  // the same for the 64-channel Winograd kernel (layers with >= 64 output channels)
    // randomize_noise (the vector path): per-sample planes drawn by gance_engine_randomize_noise, layer li at
    // noise_rand + noise_rand_off[li] as [max_batch][res][res] (SIZE_MAX: the layer's strength is zero, it reads no noise);
    // while noise_randomized the conv launches read these (a plane per sample), else the stored buffers in the pool
    float* noise_rand = nullptr;
    std::vector<size_t> noise_rand_off;
    int noise_rand_count = 0;  // samples the last draw covered
    bool noise_randomized = false;
    std::vector<size_t> wino43_w;  // F(4x4, 3x3) weights (winograd43_conv.hip), SIZE_MAX where the layer does not take that form
    std::vector<size_t> upfir_w;  // fused transposed-conv + FIR kernel's weight image of the up layers that support it (else SIZE_MAX)
    std::vector<size_t> upfir16_w;  // the same for its 16-channel, two-blocks-per-CU geometry (upfir16_fused.hip)
    std::vector<size_t> winogemm_w;  // weight image of the Winograd F(4x4,3x3) GEMM form of the stride-1 layers at 8x8, 16x16 (gemm_forms.hip; else SIZE_MAX)
    std::vector<size_t> upgemm_w;  // weight image of the scatter-form GEMM of the two smallest up layers (gemm_forms.hip; else SIZE_MAX)
    size_t up_packed_floats = 0, up_prod_floats = 0;
    size_t x_units_bytes = 0;  // the largest split input image of an up layer that can take the role-split form
    int gemm_bf16 = 0;  // experiment (GANCE_TUNE_GEMM_BF16X6 when the engine is created): the GEMM forms on the 16-bit matrix cores from split operands: 1 = bf16 x 3 (six terms), 2 = fp16 x 2 (three terms)
    std::vector<size_t> upfir16x_w;  // ... and for that geometry's pair form (F(2,2) along x: 15 MFMAs per pair of columns instead of 18)
    std::vector<size_t> upfirs_w;    // split-operand form of the fused up kernel (upfir_split.hip: three bf16 parts per value, six terms, fp32 accumulation)
    int upfir_split = 1;  // GANCE_TUNE_UPFIR_SPLIT when the engine is created: 0 never, 1 (default) where a launch fills the chip without row segments, 2 wherever supported
    int upfir_split_roles = 0;  // GANCE_TUNE_UPFIR_SPLIT_ROLES when the engine is created: 0 (default) the one-role kernel (upfir_split.hip); 1 the experiment with matrix and vector waves (upfir_split_roles.hip + a split pass over its input: slower than the one-role kernel once that pass is paid for)
    int upfir_split_max_res = 1024;  // GANCE_TUNE_UPFIR_SPLIT_MAXRES: the largest OUTPUT side that takes the split form in mode 1 (measured: DESIGN.md section 3; 512 until the staging went to 16-byte loads)
    int num_cus = 256;
    std::vector<float> conv_ns;
    std::vector<int> conv_s_off, conv_d_off;
    std::vector<size_t> rgb_w, rgb_bias;
    std::vector<int> rgb_s_off;
    int ctot = 0, dtot = 0;
    int* blk_row = nullptr;
    gance::DemodLayer* demod_layers = nullptr;

    // workspace (shared with the other engines of the same device, resolution and max_batch)
    std::shared_ptr<gance_workspace> ws;
    // captured launch sequences of the host-buffer entry points, by (batch, entry kind, psi bits, float image wanted, noise source)
    std::map<std::tuple<int, int, unsigned, int, int>, GraphEntry> graphs;
    std::vector<int> t_units;     // max_units of each up layer's parity planes
    size_t slab_floats = 0, y_floats = 0, rgb_part_floats = 0;

    // profiling / debug
    std::vector<StepRecord> steps;
    std::string profile_only;  // non-empty: bracket only launches whose name contains it
    int steps_used = 0;
    int debug_stop_after = 0;
    bool keep_skip_image = false;  // the caller will read the final fp32 skip image out of ybuf
    int last_act_layer = 0, last_act_c = 0, last_act_side = 0;
    hipStream_t last_stream = nullptr;
};

namespace {

void free_engine(gance_engine* e) {
    if (!e) return;
    gance::DeviceGuard guard(e->cfg.device);
    for (auto& s : e->steps) {
        hipEventDestroy(s.start);
        hipEventDestroy(s.stop);
    }
    hipFree(e->pool);
    hipFree(e->noise_rand);
    hipFree(e->blk_row);
    hipFree(e->demod_layers);
    if (e->ws && e->ws->used) hipEventSynchronize(e->ws->last_use);  // nothing of this engine still runs on the shared scratch
    for (auto& kv : e->graphs)
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
    e->ws.reset();  // the last engine of a (device, resolution, max_batch) frees the workspace
    delete e;
}

// Profiling bracket around one launch.
struct StepScope {
    gance_engine* e;
    hipStream_t stream;
    StepRecord* rec = nullptr;
    StepScope(gance_engine* engine, hipStream_t s, const char* name, double flops, double bytes)
        : e(engine), stream(s) {
        if (!(e->cfg.flags & GANCE_FLAG_PROFILE_STEPS)) return;
        if (!e->profile_only.empty() && std::strstr(name, e->profile_only.c_str()) == nullptr) return;
        if (e->steps_used == (int)e->steps.size()) {
            StepRecord r{};
            hipEventCreate(&r.start);
            hipEventCreate(&r.stop);
            e->steps.push_back(r);
        }
        rec = &e->steps[e->steps_used++];
        std::snprintf(rec->name, sizeof(rec->name), "%s", name);
        rec->flops = flops;
        rec->bytes = bytes;
        hipEventRecord(rec->start, stream);
    }
    ~StepScope() {
        if (rec) hipEventRecord(rec->stop, stream);
    }
};

// the ToRGB a last-layer conv launch absorbs (kEpilogueRgb)
struct FusedRgb {
    const float *w, *s, *bias, *y_prev;
    float* y;
    uint8_t* u8;
};

// Geometry of the fused up kernel: GANCE_TUNE_UPFIR16 (read once per process) = 0: always 32 channels per block, one block per
// CU (upfir_fused.hip); 1: 16 channels per block, two blocks per CU (upfir16_fused.hip) wherever that kernel supports the layer.
static int upfir16_mode() {
    static const int mode = [] {
        const char* v = std::getenv("GANCE_TUNE_UPFIR16");
        return v ? std::atoi(v) : 1;
    }();
    return mode;
}

// GANCE_TUNE_UPFIR16X (read once per process) = 0: the fused up layers whose input the 64-column strips tile stay in direct form; 1
// (default): they run in the pair form (F(2,2) along x) when their input arrives pre-scaled.
static int upfir16x_mode() {
    static const int mode = [] {
        const char* v = std::getenv("GANCE_TUNE_UPFIR16X");
        return v ? std::atoi(v) : 1;
    }();
    return mode;
}

// GANCE_TUNE_UPGEMM (read once per process): the two smallest up layers (4x4 -> 8x8, 8x8 -> 16x16) run in scatter form (gemm_forms.hip:
// one dense GEMM, no position grid to tile) when a call has at least this many GEMM columns (samples x input positions); 0 = never.
// Default 128 (one column tile): from 2 samples at 8x8, 8 at 4x4. Measured (tools/gpu_gemm_threshold_sweep.sh, frames/s at 4 ... 32 frames
// per call): every threshold from 32 to 128 within 0.3 %, 256 / 512 -0.5 ... -1.5 %, 1 (always) -2 % at one frame per call.
static int upgemm_min_columns() {
    static const int columns = [] {
        const char* v = std::getenv("GANCE_TUNE_UPGEMM");
        return v ? std::atoi(v) : 128;
    }();
    return columns;
}

// GANCE_TUNE_UPGEMM_COLUMNS (read once per process): the scatter form's product buffer, in GEMM columns of a 512-channel layer (75 MB per 4096).
// Default 16384 (302 MB): every up layer from 4 -> 8 to 128 -> 256 of a call of up to 4 ... 8 frames fits, i.e. up to where the fused up
// kernel takes over (tools/gpu_upgemm_cap_sweep.sh: +2 ... 4 % frames/s at 2 ... 7 frames per call against 4096).
static int upgemm_buffer_columns() {
    static const int columns = [] {
        const char* v = std::getenv("GANCE_TUNE_UPGEMM_COLUMNS");
        return v ? std::max(4096, std::atoi(v)) : 16384;
    }();
    return columns;
}

// GANCE_TUNE_WINOGEMM (read once per process): the stride-1 layers at 8x8 and 16x16 run in Winograd F(4x4,3x3) GEMM form (gemm_forms.hip)
// when a call has at least this many GEMM columns (samples x 4x4 output tiles); 0 = never. Default 64: from 4 samples at 16x16, 16 at
// 8x8 (the same sweep). Like every Winograd form it is off in engines created with conv_form "direct" (or "winograd": F(2x2,3x3) only).
static int winogemm_min_columns() {
    static const int columns = [] {
        const char* v = std::getenv("GANCE_TUNE_WINOGEMM");
        return v ? std::atoi(v) : 64;
    }();
    return columns;
}

// Largest resolution whose Conv1 runs in Winograd F(4x4, 3x3) form (winograd43_conv.hip) in an engine with these
// flags: GANCE_FLAG_WINOGRAD43 = every resolution the kernel supports; otherwise the default limit, which
// GANCE_TUNE_WINO43 (read once per process: 0 = off, else a resolution) overrides.
static int wino43_max_res(int flags) {
    static const int env_value = [] {
        const char* v = std::getenv("GANCE_TUNE_WINO43");
        return v ? std::atoi(v) : -1;
    }();
    if (flags & GANCE_FLAG_DIRECT_CONV) return 0;
    if (flags & GANCE_FLAG_WINOGRAD43) return 1 << 20;
    if (flags & GANCE_FLAG_FORCE_WINOGRAD) return 0;  // FORCE_WINOGRAD alone = the F(2x2,3x3) kernels on every layer (parity tests of that form)
    return env_value >= 0 ? env_value : kWino43DefaultMaxRes;
}

// The noise conv layer li adds, and the distance between the planes of two samples: the stored buffer [res][res] shared by
// the batch (stride 0), or after gance_engine_randomize_noise the drawn planes [sample][res][res]; nullptr where the
// layer's strength is zero.
const float* layer_noise(const gance_engine* e, int li, int* b_stride) {
    *b_stride = 0;
    if (e->conv_ns[li] == 0.0f) return nullptr;
    if (e->noise_randomized && e->noise_rand != nullptr && e->noise_rand_off[li] != SIZE_MAX) {
        *b_stride = 1 << (2 * e->convs[li].res_log2);
        return e->noise_rand + e->noise_rand_off[li];
    }
    return e->pool + e->conv_noise[li];
}

int run_conv(gance_engine* e, const ConvLayerHost& c, int li, const LayerPlan& p, const float* x,
             long long x_b_stride, int H, int W, float* out, int epilogue, int out_row_stride,
             int out_y_off, int out_x_off, long long out_b_stride, long long out_c_stride,
             long long slab_stride, long long cls_stride, int B, hipStream_t stream,
             const char* name, const FusedRgb* rgb = nullptr, bool winograd = false, bool wino64 = false,
             const float* s_next = nullptr, bool wino43 = false) {
    gance::ConvArgs a{};
    a.s_next = (wino64 || wino43) ? s_next : nullptr;  // (only the 16x16x4 Winograd kernels scale their stores)
    if (epilogue == gance::kEpilogueFullRgbPart) {  // (rgb->y: the partial image; the coefficient table is the workspace's)
        a.rgb_y = rgb->y;
        a.rgb_coef = e->ws->rgb_coef;
    } else if (rgb != nullptr) {
        a.rgb_w = rgb->w;
        a.rgb_s = rgb->s;
        a.rgb_bias = rgb->bias;
        a.rgb_y_prev = rgb->y_prev;
        a.rgb_y = rgb->y;
        a.rgb_u8 = rgb->u8;
    }
    a.x = x;
    a.w = e->pool + e->conv_w[li];
    a.s = e->ws->styles + e->conv_s_off[li];
    a.d = e->ws->demod + e->conv_d_off[li];
    a.noise = layer_noise(e, li, &a.noise_b_stride);
    a.bias = e->pool + e->conv_bias[li];
    a.out = out;
    a.B = B;
    a.Cin = c.cin;
    a.Cout = c.cout;
    a.H = H;
    a.W = W;
    a.OH = p.OH;
    a.OW = p.OW;
    a.s_stride = e->ctot;
    a.d_stride = e->dtot;
    a.noise_strength = e->conv_ns[li];
    a.tiles_x = p.tiles_x;
    a.tiles_y = p.tiles_y;
    a.row_tiles = p.row_tiles;
    a.col_tiles = p.col_tiles;
    a.m_tiles = p.m_tiles;
    a.nsplit = p.nsplit;
    a.chunks_per_split = p.chunks_per_split;
    a.total_chunks = p.total_chunks;
    a.epilogue = epilogue;
    a.out_row_stride = out_row_stride;
    a.out_y_off = out_y_off;
    a.out_x_off = out_x_off;
    a.out_b_stride = out_b_stride;
    a.out_c_stride = out_c_stride;
    a.slab_stride = slab_stride;
    a.cls_stride = cls_stride;
    a.x_b_stride = x_b_stride;
    static const int debug_flags = [] { const char* v = std::getenv("GANCE_DEBUG_CONV"); return v ? std::atoi(v) : 0; }();
    a.debug_flags = debug_flags;
    // Blocks per CU of the F(4x4,3x3) launches (ConvArgs::grid_rounds). Four: measured as fast as one persistent block per
    // CU (1234 ... 1236 frames/s with 2 / 4 / 8 / 16 against 1228 with 1: the prologue of a block is a few k-steps of
    // thousands), and a CU that something else holds for a while -- the RCCL copy kernels of the frame gather on a multi-GPU
    // job -- then delays a quarter of its share instead of the tail of the launch. GANCE_TUNE_W43_ROUNDS=1: one block per CU.
    static const int env_rounds = [] { const char* v = std::getenv("GANCE_TUNE_W43_ROUNDS"); return v ? std::atoi(v) : 0; }();
    a.grid_rounds = env_rounds > 0 ? env_rounds : 4;
    // GANCE_TUNE_W43_XCD=0: the channel tile fastest in the F(4x4,3x3) launches' tile order (round 3's); default: 4 x 8 blocking per XCD
    // where the layer has 16 channel tiles
    static const int env_xcd = [] { const char* v = std::getenv("GANCE_TUNE_W43_XCD"); return v ? std::atoi(v) : 1; }();
    a.xcd_blocking = env_xcd != 0 ? 1 : 0;  // (the launcher drops it where the launch's pixel tiles are not a multiple of four)
    a.fault_flag = e->ws->fault_flag;
    static unsigned long long* stamps = nullptr;
    if (debug_flags & 16) {
        if (stamps == nullptr) hipMalloc((void**)&stamps, (size_t)5 * 8 * 65536);
        a.debug_stamps = stamps;
    }
    double flops = 2.0 * 9 * (double)c.cin * c.cout * H * W * B;
    const double out_elems = (double)B * c.cout * (c.up ? 4.0 * H * W : (double)H * W) * p.nsplit;
    double bytes = 4.0 * ((double)B * c.cin * H * W + out_elems + 9.0 * c.cin * c.cout);
    if (epilogue == gance::kEpilogueFullRgbPart) {  // + the ToRGB product and its fp32 partial image; no activation when out is null
        flops += 2.0 * 3 * (double)c.cout * H * W * B;
        bytes = 4.0 * ((double)B * c.cin * H * W + (out != nullptr ? out_elems : 0.0) + 9.0 * c.cin * c.cout + 3.0 * B * H * W);
    } else if (rgb != nullptr) {  // no activation leaves the chip: the uint8 image and the half-size skip image instead
        flops += 2.0 * 3 * (double)c.cout * H * W * B;
        bytes = 4.0 * ((double)B * c.cin * H * W + 9.0 * c.cin * c.cout + 0.75 * B * H * W) + 3.0 * B * H * W;
    }
    StepScope scope(e, stream, name, flops, bytes);
    if (wino43) {  // F(4x4, 3x3); the input arrives multiplied by this layer's style (the caller arranged that)
        a.w = e->pool + e->wino43_w[li];
        GANCE_HIP_CHECK(gance::launch_winograd43_conv(a, stream));
        return GANCE_OK;
    }
    if (winograd) {
        // the kernel on 16x16x4 MFMAs (in its 32-channel geometry the input arrives multiplied by this layer's style: the
        // caller arranged that with the producing layer), else the round-1 32-channel kernel
        if (wino64) {
            a.w = e->pool + e->wino64_w[li];
            GANCE_HIP_CHECK(gance::launch_winograd64_conv(a, stream));
            return GANCE_OK;
        }
        a.w = e->pool + e->wino_w[li];
        GANCE_HIP_CHECK(gance::launch_winograd_conv(a, stream));
        return GANCE_OK;
    }
    GANCE_HIP_CHECK(gance::launch_modconv(p.tile_id, a, p.total_blocks, stream));
    if ((debug_flags & 16) && p.total_blocks <= 65536) {
        // timing experiment: dump per-block phase stamps (100 MHz clock) of this launch
        hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)5 * p.total_blocks);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t_min = ~0ull, t_max = 0;
        double pro = 0, main_ = 0, epi = 0;
        for (int i = 0; i < p.total_blocks; ++i) {
            t_min = std::min(t_min, h[5 * i]);
            t_max = std::max(t_max, h[5 * i + 3]);
            pro += (double)(h[5 * i + 1] - h[5 * i]);
            main_ += (double)(h[5 * i + 2] - h[5 * i + 1]);
            epi += (double)(h[5 * i + 3] - h[5 * i + 2]);
        }
        const double n = p.total_blocks, span = (double)(t_max - t_min) / 100.0;
        // concurrency: sum of block lifetimes / span / 256 CUs
        const double life = (pro + main_ + epi) / 100.0;
        std::fprintf(stderr, "STAMPS %s blocks %d span %.1f us | per block: prologue %.2f us, main %.2f us, epilogue %.2f us | avg resident blocks/CU %.2f\n",
                     name, p.total_blocks, span, pro / n / 100.0, main_ / n / 100.0, epi / n / 100.0, life / span / 256.0);
        if (std::getenv("GANCE_DEBUG_DUMP") && std::strstr(name, std::getenv("GANCE_DEBUG_DUMP"))) {
            for (int i = 0; i < p.total_blocks; ++i)
                std::fprintf(stderr, "BLK %d %llu %llu %llu %llu %llx\n", i, h[5 * i] - t_min, h[5 * i + 1] - t_min, h[5 * i + 2] - t_min, h[5 * i + 3] - t_min, h[5 * i + 4]);
        }
    }
    return GANCE_OK;
}

int synthesize_from_dlat(gance_engine* e, const float* d_dlat, int B, uint8_t* d_u8, float* d_f32,
                         hipStream_t stream) {
    char name[64];
    {
        StepScope scope(e, stream, "styles", 2.0 * B * kDlatent * e->ctot,
                        4.0 * ((double)kDlatent * e->ctot + (double)B * e->ctot));
        GANCE_HIP_CHECK(gance::launch_styles(d_dlat, e->pool + e->A_off, e->pool + e->bias1_off,
                                             e->blk_row, e->ws->styles, B, e->num_rows, e->ctot,
                                             stream));
    }
    {
        StepScope scope(e, stream, "demod", 0.0, 0.0);
        GANCE_HIP_CHECK(gance::launch_demod(e->ws->styles, e->pool + e->w2_off, e->demod_layers,
                                            (int)e->convs.size(), e->ws->demod, B, e->ctot, e->dtot,
                                            stream));
    }

    int ycur = 0;  // ybuf index holding the current skip image
    bool have_y = false;
    const int num_convs = (int)e->convs.size();
    const int limit = e->debug_stop_after > 0 ? std::min(e->debug_stop_after, num_convs) : num_convs;
    const float* x_in = e->pool + e->const_off;  // zero-bordered [512][6][12], shared by the batch
    long long x_b_stride = 0;

    // Which form conv layer idx (a stride-1 conv) runs in for this batch. Decided in one place because the layer BEFORE
    // a conv on the 16x16x4 Winograd kernel has to know: that kernel takes its input multiplied by its own style.
    struct ConvForm {
        bool fused_rgb, winograd, winograd_last, wino64, wino43;
    };
    auto conv_form_of = [&](int idx, bool have_y_then) -> ConvForm {
        const ConvLayerHost& c = e->convs[idx];
        const int res = 1 << c.res_log2;
        const LayerPlan p = plan_layer(c, B);
        ConvForm form{};
        // the network's last conv absorbs its ToRGB when one block holds all channels of a pixel
        // (BM = Cout = 32, i.e. the 1024^2 generator): neither its activation nor the fp32 image is
        // written, only the uint8 frame (GANCE_TUNE_FUSE_RGB=0 turns this off)
        static const bool fuse_enabled = [] { const char* v = std::getenv("GANCE_TUNE_FUSE_RGB"); return !(v && std::atoi(v) == 0); }();
        const auto& tile = gance::kConvTiles[p.tile_id];
        form.fused_rgb = fuse_enabled && c.res_log2 == e->res_log2 && limit == num_convs && p.nsplit == 1 &&
                         p.m_tiles == 1 && tile.TB == 1 && tile.BM == 32 && have_y_then;
        // ... unless the layer runs in Winograd form on the 16x16x4 kernel's 32-channel geometry with the ToRGB product
        // in its epilogue (GANCE_TUNE_LAST_WINO64 = 0 / 1; measured: see DESIGN.md §3)
        static const int last_wino64 = [] { const char* v = std::getenv("GANCE_TUNE_LAST_WINO64"); return v ? std::atoi(v) : 1; }();
        const long long last_tiles = (long long)(res / 16) * (res / 32) * B;
        if (form.fused_rgb && last_wino64 != 0 && c.cout == 32 && e->wino64_w[idx] != SIZE_MAX && !(e->cfg.flags & GANCE_FLAG_DIRECT_CONV) &&
            !(e->cfg.flags & GANCE_FLAG_FORCE_WINOGRAD) && last_tiles >= e->num_cus)
            form.fused_rgb = false;
        // Winograd F(2x2,3x3) form where the layer supports it and the launch fills the chip
        // (one block per CU). Engine flags choose: DIRECT_CONV = never, FORCE_WINOGRAD = whatever
        // the block count; GANCE_TUNE_WINOGRAD = 0 / 1 / 2 overrides them for tuning.
        static const int env_mode = [] { const char* v = std::getenv("GANCE_TUNE_WINOGRAD"); return v ? std::atoi(v) : -1; }();
        const int wino_mode = env_mode >= 0 ? env_mode
                                            : ((e->cfg.flags & GANCE_FLAG_DIRECT_CONV) ? 0 : ((e->cfg.flags & GANCE_FLAG_FORCE_WINOGRAD) ? 2 : 1));
        // (tiles of 8 x 64 pixels, or 16 x 32 on the 32-pixel-wide layer; the kernel has no split-K)
        const long long wino_tiles = (long long)(c.cout / 32) * (res % 64 == 0 ? (res / 8) * (res / 64) : (res / 16) * (res / 32)) * B;
        form.winograd = !c.up && wino_mode != 0 && e->wino_w[idx] != SIZE_MAX && (wino_mode == 2 || (p.nsplit == 1 && wino_tiles >= 256));
        // the direct-form fused last layer stays unless Winograd is forced: the 32-channel Winograd kernel's own fused variant
        // (built, parity-green) is register-starved in its epilogue and measured no faster; GANCE_TUNE_WINOGRAD_RGB=1 selects it
        static const bool wino_rgb = [] { const char* v = std::getenv("GANCE_TUNE_WINOGRAD_RGB"); return v && std::atoi(v) != 0; }();
        form.winograd_last = form.winograd && (wino_mode == 2 || wino_rgb);
        // the kernel on 16x16x4 MFMAs (GANCE_TUNE_WINO64=0 keeps the round-1 32-channel kernel): every stride-1 conv it
        // supports that follows an up layer (all of them do: Conv1 follows Conv0_up)
        static const bool wino64_on = [] { const char* v = std::getenv("GANCE_TUNE_WINO64"); return !(v && std::atoi(v) == 0); }();
        form.wino64 = !form.fused_rgb && form.winograd && wino64_on && e->wino64_w[idx] != SIZE_MAX && idx > 0 && e->convs[idx - 1].up;
        // F(4x4, 3x3) where the layer has the weights for it (engine creation: resolution limit, geometry, an up layer in
        // front) and the launch fills the chip; never the network's last layer while that one carries the fused ToRGB
        const long long w43_tiles = (long long)(c.cout / 32) * (res >= 64 ? (res / 16) * (res / 64) : 1) * B;  // (32 x 32 pixels per tile on the 32-wide layer)
        form.wino43 = !form.fused_rgb && form.winograd && e->wino43_w[idx] != SIZE_MAX && (w43_tiles >= e->num_cus || wino_mode == 2);
        if (form.wino43) form.wino64 = false;
        return form;
    };
    // Whether up layer idx runs as the fused kernel (transposed conv + FIR in one launch): where it is supported and fills
    // the chip; GANCE_TUNE_UPFIR = 0 / 1 / 2 overrides the engine flags (never / auto / always). Decided here because the
    // layer BEFORE it has to know: fed by a 16x16x4 Winograd launch the fused kernel takes its input pre-scaled by its style.
    auto up_runs_fused = [&](int idx, gance::UpFirArgs* plan) -> bool {
        const ConvLayerHost& c = e->convs[idx];
        const int H = (1 << c.res_log2) / 2;
        static const int upfir_env = [] { const char* v = std::getenv("GANCE_TUNE_UPFIR"); return v ? std::atoi(v) : -1; }();
        const int upfir_mode = upfir_env >= 0 ? upfir_env
                                              : ((e->cfg.flags & GANCE_FLAG_SPLIT_UPFIR) ? 0 : ((e->cfg.flags & GANCE_FLAG_FORCE_FUSED_UPFIR) ? 2 : 1));
        if (!c.up || upfir_mode == 0 || (e->upfir_w[idx] == SIZE_MAX && e->upfir16_w[idx] == SIZE_MAX)) return false;
        gance::UpFirArgs u{};
        u.Cin = c.cin;
        // the split-operand form (upfir_split.hip; a block sweeps the image's height, or a row segment of it where whole images would leave
        // CUs idle: upfirs_plan): where its launch has blocks for 9/16 of the CUs
        if (e->upfirs_w[idx] != SIZE_MAX && upfir_mode != 0) {
            const bool roles = e->upfir_split_roles != 0 && gance::upfirr_supported(c.cin, c.cout, H, H);
            if (roles) gance::upfirr_plan(B, c.cout, H, H, e->num_cus, &u);
            else gance::upfirs_plan(B, c.cout, H, H, e->num_cus, &u);
            // (9/16: measured without row segments, 16 blocks per frame at every layer -- whole calls of 8 / 9 / 10 / 11 frames ran at 1053 / 842 / 909 / 940
            // frames/s in the fp32 forms, at 953 / ~1000 / 1045 / 1106 in this one; with row segments 1 ... 8 frames per call take it too wherever 16-row
            // segments reach that many blocks: 645 / 895 / 899 / 1080 / 934 / 1055 / 1136 / 1202 frames/s against 614 / 817 / 861 / 960 / - / 980 / - / 1047)
            if (e->upfir_split == 2 || (u.total_blocks >= e->num_cus * 9 / 16 && 2 * H <= e->upfir_split_max_res)) {
                u.pair_form = roles ? 3 : 2;  // (marks the plan: the caller launches launch_upfir_split_roles / launch_upfir_split)
                if (plan != nullptr) *plan = u;
                return true;
            }
            u = gance::UpFirArgs{};
            u.Cin = c.cin;
        }
        if (e->upfir16_w[idx] != SIZE_MAX)
            gance::upfir16_plan(B, c.cout, H, H, e->num_cus, &u);
        else
            gance::upfir_plan(B, c.cout, H, H, e->num_cus, &u);
        const int steps_per_seg = u.rows_per_seg / u.step_rows;
        if (plan != nullptr) *plan = u;
        // (the narrow strip geometries -- inputs 32 and 16 wide -- have one or two steps per image: never cut into segments)
        return upfir_mode == 2 || (u.total_blocks >= e->num_cus * 3 / 4 && (u.segs == 1 || steps_per_seg >= 4));
    };
    static const bool prescale_up = [] { const char* v = std::getenv("GANCE_TUNE_PRESCALE_UP"); return !(v && std::atoi(v) == 0); }();
    bool x_prescaled = false;  // x_in carries the style of the layer about to read it
    bool fused_rgb = false;
    for (int li = 0; li < limit; ++li) {
        const ConvLayerHost& c = e->convs[li];
        const int res = 1 << c.res_log2;
        const LayerPlan p = plan_layer(c, B);
        float* x_out = e->ws->act[li];
        const long long out_c = (long long)act_plane(res);
        const long long out_b = out_c * c.cout;
        int noise_b_stride = 0;
        const float* noise = layer_noise(e, li, &noise_b_stride);
        const float* bias = e->pool + e->conv_bias[li];
        bool rgb_part = false;  // this layer's conv launch also did the channel sum of its ToRGB
        int rgb_partials = 1;   // ... in this many partial images (one per channel tile of a pixel)
        if (!c.up) {
            std::snprintf(name, sizeof(name), "conv%d_%dx%d_%d->%d", c.layer_idx, res, res, c.cin,
                          c.cout);
            const ConvForm form = conv_form_of(li, have_y);
            x_prescaled = false;  // (set below where this launch scales its stores for the next layer)
            // Winograd F(4x4,3x3) as 36 dense GEMMs: at 8x8 / 16x16 always (from winogemm_min_columns() columns up), at 32x32 ... 128x128 for the
            // calls too small for the fused F(4x4,3x3) kernel (one tile per CU): one frame per call at 128x128, up to 4 at 64x64, 16 at 32x32
            const int gemm_columns = B * (res / 4) * (res / 4);
            const bool gemm_form = e->winogemm_w[li] != SIZE_MAX && gemm_columns >= winogemm_min_columns() && gemm_columns <= gance::kWinoGemmMaxColumns &&
                                   !form.wino43;
            fused_rgb = form.fused_rgb && !gemm_form;
            const bool winograd = form.winograd, winograd_last = form.winograd_last;
            if (gemm_form) {
                // ("convVG": input transform + 36 GEMMs + output transform, gemm_forms.hip)
                std::snprintf(name, sizeof(name), "convVG%d_%dx%d_%d->%d", c.layer_idx, res, res, c.cin, c.cout);
                gance::WinoGemmArgs g{};
                g.x = x_in;
                g.w = e->pool + e->winogemm_w[li];
                g.s = e->ws->styles + e->conv_s_off[li];
                g.d = e->ws->demod + e->conv_d_off[li];
                g.noise = noise;
                g.bias = bias;
                g.packed = e->ws->up_packed;
                g.prod = e->ws->up_prod;
                g.out = x_out;
                g.x_b_stride = x_b_stride;
                g.out_b_stride = out_b;
                g.noise_strength = e->conv_ns[li];
                g.noise_b_stride = noise_b_stride;
                g.B = B;
                g.Cin = c.cin;
                g.Cout = c.cout;
                g.H = res;
                g.W = res;
                g.s_stride = e->ctot;
                g.d_stride = e->dtot;
                g.n_tiles = gance::winogemm_n_tiles(B, res, res);
                g.bf16_split = c.cout % 256 == 0 ? e->gemm_bf16 : 0;  // (the split GEMM's block tiles are 256 rows: as the weights were arranged)
                const double n = (double)g.n_tiles * 128;
                StepScope scope(e, stream, name, 2.0 * 9 * c.cin * c.cout * (double)B * res * res,
                                4.0 * (36.0 * c.cin * c.cout + 2.0 * 36 * (c.cin + c.cout) * n + (double)B * (c.cin + c.cout) * res * res));
                GANCE_HIP_CHECK(gance::launch_winogemm(g, stream));
            } else if (fused_rgb) {
                const int ri = c.res_log2 - 2;
                FusedRgb rgb{e->pool + e->rgb_w[ri], e->ws->styles + e->rgb_s_off[ri], e->pool + e->rgb_bias[ri],
                             e->ws->ybuf[ycur], (d_f32 != nullptr || e->keep_skip_image) ? e->ws->ybuf[1 - ycur] : nullptr, d_u8};
                std::snprintf(name, sizeof(name), "conv%s%d+torgb_%dx%d_%d->%d", winograd_last ? "W" : "", c.layer_idx, res, res, c.cin, c.cout);
                int rc = run_conv(e, c, li, p, x_in, x_b_stride, res, res, x_out, gance::kEpilogueRgb, res + 8, 1, 4,
                                  out_b, out_c, 0, 0, B, stream, name, &rgb, winograd_last);
                if (rc) return rc;
                ycur = 1 - ycur;
            } else if (p.nsplit == 1 || winograd) {
                // Where the 64-channel Winograd kernel holds every channel of a pixel in one block (Cout = 64 at 512^2,
                // Cout = 32 at 1024^2) its epilogue also does the channel sum of the layer's ToRGB on the matrix pipe; the
                // ToRGB pass below then only adds bias and skip image (and converts). The LAST layer's activation has no
                // other reader and is not stored (unless a debug tap wants it). GANCE_TUNE_W64_RGB=0 turns this off.
                static const bool w64_rgb_enabled = [] { const char* v = std::getenv("GANCE_TUNE_W64_RGB"); return !(v && std::atoi(v) == 0); }();
                rgb_part = w64_rgb_enabled && ((form.wino64 && gance::winograd64_rgb_supported(c.cout)) || (form.wino43 && gance::winograd43_rgb_supported(c.cout)));
                if (winograd) std::snprintf(name, sizeof(name), rgb_part ? "convW%d+rgb_%dx%d_%d->%d" : "convW%d_%dx%d_%d->%d", c.layer_idx, res, res, c.cin, c.cout);
                if (form.wino43) std::snprintf(name, sizeof(name), rgb_part ? "convV%d+rgb_%dx%d_%d->%d" : "convV%d_%dx%d_%d->%d", c.layer_idx, res, res, c.cin, c.cout);
                // the next layer's style rides on this launch's stores when that layer is a fused up kernel — and only when this
                // launch also does the ToRGB channel sum (from the plain values): torgb_kernel would otherwise read the scaled ones
                const float* const s_next_up =
                    (rgb_part && (form.wino64 || form.wino43) && c.cout % 64 == 0 && prescale_up && li + 1 < limit && up_runs_fused(li + 1, nullptr))
                        ? e->ws->styles + e->conv_s_off[li + 1]
                        : nullptr;
                x_prescaled = s_next_up != nullptr;
                int rc;
                if (rgb_part) {
                    const int ri = c.res_log2 - 2;
                    GANCE_HIP_CHECK(gance::launch_winograd64_rgb_coef(e->pool + e->rgb_w[ri], e->ws->styles + e->rgb_s_off[ri], e->ctot, B, c.cout,
                                                                      e->ws->rgb_coef, stream));
                    // (one partial image: straight into the skip buffer ToRGB finishes in place; several: the workspace's)
                    rgb_partials = form.wino43 ? gance::winograd43_rgb_partials(c.cout) : gance::winograd64_rgb_partials(c.cout);
                    FusedRgb part{nullptr, nullptr, nullptr, nullptr, rgb_partials == 1 ? e->ws->ybuf[have_y ? 1 - ycur : ycur] : e->ws->rgb_part, nullptr};
                    const bool last_unread = c.res_log2 == e->res_log2 && limit == num_convs && e->debug_stop_after <= 0;
                    rc = run_conv(e, c, li, p, x_in, x_b_stride, res, res, last_unread ? nullptr : x_out, gance::kEpilogueFullRgbPart,
                                  res + 8, 1, 4, out_b, out_c, 0, 0, B, stream, name, &part, true, !form.wino43, s_next_up, form.wino43);
                } else {
                    rc = run_conv(e, c, li, p, x_in, x_b_stride, res, res, x_out,
                                  gance::kEpilogueFull, res + 8, 1, 4, out_b, out_c, 0, 0, B, stream,
                                  name, nullptr, winograd, form.wino64, s_next_up, form.wino43);
                }
                if (rc) return rc;
            } else {
                const long long dense_c = (long long)res * res;
                const long long slab = dense_c * c.cout * B;
                int rc = run_conv(e, c, li, p, x_in, x_b_stride, res, res, e->ws->slabs,
                                  gance::kEpilogueRaw, res, 0, 0, dense_c * c.cout, dense_c, slab, 0,
                                  B, stream, name);
                if (rc) return rc;
                std::snprintf(name, sizeof(name), "finish%d_%dx%d", c.layer_idx, res, res);
                StepScope scope(e, stream, name, 0.0, 4.0 * (double)slab * (p.nsplit + 1));
                GANCE_HIP_CHECK(gance::launch_splitk_finish(e->ws->slabs, slab, p.nsplit, noise,
                                                            e->conv_ns[li], noise_b_stride, bias, x_out, B, c.cout,
                                                            res, res, stream));
            }
        } else {
            const int H = res / 2, W = res / 2;
            // the next layer's style rides on this layer's activation when that layer takes its input pre-scaled
            const float* const s_next =
                (li + 1 < limit && !e->convs[li + 1].up &&
                 ((conv_form_of(li + 1, true).wino64 && gance::winograd64_input_prescaled(e->convs[li + 1].cout)) || conv_form_of(li + 1, true).wino43))
                    ? e->ws->styles + e->conv_s_off[li + 1]
                    : nullptr;
            const bool input_prescaled = x_prescaled;
            x_prescaled = false;
            {
                gance::UpFirArgs u{};
                if (up_runs_fused(li, &u)) {
                    const bool roles_form = u.pair_form == 3;
                    const bool split_form = u.pair_form == 2 || roles_form;
                    const bool geometry16 = !split_form && e->upfir16_w[li] != SIZE_MAX;
                    const bool pair_form = geometry16 && e->upfir16x_w[li] != SIZE_MAX && input_prescaled;
                    u.pair_form = pair_form ? 1 : 0;
                    u.x = x_in;
                    u.w = e->pool + (split_form ? e->upfirs_w[li] : (pair_form ? e->upfir16x_w[li] : (geometry16 ? e->upfir16_w[li] : e->upfir_w[li])));
                    u.s = e->ws->styles + e->conv_s_off[li];
                    u.d = e->ws->demod + e->conv_d_off[li];
                    u.noise = noise;
                    u.bias = bias;
                    u.out = x_out;
                    u.B = B;
                    u.Cin = c.cin;
                    u.Cout = c.cout;
                    u.H = H;
                    u.W = W;
                    u.s_stride = e->ctot;
                    u.d_stride = e->dtot;
                    u.noise_strength = e->conv_ns[li];
                    u.noise_b_stride = noise_b_stride;
                    u.x_b_stride = x_b_stride;
                    u.s_next = s_next;
                    u.input_prescaled = input_prescaled ? 1 : 0;
                    u.x_units = e->ws->x_units;
                    // ("convTFp": upfir_fused_pre_kernel, the input arrives multiplied by this layer's style)
                    // (a trailing "/16": the 16-channel, two-blocks-per-CU geometry, upfir16_fused*_kernel; "/16x": its pair form)
                    // ("/s3": the split-operand form, upfirs_fused*_kernel: bf16 x 3 parts, six product terms, fp32 accumulation)
                    // ("/s3r": the same products with the block's work in two roles, upfirr_fused*_kernel: matrix waves and vector waves)
                    std::snprintf(name, sizeof(name), input_prescaled ? "convTFp%d_%dx%d_%d->%d%s" : "convTF%d_%dx%d_%d->%d%s", c.layer_idx, res, res, c.cin,
                                  c.cout, roles_form ? "/s3r" : split_form ? "/s3" : (pair_form ? "/16x" : (geometry16 ? "/16" : "")));
                    if (roles_form) {
                        // the layer's input times its style (unless the producer multiplied it in), split into three bf16 parts per value
                        char split_name[64];
                        std::snprintf(split_name, sizeof(split_name), "split%d_%dx%d_%d", c.layer_idx, H, W, c.cin);
                        StepScope scope(e, stream, split_name, 0.0, 10.0 * (double)B * c.cin * (H + 2) * (W + 8));
                        GANCE_HIP_CHECK(gance::launch_upfirr_split_activation(x_in, x_b_stride, input_prescaled ? nullptr : u.s, u.s_stride, e->ws->x_units, B, c.cin, H, W, stream));
                    }
                    {
                        const double flops = 2.0 * 9 * (double)c.cin * c.cout * H * W * B;
                        const double bytes = 4.0 * ((double)B * c.cin * H * W + (double)B * c.cout * res * res + 9.0 * c.cin * c.cout);
                        StepScope scope(e, stream, name, flops, bytes);
                        GANCE_HIP_CHECK(roles_form ? gance::launch_upfir_split_roles(u, stream) : split_form ? gance::launch_upfir_split(u, stream)
                                                   : (geometry16 ? gance::launch_upfir16_fused(u, stream) : gance::launch_upfir_fused(u, stream)));
                    }
                    x_in = x_out;
                    x_b_stride = out_b;
                    e->last_act_layer = li;
                    e->last_act_c = c.cout;
                    e->last_act_side = res;
                    continue;
                }
            }
            const long long tc = (long long)t_plane(H);
            const long long unit = tc * c.cout;
            const long long cls_stride = unit * e->t_units[li];
            // (the scatter form: at 4x4 / 8x8 inputs from upgemm_min_columns() columns up, at 32x32 / 64x64 inputs for calls this small)
            const bool scatter = e->upgemm_w[li] != SIZE_MAX && B * H * W >= upgemm_min_columns() && B * H * W <= gance::upgemm_max_columns(c.cout, upgemm_buffer_columns());
            if (scatter) {
                // ("convTG": pack + GEMM + gather, gemm_forms.hip)
                std::snprintf(name, sizeof(name), "convTG%d_%dx%d_%d->%d", c.layer_idx, res, res, c.cin, c.cout);
                gance::UpGemmArgs g{};
                g.x = x_in;
                g.w = e->pool + e->upgemm_w[li];
                g.s = e->ws->styles + e->conv_s_off[li];
                g.d = e->ws->demod + e->conv_d_off[li];
                g.packed = e->ws->up_packed;
                g.prod = e->ws->up_prod;
                g.t = e->ws->tplanes[li];
                g.x_b_stride = x_b_stride;
                g.cls_stride = cls_stride;
                g.unit_stride = unit;
                g.B = B;
                g.Cin = c.cin;
                g.Cout = c.cout;
                g.H = H;
                g.W = W;
                g.s_stride = e->ctot;
                g.d_stride = e->dtot;
                g.n_tiles = gance::upgemm_n_tiles(B, H, W);
                g.bf16_split = (9 * c.cout) % 256 == 0 ? e->gemm_bf16 : 0;
                const double n = (double)g.n_tiles * 128;
                StepScope scope(e, stream, name, 2.0 * 9 * c.cin * c.cout * (double)B * H * W,
                                4.0 * (9.0 * c.cin * c.cout + 2.0 * c.cin * n + 2.0 * 9 * c.cout * n + 4.0 * unit * B));
                GANCE_HIP_CHECK(gance::launch_upgemm(g, stream));
            } else {
                std::snprintf(name, sizeof(name), "convT%d_%dx%d_%d->%d", c.layer_idx, res, res, c.cin,
                              c.cout);
                int rc = run_conv(e, c, li, p, x_in, x_b_stride, H, W, e->ws->tplanes[li],
                                  gance::kEpilogueRaw, W + 8, 1, 4, unit, tc, unit * B, cls_stride, B,
                                  stream, name);
                if (rc) return rc;
            }
            gance::FirArgs f{};
            f.t = e->ws->tplanes[li];
            f.cls_stride = cls_stride;
            f.unit_stride = unit;
            f.noise = noise;
            f.bias = bias;
            f.out = x_out;
            f.noise_strength = e->conv_ns[li];
            f.noise_b_stride = noise_b_stride;
            f.B = B;
            f.C = c.cout;
            f.H = H;
            f.W = W;
            f.nsplit = scatter ? 1 : p.nsplit;
            f.s_next = s_next;
            f.s_next_stride = e->ctot;
            std::snprintf(name, sizeof(name), "fir%d_%dx%d", c.layer_idx, res, res);
            StepScope scope(e, stream, name, 0.0,
                            4.0 * (double)B * c.cout * res * res * (p.nsplit + 1));
            GANCE_HIP_CHECK(gance::launch_fir_epilogue(f, stream));
        }
        x_in = x_out;
        x_b_stride = out_b;
        e->last_act_layer = li;
        e->last_act_c = c.cout;
        e->last_act_side = res;

        // ToRGB after the 4x4 conv and after every Conv1
        if (!c.up && !fused_rgb) {
            int ri = c.res_log2 - 2;
            const RgbLayerHost& r = e->rgbs[ri];
            gance::ToRgbArgs t{};
            t.x = x_in;
            t.w = e->pool + e->rgb_w[ri];
            t.s = e->ws->styles + e->rgb_s_off[ri];
            t.bias = e->pool + e->rgb_bias[ri];
            t.y_prev = have_y ? e->ws->ybuf[ycur] : nullptr;
            t.y = e->ws->ybuf[have_y ? 1 - ycur : ycur];
            const bool last = (c.res_log2 == e->res_log2);
            t.u8 = last ? d_u8 : nullptr;
            t.partial = rgb_part ? (rgb_partials == 1 ? t.y : e->ws->rgb_part) : nullptr;  // (one piece: in place)
            t.partials = rgb_partials;
            t.skip_y_store = last && res > 128 && d_u8 != nullptr && d_f32 == nullptr && !e->keep_skip_image && limit == num_convs;
            t.B = B;
            t.Cin = r.cin;
            t.R = res;
            t.s_stride = e->ctot;
            std::snprintf(name, sizeof(name), "torgb_%dx%d", res, res);
            // (after a conv launch that did the channel sum: partial image in, bias and skip image added, image and/or bytes out)
            const double px = (double)B * res * res;
            StepScope scope(e, stream, name, rgb_part ? 0.0 : 2.0 * 3 * (double)r.cin * px,
                            rgb_part ? px * (12.0 * rgb_partials + 3.0 + (t.skip_y_store ? 0.0 : 12.0) + (t.u8 != nullptr ? 3.0 : 0.0))
                                     : 4.0 * px * (r.cin + 3 + 0.75) + 3.0 * px);
            GANCE_HIP_CHECK(gance::launch_torgb(t, stream));
            if (have_y) ycur = 1 - ycur;
            have_y = true;
        }
    }
    if (d_f32 != nullptr && limit == num_convs) {
        const size_t bytes = (size_t)B * 3 * e->cfg.resolution * e->cfg.resolution * sizeof(float);
        GANCE_HIP_CHECK(hipMemcpyAsync(d_f32, e->ws->ybuf[ycur], bytes, hipMemcpyDeviceToDevice, stream));
    }
    e->last_stream = stream;
    return GANCE_OK;
}

// the workspaces alive in this process, by (device, resolution, max_batch)
std::mutex g_workspace_mutex;
std::map<std::tuple<int, int, int>, std::weak_ptr<gance_workspace>> g_workspaces;

int acquire_workspace(gance_engine* e) {
    const bool shared = !(e->cfg.flags & GANCE_FLAG_PRIVATE_WORKSPACE);
    const auto key = std::make_tuple((int)e->cfg.device, (int)e->cfg.resolution, (int)e->cfg.max_batch);
    std::lock_guard<std::mutex> lock(g_workspace_mutex);
    if (shared) {
        auto it = g_workspaces.find(key);
        if (it != g_workspaces.end())
            if (auto alive = it->second.lock()) {
                e->ws = alive;
                // (a workspace made by an engine without the role-split experiment has no room for its input image: this engine goes without it too)
                if (e->x_units_bytes > alive->x_units_bytes) e->upfir_split_roles = 0;
                return GANCE_OK;
            }
    }
    auto ws = std::make_shared<gance_workspace>();
    ws->device = e->cfg.device;
    ws->resolution = e->cfg.resolution;
    ws->max_batch = e->cfg.max_batch;
    const int nconv = (int)e->convs.size(), Bmax = e->cfg.max_batch;
    ws->act.assign(nconv, nullptr);
    ws->tplanes.assign(nconv, nullptr);
    auto alloc = [&](void** ptr, size_t bytes, bool zero) {
        hipError_t err = hipMalloc(ptr, bytes);
        if (err == hipSuccess && zero) err = hipMemset(*ptr, 0, bytes);
        if (err != hipSuccess) {
            fail(err == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY : GANCE_ERR_HIP,
                 std::string("workspace allocation of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(err));
            return false;
        }
        ws->bytes += bytes;
        return true;
    };
    bool ok = alloc((void**)&ws->dlat, (size_t)Bmax * e->num_rows * kDlatent * sizeof(float), false) &&
              alloc((void**)&ws->map_a, (size_t)Bmax * kDlatent * sizeof(float), false) &&
              alloc((void**)&ws->map_b_buf, (size_t)Bmax * kDlatent * sizeof(float), false) &&
              alloc((void**)&ws->z_in, (size_t)Bmax * kDlatent * sizeof(float), false) &&
              alloc((void**)&ws->styles, (size_t)Bmax * e->ctot * sizeof(float), false) &&
              alloc((void**)&ws->demod, (size_t)Bmax * e->dtot * sizeof(float), false);
    for (int i = 0; ok && i < nconv; ++i) {
        // every layer owns its zero-bordered output (and parity planes): kernels only ever write
        // interiors, so the borders are zeroed exactly once, here
        const ConvLayerHost& c = e->convs[i];
        ok = alloc((void**)&ws->act[i], (size_t)Bmax * c.cout * act_plane(1 << c.res_log2) * sizeof(float), true);
        if (ok && c.up)
            ok = alloc((void**)&ws->tplanes[i], (size_t)4 * e->t_units[i] * c.cout * t_plane((1 << c.res_log2) / 2) * sizeof(float), true);
    }
    ok = ok && alloc((void**)&ws->x_units, std::max<size_t>(16, e->x_units_bytes), false);
    ws->x_units_bytes = e->x_units_bytes;
    ok = ok && alloc((void**)&ws->up_packed, std::max<size_t>(1, e->up_packed_floats) * sizeof(float), false) &&
         alloc((void**)&ws->up_prod, std::max<size_t>(1, e->up_prod_floats) * sizeof(float), false);
    ok = ok && alloc((void**)&ws->slabs, e->slab_floats * sizeof(float), false) &&
         alloc((void**)&ws->ybuf[0], e->y_floats * sizeof(float), false) && alloc((void**)&ws->ybuf[1], e->y_floats * sizeof(float), false) &&
         alloc((void**)&ws->rgb_coef, (size_t)Bmax * 8 * 16 * 64 * sizeof(float), false) &&
         alloc((void**)&ws->rgb_part, std::max<size_t>(1, e->rgb_part_floats) * sizeof(float), false) && alloc((void**)&ws->u8buf, e->y_floats, false);
    if (ok && (hipEventCreateWithFlags(&ws->last_use, hipEventDisableTiming) != hipSuccess ||
               hipStreamCreateWithFlags(&ws->host_stream, hipStreamNonBlocking) != hipSuccess ||
               hipHostMalloc((void**)&ws->fault_flag, sizeof(int), hipHostMallocMapped) != hipSuccess ||
               hipHostMalloc((void**)&ws->pinned_in, (size_t)Bmax * e->num_rows * kDlatent * sizeof(float), hipHostMallocDefault) != hipSuccess ||
               hipHostMalloc((void**)&ws->pinned_out, e->y_floats, hipHostMallocDefault) != hipSuccess)) {
        fail(GANCE_ERR_HIP, "stream / event / pinned staging creation failed");
        ok = false;
    }
    if (!ok) return GANCE_ERR_OUT_OF_MEMORY;  // (the partial workspace frees itself; the message is already recorded)
    *ws->fault_flag = 0;
    if (shared) g_workspaces[key] = ws;
    e->ws = ws;
    return GANCE_OK;
}

// calls that share a workspace run one after the other, whatever streams they were given
struct WorkspaceTurn {
    gance_workspace* ws;
    hipStream_t stream;
    WorkspaceTurn(gance_workspace* w, hipStream_t s) : ws(w), stream(s) {
        if (ws->used) hipStreamWaitEvent(stream, ws->last_use, 0);
    }
    ~WorkspaceTurn() {
        hipEventRecord(ws->last_use, stream);
        ws->used = true;
    }
};

int check_call(gance_engine* e, const void* in, int batch) {
    if (e == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (in == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "input pointer is NULL");
    if (batch < 1 || batch > e->cfg.max_batch)
        return fail(GANCE_ERR_INVALID_ARGUMENT,
                    "batch " + std::to_string(batch) + " outside [1, max_batch=" +
                        std::to_string(e->cfg.max_batch) + "]");
    if (e->ws && e->ws->fault_flag != nullptr && *(volatile int*)e->ws->fault_flag != 0) {
        // (sticky: the frames of the call that raised it, and of any call queued behind it, are invalid)
        return fail(GANCE_ERR_HIP, "a two-waves-per-SIMD kernel (code " + std::to_string(*(volatile int*)e->ws->fault_flag) +
                                       ") found its waves placed otherwise than its roles assume: frames since the previous successful call are "
                                       "invalid; rebuild libgance_hip.so with tools/check_w43_isa.py passing, or run with conv_form=\"winograd\"");
    }
    if (e->noise_randomized && e->noise_rand != nullptr && batch > e->noise_rand_count)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "gance_engine_randomize_noise drew noise for " + std::to_string(e->noise_rand_count) +
                                                    " samples, this call has " + std::to_string(batch));
    return GANCE_OK;
}

}  // namespace

extern "C" {

const char* gance_last_error(void) { return g_last_error.c_str(); }
int gance_abi_version(void) { return GANCE_ABI_VERSION; }

uint64_t gance_weight_blob_floats(int32_t resolution) {
    const int l = ilog2_exact(resolution);
    if (l < 3 || l > 10) return 0;
    return blob_floats(l);
}

int gance_engine_create(const gance_engine_config* config, const float* host_weights,
                        uint64_t num_floats, gance_engine** out_engine) {
    if (config == nullptr || host_weights == nullptr || out_engine == nullptr)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument to gance_engine_create");
    *out_engine = nullptr;
    const int res_log2 = ilog2_exact(config->resolution);
    if (res_log2 < 3 || res_log2 > 10)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "resolution must be a power of two in [8, 1024]");
    if (config->max_batch < 1 || config->max_batch > 64)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "max_batch must be in [1, 64]");
    if (num_floats != blob_floats(res_log2))
        return fail(GANCE_ERR_BAD_WEIGHTS,
                    "weight blob has " + std::to_string(num_floats) + " floats, expected " +
                        std::to_string(blob_floats(res_log2)));
    int device_count = 0;
    const hipError_t count_err = hipGetDeviceCount(&device_count);
    if (count_err != hipSuccess || device_count < 1)
        return fail(GANCE_ERR_NO_DEVICE,
                    std::string("no HIP device visible (hipGetDeviceCount: ") +
                        hipGetErrorString(count_err) + ", count " + std::to_string(device_count) +
                        "); libgance_hip has no CPU path");
    if (config->device < 0 || config->device >= device_count)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    gance::DeviceGuard guard(config->device);  // the caller's current device is restored on return
    GANCE_HIP_CHECK(guard.status());
    int num_cus = 0;
    GANCE_HIP_CHECK(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, config->device));

    gance_engine* e = new gance_engine();
    {  // (read per engine, not once per process: a test creates engines with and without it)
        const char* v = std::getenv("GANCE_TUNE_GEMM_BF16X6");
        e->gemm_bf16 = v != nullptr ? std::max(0, std::min(2, std::atoi(v))) : 0;
        const char* const split = std::getenv("GANCE_TUNE_UPFIR_SPLIT");
        e->upfir_split = split != nullptr ? std::max(0, std::min(2, std::atoi(split))) : 1;
        const char* const roles = std::getenv("GANCE_TUNE_UPFIR_SPLIT_ROLES");
        if (roles != nullptr) e->upfir_split_roles = std::atoi(roles) != 0 ? 1 : 0;
        const char* const split_res = std::getenv("GANCE_TUNE_UPFIR_SPLIT_MAXRES");
        if (split_res != nullptr) e->upfir_split_max_res = std::atoi(split_res);
    }
    e->cfg = *config;
    e->num_cus = num_cus > 0 ? num_cus : 256;
    e->res_log2 = res_log2;
    e->num_rows = res_log2 * 2 - 2;
    build_spec(res_log2, &e->convs, &e->rgbs);
    const int nconv = (int)e->convs.size();
    const int nrgb = (int)e->rgbs.size();

    // ---- style / demod column layout ----
    e->conv_s_off.resize(nconv);
    e->conv_d_off.resize(nconv);
    e->rgb_s_off.resize(nrgb);
    int ctot = 0, dtot = 0;
    for (int i = 0; i < nconv; ++i) {
        e->conv_s_off[i] = ctot;
        ctot += e->convs[i].cin;
        e->conv_d_off[i] = dtot;
        dtot += e->convs[i].cout;
    }
    for (int i = 0; i < nrgb; ++i) {
        e->rgb_s_off[i] = ctot;
        ctot += e->rgbs[i].cin;
    }
    e->ctot = ctot;
    e->dtot = dtot;

    // ---- build the processed weight pool on the host ----
    std::vector<float> pool;
    auto reserve = [&](size_t n) {
        const size_t off = pool.size();
        pool.resize(off + ((n + 3) & ~(size_t)3), 0.f);  // keep every array 16-B aligned
        return off;
    };
    const float* src = host_weights;
    // mapping
    const float map_coef = (float)(1.0 / std::sqrt((double)kDlatent) * kMappingLrmul);
    for (int i = 0; i < kMappingLayers; ++i) {
        e->map_w[i] = reserve((size_t)kDlatent * kDlatent);
        for (size_t k = 0; k < (size_t)kDlatent * kDlatent; ++k) pool[e->map_w[i] + k] = src[k] * map_coef;
        src += (size_t)kDlatent * kDlatent;
        e->map_b[i] = reserve(kDlatent);
        for (int k = 0; k < kDlatent; ++k) pool[e->map_b[i] + k] = src[k] * kMappingLrmul;
        src += kDlatent;
    }
    e->avg_off = reserve(kDlatent);
    std::memcpy(&pool[e->avg_off], src, kDlatent * sizeof(float));
    src += kDlatent;
    e->const_off = reserve((size_t)nf(1) * act_plane(4));  // zero-bordered [512][6][12]
    for (int ch = 0; ch < nf(1); ++ch)
        for (int y = 0; y < 4; ++y)
            for (int x = 0; x < 4; ++x)
                pool[e->const_off + (size_t)ch * act_plane(4) + (size_t)(y + 1) * 12 + x + 4] =
                    src[(size_t)ch * 16 + y * 4 + x];
    src += (size_t)nf(1) * 16;

    e->A_off = reserve((size_t)kDlatent * ctot);
    e->bias1_off = reserve(ctot);
    size_t w2_total = 0;
    for (const auto& c : e->convs) w2_total += (size_t)c.cin * c.cout;
    e->w2_off = reserve(w2_total);
    const float mod_coef = (float)(1.0 / std::sqrt((double)kDlatent));
    std::vector<gance::DemodLayer> demod_layers(nconv);
    e->conv_w.resize(nconv);
    e->conv_bias.resize(nconv);
    e->conv_noise.resize(nconv);
    e->conv_ns.resize(nconv);
    size_t w2_cursor = 0;
    for (int i = 0; i < nconv; ++i) {
        const ConvLayerHost& c = e->convs[i];
        const size_t wn = (size_t)9 * c.cin * c.cout;
        const float coef = (float)(1.0 / std::sqrt(9.0 * c.cin));
        e->conv_w[i] = reserve(wn);
        {
            // scaled HWIO weights, re-laid-out as the kernel's LDS image:
            // [m tile][K chunk][tap slot][KC][BM], slot t of an up layer = filter tap kUpTapWeight[t]
            const int BM = layer_bm(c.cout), KC = layer_kc(c.cout, c.up);
            const int m_tiles = c.cout / BM, chunks = c.cin / KC;
            float* w = &pool[e->conv_w[i]];
            float* w2 = &pool[e->w2_off + w2_cursor];
            for (int mt = 0; mt < m_tiles; ++mt)
                for (int ch = 0; ch < chunks; ++ch)
                    for (int t = 0; t < 9; ++t) {
                        const int tap = c.up ? kUpTapWeight[t] : t;
                        for (int kc = 0; kc < KC; ++kc)
                            for (int m = 0; m < BM; ++m) {
                                const int ci = ch * KC + kc, co = mt * BM + m;
                                const float v = src[((size_t)tap * c.cin + ci) * c.cout + co] * coef;
                                w[((((size_t)mt * chunks + ch) * 9 + t) * KC + kc) * BM + m] = v;
                                w2[(size_t)ci * c.cout + co] += v * v;
                            }
                    }
        }
        e->wino_w.push_back(SIZE_MAX);
        if (!c.up && gance::winograd_supported(c.cin, c.cout, 1 << c.res_log2, 1 << c.res_log2)) {
            // U = G w G^T of the scaled weights, in the Winograd kernel's LDS image [m tile][chunk][16][4][32]
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->wino_w[i] = reserve(gance::winograd_weight_floats(c.cin, c.cout));
            gance::winograd_transform_weights(scaled.data(), c.cin, c.cout, &pool[e->wino_w[i]]);
        }
        e->wino64_w.push_back(SIZE_MAX);
        if (!c.up && gance::winograd64_supported(c.cin, c.cout, 1 << c.res_log2, 1 << c.res_log2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->wino64_w[i] = reserve(gance::winograd64_weight_floats(c.cin, c.cout));
            gance::winograd64_transform_weights(scaled.data(), c.cin, c.cout, &pool[e->wino64_w[i]]);
        }
        e->wino43_w.push_back(SIZE_MAX);
        if (!c.up && i > 0 && e->convs[i - 1].up && (1 << c.res_log2) >= 32 && (1 << c.res_log2) <= wino43_max_res(e->cfg.flags) &&
            gance::winograd43_supported(c.cin, c.cout, 1 << c.res_log2, 1 << c.res_log2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->wino43_w[i] = reserve(gance::winograd43_weight_floats(c.cin, c.cout));
            gance::winograd43_transform_weights(scaled.data(), c.cin, c.cout, &pool[e->wino43_w[i]]);
        }
        e->upfir_w.push_back(SIZE_MAX);
        // (the 32-channel kernel's image only where the 16-channel kernel will not take the layer: GANCE_TUNE_UPFIR16=0)
        if (c.up && gance::upfir_supported(c.cin, c.cout, (1 << c.res_log2) / 2, (1 << c.res_log2) / 2) &&
            !(upfir16_mode() != 0 && gance::upfir16_supported(c.cin, c.cout, (1 << c.res_log2) / 2, (1 << c.res_log2) / 2))) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->upfir_w[i] = reserve(gance::upfir_weight_floats(c.cin, c.cout));
            gance::upfir_arrange_weights(scaled.data(), c.cin, c.cout, kUpTapWeight, &pool[e->upfir_w[i]]);
        }
        e->upfir16_w.push_back(SIZE_MAX);
        if (c.up && upfir16_mode() != 0 && gance::upfir16_supported(c.cin, c.cout, (1 << c.res_log2) / 2, (1 << c.res_log2) / 2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->upfir16_w[i] = reserve(gance::upfir16_weight_floats(c.cin, c.cout));
            gance::upfir16_arrange_weights(scaled.data(), c.cin, c.cout, kUpTapWeight, &pool[e->upfir16_w[i]]);
        }
        e->winogemm_w.push_back(SIZE_MAX);
        if (!c.up && i > 0 && winogemm_min_columns() > 0 && wino43_max_res(e->cfg.flags) >= 16 &&
            gance::winogemm_supported(c.cin, c.cout, 1 << c.res_log2, 1 << c.res_log2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            // (the experiment's 256-row block tiles need Cout to be a multiple of 256: a 128-channel layer of a reduced network keeps the fp32 GEMM)
            const int split_mode = c.cout % 256 == 0 ? e->gemm_bf16 : 0;
            e->winogemm_w[i] = reserve(gance::winogemm_weight_floats(c.cin, c.cout) * (split_mode ? 3 : 2) / 2);
            if (split_mode) gance::winogemm_arrange_weights_split(scaled.data(), c.cin, c.cout, split_mode, &pool[e->winogemm_w[i]]);
            else gance::winogemm_arrange_weights(scaled.data(), c.cin, c.cout, &pool[e->winogemm_w[i]]);
        }
        e->upgemm_w.push_back(SIZE_MAX);
        if (c.up && upgemm_min_columns() > 0 && gance::upgemm_supported(c.cin, c.cout, (1 << c.res_log2) / 2, (1 << c.res_log2) / 2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            // (the experiment's 256-row block tiles need 9 Cout to be a multiple of 256: the 128-channel layer keeps the fp32 GEMM)
            const int split_mode = (9 * c.cout) % 256 == 0 ? e->gemm_bf16 : 0;
            e->upgemm_w[i] = reserve(gance::upgemm_weight_floats(c.cin, c.cout) * (split_mode ? 3 : 2) / 2);
            if (split_mode) gance::upgemm_arrange_weights_split(scaled.data(), c.cin, c.cout, kUpTapWeight, split_mode, &pool[e->upgemm_w[i]]);
            else gance::upgemm_arrange_weights(scaled.data(), c.cin, c.cout, kUpTapWeight, &pool[e->upgemm_w[i]]);
        }
        e->upfirs_w.push_back(SIZE_MAX);
        if (c.up && e->upfir_split != 0 && upfir16_mode() != 0 && gance::upfirs_supported(c.cin, c.cout, (1 << c.res_log2) / 2, (1 << c.res_log2) / 2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->upfirs_w[i] = reserve(gance::upfirs_weight_floats(c.cin, c.cout));
            gance::upfirs_arrange_weights(scaled.data(), c.cin, c.cout, kUpTapWeight, &pool[e->upfirs_w[i]]);
        }
        e->upfir16x_w.push_back(SIZE_MAX);
        if (c.up && upfir16_mode() != 0 && upfir16x_mode() != 0 && gance::upfir16x_supported(c.cin, c.cout, (1 << c.res_log2) / 2, (1 << c.res_log2) / 2)) {
            std::vector<float> scaled(wn);
            for (size_t j = 0; j < wn; ++j) scaled[j] = src[j] * coef;
            e->upfir16x_w[i] = reserve(gance::upfir16x_weight_floats(c.cin, c.cout));
            gance::upfir16x_arrange_weights(scaled.data(), c.cin, c.cout, kUpTapWeight, &pool[e->upfir16x_w[i]]);
        }
        src += wn;
        demod_layers[i] = {(long long)w2_cursor, e->conv_s_off[i], e->conv_d_off[i], c.cin, c.cout};
        w2_cursor += (size_t)c.cin * c.cout;
        // mod_weight [512][cin] -> A[k][s_off + ci]
        for (int k = 0; k < kDlatent; ++k)
            for (int ci = 0; ci < c.cin; ++ci)
                pool[e->A_off + (size_t)k * ctot + e->conv_s_off[i] + ci] =
                    src[(size_t)k * c.cin + ci] * mod_coef;
        src += (size_t)kDlatent * c.cin;
        for (int ci = 0; ci < c.cin; ++ci)
            pool[e->bias1_off + e->conv_s_off[i] + ci] = src[ci] + 1.0f;
        src += c.cin;
        e->conv_ns[i] = src[0];
        src += 1;
        e->conv_bias[i] = reserve(c.cout);
        std::memcpy(&pool[e->conv_bias[i]], src, c.cout * sizeof(float));
        src += c.cout;
    }
    e->rgb_w.resize(nrgb);
    e->rgb_bias.resize(nrgb);
    for (int i = 0; i < nrgb; ++i) {
        const RgbLayerHost& r = e->rgbs[i];
        const float coef = (float)(1.0 / std::sqrt((double)r.cin));
        e->rgb_w[i] = reserve((size_t)r.cin * 3);
        for (int k = 0; k < r.cin * 3; ++k) pool[e->rgb_w[i] + k] = src[k] * coef;
        src += (size_t)r.cin * 3;
        for (int k = 0; k < kDlatent; ++k)
            for (int ci = 0; ci < r.cin; ++ci)
                pool[e->A_off + (size_t)k * ctot + e->rgb_s_off[i] + ci] =
                    src[(size_t)k * r.cin + ci] * mod_coef;
        src += (size_t)kDlatent * r.cin;
        for (int ci = 0; ci < r.cin; ++ci)
            pool[e->bias1_off + e->rgb_s_off[i] + ci] = src[ci] + 1.0f;
        src += r.cin;
        e->rgb_bias[i] = reserve(3);
        std::memcpy(&pool[e->rgb_bias[i]], src, 3 * sizeof(float));
        src += 3;
    }
    for (int i = 0; i < nconv; ++i) {
        const size_t n = (size_t)1 << (2 * e->convs[i].res_log2);
        e->conv_noise[i] = reserve(n);
        std::memcpy(&pool[e->conv_noise[i]], src, n * sizeof(float));
        src += n;
    }
    if ((uint64_t)(src - host_weights) != num_floats) {
        delete e;
        return fail(GANCE_ERR_BAD_WEIGHTS, "internal: blob walk does not match blob size");
    }

    // dlatent row of every 32-column style block
    std::vector<int> blk_row(ctot / 32);
    for (int i = 0; i < nconv; ++i)
        for (int cb = e->conv_s_off[i] / 32; cb < (e->conv_s_off[i] + e->convs[i].cin) / 32; ++cb)
            blk_row[cb] = e->convs[i].layer_idx;
    for (int i = 0; i < nrgb; ++i)
        for (int cb = e->rgb_s_off[i] / 32; cb < (e->rgb_s_off[i] + e->rgbs[i].cin) / 32; ++cb)
            blk_row[cb] = e->rgbs[i].row;

    // ---- workspace sizes ----
    const int Bmax = config->max_batch;
    e->t_units.assign(nconv, 0);
    size_t slab_max = 4;
    for (int i = 0; i < nconv; ++i) {
        const ConvLayerHost& c = e->convs[i];
        for (int B = 1; B <= Bmax; ++B) {
            const LayerPlan p = plan_layer(c, B);
            if (c.up) e->t_units[i] = std::max(e->t_units[i], p.nsplit * B);
            else if (p.nsplit > 1)
                slab_max = std::max(slab_max, (size_t)p.nsplit * B * c.cout << (2 * c.res_log2));
        }
    }
    e->slab_floats = slab_max;
    // the scatter-form up layers' GEMM buffers (whatever this engine's knobs say: the workspace is shared)
    for (int i = 0; i < nconv; ++i) {
        const ConvLayerHost& c = e->convs[i];
        const int H = (1 << c.res_log2) / 2;
        if (c.up && gance::upgemm_supported(c.cin, c.cout, H, H)) {
            // (x 3/2: room for the three bf16 parts of the experiment's operand images, whatever this engine's knobs say)
            const int samples = std::max(1, std::min(Bmax, gance::upgemm_max_columns(c.cout, upgemm_buffer_columns()) / (H * H)));
            e->up_packed_floats = std::max(e->up_packed_floats, gance::upgemm_packed_floats(samples, c.cin, H, H) * 3 / 2);
            e->up_prod_floats = std::max(e->up_prod_floats, gance::upgemm_prod_floats(samples, c.cout, H, H));
        }
        if (!c.up && i > 0 && gance::winogemm_supported(c.cin, c.cout, 2 * H, 2 * H)) {
            const int tiles = (H / 2) * (H / 2);  // 4x4 output tiles per sample
            const int samples = std::max(1, std::min(Bmax, gance::kWinoGemmMaxColumns / tiles));
            e->up_packed_floats = std::max(e->up_packed_floats, gance::winogemm_packed_floats(samples, c.cin, 2 * H, 2 * H) * 3 / 2);
            e->up_prod_floats = std::max(e->up_prod_floats, gance::winogemm_prod_floats(samples, c.cout, 2 * H, 2 * H));
        }
    }
    // the split input image of the up layers that can take the role-split form (whatever this engine's knobs say: the workspace is shared)
    for (int i = 0; i < nconv; ++i) {
        const ConvLayerHost& c = e->convs[i];
        const int H = (1 << c.res_log2) / 2;
        if (c.up && e->upfir_split_roles != 0 && gance::upfirr_supported(c.cin, c.cout, H, H)) e->x_units_bytes = std::max(e->x_units_bytes, gance::upfirr_units_bytes(Bmax, c.cin, H, H));
    }
    e->y_floats = (size_t)3 * config->resolution * config->resolution * Bmax;
    // partial ToRGB images of the Winograd conv launches whose pixels span several channel tiles: [Cout / 64][Bmax][3][R][R]
    for (int i = 0; i < nconv; ++i) {
        const ConvLayerHost& c = e->convs[i];
        if (!c.up && c.cout > 64 && gance::winograd64_rgb_supported(c.cout))
            e->rgb_part_floats = std::max(e->rgb_part_floats, (size_t)gance::winograd64_rgb_partials(c.cout) * Bmax * 3 << (2 * c.res_log2));
        // ... and of the F(4x4,3x3) launches: one partial image per block of 32 channels. Whatever THIS engine's flags say
        // about that form: the buffer belongs to the workspace, which every engine of the (device, resolution, max_batch)
        // shares -- an engine that never runs the form may be the one that allocates it for one that does.
        if (!c.up && i > 0 && e->convs[i - 1].up && (1 << c.res_log2) >= 32 &&
            gance::winograd43_supported(c.cin, c.cout, 1 << c.res_log2, 1 << c.res_log2) && gance::winograd43_rgb_supported(c.cout))
            e->rgb_part_floats = std::max(e->rgb_part_floats, (size_t)gance::winograd43_rgb_partials(c.cout) * Bmax * 3 << (2 * c.res_log2));
    }

#define GANCE_CREATE_CHECK(expr)                                                            \
    do {                                                                                    \
        hipError_t gance_err_ = (expr);                                                     \
        if (gance_err_ != hipSuccess) {                                                     \
            free_engine(e);                                                                 \
            return fail(gance_err_ == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY         \
                                                          : GANCE_ERR_HIP,                  \
                        std::string(#expr) + ": " + hipGetErrorString(gance_err_));         \
        }                                                                                   \
    } while (0)

    e->pool_floats = pool.size();
    GANCE_CREATE_CHECK(hipMalloc((void**)&e->pool, pool.size() * sizeof(float)));
    GANCE_CREATE_CHECK(hipMemcpy(e->pool, pool.data(), pool.size() * sizeof(float), hipMemcpyHostToDevice));
    GANCE_CREATE_CHECK(hipMalloc((void**)&e->blk_row, blk_row.size() * sizeof(int)));
    GANCE_CREATE_CHECK(hipMemcpy(e->blk_row, blk_row.data(), blk_row.size() * sizeof(int), hipMemcpyHostToDevice));
    GANCE_CREATE_CHECK(hipMalloc((void**)&e->demod_layers, demod_layers.size() * sizeof(gance::DemodLayer)));
    GANCE_CREATE_CHECK(hipMemcpy(e->demod_layers, demod_layers.data(),
                                 demod_layers.size() * sizeof(gance::DemodLayer), hipMemcpyHostToDevice));
#undef GANCE_CREATE_CHECK
    if (int rc = acquire_workspace(e)) {
        const std::string message = g_last_error;
        free_engine(e);
        return fail(rc, message);
    }
    *out_engine = e;
    return GANCE_OK;
}

void gance_engine_destroy(gance_engine* engine) { free_engine(engine); }

int32_t gance_engine_vector_length(const gance_engine* engine) { return engine ? kDlatent : 0; }
int32_t gance_engine_num_layers(const gance_engine* engine) { return engine ? engine->num_rows : 0; }
int32_t gance_engine_resolution(const gance_engine* engine) { return engine ? engine->cfg.resolution : 0; }
int32_t gance_engine_max_batch(const gance_engine* engine) { return engine ? engine->cfg.max_batch : 0; }

int gance_synthesize_w(gance_engine* engine, const float* d_dlatents, int32_t batch,
                       uint8_t* d_out_u8, float* d_out_f32, void* stream) {
    if (int rc = check_call(engine, d_dlatents, batch)) return rc;
    gance::DeviceGuard guard(engine->cfg.device);
    GANCE_HIP_CHECK(guard.status());
    // filtered profiling keeps its records across calls (bench.py averages them); else one call's worth
    if (engine->profile_only.empty() || engine->steps_used >= 4096) engine->steps_used = 0;
    WorkspaceTurn turn(engine->ws.get(), (hipStream_t)stream);
    return synthesize_from_dlat(engine, d_dlatents, batch, d_out_u8, d_out_f32, (hipStream_t)stream);
}

// mapping network + truncation + synthesis (no ordering against other users of the workspace: the callers do that)
static int synthesize_from_z(gance_engine* engine, const float* d_z, int32_t batch, float truncation_psi, uint8_t* d_out_u8,
                             float* d_out_f32, hipStream_t stream) {
    const float* in = d_z;
    float* bufs[2] = {engine->ws->map_a, engine->ws->map_b_buf};
    for (int i = 0; i < kMappingLayers; ++i) {
        StepScope scope(engine, stream, "mapping_dense", 2.0 * batch * kDlatent * kDlatent,
                        4.0 * kDlatent * kDlatent);
        GANCE_HIP_CHECK(gance::launch_mapping_dense(in, engine->pool + engine->map_w[i],
                                                    engine->pool + engine->map_b[i], bufs[i & 1],
                                                    batch, i == 0, stream));
        in = bufs[i & 1];
    }
    {
        StepScope scope(engine, stream, "truncate", 0.0, 0.0);
        GANCE_HIP_CHECK(gance::launch_broadcast_truncate(in, engine->pool + engine->avg_off,
                                                         truncation_psi, engine->ws->dlat, batch,
                                                         engine->num_rows, stream));
    }
    return synthesize_from_dlat(engine, engine->ws->dlat, batch, d_out_u8, d_out_f32, stream);
}

int gance_synthesize_z(gance_engine* engine, const float* d_z, int32_t batch, float truncation_psi,
                       uint8_t* d_out_u8, float* d_out_f32, void* stream_) {
    if (int rc = check_call(engine, d_z, batch)) return rc;
    gance::DeviceGuard guard(engine->cfg.device);
    GANCE_HIP_CHECK(guard.status());
    hipStream_t stream = (hipStream_t)stream_;
    // filtered profiling keeps its records across calls (bench.py averages them); else one call's worth
    if (engine->profile_only.empty() || engine->steps_used >= 4096) engine->steps_used = 0;
    WorkspaceTurn turn(engine->ws.get(), stream);
    return synthesize_from_z(engine, d_z, batch, truncation_psi, d_out_u8, d_out_f32, stream);
}

// The host-buffer entry (the reference's own call form: one frame per call, numpy in, numpy out,
// network_functions.py:289-301). Latency path: pinned staging buffers, a private stream, and the ~50-launch
// sequence of a (batch, entry, psi, outputs) combination captured ONCE into a hipGraph and replayed
// (the first call of a combination runs eagerly: it also performs the launchers' one-time attribute setup,
// which may not happen under capture). GANCE_TUNE_GRAPH=0 keeps every call eager.
static int host_call(gance_engine* e, const float* h_in, size_t in_floats, int batch, bool is_z,
                     float psi, uint8_t* h_u8, float* h_f32) {
    if (int rc = check_call(e, h_in, batch)) return rc;
    gance::DeviceGuard guard(e->cfg.device);
    GANCE_HIP_CHECK(guard.status());
    gance_workspace* ws = e->ws.get();
    const size_t px = (size_t)e->cfg.resolution * e->cfg.resolution * 3;
    float* d_in = is_z ? ws->z_in : ws->dlat;
    if (ws->used) GANCE_HIP_CHECK(hipEventSynchronize(ws->last_use));  // another engine's call may still read the shared scratch
    hipStream_t hs = ws->host_stream;
    std::memcpy(ws->pinned_in, h_in, in_floats * sizeof(float));
    GANCE_HIP_CHECK(hipMemcpyAsync(d_in, ws->pinned_in, in_floats * sizeof(float), hipMemcpyHostToDevice, hs));
    e->keep_skip_image = h_f32 != nullptr;
    struct KeepSkipReset {  // every return path below leaves the flag cleared
        gance_engine* engine;
        ~KeepSkipReset() { engine->keep_skip_image = false; }
    } keep_skip_reset{e};
    auto run = [&]() {
        return is_z ? synthesize_from_z(e, d_in, batch, psi, ws->u8buf, nullptr, hs)
                    : synthesize_from_dlat(e, d_in, batch, ws->u8buf, nullptr, hs);
    };
    static const bool graphs_enabled = [] { const char* v = std::getenv("GANCE_TUNE_GRAPH"); return !(v && std::atoi(v) == 0); }();
    const bool plain = !graphs_enabled || (e->cfg.flags & GANCE_FLAG_PROFILE_STEPS) || e->debug_stop_after > 0;
    int rc = GANCE_OK;
    if (plain) {
        if (e->profile_only.empty() || e->steps_used >= 4096) e->steps_used = 0;
        rc = run();
    } else {
        unsigned psi_bits = 0;
        std::memcpy(&psi_bits, &psi, sizeof(psi_bits));
        const auto key = std::make_tuple(batch, is_z ? 1 : 0, is_z ? psi_bits : 0u, h_f32 != nullptr ? 1 : 0, e->noise_randomized ? 1 : 0);
        GraphEntry& entry = e->graphs[key];
        if (entry.disabled) {
            rc = run();
        } else if (entry.exec != nullptr) {
            GANCE_HIP_CHECK(hipGraphLaunch(entry.exec, hs));
        } else if (!entry.warmed) {
            rc = run();
            entry.warmed = true;
        } else {
            hipGraph_t graph = nullptr;
            GANCE_HIP_CHECK(hipStreamBeginCapture(hs, hipStreamCaptureModeThreadLocal));
            rc = run();
            const hipError_t end = hipStreamEndCapture(hs, &graph);
            if (rc == GANCE_OK && end == hipSuccess && hipGraphInstantiate(&entry.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                hipGraphDestroy(graph);
                GANCE_HIP_CHECK(hipGraphLaunch(entry.exec, hs));
            } else {
                // capture refused (it never should): eager launches for this combination from now on
                if (graph != nullptr) hipGraphDestroy(graph);
                (void)hipGetLastError();
                entry.exec = nullptr;
                entry.disabled = true;
                rc = run();
            }
        }
    }
    if (rc) return rc;
    if (e->debug_stop_after > 0) {
        GANCE_HIP_CHECK(hipStreamSynchronize(hs));
        hipEventRecord(ws->last_use, hs);
        ws->used = true;
        return GANCE_OK;
    }
    if (h_u8) GANCE_HIP_CHECK(hipMemcpyAsync(ws->pinned_out, ws->u8buf, px * batch, hipMemcpyDeviceToHost, hs));
    hipEventRecord(ws->last_use, hs);
    ws->used = true;
    GANCE_HIP_CHECK(hipStreamSynchronize(hs));
    if (int rc_fault = check_call(e, h_in, batch)) return rc_fault;  // (a kernel of this very call may have raised the workspace's fault flag)
    if (h_u8) std::memcpy(h_u8, ws->pinned_out, px * batch);
    if (h_f32) {
        // the final skip image is in whichever ybuf the last ToRGB wrote
        const int n_rgb = (int)e->rgbs.size();
        const int ycur = (n_rgb - 1) & 1;
        GANCE_HIP_CHECK(hipMemcpy(h_f32, ws->ybuf[ycur], px * batch * sizeof(float), hipMemcpyDeviceToHost));
    }
    return GANCE_OK;
}

int gance_synthesize_w_host(gance_engine* engine, const float* h_dlatents, int32_t batch,
                            uint8_t* h_out_u8, float* h_out_f32) {
    if (engine == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "engine is NULL");
    return host_call(engine, h_dlatents, (size_t)batch * engine->num_rows * kDlatent, batch, false,
                     0.f, h_out_u8, h_out_f32);
}

int gance_synthesize_z_host(gance_engine* engine, const float* h_z, int32_t batch,
                            float truncation_psi, uint8_t* h_out_u8, float* h_out_f32) {
    if (engine == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "engine is NULL");
    return host_call(engine, h_z, (size_t)batch * kDlatent, batch, true, truncation_psi, h_out_u8,
                     h_out_f32);
}

int gance_engine_set_profiling(gance_engine* engine, int32_t flags, const char* only_step) {
    if (engine == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL engine");
    engine->cfg.flags = (engine->cfg.flags & ~GANCE_FLAG_PROFILE_STEPS) | (flags & GANCE_FLAG_PROFILE_STEPS);
    engine->profile_only = only_step ? only_step : "";
    engine->steps_used = 0;
    return GANCE_OK;
}

int32_t gance_engine_step_count(const gance_engine* engine) { return engine ? engine->steps_used : 0; }

int gance_engine_step_info(gance_engine* engine, int32_t index, char* name64, float* ms,
                           double* flops, double* bytes) {
    if (engine == nullptr || index < 0 || index >= engine->steps_used)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "step index out of range");
    StepRecord& r = engine->steps[index];
    GANCE_HIP_CHECK(hipEventSynchronize(r.stop));
    float elapsed = 0.f;
    GANCE_HIP_CHECK(hipEventElapsedTime(&elapsed, r.start, r.stop));
    if (name64) std::snprintf(name64, 64, "%s", r.name);
    if (ms) *ms = elapsed;
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    return GANCE_OK;
}

int gance_engine_randomize_noise(gance_engine* e, uint64_t seed, int32_t count, uint64_t first_sample, const int64_t* d_sample_ids, void* stream_) {
    if (e == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (count == 0) count = e->cfg.max_batch;
    if (count < 1 || count > e->cfg.max_batch)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "count " + std::to_string(count) + " outside [1, max_batch=" + std::to_string(e->cfg.max_batch) + "]");
    gance::DeviceGuard guard(e->cfg.device);
    GANCE_HIP_CHECK(guard.status());
    hipStream_t stream = (hipStream_t)stream_;
    const int nconv = (int)e->convs.size();
    if (e->noise_rand_off.empty()) {  // planes for max_batch samples of every layer that reads noise (none at random init)
        size_t total = 0;
        e->noise_rand_off.assign(nconv, SIZE_MAX);
        for (int i = 0; i < nconv; ++i) {
            if (e->conv_ns[i] == 0.0f) continue;
            e->noise_rand_off[i] = total;
            total += ((size_t)e->cfg.max_batch << (2 * e->convs[i].res_log2));
        }
        if (total > 0) {
            const hipError_t err = hipMalloc((void**)&e->noise_rand, total * sizeof(float));
            if (err != hipSuccess) {
                e->noise_rand_off.clear();
                return fail(err == hipErrorOutOfMemory ? GANCE_ERR_OUT_OF_MEMORY : GANCE_ERR_HIP, std::string("noise planes: ") + hipGetErrorString(err));
            }
        }
    }
    // (the planes may still be read by a call in flight on another stream of the shared workspace: order behind it)
    if (e->ws && e->ws->used) GANCE_HIP_CHECK(hipStreamWaitEvent(stream, e->ws->last_use, 0));
    for (int i = 0; i < nconv; ++i) {
        if (e->noise_rand_off[i] == SIZE_MAX) continue;  // (a layer whose strength is zero never reads its noise)
        GANCE_HIP_CHECK(gance::launch_normal_noise(e->noise_rand + e->noise_rand_off[i], (size_t)1 << (2 * e->convs[i].res_log2), count, seed,
                                                   (unsigned long long)i, first_sample, (const long long*)d_sample_ids, stream));
    }
    e->noise_rand_count = count;
    e->noise_randomized = true;
    if (stream == nullptr) GANCE_HIP_CHECK(hipStreamSynchronize(nullptr));  // (the host-buffer entries run on a private stream)
    return GANCE_OK;
}

int gance_engine_restore_noise(gance_engine* e, void* stream_) {
    if (e == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "engine is NULL");
    (void)stream_;  // (nothing to copy: the stored buffers were never overwritten, the launches go back to reading them)
    e->noise_randomized = false;
    return GANCE_OK;
}

int gance_engine_debug_read_noise(gance_engine* e, int32_t conv_layer, int32_t sample, float* h_out, uint64_t count) {
    if (e == nullptr || h_out == nullptr || conv_layer < 0 || conv_layer >= (int)e->convs.size() || sample < 0 || sample >= e->cfg.max_batch)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "bad argument to gance_engine_debug_read_noise");
    const uint64_t n = (uint64_t)1 << (2 * e->convs[conv_layer].res_log2);
    if (count != n) return fail(GANCE_ERR_INVALID_ARGUMENT, "the layer's noise buffer holds " + std::to_string(n) + " floats");
    gance::DeviceGuard guard(e->cfg.device);
    GANCE_HIP_CHECK(guard.status());
    GANCE_HIP_CHECK(hipDeviceSynchronize());
    int b_stride = 0;
    const float* src = layer_noise(e, conv_layer, &b_stride);
    if (src == nullptr) src = e->pool + e->conv_noise[conv_layer];  // (strength zero: the stored buffer, which no launch reads)
    GANCE_HIP_CHECK(hipMemcpy(h_out, src + (size_t)sample * b_stride, n * sizeof(float), hipMemcpyDeviceToHost));
    return GANCE_OK;
}

int gance_engine_debug_stop_after(gance_engine* engine, int32_t num_conv_layers) {
    if (engine == nullptr) return fail(GANCE_ERR_INVALID_ARGUMENT, "engine is NULL");
    engine->debug_stop_after = num_conv_layers;
    return GANCE_OK;
}

int gance_engine_debug_read_activation(gance_engine* engine, int32_t batch, float* h_out,
                                       uint64_t max_floats, int32_t* out_channels,
                                       int32_t* out_side) {
    if (engine == nullptr || h_out == nullptr)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "NULL argument");
    gance::DeviceGuard guard(engine->cfg.device);
    GANCE_HIP_CHECK(guard.status());
    const int C = engine->last_act_c, R = engine->last_act_side;
    const size_t n = (size_t)batch * C * R * R;
    if (n == 0 || n > max_floats) return fail(GANCE_ERR_INVALID_ARGUMENT, "activation does not fit");
    GANCE_HIP_CHECK(hipDeviceSynchronize());
    const size_t padded = (size_t)batch * C * act_plane(R);
    std::vector<float> tmp(padded);
    GANCE_HIP_CHECK(hipMemcpy(tmp.data(), engine->ws->act[engine->last_act_layer], padded * sizeof(float),
                              hipMemcpyDeviceToHost));
    for (size_t bc = 0; bc < (size_t)batch * C; ++bc)
        for (int y = 0; y < R; ++y)
            std::memcpy(h_out + (bc * R + y) * R, &tmp[bc * act_plane(R) + (size_t)(y + 1) * (R + 8) + 4],
                        R * sizeof(float));
    if (out_channels) *out_channels = C;
    if (out_side) *out_side = R;
    return GANCE_OK;
}

int gance_resize_bicubic_u8(const uint8_t* d_in, int32_t batch, int32_t src_side, uint8_t* d_out,
                            int32_t dst_side, void* stream) {
    if (d_in == nullptr || d_out == nullptr || batch < 1 || src_side < 1 || dst_side < 1)
        return fail(GANCE_ERR_INVALID_ARGUMENT, "bad argument to gance_resize_bicubic_u8");
    gance::DeviceGuard guard(gance::device_of_pointer(d_in));  // launch where the frames live
    GANCE_HIP_CHECK(guard.status());
    GANCE_HIP_CHECK(gance::launch_resize_bicubic_u8(d_in, batch, src_side, d_out, dst_side, (hipStream_t)stream));
    return GANCE_OK;
}

}  // extern "C"
