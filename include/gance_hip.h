/*
 * gance_hip.h -- C ABI of libgance_hip.so, the MI355X (gfx950) implementation of GANce's hot path
 *                audio -> latent -> StyleGAN2 frame synthesis.
 *
 * Plain pointers and sizes only; no torch / numpy types. Device pointers are HIP device pointers
 * (e.g. torch.Tensor.data_ptr() of a cuda tensor); `stream` is a hipStream_t passed as void*
 * (NULL = the null stream). All functions return GANCE_OK (0) or a negative-free status code
 * below; gance_last_error() returns a thread-local description of the last failure.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the GANce tree).
 */
#ifndef GANCE_HIP_H
#define GANCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANCE_ABI_VERSION 6

enum gance_status {
    GANCE_OK = 0,
    GANCE_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, bad size, batch > max_batch ...            */
    GANCE_ERR_BAD_WEIGHTS = 2,      /* weight blob length does not match the configuration      */
    GANCE_ERR_HIP = 3,              /* a HIP runtime call failed (see gance_last_error)         */
    GANCE_ERR_OUT_OF_MEMORY = 4,    /* hipMalloc failed                                         */
    GANCE_ERR_NO_DEVICE = 5         /* no gfx950 device visible: there is NO CPU fallback       */
};

const char* gance_last_error(void);
int gance_abi_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Latent -> frame: StyleGAN2 config-f generator engine                                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct gance_engine gance_engine; /* opaque; owns weights + workspace in HBM */

typedef struct gance_engine_config {
    int32_t resolution; /* output side, power of two in [8, 1024]; 1024 = FFHQ config-f        */
    int32_t max_batch;  /* workspace is sized for this many frames per call (>= 1)             */
    int32_t device;     /* HIP device ordinal                                                  */
    int32_t flags;      /* GANCE_FLAG_*                                                        */
} gance_engine_config;

#define GANCE_FLAG_PROFILE_STEPS 1 /* record a hipEvent pair around every kernel launch        */
/* The Conv1 layers from 32x32 up run in Winograd form whenever a launch has at least one tile per CU: F(4x4,3x3) (1/4 of
 * the multiply-adds) where the launch fills the chip with its 16 x 64 pixel tiles, else F(2x2,3x3) (4/9), else the direct
 * form; results differ between the forms by fp32 rounding only (~1e-6 of the activation range). DIRECT_CONV keeps every
 * layer in direct form; FORCE_WINOGRAD alone puts every layer that supports it on the F(2x2,3x3) kernels whatever the
 * batch, FORCE_WINOGRAD | WINOGRAD43 on the F(4x4,3x3) kernel where that supports the layer (parity tests of each form). */
#define GANCE_FLAG_DIRECT_CONV 2
#define GANCE_FLAG_FORCE_WINOGRAD 4
/* Conv0_up of the layers whose input is >= 64 wide runs as ONE kernel (transposed conv + FIR + noise + bias +
 * leaky ReLU, the (2H+1)^2 intermediate stays on chip) whenever the launch gives every CU a block without
 * cutting the image into row segments shorter than 4 steps. SPLIT_UPFIR keeps the two-pass form
 * (transposed conv, then FIR pass); FORCE_FUSED_UPFIR drops the block-count condition (parity tests). */
#define GANCE_FLAG_SPLIT_UPFIR 8
#define GANCE_FLAG_FORCE_FUSED_UPFIR 16
/* The per-call scratch (activations etc.: ~0.8 GB per frame of batch capacity at 1024^2) is shared by every
 * engine of one (device, resolution, max_batch): N resident networks cost N x 135 MB of weights + ONE workspace.
 * Calls that share it are ordered by an event, on whatever streams they run. PRIVATE_WORKSPACE gives an engine
 * its own (calls of different engines may then overlap on different streams). */
#define GANCE_FLAG_PRIVATE_WORKSPACE 32
#define GANCE_FLAG_WINOGRAD43 64 /* with FORCE_WINOGRAD: Conv1 layers from 32x32 up in Winograd F(4x4,3x3) form wherever the kernel supports the layer */

/*
 * Load a network. Replaces load_network_network + wrap_loaded_network
 * (gance/network_interface/network_functions.py:93-111,114-192): the reference unpickles a TF1
 * Network and opens a session; here the caller hands the raw variables as one float32 blob in
 * the order documented in gance_amd/stylegan2/spec.py::variable_shapes (TF variable order:
 * mapping dense 0..7 {weight,bias}, dlatent_avg, const, per conv layer {weight HWIO, mod_weight,
 * mod_bias, noise_strength, bias}, per ToRGB {weight, mod_weight, mod_bias, bias}, noise buffers).
 * The engine applies the equalised-LR runtime coefficients and re-lays the weights out for its
 * kernels. `host_weights` is host memory and is not retained.
 */
int gance_engine_create(const gance_engine_config* config, const float* host_weights,
                        uint64_t num_floats, gance_engine** out_engine);

/* Replaces the worker's stop_function (network_functions.py:303-321): frees HBM. NULL is a no-op. */
void gance_engine_destroy(gance_engine* engine);

/* network.input_shape[1] (network_functions.py:191): length of a z / w vector (512). */
int32_t gance_engine_vector_length(const gance_engine* engine);
/* Number of dlatent rows W the matrix path expects (18 at 1024, 14 at 256). */
int32_t gance_engine_num_layers(const gance_engine* engine);
int32_t gance_engine_resolution(const gance_engine* engine);
int32_t gance_engine_max_batch(const gance_engine* engine);
/* Number of floats gance_engine_create expects for `resolution` (0 if unsupported). */
uint64_t gance_weight_blob_floats(int32_t resolution);

/*
 * Matrix path. Replaces create_image_matrix (network_functions.py:160-169):
 *   G.components.synthesis.run(dlatents[1,W,512], randomize_noise=False,
 *                              output_transform=convert_images_to_uint8(nchw_to_nhwc=True))
 * batched: d_dlatents is [batch][W][512] float32 on the device. d_out_u8 receives
 * [batch][R][R][3] uint8 (RGB, NHWC) and may be NULL; d_out_f32 (optional, may be NULL) receives
 * the pre-quantisation image [batch][3][R][R] float32 for tolerance checks.
 */
int gance_synthesize_w(gance_engine* engine, const float* d_dlatents, int32_t batch,
                       uint8_t* d_out_u8, float* d_out_f32, void* stream);

/*
 * Vector path. Replaces create_image_vector (network_functions.py:144-158):
 *   G.run(z[1,512], None, truncation_psi=1.2, output_transform=...)
 * = normalise z, 8-layer mapping, broadcast to W rows, w' = avg + psi*(w - avg) on every row,
 * synthesis. d_z is [batch][512] float32 on the device. Stored noise buffers are used.
 */
int gance_synthesize_z(gance_engine* engine, const float* d_z, int32_t batch,
                       float truncation_psi, uint8_t* d_out_u8, float* d_out_f32, void* stream);

/*
 * Host-buffer forms of the two calls above for the reference's one-frame-at-a-time boundary
 * (ImageFunction.__call__, network_functions.py:51-63): H2D copy, synthesis, D2H copy,
 * synchronous. h_out_u8 is [batch][R][R][3]; h_out_f32 may be NULL.
 */
int gance_synthesize_w_host(gance_engine* engine, const float* h_dlatents, int32_t batch,
                            uint8_t* h_out_u8, float* h_out_f32);
int gance_synthesize_z_host(gance_engine* engine, const float* h_z, int32_t batch,
                            float truncation_psi, uint8_t* h_out_u8, float* h_out_f32);

/*
 * Profiling / debugging (no reference counterpart).
 * With GANCE_FLAG_PROFILE_STEPS every launch of the last synthesize call is bracketed by
 * hipEvents on the call's stream. gance_engine_step_count returns the number of launches,
 * gance_engine_step_info fills name (<=63 chars + NUL), elapsed milliseconds and the
 * algorithmic FLOPs and bytes of launch `index` of the LAST call (synchronises the stream).
 */
/* Change the profiling mode between calls: `flags` replaces the GANCE_FLAG_PROFILE_STEPS bit of the
 * engine's configuration; with a non-NULL `only_step` just the launches whose name contains that
 * string are bracketed (two events per call instead of ~100: what bench.py's timed region uses
 * for the dominant kernel) and their records accumulate over calls (up to 4096) until the next
 * gance_engine_set_profiling. */
int gance_engine_set_profiling(gance_engine* engine, int32_t flags, const char* only_step);
int32_t gance_engine_step_count(const gance_engine* engine);
int gance_engine_step_info(gance_engine* engine, int32_t index, char* name64, float* ms,
                           double* flops, double* bytes);
/*
 * randomize_noise of the generator. The reference's vector path (create_image_vector, network_functions.py:152-157)
 * leaves `randomize_noise` at the upstream default True: every call draws fresh N(0, 1) noise, one plane per layer AND
 * PER SAMPLE (upstream: tf.random_normal([N, 1, H, W])), instead of the stored noise buffers; its matrix path passes
 * randomize_noise=False (:121-125). gance_engine_randomize_noise draws planes for `count` samples (0: max_batch) of every
 * layer whose noise_strength is non-zero (none in a random-init network) into buffers of the engine's own, asynchronously
 * on `stream` (with a NULL stream it returns after completion: the host-buffer entries run on a stream of their own).
 * Sample b of the following calls (batch <= count, else GANCE_ERR_INVALID_ARGUMENT) reads the plane with the id
 * d_sample_ids[b] (device memory, int64) or, with d_sample_ids NULL, first_sample + b; a plane is a function of
 * (seed, layer, id) only, so with id = frame number a frame's noise does not depend on how frames were batched or
 * sharded. The draws stay until the next call of either function; gance_engine_restore_noise goes back to the stored
 * buffers (which are never overwritten).
 */
int gance_engine_randomize_noise(gance_engine* engine, uint64_t seed, int32_t count, uint64_t first_sample,
                                 const int64_t* d_sample_ids, void* stream);
int gance_engine_restore_noise(gance_engine* engine, void* stream);
/* Debug: the noise plane sample `sample` of conv layer `conv_layer` currently reads ([res][res] floats, count = res * res), to host memory. */
int gance_engine_debug_read_noise(gance_engine* engine, int32_t conv_layer, int32_t sample, float* h_out, uint64_t count);

/*
 * Debug: run only the first `num_steps` conv layers of the next synthesize_w calls (<=0 = all)
 * and copy the current activation tensor [batch][C][res][res] to the host.
 */
int gance_engine_debug_stop_after(gance_engine* engine, int32_t num_conv_layers);
int gance_engine_debug_read_activation(gance_engine* engine, int32_t batch, float* h_out,
                                       uint64_t max_floats, int32_t* out_channels,
                                       int32_t* out_side);

/* ------------------------------------------------------------------------------------------ */
/* Audio -> latent: spectrogram, fft-roll, alpha blend with projected latents                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct gance_blend gance_blend; /* opaque; operator tables + workspace in HBM */

typedef struct gance_blend_config {
    int32_t num_frames;            /* N output frames; audio holds N * vector_length samples      */
    int32_t vector_length;         /* L = 512                                                     */
    int32_t num_projection_frames; /* F projected latents; N % F == 0 (divisor.py:19-24)          */
    int32_t latent_depth;          /* rows per latent matrix (18)                                 */
    int32_t blend_depth;           /* rows that receive the blend (--blend-depth, <= 18)          */
    int32_t fft_roll_enabled;      /* --fft-roll-enabled                                          */
    int32_t num_networks;          /* K = len(network_indices)                                    */
    int32_t has_amplitude_range;   /* 0 = no min-max scaling                                      */
    double alpha;                  /* --alpha                                                     */
    double amplitude_lo;           /* --fft-amplitude-range                                       */
    double amplitude_hi;
    /* savgol_filter applied to the rolling RMS before it is quantised to network indices:
     * 0, 0 = (3, 2), what alpha_blend_projection_file passes (visualization_inputs.py:244-252);
     * noise-blend leaves reduce_vector_rms_rolling_average at its defaults (7, 3)
     * (visualization_inputs.py:146-151, vector_reduction.py:102-108). */
    int32_t index_savgol_window_length;
    int32_t index_savgol_polyorder;
} gance_blend_config;

/*
 * Replaces the table-free Python of alpha_blend_projection_file
 * (gance/data_into_network_visualization/visualization_inputs.py:169-270) and everything it calls:
 * compute_spectrogram_smooth_scale (gance/apply_spectrogram.py:85-118), reduce_vector_rms_rolling_average
 * + quantize_results_layers (gance/vector_sources/vector_reduction.py:102-124,161-194),
 * rotate_vectors_over_time / smooth_each_vector / duplicate_to_vector_count / promote_to_matrix_duplicate
 * (gance/vector_sources/vector_sources_common.py:408-428,169-188,298-365).
 * Raises (returns GANCE_ERR_INVALID_ARGUMENT) where the reference raises ValueError: N % F != 0.
 */
int gance_blend_create(const gance_blend_config* config, int32_t device, gance_blend** out_blend);
void gance_blend_destroy(gance_blend* blend);

/*
 * d_audio: float32 [>= N*L] time-series audio (read_wavs_scale_for_video's output,
 * gance/vector_sources/music.py:60-169). d_latent_row0: float32 [F][L], row 0 of every projected
 * final latent (the only row the reference uses, visualization_inputs.py:220-231).
 * d_dlatents (may be NULL): float32 [N][latent_depth][L], the per-frame matrices `combined` is
 * split into by _frame_inputs (network_visualization.py:233-251), cast to the float32 the
 * network is fed. d_network_indices (may be NULL): int32 [N]. With debug_stages != 0 every
 * intermediate is kept for gance_blend_read_stage. Asynchronous on `stream`.
 */
int gance_blend_run(gance_blend* blend, const float* d_audio, uint64_t num_samples,
                    const float* d_latent_row0, float* d_dlatents, int32_t* d_network_indices,
                    int32_t debug_stages, void* stream);

enum gance_blend_stage {
    GANCE_STAGE_DB = 0,            /* float64 [N][255]  compute_spectrogram, transposed          */
    GANCE_STAGE_SCALED = 1,        /* float64 [N][L]    after resample + minmax                  */
    GANCE_STAGE_SMOOTHED_TIME = 2, /* float64 [N][L]    after smooth_across_vectors(7,3)         */
    GANCE_STAGE_SMOOTHED = 3,      /* float64 [N][L]    compute_spectrogram_smooth_scale         */
    GANCE_STAGE_ROLLED = 4,        /* float64 [N][L]    after rotate_vectors_over_time           */
    GANCE_STAGE_FINAL = 5,         /* float64 [N][L]    a_vectors.data                           */
    GANCE_STAGE_BLEND_ROW = 6,     /* float64 [N][L]    combined.data[0]                         */
    GANCE_STAGE_RAW_RMS = 7,       /* float32 [N]                                                */
    GANCE_STAGE_ROLL_VALUES = 8,   /* int32   [N]       quantised roll per frame                 */
    GANCE_STAGE_ROLL_CUMULATIVE = 9, /* int32 [N]       cumsum(roll) mod L                       */
    GANCE_STAGE_NETWORK_INDICES = 10, /* int32 [N]                                               */
    GANCE_STAGE_ROLLING_AVERAGE = 11, /* float64 [N]    roll chain                               */
    GANCE_STAGE_ROLLING_SMOOTHED = 12, /* float64 [N]   roll chain                               */
    GANCE_STAGE_INDEX_SMOOTHED = 13,   /* float64 [N]   network-index chain                      */
    GANCE_STAGE_MINMAX = 14            /* float64 [3]   global max |X|, min and max of the resampled dB  */
};
/* Synchronises the device and copies one stage of the LAST run to host memory. */
int gance_blend_read_stage(gance_blend* blend, int32_t stage, void* h_out, uint64_t num_bytes);

/* ---- stand-alone forms of the audio -> latent stages -------------------------------------------
 * The reference exposes every stage of the chain as its own function; gance_blend_run fuses them, these
 * entry points run one stage on a caller-shaped array (all pointers device pointers, float64 where the
 * reference computes in float64). Each returns after `stream` has drained unless noted.
 *
 * gance_vec_savgol_f64            smooth_across_vectors (axis 0) / smooth_each_vector (axis 1)
 *                                 (gance/vector_sources/vector_sources_common.py:136-188): scipy.signal.savgol_filter,
 *                                 mode "interp", along one axis of [num_vectors][vector_length]; any polyorder < window_length
 * gance_vec_fourier_resample_f64  scale_vectors_to_length_resample (:211-230): scipy.signal.resample per vector
 * gance_vec_spectrogram_f64       compute_spectrogram (gance/apply_spectrogram.py:49-82): periodic-Hann windows of
 *                                 num_frequency_bins - 2 samples, hop num_frequency_bins, 20 log10(|X| / max |X|);
 *                                 d_out is [(bins - 2) / 2][frames] like the reference's array
 * gance_vec_spectrogram2_f64      the same with the reference's `truncate` argument (:75-78): truncate = 0 keeps all
 *                                 num_frequency_bins - 2 bins of the two-sided spectrum (the maximum is then taken over all
 *                                 of them, the Nyquist bin included); d_out is [bins - 2][frames]
 * gance_vec_minmax_scale_f64      sklearn minmax_scale of the whole array, in place (apply_spectrogram.py:43)
 * gance_vec_remap_f64             remap_values_into_range (:44-61): interp1d through two points (asynchronous)
 * gance_vec_rms_rolling_average   reduce_vector_rms_rolling_average (gance/vector_sources/vector_reduction.py:102-124):
 *                                 librosa RMS (hop 512) -> pandas rolling mean, NaN head = series mean -> savgol;
 *                                 num_values = 1 + (num_samples - vector_length) / 512 entries per output
 * gance_vec_rms_rolling_max       reduce_vector_rms_rolling_max (:38-58): the same RMS, then scipy.ndimage.maximum_filter1d
 *                                 of size num_values / 80 (mode "reflect") when that is > 0, else a copy (asynchronous)
 * gance_vec_quantize_f64          quantize_results_layers (:161-194): remap [min, max] -> [0, K - 1], np.rint (asynchronous)
 * gance_debug_fourier_resample_matrix  host only: the [in_length][out_length] operator the resample kernel applies */
int gance_vec_savgol_f64(const double* d_in, int32_t num_vectors, int32_t vector_length, int32_t axis, int32_t window_length,
                         int32_t polyorder, double* d_out, void* stream);
int gance_vec_fourier_resample_f64(const double* d_in, int32_t num_vectors, int32_t in_length, int32_t out_length, double* d_out,
                                   void* stream);
int gance_vec_spectrogram_f64(const float* d_audio, uint64_t num_samples, int32_t num_frequency_bins, double* d_out, void* stream);
int gance_vec_spectrogram2_f64(const float* d_audio, uint64_t num_samples, int32_t num_frequency_bins, int32_t truncate, double* d_out,
                               void* stream);
int gance_vec_minmax_scale_f64(double* d_data, uint64_t count, double lo, double hi, void* stream);
int gance_vec_remap_f64(const double* d_in, uint64_t count, double in_lo, double in_hi, double out_lo, double out_hi, double* d_out,
                        void* stream);
int gance_vec_rms_rolling_average(const float* d_audio, uint64_t num_samples, int32_t vector_length, int32_t rolling_window,
                                  int32_t savgol_window_length, int32_t savgol_polyorder, float* d_rms, double* d_rolling,
                                  double* d_smoothed, int32_t num_values, void* stream);
int gance_vec_rms_rolling_max(const float* d_audio, uint64_t num_samples, int32_t vector_length, float* d_rms, float* d_out,
                              int32_t num_values, void* stream);
int gance_vec_quantize_f64(const double* d_in, int32_t count, int32_t num_indices, int64_t* d_out, void* stream);
int gance_debug_fourier_resample_matrix(int32_t in_length, int32_t out_length, double* h_out);

/* ------------------------------------------------------------------------------------------ */
/* Post-synthesis resize (next row f-2)                                                        */
/* ------------------------------------------------------------------------------------------ */

/*
 * Replaces resize_source's cv2.resize(image, (side, side), interpolation=cv2.INTER_CUBIC)
 * (gance/image_sources/video_common.py:399-429) for square uint8 RGB frames resident in HBM:
 * d_in [batch][src_side][src_side][3] -> d_out [batch][dst_side][dst_side][3]. Float bicubic,
 * a = -0.75, replicated border, round half up (OpenCV's 11-bit fixed point is not reproduced:
 * parity unpinned). Asynchronous on `stream`.
 */
int gance_resize_bicubic_u8(const uint8_t* d_in, int32_t batch, int32_t src_side, uint8_t* d_out,
                            int32_t dst_side, void* stream);

/* ---- smoothed-noise vector source (noise-blend) ------------------------------------------------
 * Replaces gaussian_data (gance/vector_sources/primatives.py:49-74) after its MT19937 draw, and the
 * minmax_scale(noise, (-4, 4)) of alpha_blend_vectors_max_rms_power_audio
 * (gance/data_into_network_visualization/visualization_inputs.py:135-142).
 * d_randn  [num_vectors][vector_length] float32 standard-normal draws (device; the caller's
 *          np.random.RandomState(1234).randn(...).astype(float32), primatives.py:66-68)
 * sigma_across / sigma_within   scipy.ndimage.gaussian_filter sigmas over vectors / inside a
 *          vector (mode "wrap", truncate 4.0); 0 skips the axis, as scipy does
 * feature_range  NULL, or {lo, hi}: min-max scale the RMS-normalised field to [lo, hi]
 * d_out    [num_vectors][vector_length] float32 (device, != d_randn)
 * Runs on `stream` of the current device and returns after the stream has drained. */
int gance_gaussian_noise(const float* d_randn, int32_t num_vectors, int32_t vector_length, double sigma_across,
                         double sigma_within, const double* feature_range, float* d_out, void* stream);

/* ---- eye-tracking overlay gate (pixel work only; the landmark detector stays external) ---------
 * gance_phash_crops_u8 replaces imagehash.phash(Image.fromarray(frame).crop(box)) of
 * compute_eye_tracking_overlay (gance/overlay/overlay_eye_tracking.py:100-108): PIL convert("L"),
 * PIL resize((32, 32), LANCZOS), scipy.fftpack.dct on both axes, top-left 8x8 > median.
 * d_frames [num_frames][side][side][3] uint8 (device); h_crops [num_crops][5] int32 (host):
 * frame index, x, y, width, height (a BoundingBox, overlay_common.py:19-27), inside the frame;
 * h_hashes [num_crops] uint64 (host): bit 63 = coefficient (0,0), row-major, i.e. the integer
 * whose hex string is str(imagehash.ImageHash). The phash distance is popcount(a ^ b).
 * Returns after `stream` has drained. */
int gance_phash_crops_u8(const uint8_t* d_frames, int32_t num_frames, int32_t side, const int32_t* h_crops,
                         int32_t num_crops, uint64_t* h_hashes, void* stream);

/* gance_overlay_boxes_u8 replaces write_boxes_onto_image (gance/overlay/overlay_common.py:104-172):
 * out = background, and foreground inside the padded rectangle _draw_mask draws around each
 * bounding box (x_pad = 0.098 side, y_pad = 0.058 side around the box's vertical centre, PIL
 * polygon with outline: inclusive integer bounds, corners truncated toward zero).
 * All three frame arrays are [num_frames][side][side][3] uint8 on the device; out may alias
 * background. h_boxes [num_boxes][5] int32 (host): frame index, x, y, width, height.
 * Returns after `stream` has drained. */
int gance_overlay_boxes_u8(const uint8_t* d_foreground, const uint8_t* d_background, uint8_t* d_out, int32_t num_frames,
                           int32_t side, const int32_t* h_boxes, int32_t num_boxes, void* stream);

/* ---- audio time-stretch ---------------------------------------------------------------------
 * Replaces resampy.resample(wav, sr_orig, sr_new) (resampy 0.2.2, filter "kaiser_best") of
 * _scale_wav_to_sample_rate (gance/vector_sources/music.py:212-230) inside read_wavs_scale_for_video
 * (:60-169), restated operation for operation: the 32 769-entry Kaiser-windowed-sinc half window
 * (64 zero crossings x 512 entries, beta 14.769656459379492, roll-off 0.9475937167399596), linear
 * interpolation between table entries, the running-sum time register, left wing then right wing, and the
 * accumulation in the signal's own dtype (one rounding to float32 per tap for a float32 signal). A ratio of
 * exactly 1 still filters. num_out must equal (uint64_t)(num_in * sr_new / sr_orig), resampy's length rule
 * (test/test_vector_source_music.py:13-24). Pinned by test/test_dynamic_model_switching.py:15-39 (claps.wav,
 * RMS of the first vector = 0.00298562). Device pointers; returns after `stream` has drained. */
int gance_resample_audio_f32(const float* d_in, uint64_t num_in, double sr_orig, double sr_new, float* d_out,
                             uint64_t num_out, void* stream);
int gance_resample_audio_f64(const double* d_in, uint64_t num_in, double sr_orig, double sr_new, double* d_out,
                             uint64_t num_out, void* stream);
/* host only: the filter table the resampler interpolates (count must be 32769) */
int gance_debug_resample_filter(double* h_out, uint64_t count);

#ifdef __cplusplus
}
#endif
#endif /* GANCE_HIP_H */
