/* mi355vits.h — C ABI of libmi355vits.so, the MI355X-native VITS inference engine that stands
 * in for the one third-party call on the Mimic 3 hot path.
 *
 * Reference interface each entry point replaces (paths relative to the mimic3 repository):
 *
 *   mi355vits_create*          onnxruntime.InferenceSession(str(generator_path), sess_options=...,
 *                              providers=...)                      mimic3_tts/voice.py:403-405
 *   mi355vits_run              self.onnx_model.run(None, inputs)   mimic3_tts/voice.py:230
 *                              (feed dict built at voice.py:180-218: "input" int64 [B,Tx],
 *                              "input_lengths" int64 [B], "scales" f32[3] = noise_scale,
 *                              length_scale, noise_w; "sid" int64 [B] iff multi-speaker)
 *     + MI355VITS_WANT_PCM16   audio_float_to_int16(audio)         mimic3_tts/utils.py:237-244
 *                              (called right after run, inside the timed region, voice.py:231)
 *     + run_args.pcm_volume    audioop.mul(audio_bytes, 2, volume / 100)   mimic3_tts/tts.py:542-543
 *   mi355vits_destroy          the session's finaliser (voice.py:71-72 keeps sessions in a
 *                              process-wide cache, so they live until exit)
 *   mi355vits_get_config       TrainingConfig.model / .audio       mimic3_tts/config.py:112-143,30-60
 *
 * Plain pointers and sizes only; no torch / numpy types.  All calls return 0 on success or a
 * negative MI355VITS_ERR_* code; mi355vits_last_error() gives the message.  Never aborts the
 * process, never returns partial audio (the reference raises Python exceptions at this level,
 * SURVEY.md §8b).  `run` is serialised per handle (internal mutex), so a handle may be shared by
 * the server's worker threads like the reference's shared sessions (voice.py:277-292).
 */
#ifndef MI355VITS_H
#define MI355VITS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355VITS_MAX_STAGES 8

/* Hyper-parameters of the voice (mirror of ModelConfig + the constants upstream VITS hard-codes).
 * Stored verbatim in the .m355 weight container header. */
typedef struct mi355vits_config {
    int32_t num_symbols;
    int32_t n_speakers;
    int32_t inter_channels;
    int32_t hidden_channels;
    int32_t filter_channels;
    int32_t n_heads;
    int32_t n_layers;
    int32_t kernel_size;
    int32_t resblock; /* 1 or 2 */
    int32_t n_resblock_kernels;
    int32_t resblock_kernel_sizes[MI355VITS_MAX_STAGES];
    int32_t resblock_n_dilations[MI355VITS_MAX_STAGES];
    int32_t resblock_dilations[MI355VITS_MAX_STAGES * MI355VITS_MAX_STAGES];
    int32_t n_upsamples;
    int32_t upsample_rates[MI355VITS_MAX_STAGES];
    int32_t upsample_kernel_sizes[MI355VITS_MAX_STAGES];
    int32_t upsample_initial_channel;
    int32_t gin_channels;
    int32_t window_size;
    int32_t flow_n_flows;
    int32_t flow_wn_layers;
    int32_t flow_wn_kernel;
    int32_t flow_wn_dilation_rate;
    int32_t dp_kernel_size;
    int32_t dp_dds_layers;
    int32_t dp_n_flows;
    int32_t dp_num_bins;
    float dp_tail_bound;
    int32_t sample_rate;
    int32_t hop_length;
} mi355vits_config;

typedef struct mi355vits_engine* mi355vits_handle;

enum {
    MI355VITS_OK = 0,
    MI355VITS_ERR_INVALID = -1,  /* bad argument / shape / id out of range */
    MI355VITS_ERR_IO = -2,       /* weight file unreadable */
    MI355VITS_ERR_FORMAT = -3,   /* not an .m355 container, or tensors missing / mis-shaped */
    MI355VITS_ERR_DEVICE = -4,   /* HIP runtime error */
    MI355VITS_ERR_NOMEM = -5,
    MI355VITS_ERR_INTERNAL = -6
};

/* run flags */
#define MI355VITS_WANT_FLOAT 1u   /* float32 waveform [B, l_max]: what onnx_model.run returns */
#define MI355VITS_WANT_PCM16 2u   /* int16 [B, l_max]: audio_float_to_int16, per utterance */
#define MI355VITS_DEVICE_ONLY 4u  /* leave audio in HBM, return lengths only (see mi355vits_fetch) */
#define MI355VITS_DEBUG_TAPS 8u   /* keep named intermediates for mi355vits_get_tap (tests) */

typedef struct mi355vits_run_args {
    int32_t batch;             /* B >= 1 */
    int32_t tx_max;            /* padded phoneme length Tx >= 1 */
    const int64_t* ids;        /* [B, tx_max]  "input" */
    const int64_t* lengths;    /* [B]          "input_lengths", 0 <= len <= tx_max */
    const float* scales;       /* [3]          "scales" = noise_scale, length_scale, noise_w */
    const int64_t* sid;        /* [B] or NULL  "sid" (required iff multi-speaker) */
    uint64_t seed;             /* Philox key for the two Gaussian draws */
    uint64_t utterance_base;   /* global index of row 0: noise does not depend on batch split */
    const float* noise_w;      /* optional injected N(0,1) [B, 2, tx_max] (parity tests) */
    const float* noise_z;      /* optional injected N(0,1) [B, inter_channels, noise_z_frames] */
    int32_t noise_z_frames;
    const int32_t* forced_durations; /* optional [B, tx_max]: overrides ceil(exp(logw)*length_scale) */
    uint32_t flags;
    double pcm_volume;         /* with WANT_PCM16: audioop.mul(pcm, 2, pcm_volume) fused into the int16 kernel —
                                * what Mimic3TextToSpeechSystem._speak_sentence_phonemes does on the host with
                                * settings.volume / 100 (mimic3_tts/tts.py:542-543).  0 or 1 = leave as is. */
} mi355vits_run_args;

typedef struct mi355vits_result {
    int32_t batch;
    int64_t l_max;     /* samples per row = hop * max_b frames_b */
    int64_t ty_max;    /* latent frames of the longest row */
    float* audio;      /* [B, l_max] or NULL; row b valid up to lengths[b] (rest is padding) */
    int16_t* pcm;      /* [B, l_max] or NULL */
    int64_t* lengths;  /* [B] valid samples per row */
    float* peaks;      /* [B] max |audio| over each row's valid samples */
    void* owner_;      /* private */
} mi355vits_result;

const char* mi355vits_version(void);
/* Number of HIP devices this process can use (0 when there is none): sizes the in-process device round-robin of a
 * session serving mimic3_http's worker threads (mimic3_http/__main__.py:53-61 starts them in ONE process). */
int mi355vits_device_count(void);

/* Load a voice from an .m355 container (mimic3_amd/weights.py) onto HIP device `device`. */
int mi355vits_create(const char* weights_path, int device, mi355vits_handle* out);
int mi355vits_create_from_buffer(const void* blob, size_t blob_bytes, int device, mi355vits_handle* out);
/* Another execution lane for the voice of `src`, on the same device: own HIP stream and workspace, SHARED weight
 * replica (the weights are freed with the last lane).  The reference shares one session between the server's worker
 * threads (voice.py:277-292, mimic3_http/__main__.py:53-61); lanes are how those concurrent `run` calls overlap. */
int mi355vits_clone(mi355vits_handle src, mi355vits_handle* out);
void mi355vits_destroy(mi355vits_handle h);
/* Which matrix-core path the dense Conv1d stacks of the flow and the decoder take on this handle
 * (default: environment MI355VITS_MATH = "f32" | "bf16x3" | "bf16w" | "f16x2", read when the handle is created, else BF16X3):
 *   MI355VITS_MATH_F32     v_mfma_f32_32x32x2_f32 — f32 operands, bit-exact f32 FMA chains;
 *   MI355VITS_MATH_BF16X3  the f32 operands split EXACTLY into three bf16 terms each (x = h + m + l) and the six leading
 *                          partial products on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16 / _16x16x32_bf16) with f32 accumulation: every retained product is
 *                          exact, the three dropped ones are below 2^-24 of the product — the f32 rounding level.  Same
 *                          parity tolerances; measured against fp64 it is slightly MORE accurate than the f32 MFMA kernels
 *                          (tests/test_gpu_parity.py::test_split_bf16_staged_conv_kernel_vs_fp64), at 6/16 of their
 *                          matrix-core time (bf16 MFMA = 16 x the f32 MFMA rate on MI355X).  f32 in, f32 out, f32
 *                          accumulate: nothing is stored or rounded in bf16 except the exact split terms.
 * Results of the two differ at f32 rounding level (like two f32 BLAS builds); each is deterministic, and within a mode a
 * batched call is bitwise equal to separate calls.  The text side (encoder convs, duration-predictor stacks) follows the
 * mode as well (k_enc_b3 / k_enc_o_ln / k_dds_stack_b3 in the split modes, the f32-MFMA kernels in MATH_F32), with ONE
 * exception: it never runs with rounded weights — in BF16W it takes the exact BF16X3 path, so the durations
 * ceil(exp(logw) * length_scale), hence every utterance length, are bitwise those of the default mode (Engine::tmath;
 * tests/test_gpu_parity.py::test_bf16_weights_mode_natural_durations_200_sentences).  Attention itself is an f32-MFMA chain. */
#define MI355VITS_MATH_F32 0
#define MI355VITS_MATH_BF16X3 1
/* "bf16 weights" (BASELINE.json configs[4]): BF16X3 with the weights' leading bf16 term only — the weights are rounded to
 * bf16, the activations stay exact f32 (three terms), f32 accumulate: three MFMA products per multiply-add.  A REDUCED
 * precision variant: separate tolerance (rel. RMS <= 2e-2 vs the f32 oracle), never the default, reported separately.
 * Applies to the frame-rate convs (flow, decoder) only; the text side stays exact (see above): same lengths as the default. */
#define MI355VITS_MATH_BF16W 2
/* experimental, opt-in: every kernel of the three-term bf16 split (fused MRF stages, fused WaveNet layers, staged convs,
 * polyphase upsamplers) with both operands split into TWO fp16 terms (11 + 11 significant bits, three products per multiply-add
 * on v_mfma_f32_32x32x16_f16); every other kernel (text encoder, duration predictor) as in BF16X3.  A FIXED-SCALE mode:
 * weights are packed x 2^13 (a stage holding a weight |w| >= 7.99 silently runs as BF16X3 instead), activations are written to
 * the LDS tiles x 2^4 with a saturating round-toward-zero conversion — |x| > 4094 CLIPS (no runtime detection), and below
 * |x| = 2^-7 the second term goes subnormal (absolute instead of relative precision).  Measured on the MI355X
 * (tests/test_gpu_serving.py::test_fused_mrf_stages_on_scaled_activations_vs_fp64): at the activations' natural scale the stage
 * error against fp64 is 1.2e-6 (like the other modes); with the decoder's activations scaled by 2^+-40 the result is finite but
 * meaningless (rel. error ~1).  A 22-bit-operand mode between BF16W and the f32-grade default, with its own tests; never the
 * default.  (The 64- / 32-channel MRF stages run the round-2 kernel k_mrf_fused in this mode, not k_mrf_p.) */
#define MI355VITS_MATH_F16X2 3
int mi355vits_set_math(mi355vits_handle h, int mode);
int mi355vits_get_math(mi355vits_handle h);
int mi355vits_get_config(mi355vits_handle h, mi355vits_config* out);

/* One synthesis call.  `out` is filled with callee-allocated (pinned) host buffers; release
 * them with mi355vits_free_result.  With MI355VITS_DEVICE_ONLY the audio stays in the engine's
 * workspace until the next run; mi355vits_fetch copies it out afterwards. */
int mi355vits_run(mi355vits_handle h, const mi355vits_run_args* args, mi355vits_result* out);
int mi355vits_fetch(mi355vits_handle h, uint32_t want_flags, mi355vits_result* out);
void mi355vits_free_result(mi355vits_result* r);
/* Device pointers of the last run's results on this handle (valid until its next run; the engine's stream has been
 * synchronised when this returns): int16 [batch, row_stride] and/or float [batch, row_stride] in HBM, plus the valid
 * sample counts [batch] (int32, device).  For the optional device-side result gather over RCCL (north star; SURVEY.md
 * §8e) — the data never visits the host.  Pass NULL for what is not wanted. */
int mi355vits_device_result(mi355vits_handle h, const int16_t** pcm, const float** audio, int64_t* row_stride,
                            int32_t* batch, const int32_t** device_lengths);

/* Message of the last error on this handle (or, with h == NULL, of the last failed create on the
 * calling thread).  Valid until the next call on the same handle / thread. */
const char* mi355vits_last_error(mi355vits_handle h);

/* Per-kernel timing with HIP events on the engine's own stream (bench.py roofline leg).
 * enable(1) brackets every launch with an event pair; report() synchronises and writes one
 * line per kernel name: "name calls total_ms flops bytes\n".  Returns bytes written. */
int mi355vits_profile_enable(mi355vits_handle h, int on);
int mi355vits_profile_reset(mi355vits_handle h);
long mi355vits_profile_report(mi355vits_handle h, char* buf, size_t cap);

/* Wall time of the last run on the engine's stream, HIP events around the whole call (ms). */
float mi355vits_last_run_ms(mi355vits_handle h);

/* Debug taps (needs MI355VITS_DEBUG_TAPS on the last run): copy the named intermediate to
 * `out` (capacity in floats); dims receives up to 4 extents.  Returns element count or < 0. */
long mi355vits_get_tap(mi355vits_handle h, const char* name, float* out, size_t capacity, int64_t dims[4]);
/* The same for rows [row0, row0 + nrows) of the tap's leading (batch) extent only; dims[0] = nrows. */
long mi355vits_get_tap_rows(mi355vits_handle h, const char* name, long row0, long nrows, float* out, size_t capacity, int64_t dims[4]);
long mi355vits_list_taps(mi355vits_handle h, char* buf, size_t cap);

/* The kernel unit-test, micro-benchmark and box-probe hooks are NOT part of this library: include/mi355vits_lab.h,
 * exported by libmi355vits_hooks.so (the product's objects + the hooks) and the lab build only. */

#ifdef __cplusplus
}
#endif
#endif /* MI355VITS_H */
