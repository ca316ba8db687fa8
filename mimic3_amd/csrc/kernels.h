// kernels.h — launch wrappers of every HIP kernel on the path (definitions in kernels_*.cpp).
// Activations are channels-first fp32 [B, C, T] (the reference graph's layout);
// a tensor "view" is (pointer, batch stride, row stride) so channel halves need no copies.
#pragma once
#include "hipx.h"

namespace m355 {

struct Profiler;  // engine.h

// ---------------------------------------------------------------- Conv1d
enum ConvEpilogue { EPI_STD = 0, EPI_GATE = 1, EPI_RESSKIP = 2 };

struct ConvArgs {
    const float* x = nullptr; long x_bs = 0; int x_ld = 0;   // input  [B,Cin,T]
    float* y = nullptr;       long y_bs = 0; int y_ld = 0;   // output [B,Cout,T] (GATE: [B,H,T]; RESSKIP: h, in place)
    const float* w = nullptr;      // generic kernel: [Cout,Cin,K]; MFMA kernel: packed fragments (pack_conv_weights_mfma)
    const float* wb3 = nullptr;    // the same weights as three bf16 planes (pack_conv_weights_bf16x3_mode, layout 1) or null
    int math = 0;                  // MATH_BF16X3 + wb3: split-operand path on the bf16 matrix cores (k_conv1d_b3)
    const float* bias = nullptr;   // [Cout] or null
    const float* cond = nullptr; long cond_bs = 0;  // per-(b,co) additive term (speaker conditioning) or null
    const float* res = nullptr; long res_bs = 0; int res_ld = 0;  // residual [B,Cout,T] or null
    float* y2 = nullptr; long y2_bs = 0; int y2_ld = 0;           // RESSKIP: skip accumulator [B,H,T]
    const int* in_len = nullptr;   // [B] valid length of the input (x * mask) or null
    const int* in_len_host = nullptr;  // kernels_rbc.cpp: the host's copy of in_len (launcher: counts the items with work; null: read back)
    int nvalid = 0;                // kernels_rbc.cpp: (row, column block) items with work, filled by the launcher (mrf_valid_items)
    int rb_loop = 0;               // k_enc_b3: 64-row output blocks a workgroup walks over one staged slice (launcher: 3 on large grids)
    const int* out_len = nullptr;  // [B] valid length of the output (y * mask) or null
    int B = 1, Cin = 0, Cout = 0, T = 0, K = 1, dil = 1;
    int pad = -1;            // left zero padding; -1 = "same" ((K*dil - dil) / 2), filled in by Engine::conv
    int Tin = -1;            // input extent when it differs from the number of output positions T (-1: = T)
    float in_slope = 1.0f;   // leaky-relu slope applied to the input while staging (1 = identity)
    int relu = 0;            // relu on the output
    float out_scale = 1.0f;  // y = (...) * out_scale
    int accumulate = 0;      // y += instead of y =
    int res_sub = 0;         // v = res - v instead of res + v   (coupling: x1 = (x1 - mean) * mask)
    int mask_before_res = 0; // apply the output mask to the conv result before the residual add (FFN: x + conv(..)*mask)
    int skip_init = 0;       // RESSKIP: first WN layer writes skip instead of accumulating
    int H = 0;               // GATE / RESSKIP: hidden channels
    int epi = EPI_STD;
    // ConvTranspose1d run as a stride-1 conv over `stride` polyphase filters: output channel
    // co' = r * shuf_cout + co of position i lands at y[co][i * shuf_s + r - shuf_p]  (0 <= n < shuf_T)
    int shuf_s = 0, shuf_p = 0, shuf_cout = 0, shuf_T = 0;
    int vec = 0;   // set by the launcher: input rows are 16-byte aligned -> 16-byte staging loads
    int yvec = 0;  // set by the launcher: output rows are 16-byte aligned -> vector stores (polyphase epilogue)
    int ovec = 0;  // set by the launcher: row-major epilogue through LDS (16-byte loads / stores of y, res)
    int ablate = 0;  // timing experiments only (MI355VITS_CONV_ABLATE): 1 no MFMA loop, 2 no staging, 4 no epilogue
    int fixed_rule = 0;  // kernel choice from the layer shape alone, never from T (flow / decoder convs: T = frames depends on
                         // what a row is batched with, and a row's bits must not)
    // launch_enc_conv_b3 only: the input channels in `ksplit` slices of 192, slice s of row b -> raw sums in
    // part[((s * B + b) * Cout + co) * T + t] (no epilogue; launch_layernorm adds them up); ksplit = 1: y with the epilogue
    int ksplit = 1;
    float* part = nullptr;
    int item_order = 0;  // kernels_rbc.cpp (set by its launchers): 1 = the persistent workgroups walk their items XCD-major (rbc_item)
};

// Short-sequence dense conv of the text encoder (q/k/v, o, FFN: T = phonemes) in MATH_BF16X3 / BF16W: 64 x 64 output tiles, a
// 192-channel slice of the input staged ONCE per workgroup as three bf16 planes (k_enc_b3, kernels_conv.cpp).  Cin % 192 == 0
// (or Cin = 96: the pointwise conv in front of a coupling layer's WaveNet), K in {1, 3}, dil 1, EPI_STD, wb3 = layout-1 planes.
bool enc_conv_b3_supported(int Cin, int Cout, int K, int dil);
int enc_conv_b3_slices(int Cin);  // workgroups along the input channels = ConvArgs.ksplit (Cin / 192; 1 for the 96-channel form)
void launch_enc_conv_b3(const ConvArgs& a, hipStream_t s);
// y = LN_c(res + conv1x1(x) + bias) in one launch (the attention block's o-proj + residual + LayerNorm; 192 -> 192 channels):
// `c` as for launch_enc_conv_b3 (x, wb3, bias, res, y, in_len, math; no output mask on the conv), then the LayerNorm's
// gamma / beta / eps and its output mask.  y may be the residual buffer.
bool enc_o_ln_supported(int Cin, int Cout, int K);
void launch_enc_o_ln(const ConvArgs& c, const float* gamma, const float* beta, const int* ln_out_len, float eps, hipStream_t s);
// Generic VALU/LDS-tiled Conv1d (any shape; reference implementation + fallback).
void launch_conv1d_generic(const ConvArgs& a, hipStream_t s);
// fp32-MFMA implicit-GEMM Conv1d (v_mfma_f32_32x32x2_f32).  Needs Cin even and packed weights.
void launch_conv1d_mfma(const ConvArgs& a, hipStream_t s);
bool conv1d_mfma_supported(int Cin, int Cout, int K, int dil);
// Host-side weight repack for the MFMA kernel: [Cout,Cin,K] -> [ceil(Cout/32)][K][Cin/2][64].
size_t mfma_packed_floats(int Cout, int Cin, int K);
void pack_conv_weights_mfma(const float* w, int Cout, int Cin, int K, float* out);
void regroup_packed_x4(const float* packed, size_t n_floats, float* out);  // fused MRF stage weights (Cin % 8 == 0)
// f32 weights split into three bf16 planes (w = h + m + l exactly, each rounded to nearest) in the A-fragment order of
// v_mfma_f32_32x32x16_bf16: [Cout/32][K][Cin/16][plane h,m,l][64 lanes][8 bf16]; lane l = (half l >> 5, row l & 31)
// holds the k-slots e < 4: channel 16G + half + 2e, e >= 4: 16G + 8 + half + 2(e - 4) — the channels a lane's two
// ds_read_b128 of a packed activation tile deliver.  Needs Cin % 16 == 0 and Cout % 32 == 0; sizes in 32-bit words.
// BF16W: BF16X3 with the weights' leading bf16 term only.  F16X2 (experimental: every BF16X3 kernel; every other kernel
// runs as in BF16X3): operands split into two fp16 terms (22 significant bits), three products per multiply-add.
enum MathMode { MATH_F32 = 0, MATH_BF16X3 = 1, MATH_BF16W = 2, MATH_F16X2 = 3 };
inline bool math_on_bf16(int m) { return m == MATH_BF16X3 || m == MATH_BF16W; }
// F16X2 scaling (exact powers of two): weights x 2^13 when packed (|w| < 7.99 required), activations x 2^4 when staged
// (saturating at |x| = 4094; both terms are normal halves for |x| >= 2^-7), accumulators therefore x 2^17
constexpr float F16X2_W_SCALE = 8192.0f, F16X2_X_SCALE = 16.0f, F16X2_ACC_SCALE = 131072.0f;
size_t f16x2_packed_words(int Cout, int Cin, int K);
// false (nothing written) when a weight is too large for the fixed scale
bool pack_conv_weights_f16x2(const float* w, int Cout, int Cin, int K, uint32_t* out, int layout = 0);  // layout as for bf16x3
size_t bf16x3_packed_words(int Cout, int Cin, int K);
void pack_conv_weights_bf16x3(const float* w, int Cout, int Cin, int K, uint32_t* out);
// general form: epi selects the tile -> output-channel map (EPI_GATE: tile pairs (c, H + c)), layout the k-slot ->
// channel map (0: packed activation tiles of the fused MRF stage, 1: staged planes of k_conv1d_b3)
size_t bf16x3_packed_words_mode(int Cout, int Cin, int K, int epi);
void pack_conv_weights_bf16x3_mode(const float* w, int Cout, int Cin, int K, int epi, int layout, uint32_t* out);
bool conv1d_b3_supported(int Cin, int Cout, int K, int dil, int T_hint);
// epi = EPI_GATE packs rows as (c, H + c) tile pairs (H = Cout / 2); otherwise identical to the above.
void pack_conv_weights_mfma_mode(const float* w, int Cout, int Cin, int K, int epi, float* out);

// ---------------------------------------------------------------- ConvTranspose1d (K = 2*stride polyphase or general)
struct ConvTArgs {
    const float* x = nullptr; long x_bs = 0; int x_ld = 0;  // [B,Cin,Tin]
    float* y = nullptr; long y_bs = 0; int y_ld = 0;        // [B,Cout,Tin*stride]
    const float* w = nullptr;     // [Cin,Cout,K]
    const float* bias = nullptr;  // [Cout]
    const int* in_len = nullptr;  // [B] valid input frames per row (rows of a batch end at their own length)
    int B = 1, Cin = 0, Cout = 0, Tin = 0, K = 0, stride = 0, pad = 0;
    float in_slope = 1.0f;
};
void launch_conv_transpose1d(const ConvTArgs& a, hipStream_t s);
// Polyphase repack: ConvTranspose1d weight [Cin,Cout,K] (+bias [Cout]) -> Conv1d weight
// [stride*Cout, Cin, taps] (+bias [stride*Cout]) with taps = ceil(K/stride); see ConvArgs::shuf_*.
int convt_taps(int K, int stride);
void convt_to_polyphase(const float* w, const float* bias, int Cin, int Cout, int K, int stride, float* w_out,
                        float* bias_out);

// ---------------------------------------------------------------- fused multi-receptive-field stage (K11)
// y = (1/n) * sum_j ResBlock2_j(x),  ResBlock2(k,(d1,d2)): x1 = x + conv_{k,d1}(lrelu(x)); x2 = x1 + conv_{k,d2}(lrelu(x1)).
// One workgroup keeps the x tile (+halo) and the x1 tile in LDS, runs all 2n convolutions on the fp32 matrix
// cores and writes y once: HBM traffic = read x + write y (vs ~7 tensor passes per resblock conv-by-conv).
constexpr int MRF_MAX_RB = 4;
struct MrfArgs {
    const float* x = nullptr; long x_bs = 0; int x_ld = 0;
    float* y = nullptr; long y_bs = 0; int y_ld = 0;
    const float* w[MRF_MAX_RB][2] = {};     // math 0: MFMA-packed x4 [C/32][K][C/8][64][4]; math 1: pack_conv_weights_bf16x3
    int math = 0;                           // MATH_F32 / MATH_BF16X3 (which matrix-core path; the weights must match)
    const float* bias[MRF_MAX_RB][2] = {};
    int k[MRF_MAX_RB] = {}, d1[MRF_MAX_RB] = {}, d2[MRF_MAX_RB] = {};
    int nrb = 0;
    const int* len = nullptr;  // [B] rows end at their own length
    int B = 1, C = 0, T = 0;
    int R = 0, ldx = 0, ld1 = 0, vec = 0;  // filled by the launcher (R = staging halo, rounded up to 4)
    float out_scale = 0.0f;  // 0: y = mean of the nrb resblocks; > 0: y = out_scale * sum (a stage fused only in part:
                             // the remaining resblocks are accumulated onto y by the conv-by-conv path)
    int ablate = 0;  // profiling only (MI355VITS_MRF_ABLATE): 1 = skip MFMA loops, 2 = skip staging, 4 = skip output
    int seg = 0;     // launch_mrf_s: columns per work item (a multiple of the kernel's step), from mrf_s_segment
    unsigned* clk = nullptr;  // -DMRFP_CLOCKS lab builds only: shader-clock stamps (kernels_mrfp.cpp)
    const int* len_host = nullptr;  // the host's copy of `len` (the engine has it: the per-stage lengths are made on the host): lets the
                                    // launcher count the column blocks that have work; nullptr with len != nullptr: read back (tests)
    int nvalid = 0;  // k_mrf_p: work items with work = sum over the rows of ceil(len / T_B), filled by the launcher
    int prio = 0;    // k_mrf_s: wave-priority mode of the conv2 waves (0 none, 1-3 a fixed s_setprio, 4-6 dynamic: see the kernel)
};
bool mrf_fused_supported(int C, int nrb, const int* k, const int* d1, const int* d2);
void launch_mrf_fused(MrfArgs a, hipStream_t s);
// MATH_BF16X3 with every element split once (kernels_mrfp.cpp): bf16 planes in LDS, the running conv's weight fragments in
// registers, v_mfma_f32_16x16x32_bf16 tiles.  w[][] = pack_conv_weights_p16 fragments.  C = 32; taps in {3, 5, 7}.
size_t p16_packed_words(int Cout, int Cin, int K);
void pack_conv_weights_p16(const float* w, int Cout, int Cin, int K, uint32_t* out);
bool mrf_p_supported(int C, int nrb, const int* k, const int* d1, const int* d2);
// sum over the rows of ceil(min(len, T) / block): the (row, column block) items of a ragged batch that have work.  len_host = the
// host's copy of the device array len (nullptr: copied back, a synchronising call — unit-test hooks only); len == nullptr: every row T
// extra: positions a row has beyond its length (polyphase upsamplers: len input positions -> len + 1 output positions)
int mrf_valid_items(const int* len_host, const int* len_dev, int B, int T, int block, int extra = 0);
void launch_mrf_p(MrfArgs a, hipStream_t s);
// The same stage, same bits, as a row sweep (kernels_mrfs.cpp): work item = (row, segment), one pass per resblock with the
// waves specialised by conv, every conv's fragments in registers for the whole segment, no halo recompute; y accumulates the
// resblocks in place.  mrf_s_segment: the segment length for a grid, 0 = the stage is too small for the sweep to pay.
bool mrf_s_supported(int C, int nrb, const int* k, const int* d1, const int* d2);
int mrf_s_segment(int C, int B, int T, int cus, const int* len_host = nullptr);
void launch_mrf_s(MrfArgs a, hipStream_t s);
// One dense conv of a 128-channel ResBlock2 (HiFi-GAN stage 0) in MATH_BF16X3 with all input channels resident in LDS (kernels_rbc.cpp):
// y (+)= (res + bias + conv(lrelu(x * mask))) * out_scale; a.w = pack_conv_weights_p16 fragments; K / dilation pairs of the "_low" voices.
bool rb_conv_supported(const ConvArgs& a);
void launch_rb_conv(ConvArgs a, hipStream_t s);
// The polyphase upsamplers 128 -> 64 (x 8) and 64 -> 32 (x 4) in the same form (k_ups_pl): a = the ConvArgs of the polyphase conv
// (shuf_*, Tin, T = Tin + 1), a.w = pack_conv_weights_p16n fragments of the [stride * Cout, Cin, 2] polyphase filter.
void pack_conv_weights_p16n(const float* w, int Cout, int Cin, int K, uint32_t* out);
bool ups_pl_supported(const ConvArgs& a);
void launch_ups_pl(ConvArgs a, hipStream_t s);
int current_device_cu_count();  // compute units of the current device (persistent grids), looked up once per device
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set once per (kernel, current device)
void set_max_dynamic_lds(const void* fn, int bytes);

// ---------------------------------------------------------------- fused WaveNet layer of the coupling flow (K8)
// u = tanh(in(h)[:H] + cond) * sigmoid(in(h)[H:] + cond); rs = res_skip(u); h' = (h + rs[:H]) * mask; skip += rs[H:]
struct WnArgs {
    const float* h_in = nullptr; float* h_out = nullptr; long h_bs = 0; int h_ld = 0;  // [B,H,T], ping-pong
    float* skip = nullptr; long s_bs = 0; int s_ld = 0;                                 // [B,H,T]
    const float* w_in = nullptr;   // MFMA-packed, gate tile map [2*H/32][K][H/2][64]
    const float* b_in = nullptr;   // [2H]
    const float* w_rs = nullptr;   // MFMA-packed [Crs/32][1][H/2][64]
    const float* b_rs = nullptr;   // [Crs]
    const float* cond = nullptr; long cond_bs = 0;  // this layer's [B][2H] conditioning slice or null
    const int* len = nullptr;
    int B = 1, H = 0, T = 0, K = 1, dil = 1, Crs = 0, skip_init = 0;
    int ldx = 0, vec = 0;  // filled by the launcher
    int ablate = 0;        // timing experiments only (MI355VITS_WN_ABLATE): 1 no MFMA loops, 2 no staging, 4 no stores
    int math = 0;          // launch_wn_layer_b3: MATH_BF16X3 or MATH_BF16W
};
bool wn_layer_fused_supported(int H, int K, int dil);
void launch_wn_layer(WnArgs a, hipStream_t s);
// MATH_BF16X3 form of the same layer (H = 192): w_in / w_rs are pack_conv_weights_bf16x3_mode(..., EPI_STD, layout 1)
// fragments (plain row order: tanh rows, then sigmoid rows); 96 time columns x all rows per workgroup.
bool wn_layer_b3_supported(int H, int K, int dil);
void launch_wn_layer_b3(WnArgs a, hipStream_t s);

// ---------------------------------------------------------------- decoder tail
// y = tanh(conv_post(lrelu_0.01(x * mask))) (Cout = 1, no bias) + per-utterance max|y| over valid samples;
// mask = t < valid_len[b]: every row of a batch is synthesised as if it were alone (zero padding at its own end).
void launch_conv_post_tanh(const float* x, long x_bs, int x_ld, const float* w, int Cin, int K, int B, int L,
                           const int* valid_len, float* audio, long audio_bs, unsigned* peak_bits, hipStream_t s);
// audio_float_to_int16 per utterance (utils.py:237-244) from the peaks found above.
// volume != 1: followed by audioop.mul(pcm, 2, volume) (tts.py:542-543): sample * volume in double, clipped to
// [-32768, 32767], rounded toward minus infinity.
void launch_pcm16(const float* audio, long audio_bs, const unsigned* peak_bits, const int* valid_len, int B, int L,
                  int16_t* pcm, long pcm_bs, hipStream_t s, double volume = 1.0);

// ---------------------------------------------------------------- encoder pieces
void launch_embed(const long long* ids, const int* len, const float* emb, int B, int T, int H, int num_symbols,
                  float scale, float* y, hipStream_t s);
// y = LN_c(x (+ res)) [* gelu] [* mask]; in place allowed (y == x).
struct LNArgs {
    const float* x = nullptr; const float* res = nullptr; float* y = nullptr;
    const float* gamma = nullptr; const float* beta = nullptr;
    const int* out_len = nullptr;
    int B = 1, C = 0, T = 0;
    int gelu = 0;
    const float* add_to = nullptr;  // y = add_to + f(LN(x))   (DDS residual) or null
    float eps = 1e-5f;
    // x as `nparts` raw partial sums of a conv split over its input channels (x + p * part_stride, added in order p = 0, 1, ..),
    // then the conv's epilogue: + bias[c], zero at t >= premask_len[b] (the FFN's mask before the residual); then + res, LN
    int nparts = 1;
    long part_stride = 0;
    const float* bias = nullptr;
    const int* premask_len = nullptr;
};
void launch_layernorm(const LNArgs& a, hipStream_t s);
// relative-position multi-head attention on packed qkv [B, 3H, T] -> out [B, H, T]
void launch_rel_attention(const float* qkv, const float* emb_rel_k, const float* emb_rel_v, const int* len,
                          int B, int T, int H, int n_heads, int window, float* out, hipStream_t s);
// same contract on the fp32 matrix cores (one wave per 32 query rows; T <= 512, window <= 15), see kernels_attn.cpp
bool rel_attention_mfma_supported(int T, int H, int n_heads, int window);
void launch_rel_attention_mfma(const float* qkv, const float* emb_rel_k, const float* emb_rel_v, const int* len,
                               int B, int T, int H, int n_heads, int window, float* out, hipStream_t s);

// ---------------------------------------------------------------- stochastic duration predictor pieces
// y = gelu(LN(dwconv_k,d(x * mask)))   (first half of a DDS layer, A.6)
void launch_dds_dwconv_ln_gelu(const float* x, const float* w, const float* bias, const float* gamma,
                               const float* beta, const int* len, int B, int C, int T, int K, int dil, float* y,
                               hipStream_t s);
// one whole DDS layer: y = x + gelu(LN2(conv1x1(gelu(LN1(dwconv(x * mask))))))  (x, y different buffers; w1x1 = packed f32
// A fragments of the 1x1 conv).  C in {32, 64, 128, 192, 256}.
bool dds_layer_fused_supported(int C);
void launch_dds_layer(const float* x, float* y, const float* dw_w, const float* dw_b, const float* g1, const float* b1,
                      const float* w1x1_packed, const float* bias1x1, const float* g2, const float* b2, const int* len, int B,
                      int C, int T, int K, int dil, hipStream_t s);
// the whole DDS stack with its pre (1x1 conv + cond, or ConvFlow's affine pre + g), its proj and, for a ConvFlow, the spline:
// one launch (kernels_misc.cpp, k_dds_stack).  C = 192, K = 3, the taps' total reach <= 16 columns.
constexpr int DDS_STACK_MAX_LAYERS = 4, DDS_STACK_W = 64, DDS_STACK_OFF = 16, DDS_STACK_K = 3;
enum { DDS_PRE_CONV = 0, DDS_PRE_AFFINE = 1 };
struct DdsStackArgs {
    const float* src = nullptr;    // PRE_CONV: input of the 1x1 pre conv [B, C, T]; PRE_AFFINE: g [B, C, T]
    int pre_mode = DDS_PRE_CONV;
    const float* pre_w = nullptr;  // PRE_CONV: packed f32 A fragments [C/32][C/2][64]; PRE_AFFINE: [C]
    const float* pre_b = nullptr;  // [C]
    const float* cond = nullptr;   // PRE_CONV: + cond[b * cond_bs + c] (or null)
    long cond_bs = 0;
    float* z = nullptr;            // PRE_AFFINE: z [B, 2, T]; channel zch feeds pre, the spline rewrites channel 1 - zch
    int zch = 0;
    int n_layers = 0, K = 3;       // dilation of layer i = K^i
    const float* dw_w[DDS_STACK_MAX_LAYERS] = {};
    const float* dw_b[DDS_STACK_MAX_LAYERS] = {};
    const float* g1[DDS_STACK_MAX_LAYERS] = {};
    const float* b1[DDS_STACK_MAX_LAYERS] = {};
    const float* w1x1[DDS_STACK_MAX_LAYERS] = {};     // packed f32 A fragments
    const float* bias1x1[DDS_STACK_MAX_LAYERS] = {};
    const float* g2[DDS_STACK_MAX_LAYERS] = {};
    const float* b2[DDS_STACK_MAX_LAYERS] = {};
    float* x_out = nullptr;        // the stack's x before proj [B, C, T] (tests) or null
    const float* proj_w = nullptr; // packed f32 A fragments of proj (rows beyond proj_cout zero) or null: no proj
    const float* proj_b = nullptr;
    int proj_cout = 0;
    float* out = nullptr;          // [B, proj_cout, T] or null (a ConvFlow's theta stays on the CU)
    int spline = 0, nb = 0;
    float tail = 0.0f, inv_sqrt_fc = 0.0f;
    const int* len = nullptr;
    int B = 1, T = 0;
    int math = MATH_F32;  // MATH_BF16X3 / BF16W: pre_w (conv mode), w1x1[], proj_w are layout-1 bf16 planes (k_dds_stack_b3)
    int ablate = 0;  // lab build only
};
bool dds_stack_supported(int C, int K, int n_layers, int proj_cout);
void launch_dds_stack(const DdsStackArgs& a, int C, hipStream_t s);
// h[b,c,t] = w[c] * z[b,ch,t] + bias[c] + g[b,c,t]      (ConvFlow.pre on one channel + conditioning)
void launch_convflow_pre(const float* z, int ch, const float* w, const float* bias, const float* g, int B, int C,
                         int T, float* h, hipStream_t s);
// z[b,1-ch] <- RQS^-1(z[b,1-ch]; theta[b,:,t]) inside the tail bound; then z *= mask     (A.7)
void launch_spline_inverse(float* z, int ch_x0, const float* theta, const int* len, int B, int T, int nbins,
                           float tail, float inv_sqrt_fc, hipStream_t s);
// z init: z[b,c,t] = noise * noise_w (Philox or injected)
void launch_sdp_noise(float* z, const float* injected, int B, int T, float noise_w, unsigned long long seed,
                      unsigned long long utt_base, hipStream_t s);
// most latent frames one utterance may have (4.2 M frames = 13.5 h of audio); larger / non-finite predictions are an error
constexpr int DURATION_FRAME_CAP = 1 << 22;
// EA^-1 on channel `ch` -> logw; durations: w = ceil(exp(logw)*mask*ls) (or forced); cum = inclusive scan;
// ylen = max(1, sum)
void launch_durations(const float* z, int ch, float ea_m, float ea_logs, const int* len, const int* forced, int B,
                      int T, float length_scale, float* logw, int* w_ceil, int* cum, int* ylen, hipStream_t s);

// ---------------------------------------------------------------- length regulator + prior sampling (K6, K7)
void launch_expand_prior(const float* stats /*[B,2I,Tx]: m_p | logs_p*/, const int* cum, const int* ylen,
                         const float* injected, int injected_frames, int B, int I, int Tx, int Ty,
                         float noise_scale, unsigned long long seed, unsigned long long utt_base, float* z,
                         hipStream_t s);

// ---------------------------------------------------------------- speaker conditioning (A.11)
// out[b, co] = W[co, :] . emb_g[sid[b], :] + bias[co]
void launch_speaker_cond(const float* emb_g, const long long* sid, const float* w, const float* bias, int B,
                         int gin, int Cout, float* out, hipStream_t s);

// misc
void launch_fill(float* p, float v, size_t n, hipStream_t s);
void launch_zero_row_tails(float* p, int B, int C, long T, const int* len, int factor, hipStream_t s);  // p[b, :, len[b] * factor:] = 0
void launch_mfma_selftest(float* out /*[32*32 + 16*16 + 16*16]*/, hipStream_t s);
// box probe (mi355vits_probe_device): L2-hit 16-byte stream of one table by every CU, L2-hit dependent-load chain, HBM copy
void launch_probe_l2_stream(const void* tbl, int n16, int reps, unsigned* sink, int grid, hipStream_t s);
void launch_probe_l2_latency(const unsigned* chain, int steps, unsigned nlines, unsigned* out, int grid, hipStream_t s);
void launch_probe_copy(const void* src, void* dst, long n16, int grid, hipStream_t s);
void launch_probe_l2_stream1(const void* tbl, int n16, int reps, unsigned* sink, int grid, hipStream_t s);
void launch_probe_l2_mixed(const void* tbl, int n16, int reps, const void* src, void* dst, long slice16, unsigned* sink, int grid, hipStream_t s);

}  // namespace m355
