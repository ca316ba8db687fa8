// engine.h — host side of the MI355X VITS engine: weight container, device arenas, per-kernel
// event profiler and the launch sequence of one synthesis call.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mi355vits.h"
#include "kernels.h"

namespace m355 {

struct EngineError : std::runtime_error {
    int code;
    EngineError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---------------------------------------------------------------- .m355 container (mimic3_amd/weights.py)
struct HostTensor {
    std::vector<int> dims;
    const float* data = nullptr;
    size_t count = 0;
};
struct WeightsFile {
    mi355vits_config cfg{};
    std::map<std::string, HostTensor> tensors;
    std::vector<unsigned char> storage;      // owns the bytes when loaded from a file
    void parse(const void* blob, size_t n);  // tensors point into `blob`
    void load(const std::string& path);
    const HostTensor& get(const std::string& name, std::initializer_list<int> dims) const;
};

// ---------------------------------------------------------------- grow-only bump allocator in HBM
class DeviceArena {
  public:
    ~DeviceArena();
    void reserve(size_t bytes, hipStream_t s);  // may reallocate (synchronises the stream first)
    void reset() { off_ = 0; }
    template <typename T> T* alloc(size_t n) { return reinterpret_cast<T*>(alloc_bytes(n * sizeof(T))); }
    size_t capacity() const { return cap_; }
    size_t used() const { return off_; }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }

  private:
    void* alloc_bytes(size_t bytes);
    unsigned char* base_ = nullptr;
    size_t cap_ = 0, off_ = 0;
};

// ---------------------------------------------------------------- HIP-event profiler on the engine stream
struct Profiler {
    struct Rec { int name_id; hipEvent_t a, b; double flops, bytes; };
    bool enabled = false;
    std::vector<std::string> names;
    std::unordered_map<std::string, int> ids;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipStream_t stream = nullptr;
    ~Profiler();
    int begin(const char* name, double flops, double bytes);
    void end(int rec);
    void clear();
    std::string report();  // synchronises
  private:
    hipEvent_t get_event();
};
struct ProfScope {
    Profiler& p;
    int rec;
    ProfScope(Profiler& prof, const char* name, double flops = 0, double bytes = 0) : p(prof), rec(-1) {
        if (p.enabled) rec = p.begin(name, flops, bytes);
    }
    ~ProfScope() {
        if (rec >= 0) p.end(rec);
    }
};

struct Tap {
    std::string name;
    float* dev;
    std::vector<int64_t> dims;
    size_t count;
};

constexpr size_t NO_OFF = ~size_t(0);

// one dense Conv1d's parameters, as offsets (floats) into the device weight arena
struct ConvW {
    size_t raw = NO_OFF;     // [Cout,Cin,K]
    size_t packed = NO_OFF;  // MFMA fragment order (absent when Cin is odd)
    size_t packed4 = NO_OFF; // same records regrouped [tile][tap][4 pairs][lane][4] for 16-byte A loads (fused MRF stage)
    size_t packed_b3 = NO_OFF;  // three bf16 planes in bf16-MFMA fragment order (pack_conv_weights_bf16x3), 32-bit words
    size_t packed_b3s = NO_OFF; // the same for the staged split-bf16 conv kernel (layout 1, this conv's tile map)
    size_t packed_b3w = NO_OFF; // WaveNet in-layer convs: layout 1, 32-row tiles of (16 tanh rows, their 16 sigmoid rows) (fused split-bf16 layer kernel)
    size_t packed_h2 = NO_OFF;    // fused-MRF convs: two fp16 planes, weights x 2^13 (MATH_F16X2), when every |w| < 7.99
    size_t packed_p = NO_OFF;     // fused-MRF convs (Cin = Cout, taps 3 / 5 / 7): pack_conv_weights_p16 fragments (k_mrf_p)
    size_t packed_h2s = NO_OFF;   // WaveNet layer convs: the same in plane order (layout 1), plain rows (gate convs: packed_b3w's tile order)
    size_t bias = NO_OFF;
    int Cout = 0, Cin = 0, K = 1;
    int epi = EPI_STD;  // tile map the packed copy was built for
};

// ---------------------------------------------------------------- one weight replica on one device
// Uploaded once per (voice, device); every engine handle ("lane") created on that device through mi355vits_clone shares
// it, so N lanes cost N workspaces but ONE copy of the weights.
struct Model {
    int device = 0;
    float* dev_weights = nullptr;
    std::unordered_map<std::string, ConvW> convs;
    std::unordered_map<std::string, size_t> vecs;
    float ea_m[2] = {0, 0}, ea_logs[2] = {0, 0};
    bool flow_reversed_out = false;
    size_t bytes = 0;
    ~Model();
};

// ---------------------------------------------------------------- the engine
class Engine {
  public:
    Engine(const WeightsFile& wf, int device);
    explicit Engine(const Engine& lane0);  // another lane on lane0's device, sharing its weight replica
    ~Engine();
    // diagnostics (mi355vits_probe_weights): how fast every CU together streams 2.6 MB windows of THIS replica's weight arena out of
    // the L2 — eight loads in flight per lane (bandwidth) and one (latency per fragment); min / median / max over the windows
    void probe_weights(double out[8]);
    const std::shared_ptr<Model>& model() const { return model_; }
    // device pointers of the last run's results (valid until the next run on this handle); for device-side gathers
    void device_buffers(const int16_t** pcm, const float** audio, long* row_stride, int* batch, const int** dev_lengths);
    void run(const mi355vits_run_args& args, mi355vits_result* out);
    void fetch(uint32_t want, mi355vits_result* out);
    const mi355vits_config& config() const { return cfg_; }
    void set_math(int mode);
    int math() const { return math_; }
    Profiler& profiler() { return prof_; }
    float last_run_ms();
    long get_tap(const std::string& name, float* out, size_t cap, int64_t dims[4], long row0 = 0, long nrows = -1);
    std::string list_taps() const;
    std::string last_error;
    std::mutex mu;

  private:
    void construct(const WeightsFile& wf, int device);
    void open_device(int device);
    void release() noexcept;
    // weight staging
    size_t stage(const float* p, size_t n);
    const ConvW& add_conv(const WeightsFile& wf, const std::string& key, const std::string& tensor, int Cout, int Cin,
                          int K, bool bias, int epi = EPI_STD);
    const ConvW& add_conv_data(const std::string& key, const std::vector<float>& w, const std::vector<float>* bias,
                               int Cout, int Cin, int K, int epi = EPI_STD);
    void add_vec(const WeightsFile& wf, const std::string& name, std::initializer_list<int> dims);
    const float* vec(const std::string& name) const;
    const float* P(size_t off) const { return off == NO_OFF ? nullptr : model_->dev_weights + off; }
    const ConvW& cw(const std::string& key) const;

    // launch helpers
    void conv(const char* label, const ConvW& w, ConvArgs a);
    bool rbc_ok(const ConvW& w, const ConvArgs& a) const;    // this conv runs on k_rb_conv (128-channel resblock conv, MATH_BF16X3)
    bool enc_gemm(const ConvW& w, const ConvArgs& a) const;  // this conv runs on k_enc_b3 (phoneme-sized, split-bf16)
    // row_len / factor: frame-resolution taps of a ragged batch are zeroed past len[b] * factor (those columns are not computed)
    void tap(const char* name, const float* dev, std::initializer_list<int64_t> dims, const int* row_len = nullptr, int factor = 1);
    void text_encoder(int B, int Tx);
    void duration_predictor(int B, int Tx, const mi355vits_run_args& args);
    void dds(const std::string& key, float* X, float* Y1, float* Y2, int B, int T);
    void flow_and_decoder(int B, int Ty, const mi355vits_run_args& args);
    void copy_out(uint32_t want, mi355vits_result* out);

    mi355vits_config cfg_{};
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_start_ = nullptr, ev_end_ = nullptr;
    bool timed_ = false;
    bool phase_b_ = false;       // inside flow_and_decoder (see Engine::conv)
    bool force_generic_ = false;
    int b3_min_work_ = 256;      // MATH_BF16X3: smallest K * Cin routed to the staged split-bf16 conv kernel
    bool no_flow_gemm_ = false;  // MI355VITS_NO_FLOW_GEMM=1: flow.pre / flow.post on the general conv kernels (A/B + fallback)
    bool no_enc_o_ln_ = false;   // MI355VITS_NO_ENC_O_LN=1: o-proj and its LayerNorm as two launches (A/B + fallback)
    bool no_enc_gemm_ = false;   // MI355VITS_NO_ENC_GEMM=1: phoneme-sized convs on the general conv kernels (A/B + fallback)
    bool no_dds_stack_b3_ = false;  // MI355VITS_NO_DDS_STACK_B3=1: the f32-MFMA form of the stack in every math mode (A/B)
    bool no_dds_stack_ = false;  // MI355VITS_NO_DDS_STACK=1: one launch per DDS layer / pre / proj / spline (A/B + fallback)
    bool no_fused_dds_ = false;  // MI355VITS_NO_FUSED_DDS=1: DDS layers as three launches (A/B + fallback)
    // the math mode of the kernels that have no fp16 form of their own: in F16X2 they run as BF16X3 (the kernels that do —
    // fused MRF stages, fused WaveNet layers, staged convs, upsamplers — are switched where they are launched)
    int kmath() const { return math_ == MATH_F16X2 ? (int)MATH_BF16X3 : math_; }
    // the text side (phase A: encoder, duration predictor) never rounds its weights: ceil(exp(logw) * length_scale) is
    // discontinuous, so in MATH_BF16W it runs the exact three-term split and utterance lengths equal the default mode's
    int tmath() const { return kmath() == MATH_BF16W ? (int)MATH_BF16X3 : kmath(); }
    int pmath() const { return phase_b_ ? kmath() : tmath(); }  // math of the launch helpers shared by both phases
    bool no_f16x2_convs_ = false;  // MI355VITS_F16X2_NO_CONVS=1: in MATH_F16X2 keep the staged convs / upsamplers on bf16x3
    bool enc_b3_ = true;           // the encoder's wide FFN conv on the split-bf16 staged kernel (MI355VITS_NO_ENC_B3=1: f32 kernel)
    bool no_rbc_ = false;        // MI355VITS_NO_RBC=1: the 128-channel stage on k_mrf_fused + the staged conv (A/B against k_rb_conv)
    bool no_mrf_p_ = false;      // MI355VITS_NO_MRF_P=1: keep the on-the-fly split MRF kernel (A/B against k_mrf_p)
    bool wn_b3_ = false;         // MATH_BF16X3: WaveNet layers as two staged split-bf16 convs instead of the fused f32 layer
    int math_ = MATH_BF16X3;     // which matrix-core path the dense convs take (include/mi355vits.h: MI355VITS_MATH_*)
    bool no_fused_wn_ = false;   // MI355VITS_NO_FUSED_WN=1: in-layer + res/skip as two launches (A/B + fallback)
    bool no_fused_mrf_ = false;  // MI355VITS_NO_FUSED_MRF=1: conv-by-conv resblocks (A/B + fallback)
    Profiler prof_;

    std::vector<float> host_stage_;
    std::shared_ptr<Model> model_;

    DeviceArena arena_a_, arena_b_, arena_taps_;
    std::vector<Tap> taps_;
    bool taps_on_ = false;
    int B_ = 0, Tx_ = 0, Ty_ = 0;
    long L_ = 0;
    bool have_result_ = false, have_pcm_ = false;
    double pcm_volume_ = 1.0;
    // phase-A buffers (sized by B, Tx)
    long long *d_ids_ = nullptr, *d_sid_ = nullptr;
    int *d_len_ = nullptr, *d_wceil_ = nullptr, *d_cum_ = nullptr, *d_ylen_ = nullptr, *d_alen_ = nullptr,
        *d_forced_ = nullptr;
    float *d_x_ = nullptr, *d_x2_ = nullptr, *d_qkv_ = nullptr, *d_att_ = nullptr, *d_ffn_ = nullptr, *d_part_ = nullptr,
          *d_stats_ = nullptr;
    float *d_h_ = nullptr, *d_d0_ = nullptr, *d_d1_ = nullptr, *d_d2_ = nullptr, *d_theta_ = nullptr, *d_z2_ = nullptr,
          *d_logw_ = nullptr, *d_noise_w_ = nullptr;
    float *d_cond_dp_ = nullptr, *d_cond_dec_ = nullptr;
    std::vector<float*> d_cond_flow_;
    // phase-B buffers (sized by B, Ty)
    float* d_fh2_ = nullptr;  // second h buffer of the fused WaveNet layers (ping-pong)
    float *d_z_ = nullptr, *d_fh_ = nullptr, *d_fu_ = nullptr, *d_fskip_ = nullptr, *d_noise_z_ = nullptr;
    float *d_bufA_ = nullptr, *d_bufB_ = nullptr, *d_bufT_ = nullptr, *d_bufC_ = nullptr;
    float* d_audio_ = nullptr;
    int16_t* d_pcm_ = nullptr;
    unsigned* d_peaks_ = nullptr;
    int* d_slen_ = nullptr;  // [n_upsamples + 1][B] valid frames per decoder stage
    std::vector<int> h_ylen_;
    std::vector<unsigned char> h_in_;  // the call's host inputs, laid out like their device block (one upload)
    std::vector<int> h_slen_;         // per-stage valid lengths + audio lengths (one upload)
};

void free_result_impl(mi355vits_result* r);

}  // namespace m355
