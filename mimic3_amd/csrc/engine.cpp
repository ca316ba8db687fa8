// engine.cpp — weight upload (with the load-time repacks the kernels want) and the launch sequence of one
// synthesis call: text encoder -> stochastic duration predictor -> length regulator -> flow^-1 -> HiFi-GAN
// -> tanh / peak / int16.  Everything runs on one HIP stream; the only host round trip inside a call is
// the 4*B-byte read of the frame counts that size the second half of the graph.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace m355 {

// =================================================================================================
// .m355 container
// =================================================================================================
namespace {
struct Reader {
    const unsigned char* p;
    size_t n, pos = 0;
    void need(size_t k) const {
        if (pos + k > n) throw EngineError(MI355VITS_ERR_FORMAT, "weight container truncated");
    }
    template <typename T> T get() {
        need(sizeof(T));
        T v;
        memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
};
}  // namespace

void WeightsFile::parse(const void* blob, size_t n) {
    Reader r{static_cast<const unsigned char*>(blob), n};
    r.need(8);
    if (memcmp(r.p, "M355VITS", 8) != 0) throw EngineError(MI355VITS_ERR_FORMAT, "not an M355VITS weight container");
    r.pos = 8;
    const uint32_t version = r.get<uint32_t>();
    const uint32_t clen = r.get<uint32_t>();
    if (version != 1) throw EngineError(MI355VITS_ERR_FORMAT, "unsupported container version");
    if (clen != sizeof(mi355vits_config)) throw EngineError(MI355VITS_ERR_FORMAT, "config block size mismatch");
    r.need(clen);
    memcpy(&cfg, r.p + r.pos, clen);
    r.pos += clen;
    const uint32_t nt = r.get<uint32_t>();
    struct Ent { std::string name; std::vector<int> dims; uint64_t off; };
    std::vector<Ent> ents;
    for (uint32_t i = 0; i < nt; ++i) {
        const uint16_t nl = r.get<uint16_t>();
        r.need(nl);
        Ent e;
        e.name.assign(reinterpret_cast<const char*>(r.p + r.pos), nl);
        r.pos += nl;
        const uint32_t nd = r.get<uint32_t>();
        if (nd > 8) throw EngineError(MI355VITS_ERR_FORMAT, "tensor rank too large");
        for (uint32_t d = 0; d < nd; ++d) {
            const uint32_t v = r.get<uint32_t>();
            if (v > (1u << 28)) throw EngineError(MI355VITS_ERR_FORMAT, "implausible tensor dimension in " + e.name);
            e.dims.push_back((int)v);
        }
        e.off = r.get<uint64_t>();
        ents.push_back(std::move(e));
    }
    const uint64_t dbytes = r.get<uint64_t>();
    r.pos += (64 - r.pos % 64) % 64;
    if (r.pos > n || dbytes > n - r.pos) throw EngineError(MI355VITS_ERR_FORMAT, "weight container truncated (data section)");
    for (auto& e : ents) {
        size_t cnt = 1;
        for (int d : e.dims) {
            cnt *= (size_t)d;
            if (cnt > (size_t(1) << 34)) throw EngineError(MI355VITS_ERR_FORMAT, "tensor " + e.name + " too large");
        }
        if ((e.off & 3) || e.off > dbytes || cnt * 4 > dbytes - e.off)
            throw EngineError(MI355VITS_ERR_FORMAT, "tensor " + e.name + " out of bounds");
        HostTensor t;
        t.dims = e.dims;
        t.count = cnt;
        t.data = reinterpret_cast<const float*>(r.p + r.pos + e.off);
        tensors[e.name] = t;
    }
}

void WeightsFile::load(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw EngineError(MI355VITS_ERR_IO, "cannot open weight file: " + path);
    const std::streamsize sz = f.tellg();
    f.seekg(0);
    storage.resize((size_t)sz + 64);
    // keep float data 4-byte aligned: the data section starts on a 64-byte file offset
    unsigned char* base = storage.data();
    base += (64 - reinterpret_cast<uintptr_t>(base) % 64) % 64;
    if (!f.read(reinterpret_cast<char*>(base), sz)) throw EngineError(MI355VITS_ERR_IO, "cannot read weight file: " + path);
    parse(base, (size_t)sz);
}

const HostTensor& WeightsFile::get(const std::string& name, std::initializer_list<int> dims) const {
    auto it = tensors.find(name);
    if (it == tensors.end()) throw EngineError(MI355VITS_ERR_FORMAT, "missing tensor: " + name);
    if (it->second.dims != std::vector<int>(dims)) {
        std::ostringstream os;
        os << "tensor " << name << " has shape [";
        for (int d : it->second.dims) os << d << ",";
        os << "] expected [";
        for (int d : dims) os << d << ",";
        os << "]";
        throw EngineError(MI355VITS_ERR_FORMAT, os.str());
    }
    return it->second;
}

// =================================================================================================
// arena / profiler
// =================================================================================================
DeviceArena::~DeviceArena() {
    if (base_) (void)hipFree(base_);
}
void DeviceArena::reserve(size_t bytes, hipStream_t s) {
    if (bytes <= cap_) return;
    HIP_CHECK(hipStreamSynchronize(s));
    if (base_) HIP_CHECK(hipFree(base_));
    base_ = nullptr;
    cap_ = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) throw EngineError(MI355VITS_ERR_NOMEM, "out of device memory (workspace)");
    base_ = static_cast<unsigned char*>(p);
    cap_ = want;
    // zero-filled once per (re)allocation: the decoder / flow kernels never compute items past a row's length in a ragged batch
    // (kernels_mrfp.cpp next_item), so those columns keep what the arena held before — which must be finite, not fresh-memory NaNs
    HIP_CHECK(hipMemsetAsync(base_, 0, want, s));
}
void* DeviceArena::alloc_bytes(size_t bytes) {
    const size_t need = padded(bytes);
    if (off_ + need > cap_) throw EngineError(MI355VITS_ERR_INTERNAL, "workspace arena overflow (sizing bug)");
    void* p = base_ + off_;
    off_ += need;
    return p;
}

Profiler::~Profiler() {
    for (auto e : pool) (void)hipEventDestroy(e);
    for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
}
hipEvent_t Profiler::get_event() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}
int Profiler::begin(const char* name, double flops, double bytes) {
    auto it = ids.find(name);
    int id;
    if (it == ids.end()) {
        id = (int)names.size();
        names.emplace_back(name);
        ids[name] = id;
    } else {
        id = it->second;
    }
    Rec r{id, get_event(), get_event(), flops, bytes};
    HIP_CHECK(hipEventRecord(r.a, stream));
    recs.push_back(r);
    return (int)recs.size() - 1;
}
void Profiler::end(int rec) { (void)hipEventRecord(recs[rec].b, stream); }
void Profiler::clear() {
    (void)hipStreamSynchronize(stream);
    for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
    recs.clear();
}
std::string Profiler::report() {
    HIP_CHECK(hipStreamSynchronize(stream));
    struct Agg { long calls = 0; double ms = 0, flops = 0, bytes = 0; };
    std::vector<Agg> agg(names.size());
    for (auto& r : recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0;
        Agg& a = agg[r.name_id];
        a.calls++;
        a.ms += ms;
        a.flops += r.flops;
        a.bytes += r.bytes;
    }
    std::ostringstream os;
    os.precision(9);
    for (size_t i = 0; i < names.size(); ++i)
        if (agg[i].calls) os << names[i] << " " << agg[i].calls << " " << agg[i].ms << " " << agg[i].flops << " " << agg[i].bytes << "\n";
    return os.str();
}

// =================================================================================================
// engine: construction / weight upload
// =================================================================================================
size_t Engine::stage(const float* p, size_t n) {
    const size_t off = (host_stage_.size() + 63) & ~size_t(63);
    host_stage_.resize(off + n);
    if (n) memcpy(host_stage_.data() + off, p, n * sizeof(float));
    return off;
}

const ConvW& Engine::add_conv_data(const std::string& key, const std::vector<float>& w, const std::vector<float>* bias,
                                   int Cout, int Cin, int K, int epi) {
    if (w.size() != (size_t)Cout * Cin * K) throw EngineError(MI355VITS_ERR_INTERNAL, "add_conv_data size: " + key);
    ConvW c;
    c.Cout = Cout; c.Cin = Cin; c.K = K; c.epi = epi;
    c.raw = stage(w.data(), w.size());
    if (bias) c.bias = stage(bias->data(), bias->size());
    if (conv1d_mfma_supported(Cin, Cout, K, 1)) {
        std::vector<float> pk(mfma_packed_floats(Cout, Cin, K), 0.0f);
        pack_conv_weights_mfma_mode(w.data(), Cout, Cin, K, epi == EPI_GATE ? EPI_GATE : EPI_STD, pk.data());
        c.packed = stage(pk.data(), pk.size());
        if (key.rfind("dec.rb.", 0) == 0 && epi == EPI_STD && Cin % 8 == 0 && Cout % 32 == 0) {
            // fused MRF stage: a lane's A fragments of four consecutive channel pairs side by side
            std::vector<float> p4(pk.size());
            regroup_packed_x4(pk.data(), pk.size(), p4.data());
            c.packed4 = stage(p4.data(), p4.size());
        }
        if (key.rfind("dec.rb.", 0) == 0 && epi == EPI_STD && Cin % 16 == 0 && Cout % 32 == 0) {
            // the same weights as three bf16 planes for the split-operand path (MATH_BF16X3)
            std::vector<uint32_t> b3(bf16x3_packed_words(Cout, Cin, K));
            pack_conv_weights_bf16x3(w.data(), Cout, Cin, K, b3.data());
            static_assert(sizeof(uint32_t) == sizeof(float), "bit patterns travel in the float arena");
            c.packed_b3 = stage(reinterpret_cast<const float*>(b3.data()), b3.size());
            // ... and as two fp16 planes (MATH_F16X2, experimental), unless a weight is too large for the fixed scale
            std::vector<uint32_t> h2(f16x2_packed_words(Cout, Cin, K));
            if (pack_conv_weights_f16x2(w.data(), Cout, Cin, K, h2.data()))
                c.packed_h2 = stage(reinterpret_cast<const float*>(h2.data()), h2.size());
        }
        if (key.rfind("dec.rb.", 0) == 0 && epi == EPI_STD && Cin == Cout && Cin % 32 == 0 && (K == 3 || K == 5 || K == 7)) {
            // ... and in the fragment order of k_mrf_p (16-row tiles, 32-channel k-groups)
            std::vector<uint32_t> pp(p16_packed_words(Cout, Cin, K));
            pack_conv_weights_p16(w.data(), Cout, Cin, K, pp.data());
            c.packed_p = stage(reinterpret_cast<const float*>(pp.data()), pp.size());
        }
        if (key.rfind("dec.ups.", 0) == 0 && epi == EPI_STD && (Cin == 64 || Cin == 128 || Cin == 256) && Cout % 128 == 0 && K == 2) {
            // polyphase upsamplers 128 -> 64 / 64 -> 32: fragments of 16-row tiles in natural row order (k_ups_pl)
            std::vector<uint32_t> pp(p16_packed_words(Cout, Cin, K));
            pack_conv_weights_p16n(w.data(), Cout, Cin, K, pp.data());
            c.packed_p = stage(reinterpret_cast<const float*>(pp.data()), pp.size());
        }
        if (Cin % 32 == 0 && (epi == EPI_GATE ? (Cout / 2) % 32 == 0 : true)) {
            // ... and for the staged split-bf16 kernel (every dense conv with a multiple of 32 input channels)
            const int pe = epi == EPI_GATE ? EPI_GATE : EPI_STD;
            std::vector<uint32_t> b3(bf16x3_packed_words_mode(Cout, Cin, K, pe));
            pack_conv_weights_bf16x3_mode(w.data(), Cout, Cin, K, pe, 1, b3.data());
            c.packed_b3s = stage(reinterpret_cast<const float*>(b3.data()), b3.size());
            // k_wn_layer_b3 gates in registers: 32-row tile q of a gate conv = the tanh rows of channels 16 q .. 16 q + 15, then their sigmoid
            // rows (a lane of the 32 x 32 accumulator tile then holds both halves of its eight channels; kernels_wn.cpp)
            std::vector<float> wg;
            if (epi == EPI_GATE && Cout % 32 == 0) {
                const int Hh = Cout / 2;
                const size_t row = (size_t)Cin * K;
                wg.resize(w.size());
                for (int q = 0; q < Cout / 32; ++q)
                    for (int r = 0; r < 32; ++r) {
                        const int src = (r < 16 ? 0 : Hh) + 16 * q + (r & 15);
                        memcpy(wg.data() + ((size_t)32 * q + r) * row, w.data() + (size_t)src * row, row * sizeof(float));
                    }
                std::vector<uint32_t> bw(bf16x3_packed_words_mode(Cout, Cin, K, EPI_STD));
                pack_conv_weights_bf16x3_mode(wg.data(), Cout, Cin, K, EPI_STD, 1, bw.data());
                c.packed_b3w = stage(reinterpret_cast<const float*>(bw.data()), bw.size());
            }
            if (Cin % 16 == 0 && Cout % 32 == 0) {  // MATH_F16X2: WaveNet layers (gate convs: the tile order above), staged convs, polyphase upsamplers (plain rows)
                std::vector<uint32_t> h2(f16x2_packed_words(Cout, Cin, K));
                if (pack_conv_weights_f16x2(wg.empty() ? w.data() : wg.data(), Cout, Cin, K, h2.data(), 1))
                    c.packed_h2s = stage(reinterpret_cast<const float*>(h2.data()), h2.size());
            }
        }
    }
    return model_->convs[key] = c;
}

const ConvW& Engine::add_conv(const WeightsFile& wf, const std::string& key, const std::string& tensor, int Cout, int Cin,
                              int K, bool bias, int epi) {
    const HostTensor& w = wf.get(tensor + ".weight", {Cout, Cin, K});
    std::vector<float> wv(w.data, w.data + w.count);
    if (bias) {
        const HostTensor& b = wf.get(tensor + ".bias", {Cout});
        std::vector<float> bv(b.data, b.data + b.count);
        return add_conv_data(key, wv, &bv, Cout, Cin, K, epi);
    }
    return add_conv_data(key, wv, nullptr, Cout, Cin, K, epi);
}

void Engine::add_vec(const WeightsFile& wf, const std::string& name, std::initializer_list<int> dims) {
    const HostTensor& t = wf.get(name, dims);
    model_->vecs[name] = stage(t.data, t.count);
}
const float* Engine::vec(const std::string& name) const {
    auto it = model_->vecs.find(name);
    if (it == model_->vecs.end()) throw EngineError(MI355VITS_ERR_INTERNAL, "unknown tensor: " + name);
    return model_->dev_weights + it->second;
}
const ConvW& Engine::cw(const std::string& key) const {
    auto it = model_->convs.find(key);
    if (it == model_->convs.end()) throw EngineError(MI355VITS_ERR_INTERNAL, "unknown conv: " + key);
    return it->second;
}

static std::string S(const char* fmt, int a = 0, int b = 0, int c = 0) {
    char buf[160];
    snprintf(buf, sizeof(buf), fmt, a, b, c);
    return buf;
}

static void validate_config(const mi355vits_config& c) {
    auto bad = [](const std::string& m) { throw EngineError(MI355VITS_ERR_FORMAT, "invalid voice config: " + m); };
    if (c.num_symbols < 1 || c.n_speakers < 1) bad("num_symbols / n_speakers");
    if (c.hidden_channels < 2 || c.hidden_channels % 2 || c.inter_channels < 2 || c.inter_channels % 2) bad("channel counts must be even");
    if (c.n_heads < 1 || c.hidden_channels % c.n_heads) bad("hidden_channels % n_heads");
    if (c.resblock != 1 && c.resblock != 2) bad("resblock must be 1 or 2");
    if (c.n_upsamples < 1 || c.n_upsamples > MI355VITS_MAX_STAGES) bad("n_upsamples");
    if (c.n_resblock_kernels < 1 || c.n_resblock_kernels > MI355VITS_MAX_STAGES) bad("n_resblock_kernels");
    long hop = 1;
    int ch = c.upsample_initial_channel;
    for (int i = 0; i < c.n_upsamples; ++i) {
        if (c.upsample_rates[i] < 1 || c.upsample_kernel_sizes[i] < c.upsample_rates[i]) bad("upsample stage");
        if ((c.upsample_kernel_sizes[i] - c.upsample_rates[i]) % 2) bad("upsample kernel - rate must be even");
        if (ch % 2) bad("upsample channels must halve evenly");
        ch /= 2;
        hop *= c.upsample_rates[i];
    }
    if (ch % 2) bad("decoder channel counts must be even");
    if (hop != c.hop_length) bad("hop_length must equal the product of upsample_rates");
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
        if (c.resblock_kernel_sizes[j] < 1 || c.resblock_kernel_sizes[j] % 2 == 0) bad("resblock kernel sizes must be odd");
        if (c.resblock_n_dilations[j] < 1 || c.resblock_n_dilations[j] > MI355VITS_MAX_STAGES) bad("resblock dilations");
    }
    if (c.n_speakers > 1 && c.gin_channels < 1) bad("multi-speaker voice needs gin_channels");
    if (c.dp_num_bins < 2 || c.dp_num_bins > 16) bad("dp_num_bins");
    if (c.dp_n_flows < 2 || c.flow_n_flows < 1 || c.flow_wn_layers < 1) bad("flow depth");
    if (c.flow_wn_kernel % 2 == 0 || c.dp_kernel_size % 2 == 0) bad("flow / dp kernels must be odd");
    if (c.window_size < 0 || c.n_layers < 1 || c.kernel_size < 1) bad("encoder");
    // plausibility caps: a corrupted header must not turn into absurd loops or allocations
    if (c.num_symbols > (1 << 20) || c.n_speakers > (1 << 20) || c.hidden_channels > 8192 || c.inter_channels > 8192 ||
        c.filter_channels < 1 || c.filter_channels > 32768 || c.upsample_initial_channel < 2 || c.upsample_initial_channel > 8192 ||
        c.gin_channels < 0 || c.gin_channels > 8192)
        bad("channel / symbol counts out of range");
    if (c.n_layers > 64 || c.kernel_size > 63 || c.window_size > 15 || c.flow_n_flows > 32 || c.flow_wn_layers > 32 ||
        c.flow_wn_kernel < 1 || c.flow_wn_kernel > 63 || c.flow_wn_dilation_rate < 1 || c.flow_wn_dilation_rate > 8 ||
        c.dp_n_flows > 16 || c.dp_dds_layers < 1 || c.dp_dds_layers > 8 || c.dp_kernel_size < 1 || c.dp_kernel_size > 63 ||
        !(c.dp_tail_bound > 0.0f) || c.hop_length > (1 << 16))
        bad("depth / kernel sizes out of range");
    for (int i = 0; i < c.n_upsamples; ++i)
        if (c.upsample_rates[i] > 64 || c.upsample_kernel_sizes[i] > 256) bad("upsample stage out of range");
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
        if (c.resblock_kernel_sizes[j] > 63) bad("resblock kernel size out of range");
        for (int m = 0; m < c.resblock_n_dilations[j]; ++m)
            if (c.resblock_dilations[j * MI355VITS_MAX_STAGES + m] < 1 || c.resblock_dilations[j * MI355VITS_MAX_STAGES + m] > 256)
                bad("resblock dilation out of range");
    }
}

Engine::Engine(const WeightsFile& wf, int device) : cfg_(wf.cfg), device_(device) {
    // a throwing constructor never runs the destructor: stream, events and the weight arena are released here
    try {
        construct(wf, device);
    } catch (...) {
        release();
        throw;
    }
}

Model::~Model() {
    if (dev_weights) {
        (void)hipSetDevice(device);
        (void)hipFree(dev_weights);
    }
}

Engine::Engine(const Engine& lane0) : cfg_(lane0.cfg_), device_(lane0.device_) {
    try {
        open_device(device_);
        model_ = lane0.model_;
        math_ = lane0.math_;
        b3_min_work_ = lane0.b3_min_work_;
        wn_b3_ = lane0.wn_b3_;
        no_mrf_p_ = lane0.no_mrf_p_;
        no_rbc_ = lane0.no_rbc_;
        no_fused_dds_ = lane0.no_fused_dds_;
        no_dds_stack_ = lane0.no_dds_stack_;
        no_dds_stack_b3_ = lane0.no_dds_stack_b3_;
        no_enc_gemm_ = lane0.no_enc_gemm_;
        no_enc_o_ln_ = lane0.no_enc_o_ln_;
        no_flow_gemm_ = lane0.no_flow_gemm_;
        enc_b3_ = lane0.enc_b3_;
        no_f16x2_convs_ = lane0.no_f16x2_convs_;
    } catch (...) {
        release();
        throw;
    }
}

void Engine::open_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        throw EngineError(MI355VITS_ERR_DEVICE, "no HIP device available (the MI355X engine has no CPU fallback)");
    if (device < 0 || device >= ndev) throw EngineError(MI355VITS_ERR_INVALID, "device index out of range");
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreate(&ev_start_));
    HIP_CHECK(hipEventCreate(&ev_end_));
    prof_.stream = stream_;
    const char* fg = lab_getenv("MI355VITS_FORCE_GENERIC");
    force_generic_ = fg && fg[0] == '1';
    const char* nw = lab_getenv("MI355VITS_NO_FUSED_WN");
    no_fused_wn_ = nw && nw[0] == '1';
    const char* nf = lab_getenv("MI355VITS_NO_FUSED_MRF");
    no_fused_mrf_ = nf && nf[0] == '1';
    const char* bw = lab_getenv("MI355VITS_B3_MIN_WORK");
    b3_min_work_ = bw ? atoi(bw) : 256;
    wn_b3_ = lab_getenv("MI355VITS_WN_B3") != nullptr;
    no_mrf_p_ = lab_getenv("MI355VITS_NO_MRF_P") != nullptr;
    no_rbc_ = lab_getenv("MI355VITS_NO_RBC") != nullptr;
    no_fused_dds_ = lab_getenv("MI355VITS_NO_FUSED_DDS") != nullptr;
    no_dds_stack_ = lab_getenv("MI355VITS_NO_DDS_STACK") != nullptr;
    no_dds_stack_b3_ = lab_getenv("MI355VITS_NO_DDS_STACK_B3") != nullptr;
    no_enc_gemm_ = lab_getenv("MI355VITS_NO_ENC_GEMM") != nullptr;
    no_enc_o_ln_ = lab_getenv("MI355VITS_NO_ENC_O_LN") != nullptr;
    no_flow_gemm_ = lab_getenv("MI355VITS_NO_FLOW_GEMM") != nullptr;
    enc_b3_ = lab_getenv("MI355VITS_NO_ENC_B3") == nullptr;
    no_f16x2_convs_ = lab_getenv("MI355VITS_F16X2_NO_CONVS") != nullptr;
    math_ = MATH_BF16X3;  // default (see include/mi355vits.h: f32-grade results; MI355VITS_MATH=f32 for v_mfma_f32_*)
    const char* mm = getenv("MI355VITS_MATH");
    if (mm && mm[0]) {
        if (!strcmp(mm, "bf16x3")) math_ = MATH_BF16X3;
        else if (!strcmp(mm, "f32")) math_ = MATH_F32;
        else if (!strcmp(mm, "bf16w")) math_ = MATH_BF16W;
        else if (!strcmp(mm, "f16x2")) math_ = MATH_F16X2;
        else throw EngineError(MI355VITS_ERR_INVALID, std::string("MI355VITS_MATH: unknown mode '") + mm + "' (f32 | bf16x3 | bf16w | f16x2)");
    }
}

void Engine::set_math(int mode) {
    if (mode != MATH_F32 && mode != MATH_BF16X3 && mode != MATH_BF16W && mode != MATH_F16X2) throw EngineError(MI355VITS_ERR_INVALID, "unknown math mode");
    math_ = mode;
}

void Engine::construct(const WeightsFile& wf, int device) {
    validate_config(cfg_);
    open_device(device);
    model_ = std::make_shared<Model>();
    model_->device = device;

    const mi355vits_config& c = cfg_;
    const int H = c.hidden_channels, F = c.filter_channels, I = c.inter_channels, half = I / 2;
    const int hd = H / c.n_heads, nrel = 2 * c.window_size + 1;
    const int gin = c.n_speakers > 1 ? c.gin_channels : 0;

    auto add_dds = [&](const std::string& p, int C) {
        for (int i = 0; i < c.dp_dds_layers; ++i) {
            add_vec(wf, p + S(".convs_sep.%d.weight", i), {C, 1, c.dp_kernel_size});
            add_vec(wf, p + S(".convs_sep.%d.bias", i), {C});
            add_conv(wf, p + S(".convs_1x1.%d", i), p + S(".convs_1x1.%d", i), C, C, 1, true);
            add_vec(wf, p + S(".norms_1.%d.gamma", i), {C});
            add_vec(wf, p + S(".norms_1.%d.beta", i), {C});
            add_vec(wf, p + S(".norms_2.%d.gamma", i), {C});
            add_vec(wf, p + S(".norms_2.%d.beta", i), {C});
        }
    };

    // ---- text encoder
    add_vec(wf, "enc_p.emb.weight", {c.num_symbols, H});
    for (int i = 0; i < c.n_layers; ++i) {
        const std::string a = S("enc_p.encoder.attn_layers.%d", i);
        std::vector<float> wq(3 * (size_t)H * H), bq(3 * (size_t)H);
        const char* names[3] = {".conv_q", ".conv_k", ".conv_v"};
        for (int q = 0; q < 3; ++q) {
            const HostTensor& w = wf.get(a + names[q] + ".weight", {H, H, 1});
            const HostTensor& b = wf.get(a + names[q] + ".bias", {H});
            memcpy(wq.data() + (size_t)q * H * H, w.data, sizeof(float) * H * H);
            memcpy(bq.data() + (size_t)q * H, b.data, sizeof(float) * H);
        }
        add_conv_data(S("enc.%d.qkv", i), wq, &bq, 3 * H, H, 1);
        add_conv(wf, S("enc.%d.o", i), a + ".conv_o", H, H, 1, true);
        add_vec(wf, a + ".emb_rel_k", {1, nrel, hd});
        add_vec(wf, a + ".emb_rel_v", {1, nrel, hd});
        add_vec(wf, S("enc_p.encoder.norm_layers_1.%d.gamma", i), {H});
        add_vec(wf, S("enc_p.encoder.norm_layers_1.%d.beta", i), {H});
        add_conv(wf, S("enc.%d.ffn1", i), S("enc_p.encoder.ffn_layers.%d.conv_1", i), F, H, c.kernel_size, true);
        add_conv(wf, S("enc.%d.ffn2", i), S("enc_p.encoder.ffn_layers.%d.conv_2", i), H, F, c.kernel_size, true);
        add_vec(wf, S("enc_p.encoder.norm_layers_2.%d.gamma", i), {H});
        add_vec(wf, S("enc_p.encoder.norm_layers_2.%d.beta", i), {H});
    }
    add_conv(wf, "enc.proj", "enc_p.proj", 2 * I, H, 1, true);

    // ---- stochastic duration predictor (inference half)
    add_conv(wf, "dp.pre", "dp.pre", H, H, 1, true);
    add_conv(wf, "dp.proj", "dp.proj", H, H, 1, true);
    add_dds("dp.convs", H);
    if (gin) {
        add_vec(wf, "dp.cond.weight", {H, gin, 1});
        add_vec(wf, "dp.cond.bias", {H});
    }
    {
        const HostTensor& m = wf.get("dp.flows.0.m", {2, 1});
        const HostTensor& l = wf.get("dp.flows.0.logs", {2, 1});
        model_->ea_m[0] = m.data[0]; model_->ea_m[1] = m.data[1];
        model_->ea_logs[0] = l.data[0]; model_->ea_logs[1] = l.data[1];
    }
    for (int j = 1; j < c.dp_n_flows; ++j) {
        const std::string p = S("dp.flows.%d", 1 + 2 * j);
        add_vec(wf, p + ".pre.weight", {H, 1, 1});
        add_vec(wf, p + ".pre.bias", {H});
        add_dds(p + ".convs", H);
        add_conv(wf, p + ".proj", p + ".proj", 3 * c.dp_num_bins - 1, H, 1, true);
    }

    // ---- residual coupling flow.  The channel flips between couplings are folded into the weights:
    // after an odd number of flips the logical tensor is the physical one reversed, so that coupling reads
    // its x0 from the physical upper half through a channel-reversed `pre` and writes its x1 into the
    // physical lower half through a channel-reversed `post`; after an even number it is the identity.
    for (int j = 0; j < c.flow_n_flows; ++j) {
        const int e = c.flow_n_flows - 1 - j;  // execution index (reverse order)
        const bool rev = (e % 2) == 0;
        const std::string f = S("flow.flows.%d", 2 * j);
        {
            const HostTensor& w = wf.get(f + ".pre.weight", {H, half, 1});
            const HostTensor& b = wf.get(f + ".pre.bias", {H});
            std::vector<float> wv(w.count), bv(b.data, b.data + b.count);
            for (int co = 0; co < H; ++co)
                for (int q = 0; q < half; ++q) wv[(size_t)co * half + q] = w.data[(size_t)co * half + (rev ? half - 1 - q : q)];
            add_conv_data(S("flow.%d.pre", j), wv, &bv, H, half, 1);
        }
        for (int l = 0; l < c.flow_wn_layers; ++l) {
            add_conv(wf, S("flow.%d.in.%d", j, l), f + S(".enc.in_layers.%d", l), 2 * H, H, c.flow_wn_kernel, true, EPI_GATE);
            const int rs = l < c.flow_wn_layers - 1 ? 2 * H : H;
            add_conv(wf, S("flow.%d.rs.%d", j, l), f + S(".enc.res_skip_layers.%d", l), rs, H, 1, true);
        }
        if (gin) {
            add_vec(wf, f + ".enc.cond_layer.weight", {2 * H * c.flow_wn_layers, gin, 1});
            add_vec(wf, f + ".enc.cond_layer.bias", {2 * H * c.flow_wn_layers});
        }
        {
            const HostTensor& w = wf.get(f + ".post.weight", {half, H, 1});
            const HostTensor& b = wf.get(f + ".post.bias", {half});
            std::vector<float> wv(w.count), bv(b.count);
            for (int p = 0; p < half; ++p) {
                const int src = rev ? half - 1 - p : p;
                memcpy(wv.data() + (size_t)p * H, w.data + (size_t)src * H, sizeof(float) * H);
                bv[p] = b.data[src];
            }
            add_conv_data(S("flow.%d.post", j), wv, &bv, half, H, 1);
        }
    }
    model_->flow_reversed_out = (c.flow_n_flows % 2) == 1;

    // ---- HiFi-GAN decoder
    const int C0 = c.upsample_initial_channel;
    {
        const HostTensor& w = wf.get("dec.conv_pre.weight", {C0, I, 7});
        const HostTensor& b = wf.get("dec.conv_pre.bias", {C0});
        std::vector<float> wv(w.count), bv(b.data, b.data + b.count);
        for (int co = 0; co < C0; ++co)
            for (int ci = 0; ci < I; ++ci)
                memcpy(wv.data() + ((size_t)co * I + ci) * 7, w.data + ((size_t)co * I + (model_->flow_reversed_out ? I - 1 - ci : ci)) * 7,
                       sizeof(float) * 7);
        add_conv_data("dec.conv_pre", wv, &bv, C0, I, 7);
    }
    int ch = C0;
    for (int i = 0; i < c.n_upsamples; ++i) {
        add_vec(wf, S("dec.ups.%d.weight", i), {ch, ch / 2, c.upsample_kernel_sizes[i]});
        add_vec(wf, S("dec.ups.%d.bias", i), {ch / 2});
        {
            // the transposed conv as `rate` polyphase stride-1 filters on the matrix cores (kernels.h: shuf_*)
            const int uk = c.upsample_kernel_sizes[i], ur = c.upsample_rates[i], taps = convt_taps(uk, ur);
            const HostTensor& w = wf.get(S("dec.ups.%d.weight", i), {ch, ch / 2, uk});
            const HostTensor& b = wf.get(S("dec.ups.%d.bias", i), {ch / 2});
            std::vector<float> wv((size_t)ur * (ch / 2) * ch * taps), bv((size_t)ur * (ch / 2));
            convt_to_polyphase(w.data, b.data, ch, ch / 2, uk, ur, wv.data(), bv.data());
            add_conv_data(S("dec.ups.%d.poly", i), wv, &bv, ur * (ch / 2), ch, taps);
        }
        ch /= 2;
        for (int j = 0; j < c.n_resblock_kernels; ++j) {
            const int n = i * c.n_resblock_kernels + j;
            const int rk = c.resblock_kernel_sizes[j];
            for (int m = 0; m < c.resblock_n_dilations[j]; ++m) {
                if (c.resblock == 2) {
                    add_conv(wf, S("dec.rb.%d.c.%d", n, m), S("dec.resblocks.%d.convs.%d", n, m), ch, ch, rk, true);
                } else {
                    add_conv(wf, S("dec.rb.%d.c1.%d", n, m), S("dec.resblocks.%d.convs1.%d", n, m), ch, ch, rk, true);
                    add_conv(wf, S("dec.rb.%d.c2.%d", n, m), S("dec.resblocks.%d.convs2.%d", n, m), ch, ch, rk, true);
                }
            }
        }
    }
    add_vec(wf, "dec.conv_post.weight", {1, ch, 7});
    if (gin) {
        add_vec(wf, "dec.cond.weight", {C0, gin, 1});
        add_vec(wf, "dec.cond.bias", {C0});
        add_vec(wf, "emb_g.weight", {c.n_speakers, gin});
    }

    // ---- one upload
    void* p = nullptr;
    if (hipMalloc(&p, host_stage_.size() * sizeof(float) + 256) != hipSuccess)
        throw EngineError(MI355VITS_ERR_NOMEM, "out of device memory (weights)");
    model_->dev_weights = static_cast<float*>(p);
    model_->bytes = host_stage_.size() * sizeof(float);
    HIP_CHECK(hipMemcpy(model_->dev_weights, host_stage_.data(), host_stage_.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float>().swap(host_stage_);
}

void Engine::probe_weights(double out[8]) {
    HIP_CHECK(hipSetDevice(device_));
    const int cus = current_device_cu_count();
    const int n16 = 256 * 8 * 80;  // 2.62 MB windows
    const size_t win = (size_t)n16 * 16;
    const int nwin = (int)std::min<size_t>(model_->bytes / win, 24);
    if (nwin < 1) throw EngineError(MI355VITS_ERR_INVALID, "weight arena smaller than one probe window");
    // (diagnostics of include/mi355vits_lab.h: hipMalloc / hipFree synchronise the device — not for a handle that is serving.)
    // RAII: a HIP error in between must not leak the events or the sink
    struct Ev {
        hipEvent_t e = nullptr;
        Ev() { HIP_CHECK(hipEventCreate(&e)); }
        ~Ev() { if (e) (void)hipEventDestroy(e); }
    } ev0, ev1;
    struct Sink {
        unsigned* p = nullptr;
        explicit Sink(size_t n) { HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), n)); }
        ~Sink() { (void)hipFree(p); }
    } sk((size_t)cus * 4 + 16);
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    unsigned* sink = sk.p;
    std::vector<double> g8, g1;
    for (int wdx = 0; wdx < nwin; ++wdx) {
        const char* base = reinterpret_cast<const char*>(model_->dev_weights) + (size_t)wdx * win;
        for (int mode = 0; mode < 2; ++mode) {
            auto go = [&] {
                if (mode == 0) launch_probe_l2_stream(base, n16, 4, sink, cus, stream_);
                else launch_probe_l2_stream1(base, n16, 1, sink, cus, stream_);
            };
            go();
            HIP_CHECK(hipEventRecord(e0, stream_));
            go();
            go();
            HIP_CHECK(hipEventRecord(e1, stream_));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = (double)cus * (mode == 0 ? 4 : 1) * win / (ms * 0.5e-3) / 1e9;
            (mode == 0 ? g8 : g1).push_back(gbs);
        }
    }
    std::sort(g8.begin(), g8.end());
    std::sort(g1.begin(), g1.end());
    out[0] = g8.front(); out[1] = g8[g8.size() / 2]; out[2] = g8.back();
    out[3] = g1.front(); out[4] = g1[g1.size() / 2]; out[5] = g1.back();
    out[6] = (double)nwin;
    out[7] = (double)(reinterpret_cast<uintptr_t>(model_->dev_weights) & 0xfffffffffULL);  // low bits of the arena's virtual address
}

void Engine::release() noexcept {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& t : taps_) (void)hipFree(t.dev);
    taps_.clear();
    model_.reset();  // the replica is freed with its last lane
    if (ev_start_) (void)hipEventDestroy(ev_start_);
    if (ev_end_) (void)hipEventDestroy(ev_end_);
    ev_start_ = ev_end_ = nullptr;
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
}

Engine::~Engine() { release(); }

// =================================================================================================
// launch helpers
// =================================================================================================
bool Engine::enc_gemm(const ConvW& w, const ConvArgs& a) const {
    // phoneme-sized convs, and the pointwise convs around the coupling layers' WaveNet stacks (frames: K = 1 only)
    return !force_generic_ && !no_enc_gemm_ && (!phase_b_ || (w.K == 1 && !no_flow_gemm_)) && math_on_bf16(pmath()) && w.packed_b3s != NO_OFF &&
           a.epi == EPI_STD && w.epi == EPI_STD && !a.shuf_s && a.Tin < 0 && !a.accumulate && enc_conv_b3_supported(w.Cin, w.Cout, w.K, a.dil) &&
           a.ksplit == enc_conv_b3_slices(w.Cin) && (a.ksplit == 1 || a.part);  // a split conv only where the caller adds up the slices
}

// a filled-in conv of the decoder that k_rb_conv takes (MATH_BF16X3 only: the other modes keep their kernels)
bool Engine::rbc_ok(const ConvW& w, const ConvArgs& a) const {
    return phase_b_ && !force_generic_ && !no_rbc_ && math_ == MATH_BF16X3 && w.packed_p != NO_OFF && rb_conv_supported(a);
}

void Engine::conv(const char* label, const ConvW& w, ConvArgs a) {
    // frames-sized tensors: the kernel is a function of the layer, not of the batch's padding.  Phoneme-sized ones (the text
    // encoder) stay on the f32 kernels except the wide FFN conv (192 -> 768, k3: K Cin >= 512 and >= 4 row blocks), whose
    // grid fills most of the chip even at one column tile per row — again a rule of the layer alone
    const bool wide_enc = !phase_b_ && enc_b3_ && w.K * w.Cin >= 512 && w.Cout >= 512 && a.epi == EPI_STD;
    a.fixed_rule = (phase_b_ || wide_enc) ? 1 : 0;
    a.Cin = w.Cin;
    a.Cout = w.Cout;
    a.K = w.K;
    a.bias = P(w.bias);
    if (a.pad < 0) a.pad = (w.K * a.dil - a.dil) / 2;
    if ((a.epi == EPI_GATE) != (w.epi == EPI_GATE)) throw EngineError(MI355VITS_ERR_INTERNAL, "conv epilogue / packing mismatch");
    const double flops = 2.0 * a.B * (double)a.T * w.Cout * w.Cin * w.K;
    double ch_io = (double)w.Cin + (a.epi == EPI_GATE ? a.H : w.Cout);
    if (a.res) ch_io += w.Cout;
    if (a.accumulate) ch_io += w.Cout;
    if (a.epi == EPI_RESSKIP) ch_io += w.Cout;  // h and skip are read-modify-write
    const double bytes = 4.0 * a.B * (double)a.T * ch_io + 4.0 * (double)w.Cout * w.Cin * w.K;
    ProfScope ps(prof_, label, flops, bytes);
    if (a.shuf_s && phase_b_ && !force_generic_ && !no_rbc_ && math_ == MATH_BF16X3 && w.packed_p != NO_OFF && ups_pl_supported(a)) {
        a.w = P(w.packed_p);  // polyphase upsamplers 128 -> 64, 64 -> 32: every input channel resident (k_ups_pl)
        a.math = MATH_BF16X3;
        if (d_slen_ && a.in_len >= d_slen_ && a.in_len < d_slen_ + h_slen_.size()) a.in_len_host = h_slen_.data() + (a.in_len - d_slen_);
        launch_ups_pl(a, stream_);
        return;
    }
    if (rbc_ok(w, a)) {  // 128-channel resblock convs: every input channel resident, one launch per conv (k_rb_conv)
        a.w = P(w.packed_p);
        a.math = MATH_BF16X3;
        // the host's copy of the row lengths (the per-stage lengths are made on the host and uploaded as one table): the launcher
        // counts the (row, column block) items that have work from it
        if (d_slen_ && a.in_len >= d_slen_ && a.in_len < d_slen_ + h_slen_.size()) a.in_len_host = h_slen_.data() + (a.in_len - d_slen_);
        launch_rb_conv(a, stream_);
        return;
    }
    if (enc_gemm(w, a)) {  // phoneme-sized dense convs: one 192-channel slice per workgroup, staged once (k_enc_b3)
        a.wb3 = P(w.packed_b3s);
        a.math = pmath();
        launch_enc_conv_b3(a, stream_);
        return;
    }
    if (!force_generic_ && w.packed != NO_OFF) {
        a.w = P(w.packed);
        // split-bf16 staged kernel where it pays: convs with little work per staged chunk (1x1 convs, the last
        // upsampler: K * Cin < 256) spend more on splitting the chunk than the faster matrix-core loop saves
        // (measured: flow.pre / post, res_skip, upsample 64 -> 32); MI355VITS_B3_MIN_WORK overrides the threshold (tests)
        if (math_on_bf16(pmath()) && w.packed_b3s != NO_OFF &&
            (math_on_bf16(a.math) || ((w.K * w.Cin >= b3_min_work_ || (a.shuf_s && w.Cin % 64 == 0)) && a.epi == EPI_STD))) {
            a.wb3 = P(w.packed_b3s);
            a.math = pmath();
            if (math_ == MATH_F16X2 && w.packed_h2s != NO_OFF && a.epi == EPI_STD && !no_f16x2_convs_) {  // two fp16 terms per operand
                a.wb3 = P(w.packed_h2s);
                a.math = MATH_F16X2;
            }
        } else {
            a.math = MATH_F32;
        }
        launch_conv1d_mfma(a, stream_);
    } else {
        a.w = P(w.raw);
        launch_conv1d_generic(a, stream_);
    }
}

void Engine::tap(const char* name, const float* dev, std::initializer_list<int64_t> dims, const int* row_len, int factor) {
    if (!taps_on_) return;
    size_t cnt = 1;
    for (auto d : dims) cnt *= (size_t)d;
    for (auto& t : taps_)
        if (t.name == name) return;
    Tap t;
    t.name = name;
    t.dims.assign(dims.begin(), dims.end());
    t.count = cnt;
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, cnt * sizeof(float) + 16));
    t.dev = static_cast<float*>(p);
    HIP_CHECK(hipMemcpyAsync(t.dev, dev, cnt * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    if (row_len && t.dims.size() == 3) launch_zero_row_tails(t.dev, (int)t.dims[0], (int)t.dims[1], (long)t.dims[2], row_len, factor, stream_);
    taps_.push_back(std::move(t));
}

// rows [row0, row0 + nrows) of the leading (batch) extent; nrows < 0: all rows (a batch-256 stage tap is 6.4 GB: the tests fetch the rows they check)
long Engine::get_tap(const std::string& name, float* out, size_t cap, int64_t dims[4], long row0, long nrows) {
    for (auto& t : taps_)
        if (t.name == name) {
            for (int i = 0; i < 4; ++i) dims[i] = i < (int)t.dims.size() ? t.dims[i] : 1;
            const long rows = t.dims.empty() ? 1 : (long)t.dims[0];
            if (nrows < 0) { row0 = 0; nrows = rows; }
            if (row0 < 0 || nrows < 0 || row0 + nrows > rows) throw EngineError(MI355VITS_ERR_INVALID, "tap rows out of range");
            const size_t per_row = rows > 0 ? t.count / (size_t)rows : 0, cnt = per_row * (size_t)nrows;
            dims[0] = nrows;
            if (out) {
                if (cap < cnt) throw EngineError(MI355VITS_ERR_INVALID, "tap buffer too small");
                HIP_CHECK(hipStreamSynchronize(stream_));
                if (cnt) HIP_CHECK(hipMemcpy(out, t.dev + per_row * (size_t)row0, cnt * sizeof(float), hipMemcpyDeviceToHost));
            }
            return (long)cnt;
        }
    throw EngineError(MI355VITS_ERR_INVALID, "no such tap: " + name);
}
std::string Engine::list_taps() const {
    std::string s;
    for (auto& t : taps_) s += t.name + "\n";
    return s;
}

float Engine::last_run_ms() {
    if (!timed_) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ev_start_, ev_end_) != hipSuccess) return -1.0f;
    return ms;
}

// =================================================================================================
// K1-K4 text encoder
// =================================================================================================
void Engine::text_encoder(int B, int Tx) {
    const mi355vits_config& c = cfg_;
    const int H = c.hidden_channels, F = c.filter_channels, I = c.inter_channels;
    const long xbs = (long)H * Tx;
    {
        ProfScope ps(prof_, "embed", 0, 4.0 * B * H * Tx);
        launch_embed(d_ids_, d_len_, vec("enc_p.emb.weight"), B, Tx, H, c.num_symbols, sqrtf((float)H), d_x_, stream_);
    }
    for (int i = 0; i < c.n_layers; ++i) {
        const std::string a = S("enc_p.encoder.attn_layers.%d", i);
        ConvArgs q;
        q.x = d_x_; q.x_bs = xbs; q.x_ld = Tx;
        q.y = d_qkv_; q.y_bs = 3 * xbs; q.y_ld = Tx;
        q.B = B; q.T = Tx;
        conv("enc.qkv", cw(S("enc.%d.qkv", i)), q);
        {
            const double fl = 4.0 * B * (double)Tx * Tx * H;
            ProfScope ps(prof_, "enc.attention", fl, 4.0 * B * 4 * H * Tx);
            if (!force_generic_ && rel_attention_mfma_supported(Tx, H, c.n_heads, c.window_size))
                launch_rel_attention_mfma(d_qkv_, vec(a + ".emb_rel_k"), vec(a + ".emb_rel_v"), d_len_, B, Tx, H, c.n_heads,
                                          c.window_size, d_att_, stream_);
            else
                launch_rel_attention(d_qkv_, vec(a + ".emb_rel_k"), vec(a + ".emb_rel_v"), d_len_, B, Tx, H, c.n_heads,
                                     c.window_size, d_att_, stream_);
        }
        ConvArgs o;
        o.x = d_att_; o.x_bs = xbs; o.x_ld = Tx;
        o.y = d_x2_; o.y_bs = xbs; o.y_ld = Tx;
        o.res = d_x_; o.res_bs = xbs; o.res_ld = Tx;
        o.B = B; o.T = Tx;
        const ConvW& wo = cw(S("enc.%d.o", i));
        if (enc_gemm(wo, o) && !no_enc_o_ln_ && enc_o_ln_supported(wo.Cin, wo.Cout, wo.K)) {
            // o-proj + residual + LayerNorm in one launch, in place on x (k_enc_o_ln)
            o.y = d_x_;
            o.Cin = wo.Cin; o.Cout = wo.Cout; o.K = wo.K; o.bias = P(wo.bias);
            o.wb3 = P(wo.packed_b3s);
            o.math = tmath();
            ProfScope ps(prof_, "enc.o_ln", 2.0 * B * (double)Tx * H * H, 4.0 * B * 3 * H * Tx);
            launch_enc_o_ln(o, vec(S("enc_p.encoder.norm_layers_1.%d.gamma", i)), vec(S("enc_p.encoder.norm_layers_1.%d.beta", i)), nullptr,
                            1e-5f, stream_);
        } else {
            conv("enc.o", wo, o);
            LNArgs ln;
            ln.x = d_x2_; ln.y = d_x_; ln.B = B; ln.C = H; ln.T = Tx;
            ln.gamma = vec(S("enc_p.encoder.norm_layers_1.%d.gamma", i));
            ln.beta = vec(S("enc_p.encoder.norm_layers_1.%d.beta", i));
            ProfScope ps(prof_, "layernorm", 0, 8.0 * B * H * Tx);
            launch_layernorm(ln, stream_);
        }
        ConvArgs f1;
        f1.x = d_x_; f1.x_bs = xbs; f1.x_ld = Tx;
        f1.y = d_ffn_; f1.y_bs = (long)F * Tx; f1.y_ld = Tx;
        f1.in_len = d_len_; f1.relu = 1;
        f1.B = B; f1.T = Tx;
        conv("enc.ffn1", cw(S("enc.%d.ffn1", i)), f1);
        ConvArgs f2;
        f2.x = d_ffn_; f2.x_bs = (long)F * Tx; f2.x_ld = Tx;
        f2.y = d_x2_; f2.y_bs = xbs; f2.y_ld = Tx;
        f2.in_len = d_len_; f2.out_len = d_len_; f2.mask_before_res = 1;
        f2.res = d_x_; f2.res_bs = xbs; f2.res_ld = Tx;
        f2.B = B; f2.T = Tx;
        const ConvW& w2 = cw(S("enc.%d.ffn2", i));
        bool split2 = false;  // conv_2 as 192-channel slices: raw sums, added up by the LayerNorm launch below
        if (d_part_ && w2.Cout == H && enc_conv_b3_slices(w2.Cin) > 1 && enc_conv_b3_slices(w2.Cin) * 192 == F) {
            f2.ksplit = enc_conv_b3_slices(w2.Cin);
            f2.part = d_part_;
            split2 = enc_gemm(w2, f2);
            if (!split2) { f2.ksplit = 1; f2.part = nullptr; }
        }
        conv("enc.ffn2", w2, f2);
        {
            LNArgs ln;
            ln.x = d_x2_; ln.y = d_x_; ln.B = B; ln.C = H; ln.T = Tx;
            if (split2) {
                ln.x = d_part_; ln.nparts = f2.ksplit; ln.part_stride = (long)B * xbs;
                ln.bias = P(w2.bias); ln.premask_len = d_len_;
                ln.res = d_x_;
            }
            ln.gamma = vec(S("enc_p.encoder.norm_layers_2.%d.gamma", i));
            ln.beta = vec(S("enc_p.encoder.norm_layers_2.%d.beta", i));
            if (i == c.n_layers - 1) ln.out_len = d_len_;  // x = x * x_mask after the last layer
            ProfScope ps(prof_, "layernorm", 0, 8.0 * B * H * Tx);
            launch_layernorm(ln, stream_);
        }
    }
    tap("x", d_x_, {B, H, Tx});
    ConvArgs p;
    p.x = d_x_; p.x_bs = xbs; p.x_ld = Tx;
    p.y = d_stats_; p.y_bs = 2L * I * Tx; p.y_ld = Tx;
    p.out_len = d_len_;
    p.B = B; p.T = Tx;
    conv("enc.proj", cw("enc.proj"), p);
    tap("stats", d_stats_, {B, 2 * I, Tx});
}

// =================================================================================================
// K5 stochastic duration predictor (reverse) + K6 durations
// =================================================================================================
void Engine::dds(const std::string& key, float* X, float* Y1, float* Y2, int B, int T) {
    const mi355vits_config& c = cfg_;
    const int C = c.hidden_channels;
    const long bs = (long)C * T;
    int dil = 1;
    if (!force_generic_ && !no_fused_dds_ && c.dp_dds_layers >= 2 && dds_layer_fused_supported(C) &&
        cw(key + S(".convs_1x1.%d", 0)).packed != NO_OFF) {
        // one launch per layer; x ping-pongs through the scratch buffers and the last layer lands in X again
        const float* src = X;
        for (int i = 0; i < c.dp_dds_layers; ++i) {
            float* dst = (i == c.dp_dds_layers - 1) ? X : ((i & 1) ? Y2 : Y1);
            const ConvW& w = cw(key + S(".convs_1x1.%d", i));
            ProfScope ps(prof_, "dds.layer", 2.0 * B * (double)T * C * C, 8.0 * B * C * T);
            launch_dds_layer(src, dst, vec(key + S(".convs_sep.%d.weight", i)), vec(key + S(".convs_sep.%d.bias", i)),
                             vec(key + S(".norms_1.%d.gamma", i)), vec(key + S(".norms_1.%d.beta", i)), P(w.packed), P(w.bias),
                             vec(key + S(".norms_2.%d.gamma", i)), vec(key + S(".norms_2.%d.beta", i)), d_len_, B, C, T,
                             c.dp_kernel_size, dil, stream_);
            src = dst;
            dil *= c.dp_kernel_size;
        }
        return;
    }
    for (int i = 0; i < c.dp_dds_layers; ++i) {
        {
            ProfScope ps(prof_, "dds.dwconv_ln_gelu", 0, 8.0 * B * C * T);
            launch_dds_dwconv_ln_gelu(X, vec(key + S(".convs_sep.%d.weight", i)), vec(key + S(".convs_sep.%d.bias", i)),
                                      vec(key + S(".norms_1.%d.gamma", i)), vec(key + S(".norms_1.%d.beta", i)), d_len_,
                                      B, C, T, c.dp_kernel_size, dil, Y1, stream_);
        }
        ConvArgs a;
        a.x = Y1; a.x_bs = bs; a.x_ld = T;
        a.y = Y2; a.y_bs = bs; a.y_ld = T;
        a.B = B; a.T = T;
        conv("dds.1x1", cw(key + S(".convs_1x1.%d", i)), a);
        {
            LNArgs ln;
            ln.x = Y2; ln.y = X; ln.add_to = X; ln.gelu = 1;
            ln.B = B; ln.C = C; ln.T = T;
            ln.gamma = vec(key + S(".norms_2.%d.gamma", i));
            ln.beta = vec(key + S(".norms_2.%d.beta", i));
            ProfScope ps(prof_, "layernorm", 0, 12.0 * B * C * T);
            launch_layernorm(ln, stream_);
        }
        dil *= c.dp_kernel_size;
    }
}

void Engine::duration_predictor(int B, int Tx, const mi355vits_run_args& args) {
    const mi355vits_config& c = cfg_;
    const int H = c.hidden_channels;
    const long bs = (long)H * Tx;
    const int nth = 3 * c.dp_num_bins - 1;
    // the whole stack in one launch each (k_dds_stack): pre + DDS layers + proj, and for a ConvFlow the spline too
    const bool stack = !force_generic_ && !no_fused_dds_ && !no_dds_stack_ &&
                       dds_stack_supported(H, c.dp_kernel_size, c.dp_dds_layers, H) && nth <= H && c.dp_num_bins <= 16 &&
                       cw("dp.pre").packed != NO_OFF && cw("dp.proj").packed != NO_OFF && cw("dp.pre").packed_b3s != NO_OFF &&
                       cw("dp.proj").packed_b3s != NO_OFF;
    // the stack's 1x1 convs: bf16 planes in the split-bf16 math modes (k_dds_stack_b3), f32 A fragments in MATH_F32
    const bool stack_b3 = math_on_bf16(tmath()) && !no_dds_stack_b3_;
    auto stack_w = [&](const ConvW& w) -> const float* {
        const size_t off = stack_b3 ? w.packed_b3s : w.packed;
        return off == NO_OFF ? nullptr : P(off);
    };
    auto stack_layers = [&](DdsStackArgs& a, const std::string& key) {
        a.math = stack_b3 ? tmath() : (int)MATH_F32;  // exact in MATH_BF16W too: durations must not move
        a.n_layers = c.dp_dds_layers;
        a.K = c.dp_kernel_size;
        for (int i = 0; i < c.dp_dds_layers; ++i) {
            const ConvW& w = cw(key + S(".convs_1x1.%d", i));
            if (stack_w(w) == nullptr) throw EngineError(MI355VITS_ERR_INTERNAL, "dds stack: 1x1 conv without packed weights");
            a.dw_w[i] = vec(key + S(".convs_sep.%d.weight", i));
            a.dw_b[i] = vec(key + S(".convs_sep.%d.bias", i));
            a.g1[i] = vec(key + S(".norms_1.%d.gamma", i));
            a.b1[i] = vec(key + S(".norms_1.%d.beta", i));
            a.w1x1[i] = stack_w(w);
            a.bias1x1[i] = P(w.bias);
            a.g2[i] = vec(key + S(".norms_2.%d.gamma", i));
            a.b2[i] = vec(key + S(".norms_2.%d.beta", i));
        }
        a.len = d_len_;
        a.B = B;
        a.T = Tx;
    };
    const double stack_flops = 2.0 * B * (double)Tx * H * H;
    if (stack) {
        DdsStackArgs a;
        a.src = d_x_;
        a.pre_mode = DDS_PRE_CONV;
        a.pre_w = stack_w(cw("dp.pre"));
        a.pre_b = P(cw("dp.pre").bias);
        a.cond = d_cond_dp_;
        a.cond_bs = H;
        stack_layers(a, "dp.convs");
        a.proj_w = stack_w(cw("dp.proj"));
        a.proj_b = P(cw("dp.proj").bias);
        a.proj_cout = H;
        a.out = d_h_;
        ProfScope ps(prof_, "dp.stack", stack_flops * (c.dp_dds_layers + 2), 8.0 * B * H * Tx);
        launch_dds_stack(a, H, stream_);
    }
    // h = proj(DDS(pre(x) [+ cond(g)])) * mask
    ConvArgs pre;
    pre.x = d_x_; pre.x_bs = bs; pre.x_ld = Tx;
    pre.y = d_d0_; pre.y_bs = bs; pre.y_ld = Tx;
    pre.cond = d_cond_dp_; pre.cond_bs = H;
    pre.B = B; pre.T = Tx;
    if (!stack) {
        conv("dp.pre", cw("dp.pre"), pre);
        dds("dp.convs", d_d0_, d_d1_, d_d2_, B, Tx);
    }
    ConvArgs pr;
    pr.x = d_d0_; pr.x_bs = bs; pr.x_ld = Tx;
    pr.y = d_h_; pr.y_bs = bs; pr.y_ld = Tx;
    pr.in_len = d_len_; pr.out_len = d_len_;
    pr.B = B; pr.T = Tx;
    if (!stack) conv("dp.proj", cw("dp.proj"), pr);
    tap("dp.h", d_h_, {B, H, Tx});

    {
        ProfScope ps(prof_, "sdp.noise");
        launch_sdp_noise(d_z2_, d_noise_w_, B, Tx, args.scales[2], args.seed, args.utterance_base, stream_);
    }
    int ch0 = 0;  // physical channel that is logical channel 0
    for (int j = c.dp_n_flows - 1; j >= 1; --j) {
        ch0 ^= 1;  // Flip
        const std::string p = S("dp.flows.%d", 1 + 2 * j);
        if (stack && stack_w(cw(p + ".proj")) != nullptr) {
            DdsStackArgs a;
            a.src = d_h_;
            a.pre_mode = DDS_PRE_AFFINE;
            a.pre_w = vec(p + ".pre.weight");
            a.pre_b = vec(p + ".pre.bias");
            a.z = d_z2_;
            a.zch = ch0;
            stack_layers(a, p + ".convs");
            a.proj_w = stack_w(cw(p + ".proj"));
            a.proj_b = P(cw(p + ".proj").bias);
            a.proj_cout = nth;
            a.spline = 1;
            a.nb = c.dp_num_bins;
            a.tail = c.dp_tail_bound;
            a.inv_sqrt_fc = 1.0f / sqrtf((float)H);
            ProfScope ps(prof_, "convflow.stack", stack_flops * c.dp_dds_layers + 2.0 * B * (double)Tx * H * nth, 12.0 * B * H * Tx);
            launch_dds_stack(a, H, stream_);
            continue;
        }
        {
            ProfScope ps(prof_, "convflow.pre", 0, 12.0 * B * H * Tx);
            launch_convflow_pre(d_z2_, ch0, vec(p + ".pre.weight"), vec(p + ".pre.bias"), d_h_, B, H, Tx, d_d0_, stream_);
        }
        dds(p + ".convs", d_d0_, d_d1_, d_d2_, B, Tx);
        ConvArgs a;
        a.x = d_d0_; a.x_bs = bs; a.x_ld = Tx;
        a.y = d_theta_; a.y_bs = (long)nth * Tx; a.y_ld = Tx;
        a.in_len = d_len_; a.out_len = d_len_;
        a.B = B; a.T = Tx;
        conv("convflow.proj", cw(p + ".proj"), a);
        {
            ProfScope ps(prof_, "spline");
            launch_spline_inverse(d_z2_, ch0, d_theta_, d_len_, B, Tx, c.dp_num_bins, c.dp_tail_bound,
                                  1.0f / sqrtf((float)H), stream_);
        }
    }
    ch0 ^= 1;  // Flip before the ElementwiseAffine
    {
        ProfScope ps(prof_, "durations");
        // logical channel 0 uses EA parameters [0]
        launch_durations(d_z2_, ch0, model_->ea_m[0], model_->ea_logs[0], d_len_, d_forced_, B, Tx, args.scales[1], d_logw_, d_wceil_,
                         d_cum_, d_ylen_, stream_);
    }
    tap("logw", d_logw_, {B, 1, Tx});
}

// =================================================================================================
// K6/K7 expand + K8 flow^-1 + K9-K12 decoder
// =================================================================================================
void Engine::flow_and_decoder(int B, int Ty, const mi355vits_run_args& args) {
    struct PhaseB { bool& f; explicit PhaseB(bool& r) : f(r) { f = true; } ~PhaseB() { f = false; } } phase_guard(phase_b_);
    const mi355vits_config& c = cfg_;
    const int H = c.hidden_channels, I = c.inter_channels, half = I / 2;
    const int Tx = Tx_;
    const long zbs = (long)I * Ty, hbs = (long)H * Ty;
    {
        ProfScope ps(prof_, "expand_prior", 0, 4.0 * B * I * Ty * 3);
        launch_expand_prior(d_stats_, d_cum_, d_ylen_, d_noise_z_, args.noise_z_frames, B, I, Tx, Ty, args.scales[0],
                            args.seed, args.utterance_base, d_z_, stream_);
    }
    tap("z_p", d_z_, {B, I, Ty}, d_ylen_, 1);

    for (int j = c.flow_n_flows - 1; j >= 0; --j) {
        const int e = c.flow_n_flows - 1 - j;
        const bool rev = (e % 2) == 0;
        float* x0 = d_z_ + (rev ? (long)half * Ty : 0);
        float* x1 = d_z_ + (rev ? 0 : (long)half * Ty);
        ConvArgs pre;
        pre.x = x0; pre.x_bs = zbs; pre.x_ld = Ty;
        pre.y = d_fh_; pre.y_bs = hbs; pre.y_ld = Ty;
        pre.out_len = d_ylen_;
        pre.B = B; pre.T = Ty;
        conv("flow.pre", cw(S("flow.%d.pre", j)), pre);
        float* hcur = d_fh_;
        float* hnext = d_fh2_;
        for (int l = 0; l < c.flow_wn_layers; ++l) {
            int dil = 1;
            for (int q = 0; q < l; ++q) dil *= c.flow_wn_dilation_rate;
            const ConvW& win = cw(S("flow.%d.in.%d", j, l));
            const ConvW& wrs = cw(S("flow.%d.rs.%d", j, l));
            const float* cond_l = d_cond_flow_.empty() ? nullptr : d_cond_flow_[j] + (long)l * 2 * H;
            // split-bf16 math: the in-layer and the res/skip convs go through the staged bf16 kernel (two launches, `u`
            // through HBM: 38 MB per layer, nothing next to the matrix-core time saved); the fused kernel is f32-MFMA
            // in-layer + res/skip through the staged split-bf16 kernel instead of the fused f32 layer: measured 2.09 + 1.70 ms
            // vs 3.65 ms per step — no gain (small grids, scalar res/skip epilogue); opt-in for A/B and for the tests
            const bool wn_b3 = wn_b3_ && math_on_bf16(kmath()) && win.packed_b3s != NO_OFF && wrs.packed_b3s != NO_OFF &&
                               conv1d_b3_supported(win.Cin, win.Cout, win.K, dil, Ty) && !force_generic_;
            if (!force_generic_ && !no_fused_wn_ && !wn_b3 && math_on_bf16(kmath()) && win.packed_b3w != NO_OFF &&
                wrs.packed_b3s != NO_OFF && wn_layer_b3_supported(H, win.K, dil)) {
                // fused layer on the bf16 matrix cores.  Chosen by the layer shape alone (never by the grid size), so a
                // row's bits do not depend on what it is batched with.
                WnArgs w;
                w.h_in = hcur; w.h_out = hnext; w.h_bs = hbs; w.h_ld = Ty;
                w.skip = d_fskip_; w.s_bs = hbs; w.s_ld = Ty;
                w.w_in = P(win.packed_b3w); w.b_in = P(win.bias);
                w.w_rs = P(wrs.packed_b3s); w.b_rs = P(wrs.bias);
                w.cond = cond_l; w.cond_bs = 2L * H * c.flow_wn_layers;
                w.len = d_ylen_;
                w.B = B; w.H = H; w.T = Ty; w.K = win.K; w.dil = dil; w.Crs = wrs.Cout; w.skip_init = (l == 0);
                w.math = kmath();
                if (math_ == MATH_F16X2 && win.packed_h2s != NO_OFF && wrs.packed_h2s != NO_OFF) {  // two fp16 terms per operand
                    w.w_in = P(win.packed_h2s);
                    w.w_rs = P(wrs.packed_h2s);
                    w.math = MATH_F16X2;
                }
                const double fl = 2.0 * B * (double)Ty * H * ((double)win.Cout * win.K + wrs.Cout);
                ProfScope ps(prof_, "flow.wn_layer_b3", fl, 4.0 * B * (double)Ty * H * 4);
                launch_wn_layer_b3(w, stream_);
                if (wrs.Cout == 2 * H) std::swap(hcur, hnext);
                continue;
            }
            if (!force_generic_ && !no_fused_wn_ && !wn_b3 && wn_layer_fused_supported(H, win.K, dil)) {
                WnArgs w;
                w.h_in = hcur; w.h_out = hnext; w.h_bs = hbs; w.h_ld = Ty;
                w.skip = d_fskip_; w.s_bs = hbs; w.s_ld = Ty;
                w.w_in = P(win.packed); w.b_in = P(win.bias);
                w.w_rs = P(wrs.packed); w.b_rs = P(wrs.bias);
                w.cond = cond_l; w.cond_bs = 2L * H * c.flow_wn_layers;
                w.len = d_ylen_;
                w.B = B; w.H = H; w.T = Ty; w.K = win.K; w.dil = dil; w.Crs = wrs.Cout; w.skip_init = (l == 0);
                const double fl = 2.0 * B * (double)Ty * H * ((double)win.Cout * win.K + wrs.Cout);
                ProfScope ps(prof_, "flow.wn_layer", fl, 4.0 * B * (double)Ty * H * 4);
                launch_wn_layer(w, stream_);
                if (wrs.Cout == 2 * H) std::swap(hcur, hnext);  // the last layer leaves h untouched
                continue;
            }
            ConvArgs in;
            in.x = hcur; in.x_bs = hbs; in.x_ld = Ty;
            in.y = d_fu_; in.y_bs = hbs; in.y_ld = Ty;
            in.epi = EPI_GATE; in.H = H; in.dil = dil;
            if (cond_l) {
                in.cond = cond_l;
                in.cond_bs = 2L * H * c.flow_wn_layers;
            }
            in.B = B; in.T = Ty;
            if (wn_b3) in.math = kmath();
            conv("flow.in_gate", win, in);
            ConvArgs rs;
            rs.x = d_fu_; rs.x_bs = hbs; rs.x_ld = Ty;
            rs.y = hcur; rs.y_bs = hbs; rs.y_ld = Ty;
            rs.y2 = d_fskip_; rs.y2_bs = hbs; rs.y2_ld = Ty;
            rs.epi = EPI_RESSKIP; rs.H = H; rs.skip_init = (l == 0);
            rs.out_len = d_ylen_;
            rs.B = B; rs.T = Ty;
            if (wn_b3) rs.math = kmath();
            conv("flow.res_skip", wrs, rs);
        }
        ConvArgs post;
        post.x = d_fskip_; post.x_bs = hbs; post.x_ld = Ty;
        post.y = x1; post.y_bs = zbs; post.y_ld = Ty;
        post.res = x1; post.res_bs = zbs; post.res_ld = Ty; post.res_sub = 1;
        post.in_len = d_ylen_; post.out_len = d_ylen_;
        post.B = B; post.T = Ty;
        conv("flow.post_couple", cw(S("flow.%d.post", j)), post);
    }
    tap("z", d_z_, {B, I, Ty}, d_ylen_, 1);

    // ---- decoder
    const int C0 = c.upsample_initial_channel;
    ConvArgs cp;
    cp.x = d_z_; cp.x_bs = zbs; cp.x_ld = Ty;
    cp.y = d_bufC_; cp.y_bs = (long)C0 * Ty; cp.y_ld = Ty;
    cp.in_len = d_ylen_;
    cp.cond = d_cond_dec_; cp.cond_bs = C0;
    cp.B = B; cp.T = Ty;
    conv("dec.conv_pre", cw("dec.conv_pre"), cp);
    tap("dec.conv_pre", d_bufC_, {B, C0, Ty}, d_ylen_, 1);

    int ch = C0;
    long T = Ty;
    const int nk = c.n_resblock_kernels;
    HIP_CHECK(hipMemsetAsync(d_peaks_, 0, sizeof(unsigned) * B, stream_));
    for (int i = 0; i < c.n_upsamples; ++i) {
        const int r = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
        const ConvW& up = cw(S("dec.ups.%d.poly", i));
        if (!force_generic_ && up.packed != NO_OFF) {
            ConvArgs u;
            u.x = d_bufC_; u.x_bs = (long)ch * T; u.x_ld = (int)T;
            u.y = d_bufA_; u.y_bs = (long)(ch / 2) * T * r; u.y_ld = (int)(T * r);
            u.in_slope = 0.1f; u.in_len = d_slen_ + (long)i * B;
            u.pad = up.K - 1; u.Tin = (int)T;
            u.shuf_s = r; u.shuf_p = (k - r) / 2; u.shuf_cout = ch / 2; u.shuf_T = (int)(T * r);
            u.B = B; u.T = (int)T + up.K - 1;
            conv(i == 0 ? "dec.upsample.s0" : (i == 1 ? "dec.upsample.s1" : "dec.upsample.s2+"), up, u);
        } else {
            ConvTArgs u;
            u.x = d_bufC_; u.x_bs = (long)ch * T; u.x_ld = (int)T;
            u.y = d_bufA_; u.y_bs = (long)(ch / 2) * T * r; u.y_ld = (int)(T * r);
            u.w = vec(S("dec.ups.%d.weight", i)); u.bias = vec(S("dec.ups.%d.bias", i));
            u.B = B; u.Cin = ch; u.Cout = ch / 2; u.Tin = (int)T; u.K = k; u.stride = r; u.pad = (k - r) / 2;
            u.in_slope = 0.1f;
            u.in_len = d_slen_ + (long)i * B;
            const int taps = (k + r - 1) / r;
            ProfScope ps(prof_, "dec.upsample", 2.0 * B * (double)T * r * ch * (ch / 2) * taps,
                         4.0 * B * ((double)ch * T + (double)(ch / 2) * T * r));
            launch_conv_transpose1d(u, stream_);
        }
        ch /= 2;
        T *= r;
        const long sbs = (long)ch * T;
        const int* slen = d_slen_ + (long)(i + 1) * B;  // rows end at their own length (batched == unbatched)
        tap(S("dec.ups.%d", i).c_str(), d_bufA_, {B, ch, T}, d_ylen_, (int)(T / Ty));
        int n_fused = 0;  // resblocks 0 .. n_fused-1 of this stage run in the fused kernel, the rest conv by conv
        if (!force_generic_ && !no_fused_mrf_ && c.resblock == 2 && nk <= MRF_MAX_RB) {
            MrfArgs m;
            bool two = true;
            for (int j = 0; j < nk; ++j) two = two && c.resblock_n_dilations[j] == 2;
            if (two) {
                for (int j = 0; j < nk; ++j) {
                    m.k[j] = c.resblock_kernel_sizes[j];
                    m.d1[j] = c.resblock_dilations[j * MI355VITS_MAX_STAGES + 0];
                    m.d2[j] = c.resblock_dilations[j * MI355VITS_MAX_STAGES + 1];
                }
                // MATH_BF16X3: the whole stage on planes split once, weights in registers (k_mrf_p)
                bool all_p = math_ == MATH_BF16X3 && !no_mrf_p_ && mrf_p_supported(ch, nk, m.k, m.d1, m.d2);
                for (int j = 0; j < nk && all_p; ++j)
                    for (int q = 0; q < 2; ++q) all_p = all_p && cw(S("dec.rb.%d.c.%d", i * nk + j, q)).packed_p != NO_OFF;
                if (all_p) {
                    double flops = 0;
                    for (int j = 0; j < nk; ++j) {
                        for (int q = 0; q < 2; ++q) {
                            const ConvW& w = cw(S("dec.rb.%d.c.%d", i * nk + j, q));
                            m.w[j][q] = P(w.packed_p);
                            m.bias[j][q] = P(w.bias);
                        }
                        flops += 2.0 * 2.0 * B * (double)T * ch * ch * m.k[j];
                    }
                    m.nrb = nk;
                    m.math = MATH_BF16X3;
                    m.x = d_bufA_; m.x_bs = sbs; m.x_ld = (int)T;
                    m.y = d_bufC_; m.y_bs = sbs; m.y_ld = (int)T;
                    m.len = slen; m.B = B; m.C = ch; m.T = (int)T;
                    m.len_host = h_slen_.data() + (size_t)(i + 1) * B;
                    // large grids: the row sweep (k_mrf_s: fragments register-resident per segment, no halo recompute); small ones:
                    // (row, column block) items (k_mrf_p).  The two agree bit for bit, so the choice may follow the grid.
                    // 64 channels: one pass per resblock (k_mrf_s).  32 channels: k_mrf_p (its single-pass sweep, k_mrf_s1, measured equal in
                    // round 4 — 2.33 vs 2.34 ms, 2.37 vs 2.36 with 48-column steps — and was deleted in round 5: DESIGN.md §6)
                    bool sw_ok = ch != 32 && mrf_s_supported(ch, nk, m.k, m.d1, m.d2);
                    int seg = !sw_ok ? 0 : mrf_s_segment(ch, B, (int)T, current_device_cu_count(), lab_getenv("MI355VITS_MRFS_UNIFORM_SEG") ? nullptr : m.len_host);
                    if (const char* f = lab_getenv("MI355VITS_MRF_SWEEP_SEG")) seg = sw_ok ? atoi(f) : 0;  // lab / tests
                    if (seg > 0) {
                        m.seg = seg;
                        ProfScope ps(prof_, i == 1 ? "dec.mrf_s.s1" : (i == 2 ? "dec.mrf_s.s2" : "dec.mrf_s"), flops, 8.0 * B * (double)T * ch);
                        launch_mrf_s(m, stream_);
                    } else {
                        ProfScope ps(prof_, i == 1 ? "dec.mrf_p.s1" : (i == 2 ? "dec.mrf_p.s2" : "dec.mrf_p"), flops, 8.0 * B * (double)T * ch);
                        launch_mrf_p(m, stream_);
                    }
                    n_fused = nk;
                }
                // 128 channels in MATH_BF16X3: conv by conv on k_rb_conv (every input channel resident in LDS, kernels_rbc.cpp)
                bool rbc_stage = !n_fused && math_ == MATH_BF16X3 && !no_rbc_;
                for (int j = 0; j < nk && rbc_stage; ++j)
                    for (int q = 0; q < 2; ++q) {
                        const ConvW& w = cw(S("dec.rb.%d.c.%d", i * nk + j, q));
                        ConvArgs probe;
                        probe.Cin = w.Cin; probe.Cout = w.Cout; probe.K = w.K; probe.dil = q == 0 ? m.d1[j] : m.d2[j];
                        probe.pad = (w.K - 1) / 2 * probe.dil; probe.res = d_bufA_; probe.in_len = slen; probe.T = (int)T;
                        probe.x_ld = probe.res_ld = probe.y_ld = (int)T;  // (the stage's tensors are dense rows: what the launches below pass)
                        rbc_stage = rbc_stage && rbc_ok(w, probe);
                    }
                // the longest prefix of resblocks whose tiles fit LDS together (128 channels: only the narrow ones)
                int p = (n_fused || rbc_stage) ? 0 : nk;
                while (p > 0 && !(mrf_fused_supported(ch, p, m.k, m.d1, m.d2) && cw(S("dec.rb.%d.c.%d", i * nk, 0)).packed4 != NO_OFF)) --p;
                if (p > 0) {
                    double flops = 0;
                    bool b3 = math_on_bf16(kmath());
                    bool h2 = math_ == MATH_F16X2;
                    for (int j = 0; j < p; ++j)
                        for (int q = 0; q < 2; ++q) {
                            b3 = b3 && cw(S("dec.rb.%d.c.%d", i * nk + j, q)).packed_b3 != NO_OFF;
                            h2 = h2 && cw(S("dec.rb.%d.c.%d", i * nk + j, q)).packed_h2 != NO_OFF;
                        }
                    m.math = h2 ? (int)MATH_F16X2 : (b3 ? kmath() : (int)MATH_F32);
                    for (int j = 0; j < p; ++j) {
                        for (int q = 0; q < 2; ++q) {
                            const ConvW& w = cw(S("dec.rb.%d.c.%d", i * nk + j, q));
                            m.w[j][q] = P(h2 ? w.packed_h2 : (b3 ? w.packed_b3 : w.packed4));
                            m.bias[j][q] = P(w.bias);
                        }
                        flops += 2.0 * 2.0 * B * (double)T * ch * ch * m.k[j];
                    }
                    m.nrb = p;
                    if (p < nk) m.out_scale = 1.0f / nk;
                    m.x = d_bufA_; m.x_bs = sbs; m.x_ld = (int)T;
                    m.y = d_bufC_; m.y_bs = sbs; m.y_ld = (int)T;
                    m.len = slen; m.B = B; m.C = ch; m.T = (int)T;
                    m.len_host = h_slen_.data() + (size_t)(i + 1) * B;
                    const double bytes = 8.0 * B * (double)T * ch;
                    ProfScope ps(prof_, i == 1 ? "dec.mrf_fused.s1" : (i == 2 ? "dec.mrf_fused.s2" : (i == 0 ? "dec.mrf_fused.s0" : "dec.mrf_fused")), flops,
                                 bytes);
                    launch_mrf_fused(m, stream_);
                    n_fused = p;
                }
            }
        }
        for (int j = n_fused; j < nk; ++j) {
            const int n = i * nk + j;
            const int nd = c.resblock_n_dilations[j];
            const float* src = d_bufA_;
            float* pp[2] = {d_bufB_, d_bufT_};
            for (int m = 0; m < nd; ++m) {
                const int dil = c.resblock_dilations[j * MI355VITS_MAX_STAGES + m];
                const bool last = (m == nd - 1);
                if (c.resblock == 2) {
                    // x = x + conv_{k,d}(lrelu(x))
                    float* dst = last ? d_bufC_ : pp[m & 1];
                    ConvArgs a;
                    a.x = src; a.x_bs = sbs; a.x_ld = (int)T;
                    a.y = dst; a.y_bs = sbs; a.y_ld = (int)T;
                    a.res = src; a.res_bs = sbs; a.res_ld = (int)T;
                    a.in_slope = 0.1f; a.dil = dil; a.in_len = slen;
                    if (last) { a.out_scale = 1.0f / nk; a.accumulate = (j > 0); }
                    a.B = B; a.T = (int)T;
                    conv(i == 0 ? "dec.rb.s0" : (i == 1 ? "dec.rb.s1" : "dec.rb.s2+"), cw(S("dec.rb.%d.c.%d", n, m)), a);
                    src = dst;
                } else {
                    // xt = c2(lrelu(c1(lrelu(x)))); x = xt + x
                    // the pair's output may overwrite its own residual source in place (c2 reads `mid`, and each
                    // thread reads res[co,t] before writing y[co,t]); only the stage input bufA must survive.
                    float* mid = d_bufT_;
                    float* dst = last ? d_bufC_ : d_bufB_;
                    ConvArgs a1;
                    a1.x = src; a1.x_bs = sbs; a1.x_ld = (int)T;
                    a1.y = mid; a1.y_bs = sbs; a1.y_ld = (int)T;
                    a1.in_slope = 0.1f; a1.dil = dil; a1.in_len = slen;
                    a1.B = B; a1.T = (int)T;
                    conv("dec.rb1.c1", cw(S("dec.rb.%d.c1.%d", n, m)), a1);
                    ConvArgs a2;
                    a2.x = mid; a2.x_bs = sbs; a2.x_ld = (int)T;
                    a2.y = dst; a2.y_bs = sbs; a2.y_ld = (int)T;
                    a2.res = src; a2.res_bs = sbs; a2.res_ld = (int)T;
                    a2.in_slope = 0.1f; a2.dil = 1; a2.in_len = slen;
                    if (last) { a2.out_scale = 1.0f / nk; a2.accumulate = (j > 0); }
                    a2.B = B; a2.T = (int)T;
                    conv("dec.rb1.c2", cw(S("dec.rb.%d.c2.%d", n, m)), a2);
                    src = dst;
                }
            }
        }
        tap(S("dec.mrf.%d", i).c_str(), d_bufC_, {B, ch, T}, d_ylen_, (int)(T / Ty));
    }
    {
        ProfScope ps(prof_, "dec.conv_post_tanh", 2.0 * B * (double)T * ch * 7, 4.0 * B * (double)T * (ch + 1));
        launch_conv_post_tanh(d_bufC_, (long)ch * T, (int)T, vec("dec.conv_post.weight"), ch, 7, B, (int)T, d_alen_, d_audio_,
                              T, d_peaks_, stream_);
    }
}

// =================================================================================================
// one synthesis call
// =================================================================================================
namespace {
// Pinned host buffers for results, recycled process-wide: hipHostMalloc costs milliseconds for a 12 MB block (page
// pinning + IOMMU mapping), more than the D2H itself, so a buffer released by mi355vits_free_result goes back on a
// free list and the next call of any handle takes the smallest one that fits.  Buffers are handed out exclusively
// (a result stays valid until its free_result, whatever runs meanwhile); at most POOL_KEEP_BYTES stay cached.
class PinnedPool {
  public:
    static PinnedPool& get() {
        static PinnedPool* p = new PinnedPool();  // never destroyed: no hipHostFree after the runtime has shut down
        return *p;
    }
    void* take(size_t bytes, size_t* cap) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = free_.lower_bound(bytes);
            // a block more than twice as large as needed is left for a caller that needs it
            if (it != free_.end() && it->first <= 2 * bytes + (1 << 16)) {
                void* p = it->second;
                *cap = it->first;
                cached_ -= it->first;
                free_.erase(it);
                return p;
            }
        }
        const size_t want = ((bytes + (1 << 16) - 1) >> 16) << 16;  // 64 KiB granules: nearby sizes share blocks
        void* p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            trim(0);
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess)
                throw EngineError(MI355VITS_ERR_NOMEM, "out of pinned host memory (result buffer)");
        }
        *cap = want;
        return p;
    }
    void give(void* p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            free_.emplace(cap, p);
            cached_ += cap;
        }
        trim(POOL_KEEP_BYTES);
    }

  private:
    static constexpr size_t POOL_KEEP_BYTES = size_t(1) << 30;
    void trim(size_t keep) {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (cached_ > keep && !free_.empty()) {
                auto it = std::prev(free_.end());
                cached_ -= it->first;
                drop.push_back(it->second);
                free_.erase(it);
            }
        }
        for (void* p : drop) (void)hipHostFree(p);
    }
    std::mutex mu_;
    std::multimap<size_t, void*> free_;
    size_t cached_ = 0;
};

struct ResultOwner {
    void* audio = nullptr;
    size_t audio_cap = 0;
    void* pcm = nullptr;
    size_t pcm_cap = 0;
    void* lengths = nullptr;
    void* peaks = nullptr;
};
}  // namespace

void Engine::run(const mi355vits_run_args& args, mi355vits_result* out) {
    const mi355vits_config& c = cfg_;
    if (!out) throw EngineError(MI355VITS_ERR_INVALID, "result pointer is null");
    memset(out, 0, sizeof(*out));
    if (args.batch < 1 || args.tx_max < 1) throw EngineError(MI355VITS_ERR_INVALID, "batch and tx_max must be >= 1");
    if (!args.ids || !args.lengths || !args.scales) throw EngineError(MI355VITS_ERR_INVALID, "input, input_lengths and scales are required");
    const bool multi = c.n_speakers > 1;
    if (multi && !args.sid) throw EngineError(MI355VITS_ERR_INVALID, "multi-speaker voice: feed 'sid' is required");
    const int B = args.batch, Tx = args.tx_max;
    if ((long)B * Tx > (1L << 26)) throw EngineError(MI355VITS_ERR_INVALID, "batch * tx_max too large");
    {
        // beyond 512 ids the attention falls back to a kernel that keeps one score row per wave in LDS (64 KiB)
        const int d = c.hidden_channels / c.n_heads, tx_cap = (64 * 1024) / 16 - d - (2 * c.window_size + 1);
        if (Tx > 512 && Tx > tx_cap)
            throw EngineError(MI355VITS_ERR_INVALID, "phoneme sequence too long: at most " + std::to_string(tx_cap) +
                                                     " ids per utterance (split the text into sentences, as Mimic 3 does)");
    }
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(args.scales[i])) throw EngineError(MI355VITS_ERR_INVALID, "scales must be finite");
    if (args.scales[0] < 0 || args.scales[2] < 0) throw EngineError(MI355VITS_ERR_INVALID, "noise scales must be >= 0");
    if (!(args.scales[1] > 0)) throw EngineError(MI355VITS_ERR_INVALID, "length_scale must be > 0");
    std::vector<int> len32(B);
    for (int b = 0; b < B; ++b) {
        if (args.lengths[b] < 0 || args.lengths[b] > Tx) throw EngineError(MI355VITS_ERR_INVALID, "input_lengths out of range");
        len32[b] = (int)args.lengths[b];
        for (int t = 0; t < len32[b]; ++t) {
            const int64_t id = args.ids[(long)b * Tx + t];
            if (id < 0 || id >= c.num_symbols) throw EngineError(MI355VITS_ERR_INVALID, "phoneme id out of range");
        }
        if (multi && (args.sid[b] < 0 || args.sid[b] >= c.n_speakers)) throw EngineError(MI355VITS_ERR_INVALID, "speaker id out of range");
    }
    if (args.noise_z && args.noise_z_frames < 1) throw EngineError(MI355VITS_ERR_INVALID, "noise_z_frames must be >= 1");

    HIP_CHECK(hipSetDevice(device_));
    have_result_ = false;
    taps_on_ = (args.flags & MI355VITS_DEBUG_TAPS) != 0;
    for (auto& t : taps_) (void)hipFree(t.dev);
    taps_.clear();

    const int H = c.hidden_channels, F = c.filter_channels, I = c.inter_channels;
    const int nth = 3 * c.dp_num_bins - 1;
    const int C0 = c.upsample_initial_channel;
    const int gin = multi ? c.gin_channels : 0;
    auto pad = [](size_t bytes) { return DeviceArena::padded(bytes); };

    // ---------------- phase A workspace
    const size_t fBT = (size_t)B * Tx;
    size_t need_a = pad(fBT * 8) + pad((size_t)B * 8) + 6 * pad((size_t)B * 4) + 3 * pad(fBT * 4);
    need_a += 3 * pad(fBT * H * 4) + pad(fBT * 3 * H * 4) + pad(fBT * F * 4) + pad(fBT * 2 * I * 4);  // x,x2,att,qkv,ffn,stats
    need_a += 4 * pad(fBT * H * 4) + pad(fBT * nth * 4) + 2 * pad(fBT * 2 * 4) + pad(fBT * 4);          // h,d0,d1,d2,theta,z2,noise_w,logw
    if (F % 192 == 0 && F / 192 > 1 && H <= 256) need_a += pad(fBT * H * (F / 192) * 4);  // slice sums of the FFN's second conv
    if (gin) need_a += pad((size_t)B * H * 4) + pad((size_t)B * C0 * 4) + (size_t)c.flow_n_flows * pad((size_t)B * 2 * H * c.flow_wn_layers * 4);
    arena_a_.reserve(need_a + 4096, stream_);
    arena_a_.reset();
    // the call's host inputs sit side by side so that ONE host-to-device copy brings them all (each copy from pageable memory
    // is a staging pass + a copy kernel: ~35 us apiece in the stream of a single utterance)
    const size_t in_off = arena_a_.used();
    d_ids_ = arena_a_.alloc<long long>(fBT);
    d_sid_ = arena_a_.alloc<long long>(B);
    d_len_ = arena_a_.alloc<int>(B);
    d_forced_ = nullptr;
    if (args.forced_durations) d_forced_ = arena_a_.alloc<int>(fBT);
    d_noise_w_ = nullptr;
    if (args.noise_w) d_noise_w_ = arena_a_.alloc<float>(fBT * 2);
    const size_t in_bytes = arena_a_.used() - in_off;
    d_ylen_ = arena_a_.alloc<int>(B);
    d_peaks_ = arena_a_.alloc<unsigned>(B);
    d_wceil_ = arena_a_.alloc<int>(fBT);
    d_cum_ = arena_a_.alloc<int>(fBT);
    d_x_ = arena_a_.alloc<float>(fBT * H);
    d_x2_ = arena_a_.alloc<float>(fBT * H);
    d_att_ = arena_a_.alloc<float>(fBT * H);
    d_qkv_ = arena_a_.alloc<float>(fBT * 3 * H);
    d_ffn_ = arena_a_.alloc<float>(fBT * F);
    d_part_ = (F % 192 == 0 && F / 192 > 1 && H <= 256) ? arena_a_.alloc<float>(fBT * H * (F / 192)) : nullptr;
    d_stats_ = arena_a_.alloc<float>(fBT * 2 * I);
    d_h_ = arena_a_.alloc<float>(fBT * H);
    d_d0_ = arena_a_.alloc<float>(fBT * H);
    d_d1_ = arena_a_.alloc<float>(fBT * H);
    d_d2_ = arena_a_.alloc<float>(fBT * H);
    d_theta_ = arena_a_.alloc<float>(fBT * nth);
    d_z2_ = arena_a_.alloc<float>(fBT * 2);
    d_logw_ = arena_a_.alloc<float>(fBT);
    d_cond_dp_ = nullptr;
    d_cond_dec_ = nullptr;
    d_cond_flow_.clear();
    if (gin) {
        d_cond_dp_ = arena_a_.alloc<float>((size_t)B * H);
        d_cond_dec_ = arena_a_.alloc<float>((size_t)B * C0);
        for (int j = 0; j < c.flow_n_flows; ++j) d_cond_flow_.push_back(arena_a_.alloc<float>((size_t)B * 2 * H * c.flow_wn_layers));
    }

    HIP_CHECK(hipEventRecord(ev_start_, stream_));
    timed_ = false;
    {
        h_in_.assign(in_bytes, 0);
        unsigned char* h0 = h_in_.data();
        auto put = [&](const void* dev, const void* src, size_t bytes) {
            memcpy(h0 + (reinterpret_cast<const unsigned char*>(dev) - reinterpret_cast<const unsigned char*>(d_ids_)), src, bytes);
        };
        put(d_ids_, args.ids, fBT * 8);
        put(d_len_, len32.data(), (size_t)B * 4);
        if (multi) put(d_sid_, args.sid, (size_t)B * 8);
        if (args.forced_durations) put(d_forced_, args.forced_durations, fBT * 4);
        if (args.noise_w) put(d_noise_w_, args.noise_w, fBT * 2 * 4);
        // h_in_ is a member: it outlives the copy whatever HIP does with pageable sources
        HIP_CHECK(hipMemcpyAsync(d_ids_, h0, in_bytes, hipMemcpyHostToDevice, stream_));
    }

    if (gin) {
        ProfScope ps(prof_, "speaker_cond");
        launch_speaker_cond(vec("emb_g.weight"), d_sid_, vec("dp.cond.weight"), vec("dp.cond.bias"), B, gin, H, d_cond_dp_, stream_);
        launch_speaker_cond(vec("emb_g.weight"), d_sid_, vec("dec.cond.weight"), vec("dec.cond.bias"), B, gin, C0, d_cond_dec_, stream_);
        for (int j = 0; j < c.flow_n_flows; ++j) {
            const std::string f = S("flow.flows.%d", 2 * j);
            launch_speaker_cond(vec("emb_g.weight"), d_sid_, vec(f + ".enc.cond_layer.weight"), vec(f + ".enc.cond_layer.bias"), B, gin,
                                2 * H * c.flow_wn_layers, d_cond_flow_[j], stream_);
        }
    }

    B_ = B;
    Tx_ = Tx;
    text_encoder(B, Tx);
    duration_predictor(B, Tx, args);

    // ---------------- the one host round trip: frame counts size the rest of the graph
    h_ylen_.resize(B);
    HIP_CHECK(hipMemcpyAsync(h_ylen_.data(), d_ylen_, (size_t)B * 4, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    int Ty = 1;
    for (int b = 0; b < B; ++b) {
        if (h_ylen_[b] < 1 || h_ylen_[b] > DURATION_FRAME_CAP) throw EngineError(MI355VITS_ERR_INVALID, "predicted duration out of range (bad weights or length_scale?)");
        Ty = std::max(Ty, h_ylen_[b]);
    }
    if (args.noise_z && args.scales[0] != 0.0f && args.noise_z_frames < Ty)
        throw EngineError(MI355VITS_ERR_INVALID, "noise_z has fewer frames than the utterance needs");
    if (taps_on_) {
        // w_ceil as floats for the tap interface
        std::vector<int> wc(fBT);
        HIP_CHECK(hipMemcpy(wc.data(), d_wceil_, fBT * 4, hipMemcpyDeviceToHost));
        std::vector<float> wf32(wc.begin(), wc.end());
        Tap t;
        t.name = "w_ceil";
        t.dims = {B, 1, Tx};
        t.count = fBT;
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, fBT * 4 + 16));
        t.dev = static_cast<float*>(p);
        HIP_CHECK(hipMemcpy(t.dev, wf32.data(), fBT * 4, hipMemcpyHostToDevice));
        taps_.push_back(std::move(t));
    }
    Ty_ = Ty;
    const long hop = c.hop_length;
    L_ = (long)Ty * hop;

    // ---------------- phase B workspace
    size_t max_stage = (size_t)C0 * Ty;
    {
        size_t ch = C0, T = Ty;
        for (int i = 0; i < c.n_upsamples; ++i) {
            ch /= 2;
            T *= c.upsample_rates[i];
            max_stage = std::max(max_stage, ch * T);
        }
    }
    const size_t fBTy = (size_t)B * Ty;
    size_t need_b = pad(fBTy * I * 4) + 4 * pad(fBTy * H * 4);
    if (args.noise_z && args.scales[0] != 0.0f) need_b += pad((size_t)B * I * args.noise_z_frames * 4);
    need_b += 4 * pad((size_t)B * max_stage * 4) + pad((size_t)B * L_ * 4) + pad((size_t)B * L_ * 2);
    need_b += pad((size_t)(c.n_upsamples + 2) * B * 4);
    arena_b_.reserve(need_b + 4096, stream_);
    arena_b_.reset();
    d_z_ = arena_b_.alloc<float>(fBTy * I);
    d_fh_ = arena_b_.alloc<float>(fBTy * H);
    d_fh2_ = arena_b_.alloc<float>(fBTy * H);
    d_fskip_ = arena_b_.alloc<float>(fBTy * H);
    d_fu_ = arena_b_.alloc<float>(fBTy * H);
    d_noise_z_ = nullptr;
    if (args.noise_z && args.scales[0] != 0.0f) {
        d_noise_z_ = arena_b_.alloc<float>((size_t)B * I * args.noise_z_frames);
        HIP_CHECK(hipMemcpyAsync(d_noise_z_, args.noise_z, (size_t)B * I * args.noise_z_frames * 4, hipMemcpyHostToDevice, stream_));
    }
    d_bufA_ = arena_b_.alloc<float>((size_t)B * max_stage);
    d_bufB_ = arena_b_.alloc<float>((size_t)B * max_stage);
    d_bufT_ = arena_b_.alloc<float>((size_t)B * max_stage);
    d_bufC_ = arena_b_.alloc<float>((size_t)B * max_stage);
    d_audio_ = arena_b_.alloc<float>((size_t)B * L_);
    d_pcm_ = arena_b_.alloc<int16_t>((size_t)B * L_);
    // per-stage valid lengths and (last row) the audio lengths: one copy
    h_slen_.assign((size_t)(c.n_upsamples + 2) * B, 0);
    {
        long f = 1;
        for (int i = 0; i <= c.n_upsamples; ++i) {
            for (int b = 0; b < B; ++b) h_slen_[(size_t)i * B + b] = (int)(h_ylen_[b] * f);
            if (i < c.n_upsamples) f *= c.upsample_rates[i];
        }
        for (int b = 0; b < B; ++b) h_slen_[(size_t)(c.n_upsamples + 1) * B + b] = (int)(h_ylen_[b] * hop);
    }
    d_slen_ = arena_b_.alloc<int>(h_slen_.size());
    d_alen_ = d_slen_ + (size_t)(c.n_upsamples + 1) * B;
    HIP_CHECK(hipMemcpyAsync(d_slen_, h_slen_.data(), h_slen_.size() * 4, hipMemcpyHostToDevice, stream_));

    flow_and_decoder(B, Ty, args);

    have_pcm_ = false;
    if (args.flags & MI355VITS_WANT_PCM16) {
        ProfScope ps(prof_, "pcm16", 0, 6.0 * B * (double)L_);
        pcm_volume_ = (args.pcm_volume > 0.0) ? args.pcm_volume : 1.0;
        launch_pcm16(d_audio_, L_, d_peaks_, d_alen_, B, (int)L_, d_pcm_, L_, stream_, pcm_volume_);
        have_pcm_ = true;
    }
    HIP_CHECK(hipEventRecord(ev_end_, stream_));
    timed_ = true;
    have_result_ = true;
    copy_out(args.flags, out);
}

void Engine::copy_out(uint32_t want, mi355vits_result* out) {
    const int B = B_;
    auto* own = new ResultOwner();
    out->owner_ = own;
    out->batch = B;
    out->l_max = L_;
    out->ty_max = Ty_;
    own->lengths = malloc(sizeof(int64_t) * B);
    own->peaks = malloc(sizeof(float) * B);
    out->lengths = static_cast<int64_t*>(own->lengths);
    out->peaks = static_cast<float*>(own->peaks);
    if (!out->lengths || !out->peaks) throw EngineError(MI355VITS_ERR_NOMEM, "out of host memory");
    for (int b = 0; b < B; ++b) out->lengths[b] = (int64_t)h_ylen_[b] * cfg_.hop_length;
    std::vector<unsigned> pk(B);
    HIP_CHECK(hipMemcpyAsync(pk.data(), d_peaks_, sizeof(unsigned) * B, hipMemcpyDeviceToHost, stream_));
    const bool dev_only = (want & MI355VITS_DEVICE_ONLY) != 0;
    if (!dev_only && (want & MI355VITS_WANT_FLOAT)) {
        own->audio = PinnedPool::get().take(sizeof(float) * (size_t)B * L_ + 16, &own->audio_cap);
        out->audio = static_cast<float*>(own->audio);
        HIP_CHECK(hipMemcpyAsync(out->audio, d_audio_, sizeof(float) * (size_t)B * L_, hipMemcpyDeviceToHost, stream_));
    }
    if (!dev_only && (want & MI355VITS_WANT_PCM16)) {
        if (!have_pcm_) {
            launch_pcm16(d_audio_, L_, d_peaks_, d_alen_, B, (int)L_, d_pcm_, L_, stream_, 1.0);
            have_pcm_ = true;
        }
        own->pcm = PinnedPool::get().take(sizeof(int16_t) * (size_t)B * L_ + 16, &own->pcm_cap);
        out->pcm = static_cast<int16_t*>(own->pcm);
        HIP_CHECK(hipMemcpyAsync(out->pcm, d_pcm_, sizeof(int16_t) * (size_t)B * L_, hipMemcpyDeviceToHost, stream_));
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    for (int b = 0; b < B; ++b) memcpy(&out->peaks[b], &pk[b], 4);
}

void Engine::device_buffers(const int16_t** pcm, const float** audio, long* row_stride, int* batch, const int** dev_lengths) {
    if (!have_result_) throw EngineError(MI355VITS_ERR_INVALID, "device_buffers: no completed run on this handle");
    HIP_CHECK(hipSetDevice(device_));
    if (pcm && !have_pcm_) {
        launch_pcm16(d_audio_, L_, d_peaks_, d_alen_, B_, (int)L_, d_pcm_, L_, stream_, 1.0);
        have_pcm_ = true;
    }
    HIP_CHECK(hipStreamSynchronize(stream_));  // the caller reads them from another stream (RCCL)
    if (pcm) *pcm = d_pcm_;
    if (audio) *audio = d_audio_;
    if (row_stride) *row_stride = L_;
    if (batch) *batch = B_;
    if (dev_lengths) *dev_lengths = d_alen_;
}

void Engine::fetch(uint32_t want, mi355vits_result* out) {
    if (!out) throw EngineError(MI355VITS_ERR_INVALID, "result pointer is null");
    memset(out, 0, sizeof(*out));
    if (!have_result_) throw EngineError(MI355VITS_ERR_INVALID, "fetch: no completed run on this handle");
    HIP_CHECK(hipSetDevice(device_));
    copy_out(want & ~MI355VITS_DEVICE_ONLY, out);
}

void free_result_impl(mi355vits_result* r) {
    if (!r || !r->owner_) return;
    auto* own = static_cast<ResultOwner*>(r->owner_);
    PinnedPool::get().give(own->audio, own->audio_cap);
    PinnedPool::get().give(own->pcm, own->pcm_cap);
    free(own->lengths);
    free(own->peaks);
    delete own;
    memset(r, 0, sizeof(*r));
}

}  // namespace m355
