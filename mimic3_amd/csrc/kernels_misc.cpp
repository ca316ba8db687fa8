// kernels_misc.cpp — the small kernels of the path: embedding, channel LayerNorm, relative-position
// attention, the duration predictor's depthwise/spline pieces, length regulator + prior sampling,
// speaker conditioning.  All HBM/latency-bound; one pass over their tensors, pointwise work fused.
#include "kernels.h"
#include "b3.h"

namespace m355 {

// ------------------------------------------------------------------------------------------------
// counter-based Gaussian noise: Philox4x32-10 keyed by the seed, counter =
// (time index, channel, global utterance index, tensor id) -> the draw for an element depends neither on
// padding, batch composition nor on which GPU the utterance landed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long utt, unsigned tensor,
                                               unsigned c, unsigned t) {
    unsigned r[4];
    philox4x32_10(t, c, (unsigned)utt, tensor ^ (unsigned)(utt >> 32) * 0x9E3779B9u, (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// ------------------------------------------------------------------------------------------------
// K1: x[b,c,t] = Emb[ids[b,t]][c] * sqrt(H) * mask
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed(const long long* ids, const int* len, const float* emb, int T, int H,
                                               int num_symbols, float scale, float* y) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cg = threadIdx.x >> 6;
    if (t >= T) return;
    const bool valid = t < len[b];
    long long id = ids[(long)b * T + t];
    if (id < 0 || id >= num_symbols) id = 0;  // host validates; belt and braces
    for (int c = cg; c < H; c += 4) y[((long)b * H + c) * T + t] = valid ? emb[id * H + c] * scale : 0.0f;
}
void launch_embed(const long long* ids, const int* len, const float* emb, int B, int T, int H, int num_symbols,
                  float scale, float* y, hipStream_t s) {
    LAUNCH_KERNEL(k_embed, dim3((T + 63) / 64, B), dim3(256), 0, s, ids, len, emb, T, H, num_symbols, scale, y);
}

// ------------------------------------------------------------------------------------------------
// channel LayerNorm (A.3) over [B,C,T], optional residual input, GELU, additive output, mask.
// Every thread owns one (time column, channel group) slice, so global accesses run along time; three cached sweeps
// (mean, biased variance, write).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Workgroup = LN_TT time columns x LN_CG channel groups (256 threads).  16 x 16 keeps >= 256 workgroups in flight
// for the encoder-sized tensors ([B,192,128]) while a row of 16 consecutive samples is still a 64-byte segment.
constexpr int LN_TT = 16, LN_CG = 16;
__device__ __forceinline__ float block_cg_sum(float v, float* red, int tl, int cg) {
    __syncthreads();
    red[cg * LN_TT + tl] = v;
    __syncthreads();
    float s = 0.0f;
    MI355_UNROLL
    for (int g = 0; g < LN_CG; ++g) s += red[g * LN_TT + tl];
    return s;
}

// Channels per thread held in registers (C <= LN_CG * LN_NPT): the input is read ONCE, with all loads of a thread in flight
// together, and mean / variance / output come from the registers.  (Three sweeps with a test per element — `if (a.res)` —
// make hipcc wait for every load in turn: 36 dependent L2 round trips per thread for C = 192, 13 us per launch at batch 1.)
constexpr int LN_NPT = 16;

template <bool CACHED>
__global__ __launch_bounds__(256) void k_layernorm(LNArgs a) {
    DYN_SMEM(float, red);
    const int b = blockIdx.y;
    const int tl = threadIdx.x % LN_TT, cg = threadIdx.x / LN_TT;
    const int t = blockIdx.x * LN_TT + tl;
    const bool live = t < a.T;
    const long base = (long)b * a.C * a.T + (live ? t : 0);
    if (CACHED) {
        float v[LN_NPT];
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i) {
            const int c = cg + i * LN_CG;
            v[i] = a.x[base + (long)(c < a.C ? c : a.C - 1) * a.T];  // clamped: unconditional loads, batched
        }
        for (int p = 1; p < a.nparts; ++p) {  // split conv: the slices' sums in slice order
            float q[LN_NPT];
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) {
                const int c = cg + i * LN_CG;
                q[i] = a.x[(long)p * a.part_stride + base + (long)(c < a.C ? c : a.C - 1) * a.T];
            }
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) v[i] += q[i];
        }
        if (a.bias) {
            float q[LN_NPT];
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) {
                const int c = cg + i * LN_CG;
                q[i] = a.bias[c < a.C ? c : a.C - 1];
            }
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) v[i] += q[i];
        }
        if (a.premask_len) {
            const bool dead = t >= a.premask_len[b];
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) v[i] = dead ? 0.0f : v[i];
        }
        if (a.res) {
            float r[LN_NPT];
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) {
                const int c = cg + i * LN_CG;
                r[i] = a.res[base + (long)(c < a.C ? c : a.C - 1) * a.T];
            }
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) v[i] += r[i];
        }
        float sum = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i)
            if (live && cg + i * LN_CG < a.C) sum += v[i];  // same order of additions as the sweep version: c ascending
        const float mean = block_cg_sum(sum, red, tl, cg) / (float)a.C;
        float sq = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i)
            if (live && cg + i * LN_CG < a.C) { const float d = v[i] - mean; sq += d * d; }
        const float var = block_cg_sum(sq, red, tl, cg) / (float)a.C;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        __syncthreads();  // in-place use: all reads of x above are done before anyone writes y
        if (!live) return;
        const bool masked = a.out_len && t >= a.out_len[b];
        float g[LN_NPT], be[LN_NPT], ad[LN_NPT];
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i) {
            const int c = cg + i * LN_CG, cc = c < a.C ? c : a.C - 1;
            g[i] = a.gamma[cc];
            be[i] = a.beta[cc];
        }
        if (a.add_to) {
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) {
                const int c = cg + i * LN_CG;
                ad[i] = a.add_to[base + (long)(c < a.C ? c : a.C - 1) * a.T];
            }
        }
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i) {
            const int c = cg + i * LN_CG;
            float y = (v[i] - mean) * rstd * g[i] + be[i];
            if (a.gelu) y = gelu_erf(y);
            if (a.add_to) y += ad[i];
            if (masked) y = 0.0f;
            if (c < a.C) a.y[base + (long)c * a.T] = y;
        }
        return;
    }
    float sum = 0.0f;
    if (live)
        for (int c = cg; c < a.C; c += LN_CG) {
            float v = a.x[base + (long)c * a.T];
            if (a.res) v += a.res[base + (long)c * a.T];
            sum += v;
        }
    const float mean = block_cg_sum(sum, red, tl, cg) / (float)a.C;
    float sq = 0.0f;
    if (live)
        for (int c = cg; c < a.C; c += LN_CG) {
            float v = a.x[base + (long)c * a.T];
            if (a.res) v += a.res[base + (long)c * a.T];
            v -= mean;
            sq += v * v;
        }
    const float var = block_cg_sum(sq, red, tl, cg) / (float)a.C;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    __syncthreads();  // in-place use: all reads of x above are done before anyone writes y
    if (!live) return;
    const bool masked = a.out_len && t >= a.out_len[b];
    for (int c = cg; c < a.C; c += LN_CG) {
        float v = a.x[base + (long)c * a.T];
        if (a.res) v += a.res[base + (long)c * a.T];
        v = (v - mean) * rstd * a.gamma[c] + a.beta[c];
        if (a.gelu) v = gelu_erf(v);
        if (a.add_to) v += a.add_to[base + (long)c * a.T];
        if (masked) v = 0.0f;
        a.y[base + (long)c * a.T] = v;
    }
}
void launch_layernorm(const LNArgs& a, hipStream_t s) {
    if (a.T <= 0) return;
    if ((a.nparts > 1 || a.bias || a.premask_len) && a.C > LN_CG * LN_NPT) throw std::runtime_error("layernorm: split-conv input needs C <= 256");
    if (a.C <= LN_CG * LN_NPT) LAUNCH_KERNEL(k_layernorm<true>, dim3((a.T + LN_TT - 1) / LN_TT, a.B), dim3(256), 256 * sizeof(float), s, a);
    else LAUNCH_KERNEL(k_layernorm<false>, dim3((a.T + LN_TT - 1) / LN_TT, a.B), dim3(256), 256 * sizeof(float), s, a);
}

// ------------------------------------------------------------------------------------------------
// first half of a DDS layer (A.6): y = gelu(LN_c(depthwise_conv_{K,dil}(x * mask)))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dwconv_at(const float* xrow, const float* w, float bias, int t, int T, int len, int K,
                                           int dil) {
    float acc = bias;
    const int pad = (K * dil - dil) / 2;
    for (int k = 0; k < K; ++k) {
        const int tt = t - pad + k * dil;
        if (tt >= 0 && tt < T && tt < len) acc = fmaf(w[k], xrow[tt], acc);
    }
    return acc;
}
// CACHED (C <= LN_CG * LN_NPT, K <= 3... any K via the inner loop): the depthwise result of a thread's channels is computed
// once into registers — taps read through clamped indices and a 0/1 factor instead of a test per tap, so all loads of a
// thread are in flight together — and mean / variance / output come from the registers (same arithmetic, same order).
template <bool CACHED>
__global__ __launch_bounds__(256) void k_dds_dwconv_ln_gelu(const float* x, const float* w, const float* bias,
                                                            const float* gamma, const float* beta, const int* len,
                                                            int C, int T, int K, int dil, float* y) {
    DYN_SMEM(float, red);
    const int b = blockIdx.y;
    const int tl = threadIdx.x % LN_TT, cg = threadIdx.x / LN_TT;
    const int t = blockIdx.x * LN_TT + tl;
    const bool live = t < T;
    const int L = len[b];
    const float* xb = x + (long)b * C * T;
    if (CACHED) {
        const int pad = (K * dil - dil) / 2;
        const int tend = L < T ? L : T;
        float v[LN_NPT];
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i) {
            const int c = cg + i * LN_CG, cc = c < C ? c : C - 1;
            v[i] = bias[cc];
        }
        for (int k = 0; k < K; ++k) {
            const int tt = (live ? t : 0) - pad + k * dil;
            const bool in = tt >= 0 && tt < tend;
            const int tc = in ? tt : 0;
            float xv[LN_NPT], wv[LN_NPT];
            MI355_UNROLL
            for (int i = 0; i < LN_NPT; ++i) {
                const int c = cg + i * LN_CG, cc = c < C ? c : C - 1;
                xv[i] = xb[(long)cc * T + tc];
                wv[i] = w[cc * K + k];
            }
            if (in) {  // (a skipped tap leaves acc untouched, exactly like the test in dwconv_at)
                MI355_UNROLL
                for (int i = 0; i < LN_NPT; ++i) v[i] = fmaf(wv[i], xv[i], v[i]);
            }
        }
        float sum = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i)
            if (live && cg + i * LN_CG < C) sum += v[i];
        const float mean = block_cg_sum(sum, red, tl, cg) / (float)C;
        float sq = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i)
            if (live && cg + i * LN_CG < C) { const float d = v[i] - mean; sq += d * d; }
        const float var = block_cg_sum(sq, red, tl, cg) / (float)C;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (!live) return;
        float g[LN_NPT], be[LN_NPT];
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i) {
            const int c = cg + i * LN_CG, cc = c < C ? c : C - 1;
            g[i] = gamma[cc];
            be[i] = beta[cc];
        }
        MI355_UNROLL
        for (int i = 0; i < LN_NPT; ++i) {
            const int c = cg + i * LN_CG;
            if (c < C) y[((long)b * C + c) * T + t] = gelu_erf((v[i] - mean) * rstd * g[i] + be[i]);
        }
        return;
    }
    float sum = 0.0f;
    if (live)
        for (int c = cg; c < C; c += LN_CG) sum += dwconv_at(xb + (long)c * T, w + c * K, bias[c], t, T, L, K, dil);
    const float mean = block_cg_sum(sum, red, tl, cg) / (float)C;
    float sq = 0.0f;
    if (live)
        for (int c = cg; c < C; c += LN_CG) {
            const float v = dwconv_at(xb + (long)c * T, w + c * K, bias[c], t, T, L, K, dil) - mean;
            sq += v * v;
        }
    const float var = block_cg_sum(sq, red, tl, cg) / (float)C;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (!live) return;
    for (int c = cg; c < C; c += LN_CG) {
        float v = dwconv_at(xb + (long)c * T, w + c * K, bias[c], t, T, L, K, dil);
        v = (v - mean) * rstd * gamma[c] + beta[c];
        y[((long)b * C + c) * T + t] = gelu_erf(v);
    }
}
void launch_dds_dwconv_ln_gelu(const float* x, const float* w, const float* bias, const float* gamma,
                               const float* beta, const int* len, int B, int C, int T, int K, int dil, float* y,
                               hipStream_t s) {
    if (C <= LN_CG * LN_NPT)
        LAUNCH_KERNEL(k_dds_dwconv_ln_gelu<true>, dim3((T + LN_TT - 1) / LN_TT, B), dim3(256), 256 * sizeof(float), s, x, w, bias,
                      gamma, beta, len, C, T, K, dil, y);
    else
        LAUNCH_KERNEL(k_dds_dwconv_ln_gelu<false>, dim3((T + LN_TT - 1) / LN_TT, B), dim3(256), 256 * sizeof(float), s, x, w, bias,
                      gamma, beta, len, C, T, K, dil, y);
}

// ------------------------------------------------------------------------------------------------
// One whole DDS layer (A.6) in ONE launch:  x += gelu(LN2(conv1x1(gelu(LN1(dwconv_{K,dil}(x * mask))))))
// A workgroup owns 32 time columns of all C channels (C / 32 waves): the depthwise conv + LN1 + GELU result goes to LDS
// ([C][32], the B operand of the 1x1 conv), wave w computes output rows 32 w .. on the f32 matrix cores (weights: the
// packed A fragments of the conv, streamed from L2), the raw result replaces the tile in LDS, LN2 + GELU + residual are
// applied from there.  Three launches (9 + 18 + 7 us at batch 1) become one; the text side of the graph is bound by
// launch latency, not by bytes (a [32, 192, 128] tensor is 3 MB).  x is updated in place (a column's inputs are the
// neighbouring columns of the OLD x: they are read before any workgroup writes only within its own columns — the
// depthwise halo of neighbouring workgroups therefore needs the OLD values: the kernel reads x and writes y = x_new to a
// second buffer; the engine ping-pongs).
// ------------------------------------------------------------------------------------------------
struct DdsLayerArgs {
    const float* x; float* y;          // [B, C, T] in / out (different buffers)
    const float* dw_w; const float* dw_b;   // depthwise [C, K], [C]
    const float* g1; const float* b1;       // LN1
    const float* w1x1;                      // packed f32 A fragments [C/32][1][C/2][64]
    const float* bias1x1;                   // [C]
    const float* g2; const float* b2;       // LN2
    const int* len;
    int B, C, T, K, dil;
};

template <int NW>  // C = 32 NW
__global__ __launch_bounds__(64 * NW) void k_dds_layer(DdsLayerArgs a) {
    constexpr int C = 32 * NW, NCG = 2 * NW, NTH = 64 * NW, NPT = 16;  // 16 channels per thread: c = cg + NCG i
    DYN_SMEM(float, smem);
    float* Y = smem;            // [C][32]
    float* red = smem + C * 32; // [NCG][32]
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int col = tid & 31, cg = tid >> 5;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 32;
    const int t = t0 + col;
    const bool live = t < a.T;
    const int L = a.len[b];
    const int tend = L < a.T ? L : a.T;
    const float* xb = a.x + (long)b * C * a.T;
    auto col_sum = [&](float v) {  // sum over the NCG channel groups of this column (fixed order)
        __syncthreads();
        red[cg * 32 + col] = v;
        __syncthreads();
        float s = 0.0f;
        MI355_UNROLL
        for (int g = 0; g < NCG; ++g) s += red[g * 32 + col];
        return s;
    };
    // ---- depthwise conv (taps through clamped indices: all loads in flight together) + LN1 + GELU -> Y
    float v[NPT];
    {
        const int pad = (a.K * a.dil - a.dil) / 2;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) v[i] = a.dw_b[cg + NCG * i];
        for (int k = 0; k < a.K; ++k) {
            const int tt = (live ? t : 0) - pad + k * a.dil;
            const bool in = tt >= 0 && tt < tend;
            const int tc = in ? tt : 0;
            float xv[NPT], wv[NPT];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                const int c = cg + NCG * i;
                xv[i] = xb[(long)c * a.T + tc];
                wv[i] = a.dw_w[c * a.K + k];
            }
            if (in) {
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) v[i] = fmaf(wv[i], xv[i], v[i]);
            }
        }
        float sum = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) sum += v[i];
        const float mean = col_sum(sum) / (float)C;
        float sq = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) { const float d = v[i] - mean; sq += d * d; }
        const float rstd = 1.0f / sqrtf(col_sum(sq) / (float)C + 1e-5f);
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) {
            const int c = cg + NCG * i;
            Y[c * 32 + col] = live ? gelu_erf((v[i] - mean) * rstd * a.g1[c] + a.b1[c]) : 0.0f;
        }
    }
    __syncthreads();
    // ---- 1x1 conv on the f32 matrix cores: wave w -> rows 32 w .. 32 w + 31, the 32 columns; k-steps = channel pairs
    const int brow = lane >> 5, bcol = lane & 31;
    f32x16 acc;
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
        const float* wp = a.w1x1 + (long)w * (C / 2) * 64 + lane;
        const float* yb = Y + brow * 32 + bcol;
        float ra[8];
        MI355_UNROLL
        for (int u = 0; u < 4; ++u) ra[u] = wp[u * 64];
        for (int cp0 = 0; cp0 < C / 2; cp0 += 8) {
            MI355_UNROLL
            for (int u = 0; u < 8; ++u) {
                const int nxt = cp0 + u + 4 < C / 2 ? cp0 + u + 4 : C / 2 - 1;  // the tail re-reads the last record
                ra[(u + 4) & 7] = wp[nxt * 64];
                acc = MFMA_32x32x2_F32(ra[u], yb[(cp0 + u) * 64], acc);
            }
        }
    }
    float bias[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) bias[r] = a.bias1x1[32 * w + (r & 3) + 8 * (r >> 2) + 4 * brow];
    __syncthreads();  // every wave is done reading Y: the raw conv result takes its place
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) Y[(32 * w + (r & 3) + 8 * (r >> 2) + 4 * brow) * 32 + bcol] = acc[r] + bias[r];
    __syncthreads();
    // ---- LN2 + GELU + residual
    {
        float z[NPT], xr[NPT];
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) {
            const int c = cg + NCG * i;
            z[i] = Y[c * 32 + col];
            xr[i] = xb[(long)c * a.T + (live ? t : 0)];
        }
        float sum = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) sum += z[i];
        const float mean = col_sum(sum) / (float)C;
        float sq = 0.0f;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) { const float d = z[i] - mean; sq += d * d; }
        const float rstd = 1.0f / sqrtf(col_sum(sq) / (float)C + 1e-5f);
        if (live) {
            float* yo = a.y + (long)b * C * a.T + t;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                const int c = cg + NCG * i;
                yo[(long)c * a.T] = xr[i] + gelu_erf((z[i] - mean) * rstd * a.g2[c] + a.b2[c]);
            }
        }
    }
}

bool dds_layer_fused_supported(int C) { return C == 32 || C == 64 || C == 128 || C == 192 || C == 256; }

void launch_dds_layer(const float* x, float* y, const float* dw_w, const float* dw_b, const float* g1, const float* b1,
                      const float* w1x1_packed, const float* bias1x1, const float* g2, const float* b2, const int* len, int B,
                      int C, int T, int K, int dil, hipStream_t s) {
    if (T <= 0 || B <= 0) return;
    DdsLayerArgs a{x, y, dw_w, dw_b, g1, b1, w1x1_packed, bias1x1, g2, b2, len, B, C, T, K, dil};
    dim3 grid((T + 31) / 32, B);
    const size_t sh = ((size_t)C * 32 + (size_t)(C / 16) * 32) * sizeof(float);
    switch (C / 32) {
        case 1: LAUNCH_KERNEL(k_dds_layer<1>, grid, dim3(64), sh, s, a); break;
        case 2: LAUNCH_KERNEL(k_dds_layer<2>, grid, dim3(128), sh, s, a); break;
        case 4: LAUNCH_KERNEL(k_dds_layer<4>, grid, dim3(256), sh, s, a); break;
        case 6: LAUNCH_KERNEL(k_dds_layer<6>, grid, dim3(384), sh, s, a); break;
        case 8: LAUNCH_KERNEL(k_dds_layer<8>, grid, dim3(512), sh, s, a); break;
        default: throw std::runtime_error("dds_layer: unsupported channel count");
    }
}

// ------------------------------------------------------------------------------------------------
// relative-position multi-head attention (A.4).  One wave per query row; scores and probabilities of the
// row live in LDS; window terms E_k / E_v enter as 2W+1 extra logits / values.
//   s[i,j] = (q_i/sqrt(d)) . k_j + [|j-i|<=W] (q_i/sqrt(d)) . E_k[j-i+W];  masked -> -1e4;  softmax_j
//   o_i    = sum_j p[i,j] v_j + sum_{|j-i|<=W} p[i,j] E_v[j-i+W]
// ------------------------------------------------------------------------------------------------
constexpr int ATT_ROWS = 16;  // query rows per workgroup (4 per wave)

__global__ __launch_bounds__(256) void k_rel_attention(const float* qkv, const float* ek, const float* ev,
                                                       const int* len, int T, int H, int nh, int W, float* out) {
    DYN_SMEM(float, smem);
    const int d = H / nh;
    const int nrel = 2 * W + 1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int per_wave = d + T + nrel;
    float* qs = smem + wid * per_wave;  // [d]
    float* p = qs + d;                  // [T]
    float* rl = p + T;                  // [2W+1]
    const int b = blockIdx.z, h = blockIdx.y;
    const int L = len[b];
    const float scale = 1.0f / sqrtf((float)d);
    const float* qb = qkv + ((long)b * 3 * H + h * d) * T;
    const float* kb = qb + (long)H * T;
    const float* vb = qb + (long)2 * H * T;
    for (int n = 0; n < ATT_ROWS / 4; ++n) {
        const int i_raw = blockIdx.x * ATT_ROWS + n * 4 + wid;
        const bool row_live = i_raw < T;
        const int i = row_live ? i_raw : T - 1;
        for (int c = lane; c < d; c += 64) qs[c] = qb[(long)c * T + i] * scale;
        __syncthreads();
        for (int r = 0; r < nrel; ++r) {
            float part = 0.0f;
            for (int c = lane; c < d; c += 64) part = fmaf(qs[c], ek[r * d + c], part);
            part = wave_reduce_sum(part);
            if (lane == 0) rl[r] = part;
        }
        __syncthreads();
        float mx = -3.0e38f;
        for (int j0 = 0; j0 < T; j0 += 64) {
            const int j = j0 + lane;
            if (j < T) {
                float sc = 0.0f;
                for (int c = 0; c < d; ++c) sc = fmaf(qs[c], kb[(long)c * T + j], sc);
                const int rel = j - i;
                if (rel >= -W && rel <= W) sc += rl[rel + W];
                if (j >= L || i >= L) sc = -1e4f;
                p[j] = sc;
                mx = fmaxf(mx, sc);
            }
        }
        mx = wave_reduce_max(mx);
        float sum = 0.0f;
        for (int j0 = 0; j0 < T; j0 += 64) {
            const int j = j0 + lane;
            if (j < T) {
                const float e = expf(p[j] - mx);
                p[j] = e;
                sum += e;
            }
        }
        sum = wave_reduce_sum(sum);
        __syncthreads();
        const float inv = 1.0f / sum;
        for (int c = lane; c < d; c += 64) {
            const float* vr = vb + (long)c * T;
            float o = 0.0f;
            for (int j = 0; j < T; ++j) o = fmaf(p[j], vr[j], o);
            for (int r = 0; r < nrel; ++r) {
                const int j = i + r - W;
                if (j >= 0 && j < T) o = fmaf(p[j], ev[r * d + c], o);
            }
            if (row_live) out[((long)b * H + h * d + c) * T + i] = o * inv;
        }
        __syncthreads();
    }
}
void launch_rel_attention(const float* qkv, const float* emb_rel_k, const float* emb_rel_v, const int* len, int B,
                          int T, int H, int n_heads, int window, float* out, hipStream_t s) {
    const int d = H / n_heads;
    const size_t shmem = sizeof(float) * 4 * (size_t)(d + T + 2 * window + 1);
    if (shmem > 64 * 1024) throw std::runtime_error("rel_attention: phoneme sequence too long for the LDS row buffer");
    LAUNCH_KERNEL(k_rel_attention, dim3((T + ATT_ROWS - 1) / ATT_ROWS, n_heads, B), dim3(256), shmem, s, qkv, emb_rel_k,
                  emb_rel_v, len, T, H, n_heads, window, out);
}

// ------------------------------------------------------------------------------------------------
// stochastic duration predictor pieces (A.7)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_convflow_pre(const float* z, int ch, const float* w, const float* bias,
                                                      const float* g, int C, int T, float* h) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cg = threadIdx.x >> 6;
    if (t >= T) return;
    const float zv = z[((long)b * 2 + ch) * T + t];
    for (int c = cg; c < C; c += 4) {
        const long o = ((long)b * C + c) * T + t;
        h[o] = fmaf(w[c], zv, bias[c]) + g[o];
    }
}
void launch_convflow_pre(const float* z, int ch, const float* w, const float* bias, const float* g, int B, int C,
                         int T, float* h, hipStream_t s) {
    LAUNCH_KERNEL(k_convflow_pre, dim3((T + 63) / 64, B), dim3(256), 0, s, z, ch, w, bias, g, C, T, h);
}

constexpr int NB_MAX = 16;

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// Rational-quadratic spline inverse with linear tails (HF modeling_vits.py:93-302) of one value x1 inside the tail
// bound; th[i * stride] = the 3 nb - 1 unnormalised parameters of this (b, t).
__device__ __forceinline__ float spline_inverse_at(const float* th, long stride, float x1, int nb, float tb, float inv_sqrt_fc) {
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    float cw[NB_MAX + 1], chh[NB_MAX + 1], dv[NB_MAX + 1];
    // widths
    {
        float u[NB_MAX], mx = -3.0e38f, sum = 0.0f;
        for (int i = 0; i < nb; ++i) { u[i] = th[(long)i * stride] * inv_sqrt_fc; mx = fmaxf(mx, u[i]); }
        for (int i = 0; i < nb; ++i) { u[i] = expf(u[i] - mx); sum += u[i]; }
        float cum = 0.0f;
        cw[0] = -tb;
        for (int i = 0; i < nb; ++i) {
            cum += min_w + (1.0f - min_w * nb) * (u[i] / sum);
            cw[i + 1] = 2.0f * tb * cum - tb;
        }
        cw[nb] = tb;
    }
    // heights
    {
        float u[NB_MAX], mx = -3.0e38f, sum = 0.0f;
        for (int i = 0; i < nb; ++i) { u[i] = th[(long)(nb + i) * stride] * inv_sqrt_fc; mx = fmaxf(mx, u[i]); }
        for (int i = 0; i < nb; ++i) { u[i] = expf(u[i] - mx); sum += u[i]; }
        float cum = 0.0f;
        chh[0] = -tb;
        for (int i = 0; i < nb; ++i) {
            cum += min_h + (1.0f - min_h * nb) * (u[i] / sum);
            chh[i + 1] = 2.0f * tb * cum - tb;
        }
        chh[nb] = tb;
    }
    // derivatives: interior from theta, both ends = min_d + softplus(log(exp(1 - min_d) - 1))
    {
        const float cst = logf(expf(1.0f - min_d) - 1.0f);
        dv[0] = min_d + softplus_f(cst);
        dv[nb] = dv[0];
        for (int i = 1; i < nb; ++i) dv[i] = min_d + softplus_f(th[(long)(2 * nb + i - 1) * stride]);
    }
    int bin = -1;
    for (int i = 0; i <= nb; ++i) {
        const float loc = (i == nb) ? chh[i] + 1e-6f : chh[i];
        bin += (x1 >= loc) ? 1 : 0;
    }
    bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
    const float in_cw = cw[bin], in_w = cw[bin + 1] - cw[bin];
    const float in_ch = chh[bin], in_h = chh[bin + 1] - chh[bin];
    const float delta = in_h / in_w;
    const float d0 = dv[bin], d1 = dv[bin + 1];
    const float t1 = d0 + d1 - 2.0f * delta;
    const float u = x1 - in_ch;
    const float t3 = u * t1;
    const float qa = in_h * (delta - d0) + t3;
    const float qb = in_h * d0 - t3;
    const float qc = -delta * u;
    const float disc = qb * qb - 4.0f * qa * qc;
    const float root = (2.0f * qc) / (-qb - sqrtf(disc));
    return root * in_w + in_cw;
}

// One thread per (b,t)
__global__ __launch_bounds__(64) void k_spline_inverse(float* z, int ch_x0, const float* theta, const int* len, int T,
                                                       int nb, float tb, float inv_sqrt_fc) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= T) return;
    const int ch_x1 = 1 - ch_x0;
    float* z0 = z + ((long)b * 2 + ch_x0) * T + t;
    float* z1 = z + ((long)b * 2 + ch_x1) * T + t;
    const bool valid = t < len[b];
    const float x1 = *z1;
    float outv = x1;
    if (x1 >= -tb && x1 <= tb) outv = spline_inverse_at(theta + (long)b * (3 * nb - 1) * T + t, T, x1, nb, tb, inv_sqrt_fc);
    *z1 = valid ? outv : 0.0f;
    if (!valid) *z0 = 0.0f;
}
void launch_spline_inverse(float* z, int ch_x0, const float* theta, const int* len, int B, int T, int nbins,
                           float tail, float inv_sqrt_fc, hipStream_t s) {
    if (nbins > NB_MAX) throw std::runtime_error("spline: too many bins");
    LAUNCH_KERNEL(k_spline_inverse, dim3((T + 63) / 64, B), dim3(64), 0, s, z, ch_x0, theta, len, T, nbins, tail,
                  inv_sqrt_fc);
}

// ------------------------------------------------------------------------------------------------
// The whole DDS stack of the duration predictor (A.6 / A.7) in ONE launch, with what surrounds it:
//   pre:    x = conv1x1(src) + bias (+ cond[b])                 (StochasticDurationPredictor.pre + cond)
//        or x = w[c] * z[b, zch, t] + bias[c] + src[b, c, t]    (ConvFlow.pre on one channel + conditioning g)
//   layers: x += gelu(LN2(conv1x1(gelu(LN1(dwconv_{K, K^i}(x * mask))))))        i = 0 .. n_layers - 1
//   proj:   out = (conv1x1(x * mask) + bias) * mask             (dp.proj -> h, ConvFlow.proj -> theta)
//   spline: z[b, 1 - zch] <- RQS^-1(z[b, 1 - zch]; theta), z *= mask            (theta never leaves the CU)
// The text side of the graph is bound by launch latency, not by bytes ([32, 192, 128] is 3 MB): 5 + 3 x 6 launches become
// 1 + 3.  A workgroup owns 32 output columns of all C channels and keeps a 64-column window of x in LDS (16 columns of halo
// each side; the depthwise taps reach sum_i K^i (K - 1) / 2 = 13 columns at K = 3, three layers): every layer but the last
// is computed on the whole window — a column of the window is right as long as its receptive field lies inside it, and the
// 32 owned columns' fields do — the last one on the owned columns only.  Per element the arithmetic (tap order, the channel
// groups' order in the LayerNorm sums, k order of the matrix-core 1x1 convs) is that of k_dds_layer / k_convflow_pre /
// the pointwise conv kernels / k_spline_inverse, whatever the column's position in a window: results do not depend on
// the tiling, the batch or the neighbours.
// ------------------------------------------------------------------------------------------------
template <int C, int W, int NCT>
__device__ __forceinline__ void dds_mm(const float* __restrict__ wq /* packed A fragments of the row tile + lane */,
                                       const float* __restrict__ yb /* LDS B operand: row brow, first column + bcol */,
                                       f32x16 (&acc)[NCT], bool skip = false) {
    MI355_UNROLL
    for (int j = 0; j < NCT; ++j)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    if (skip) return;  // lab build: timing without the matrix work
    // weight fragments through a 16-register ring, 8 k-steps (>= 512 cycles of matrix work) ahead of their use
    float ra[16];
    MI355_UNROLL
    for (int u = 0; u < 8; ++u) ra[u] = wq[u * 64];
    for (int cp0 = 0; cp0 < C / 2; cp0 += 16) {
        MI355_UNROLL
        for (int u = 0; u < 16; ++u) {
            const int nxt = cp0 + u + 8 < C / 2 ? cp0 + u + 8 : C / 2 - 1;  // the tail re-reads the last record
            ra[(u + 8) & 15] = wq[nxt * 64];
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j) acc[j] = MFMA_32x32x2_F32(ra[u], yb[(cp0 + u) * 2 * W + 32 * j], acc[j]);
        }
    }
}

// Every global load of the kernel is unconditional (clamped index, value selected afterwards) and sits in a batch ahead of
// its first use: a load under a test makes hipcc branch around it and wait for each one in turn.
template <int NW>  // C = 32 NW
__global__ __launch_bounds__(64 * NW) void k_dds_stack(DdsStackArgs a) {
    constexpr int C = 32 * NW, NCG = 2 * NW, NPT = 16, W = DDS_STACK_W, OFF = DDS_STACK_OFF, K = DDS_STACK_K;
    static_assert((C / 2) % 16 == 0, "dds_mm walks the channel pairs 16 at a time");
    DYN_SMEM(float, smem);
    float* X = smem;                // [C][W] the running x of the stack (after proj: theta [nth][32])
    float* Y = smem + C * W;        // [C][W] B operand of the 1x1 convs, then their raw result
    float* red = smem + 2 * C * W;  // [NCG][W]
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int col = tid & 31, cg = tid >> 5;
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 32 - OFF;  // time of window column 0
    const int L = a.len[b];
    const int tend = L < a.T ? L : a.T;
    auto GELU = [&](float x) { return (LAB_ABLATE(a) & 1) ? x : gelu_erf(x); };
    // per column (nh halves of the window per thread): sum over the channel groups in a fixed order
    auto col_sum = [&](float (&v)[2], int nh, int cb) {
        if (LAB_ABLATE(a) & 16) return;
        __syncthreads();
        MI355_UNROLL
        for (int h = 0; h < 2; ++h)
            if (h < nh) red[cg * W + cb + h * 32 + col] = v[h];
        __syncthreads();
        MI355_UNROLL
        for (int h = 0; h < 2; ++h)
            if (h < nh) {
                float s = 0.0f;
                MI355_UNROLL
                for (int g = 0; g < NCG; ++g) s += red[g * W + cb + h * 32 + col];
                v[h] = s;
            }
    };
    int rowc[16];  // the 16 rows of a 32 x 32 accumulator tile this lane holds (tile row 0 = channel 32 w)
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) rowc[r] = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * brow;
    // ---- pre
    if (a.pre_mode == DDS_PRE_AFFINE) {
        float pw[NPT], pb[NPT], zv[2], g[2][NPT];
        bool in[2];
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) {
            pw[i] = a.pre_w[cg + NCG * i];
            pb[i] = a.pre_b[cg + NCG * i];
        }
        MI355_UNROLL
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + h * 32 + col;
            in[h] = t >= 0 && t < a.T;
            const int tc = in[h] ? t : 0;
            zv[h] = a.z[((long)b * 2 + a.zch) * a.T + tc];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) g[h][i] = a.src[((long)b * C + cg + NCG * i) * a.T + tc];
        }
        SCHED_FENCE();
        MI355_UNROLL
        for (int h = 0; h < 2; ++h)
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                const float v = fmaf(pw[i], zv[h], pb[i]) + g[h][i];
                X[(cg + NCG * i) * W + h * 32 + col] = in[h] ? v : 0.0f;
            }
    } else {
        float xv[2][NPT];
        bool in[2];
        MI355_UNROLL
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + h * 32 + col;
            in[h] = t >= 0 && t < a.T;
            const int tc = in[h] ? t : 0;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) xv[h][i] = a.src[((long)b * C + cg + NCG * i) * a.T + tc];
        }
        float pb[16], cd[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            pb[r] = a.pre_b[rowc[r]];
            cd[r] = a.cond ? a.cond[(long)b * a.cond_bs + rowc[r]] : 0.0f;
        }
        SCHED_FENCE();
        MI355_UNROLL
        for (int h = 0; h < 2; ++h)
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) Y[(cg + NCG * i) * W + h * 32 + col] = in[h] ? xv[h][i] : 0.0f;
        __syncthreads();
        f32x16 acc[2];
        dds_mm<C, W, 2>(a.pre_w + (long)w * (C / 2) * 64 + lane, Y + brow * W + bcol, acc, LAB_ABLATE(a) & 2);
        MI355_UNROLL
        for (int r = 0; r < 16; ++r)
            MI355_UNROLL
            for (int j = 0; j < 2; ++j) {
                float v = acc[j][r] + pb[r];
                if (a.cond) v += cd[r];
                X[rowc[r] * W + 32 * j + bcol] = v;
            }
    }
    __syncthreads();
    if (LAB_ABLATE(a) & 32) return;
    // ---- DDS layers
    int dil = 1;
    for (int li = 0; li < ((LAB_ABLATE(a) & 64) ? 0 : a.n_layers); ++li) {
        const bool last = li == a.n_layers - 1;
        const int nh = last ? 1 : 2;   // halves of the window computed: all 64 columns, or the 32 owned ones
        const int cb = last ? OFF : 0;  // first computed column
        // the layer's per-channel parameters: one batch of loads
        float dwb[NPT], dww[NPT][K], g1[NPT], b1[NPT];
        {
            const float* dw_w = a.dw_w[li];
            const float* dw_b = a.dw_b[li];
            const float* pg1 = a.g1[li];
            const float* pb1 = a.b1[li];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                const int c = cg + NCG * i;
                dwb[i] = dw_b[c];
                MI355_UNROLL
                for (int k = 0; k < K; ++k) dww[i][k] = dw_w[c * K + k];
                g1[i] = pg1[c];
                b1[i] = pb1[c];
            }
        }
        // depthwise conv of x * mask + LN1 + GELU -> Y
        if (!(LAB_ABLATE(a) & 4)) {
            float v[2][NPT];
            const int pad = (K * dil - dil) / 2;
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                const int j = cb + h * 32 + col;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) v[h][i] = dwb[i];
                MI355_UNROLL
                for (int k = 0; k < K; ++k) {
                    const int jj = j - pad + k * dil, tt = t0 + jj;
                    const bool in = tt >= 0 && tt < tend && jj >= 0 && jj < W;
                    const int jc = in ? jj : 0;
                    MI355_UNROLL
                    for (int i = 0; i < NPT; ++i) {
                        const float xv = X[(cg + NCG * i) * W + jc];
                        v[h][i] = fmaf(dww[i][k], in ? xv : 0.0f, v[h][i]);  // an absent tap adds w * 0: the value is unchanged
                    }
                }
            }
            float st[2] = {0.0f, 0.0f};
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                float sum = 0.0f;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) sum += v[h][i];
                st[h] = sum;
            }
            col_sum(st, nh, cb);
            float mean[2] = {0.0f, 0.0f}, sq[2] = {0.0f, 0.0f};
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                mean[h] = st[h] / (float)C;
                float q = 0.0f;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) { const float d = v[h][i] - mean[h]; q += d * d; }
                sq[h] = q;
            }
            col_sum(sq, nh, cb);
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                const float rstd = 1.0f / sqrtf(sq[h] / (float)C + 1e-5f);
                const int j = cb + h * 32 + col;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) Y[(cg + NCG * i) * W + j] = GELU((v[h][i] - mean[h]) * rstd * g1[i] + b1[i]);
            }
        }
        // the second half's parameters: their latency hides behind the matrix work
        float bias[16], g2[NPT], b2[NPT];
        {
            const float* pbias = a.bias1x1[li];
            const float* pg2 = a.g2[li];
            const float* pb2 = a.b2[li];
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) bias[r] = pbias[rowc[r]];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                g2[i] = pg2[cg + NCG * i];
                b2[i] = pb2[cg + NCG * i];
            }
        }
        __syncthreads();
        // 1x1 conv on the f32 matrix cores: wave w -> rows 32 w .., the computed columns
        {
            f32x16 acc[2];
            const float* wq = a.w1x1[li] + (long)w * (C / 2) * 64 + lane;
            if (last) {
                f32x16 a1[1];
                dds_mm<C, W, 1>(wq, Y + brow * W + cb + bcol, a1, LAB_ABLATE(a) & 2);
                acc[0] = a1[0];
                acc[1] = a1[0];
            } else {
                dds_mm<C, W, 2>(wq, Y + brow * W + cb + bcol, acc, LAB_ABLATE(a) & 2);
            }
            __syncthreads();  // every wave is done reading Y: the raw conv result takes its place
            MI355_UNROLL
            for (int j = 0; j < 2; ++j) {
                if (j >= nh) continue;
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) Y[rowc[r] * W + cb + 32 * j + bcol] = acc[j][r] + bias[r];
            }
        }
        __syncthreads();
        // LN2 + GELU + residual -> X
        if (!(LAB_ABLATE(a) & 8)) {
            float z[2][NPT];
            float st[2] = {0.0f, 0.0f};
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                const int j = cb + h * 32 + col;
                float sum = 0.0f;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) {
                    z[h][i] = Y[(cg + NCG * i) * W + j];
                    sum += z[h][i];
                }
                st[h] = sum;
            }
            col_sum(st, nh, cb);
            float mean[2] = {0.0f, 0.0f}, sq[2] = {0.0f, 0.0f};
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                mean[h] = st[h] / (float)C;
                float q = 0.0f;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) { const float d = z[h][i] - mean[h]; q += d * d; }
                sq[h] = q;
            }
            col_sum(sq, nh, cb);
            MI355_UNROLL
            for (int h = 0; h < 2; ++h) {
                if (h >= nh) continue;
                const float rstd = 1.0f / sqrtf(sq[h] / (float)C + 1e-5f);
                const int j = cb + h * 32 + col;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) {
                    const int c = cg + NCG * i;
                    X[c * W + j] += GELU((z[h][i] - mean[h]) * rstd * g2[i] + b2[i]);
                }
            }
        }
        __syncthreads();
        dil *= K;
    }
    // ---- x (the stack's result) to global, where somebody wants it (tests; the engine keeps it on chip)
    const int t_own = t0 + OFF + col;
    if (a.x_out && t_own < a.T) {
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) {
            const int c = cg + NCG * i;
            a.x_out[((long)b * C + c) * a.T + t_own] = X[c * W + OFF + col];
        }
    }
    if (!a.proj_w) return;
    // ---- proj: B operand = x * mask on the owned columns
    {
        const bool in = t_own < tend;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) {
            const int c = cg + NCG * i;
            Y[c * W + OFF + col] = in ? X[c * W + OFF + col] : 0.0f;
        }
    }
    const int ntp = (a.proj_cout + 31) / 32;
    float pjb[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) pjb[r] = a.proj_b[rowc[r] < a.proj_cout ? rowc[r] : 0];
    float x1 = 0.0f;  // the spline's input of this thread's column (threads 0 .. 31)
    const int t_sp = t0 + OFF + (tid & 31);
    if (a.spline) x1 = a.z[((long)b * 2 + (1 - a.zch)) * a.T + (t_sp < a.T ? t_sp : 0)];
    __syncthreads();
    {
        f32x16 a1[1];
        if (w < ntp) dds_mm<C, W, 1>(a.proj_w + (long)w * (C / 2) * 64 + lane, Y + brow * W + OFF + bcol, a1, LAB_ABLATE(a) & 2);
        // theta takes the place of x (every thread has read its x above; the barrier below orders the spline's reads)
        const int t = t0 + OFF + bcol;
        if (w < ntp) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int co = rowc[r];
                if (co < a.proj_cout) {
                    float v = a1[0][r] + pjb[r];
                    if (t >= tend) v = 0.0f;
                    if (a.out && t < a.T) a.out[((long)b * a.proj_cout + co) * a.T + t] = v;
                    if (a.spline) X[co * 32 + bcol] = v;
                }
            }
        }
    }
    if (!a.spline) return;
    __syncthreads();
    if (tid < 32 && t_sp < a.T) {
        float* z0 = a.z + ((long)b * 2 + a.zch) * a.T + t_sp;
        float* z1 = a.z + ((long)b * 2 + (1 - a.zch)) * a.T + t_sp;
        const bool valid = t_sp < L;
        float outv = x1;
        if (x1 >= -a.tail && x1 <= a.tail) outv = spline_inverse_at(X + tid, 32, x1, a.nb, a.tail, a.inv_sqrt_fc);
        *z1 = valid ? outv : 0.0f;
        if (!valid) *z0 = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// The same stack in MATH_BF16X3 / BF16W (the default math): twelve waves per workgroup and the 1x1 convs on the bf16 matrix
// cores.  k_dds_stack above spends a layer's 22 us as 6 + 5 us of VALU phases on six waves (two SIMDs carry two of them) and
// 9 us of f32 MFMA (a sixteenth of the bf16 rate); here
//   * a thread owns ONE window column and the 16 channels of one 16-channel group (= wave w: channels 16 w ..), i.e. exactly the
//     two B-operand records (group w, halves 0 / 1) of that column: LayerNorm + GELU results are split into their three bf16
//     terms in registers and stored as planes [plane][group][half][64 columns] with 16-byte stores, no transpose through LDS;
//   * the 1x1 conv is b3_chunk (K = 1, twelve groups, six products per step) on 32 x 32 tiles: wave w -> row tile w % 6 of
//     column tile w / 6 (the last layer and proj have six tiles: waves 0 .. 5); its f32 result overlays the dead planes;
//   * per-channel parameters are wave-uniform (one channel group per wave).
// Arithmetic per element does not depend on the tiling (fixed channel order inside a thread, fixed wave order across), so
// "batched == unbatched" holds as before; results differ from the f32-MFMA stack by f32 rounding only (operands are split
// exactly, f32 accumulate).
// ------------------------------------------------------------------------------------------------
template <bool W1>
__global__ __launch_bounds__(768) void k_dds_stack_b3(DdsStackArgs a) {
    constexpr int C = 192, NG = 12, NPT = 16, W = DDS_STACK_W, OFF = DDS_STACK_OFF, K = DDS_STACK_K, PS = NG * 2 * W;
    DYN_SMEM(float, smem);
    float* X = smem;                                             // [C][W] the running x of the stack (after proj: theta [nth][32])
    uint4* planes = reinterpret_cast<uint4*>(smem + C * W);      // 3 x [NG][2][W] records (72 KiB)
    float* R = smem + C * W;                                     // [C][W] raw 1x1 result, over the dead planes
    float* red = smem + C * W + 3 * PS * 4;                      // [NG][W]
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int col = lane;  // window column of this thread in the VALU phases; its channels: 16 w + i
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 32 - OFF;
    const int L = a.len[b];
    const int tend = L < a.T ? L : a.T;
    const int rt = w % 6, ct = w / 6;  // matrix phases: 32-row tile, 32-column tile
    auto GELU = [&](float x) { return (LAB_ABLATE(a) & 1) ? x : gelu_erf(x); };
    auto col_sum = [&](float v, bool on) {  // per column: sum over the twelve channel groups (waves), fixed order
        __syncthreads();
        if (on) red[w * W + col] = v;
        __syncthreads();
        float s = 0.0f;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) s += red[g * W + col];
        return s;
    };
    // a column's 16 values of channel group w -> the two records (halves 0 / 1) of the three planes
    auto store_planes = [&](const float (&v)[NPT], int j) {
        MI355_UNROLL
        for (int h = 0; h < 2; ++h) {
            // record slot e <-> channel 16 g + 8 (e >> 2) + 4 h + (e & 3)   (layout 1, b3.h / pack_conv_weights_bf16x3_mode)
            uint4 hi, mi, lo;
            split3_pk(v[4 * h + 0], v[4 * h + 1], hi.x, mi.x, lo.x);
            split3_pk(v[4 * h + 2], v[4 * h + 3], hi.y, mi.y, lo.y);
            split3_pk(v[8 + 4 * h + 0], v[8 + 4 * h + 1], hi.z, mi.z, lo.z);
            split3_pk(v[8 + 4 * h + 2], v[8 + 4 * h + 3], hi.w, mi.w, lo.w);
            const int o = (w * 2 + h) * W + j;
            planes[o] = hi;
            planes[PS + o] = mi;
            planes[2 * PS + o] = lo;
        }
    };
    // one 32 x 32 tile of W x planes: rows 32 q .., window columns c0 .. c0 + 31
    auto mm = [&](const float* wb3, int q, int c0, f32x16& acc) {
        f32x16 t[1][1];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) t[0][0][r] = 0.0f;
        if (!(LAB_ABLATE(a) & 2)) {
            const uint4* wp[1] = {reinterpret_cast<const uint4*>(wb3) + (long)q * NG * 192 + lane};
            b3_chunk<1, 1, NG, 1, W1>(t, wp, planes + brow * W + bcol + c0, PS, W, 1, NG, 1);
        }
        acc = t[0][0];
    };
    int rowc[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) rowc[r] = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * brow;
    // ---- pre
    {
        const int t = t0 + col;
        const bool in = t >= 0 && t < a.T;
        const int tc = in ? t : 0;
        float sv[NPT];
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) sv[i] = a.src[((long)b * C + 16 * w + i) * a.T + tc];
        if (a.pre_mode == DDS_PRE_AFFINE) {
            const float zv = a.z[((long)b * 2 + a.zch) * a.T + tc];
            float pw[NPT], pb[NPT];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) { pw[i] = a.pre_w[16 * w + i]; pb[i] = a.pre_b[16 * w + i]; }
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) X[(16 * w + i) * W + col] = in ? fmaf(pw[i], zv, pb[i]) + sv[i] : 0.0f;
        } else {
            float pb[16], cd[16];
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                pb[r] = a.pre_b[rowc[r]];
                cd[r] = a.cond ? a.cond[(long)b * a.cond_bs + rowc[r]] : 0.0f;
            }
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) sv[i] = in ? sv[i] : 0.0f;
            store_planes(sv, col);
            __syncthreads();
            f32x16 acc;
            mm(a.pre_w, rt, 32 * ct, acc);
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                float v = acc[r] + pb[r];
                if (a.cond) v += cd[r];
                X[rowc[r] * W + 32 * ct + bcol] = v;
            }
        }
    }
    __syncthreads();
    if (LAB_ABLATE(a) & 32) return;
    // ---- DDS layers
    int dil = 1;
    for (int li = 0; li < ((LAB_ABLATE(a) & 64) ? 0 : a.n_layers); ++li) {
        const bool last = li == a.n_layers - 1;
        // the last layer is computed on the 32 owned columns only: window columns OFF .. OFF + 31
        const bool on = !last || (col >= OFF && col < OFF + 32);
        float dwb[NPT], dww[NPT][K], g1[NPT], b1[NPT];
        {
            const float* dw_w = a.dw_w[li];
            const float* dw_b = a.dw_b[li];
            const float* pg1 = a.g1[li];
            const float* pb1 = a.b1[li];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                const int c = 16 * w + i;
                dwb[i] = dw_b[c];
                MI355_UNROLL
                for (int k = 0; k < K; ++k) dww[i][k] = dw_w[c * K + k];
                g1[i] = pg1[c];
                b1[i] = pb1[c];
            }
        }
        // depthwise conv of x * mask + LN1 + GELU -> planes
        {
            float v[NPT];
            const int pad = (K * dil - dil) / 2;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) v[i] = dwb[i];
            MI355_UNROLL
            for (int k = 0; k < K; ++k) {
                const int jj = col - pad + k * dil, tt = t0 + jj;
                const bool in = tt >= 0 && tt < tend && jj >= 0 && jj < W;
                const int jc = in ? jj : 0;
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) {
                    const float xv = X[(16 * w + i) * W + jc];
                    v[i] = fmaf(dww[i][k], in ? xv : 0.0f, v[i]);
                }
            }
            float sum = 0.0f;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) sum += v[i];
            const float mean = col_sum(sum, true) / (float)C;
            float sq = 0.0f;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) { const float d = v[i] - mean; sq += d * d; }
            const float rstd = 1.0f / sqrtf(col_sum(sq, true) / (float)C + 1e-5f);
            if (!(LAB_ABLATE(a) & 4)) {
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) v[i] = GELU((v[i] - mean) * rstd * g1[i] + b1[i]);
            }
            store_planes(v, col);
        }
        float bias[16], g2[NPT], b2[NPT];
        {
            const float* pbias = a.bias1x1[li];
            const float* pg2 = a.g2[li];
            const float* pb2 = a.b2[li];
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) bias[r] = pbias[rowc[r]];
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                g2[i] = pg2[16 * w + i];
                b2[i] = pb2[16 * w + i];
            }
        }
        __syncthreads();
        // 1x1 conv: wave -> (row tile, column tile); the last layer's six tiles (columns OFF ..) on waves 0 .. 5
        {
            f32x16 acc;
            const bool mine = !last || ct == 0;
            const int c0 = last ? OFF : 32 * ct;
            if (mine) mm(a.w1x1[li], rt, c0, acc);
            __syncthreads();  // every wave is done reading the planes: the raw result takes their place
            if (mine) {
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) R[rowc[r] * W + c0 + bcol] = acc[r] + bias[r];
            }
        }
        __syncthreads();
        // LN2 + GELU + residual -> X
        {
            float z[NPT];
            float sum = 0.0f;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) {
                z[i] = R[(16 * w + i) * W + col];
                sum += z[i];
            }
            const float mean = col_sum(sum, true) / (float)C;
            float sq = 0.0f;
            MI355_UNROLL
            for (int i = 0; i < NPT; ++i) { const float d = z[i] - mean; sq += d * d; }
            const float rstd = 1.0f / sqrtf(col_sum(sq, true) / (float)C + 1e-5f);
            if (on && !(LAB_ABLATE(a) & 8)) {
                MI355_UNROLL
                for (int i = 0; i < NPT; ++i) X[(16 * w + i) * W + col] += GELU((z[i] - mean) * rstd * g2[i] + b2[i]);
            }
        }
        __syncthreads();
        dil *= K;
    }
    const bool own = col >= OFF && col < OFF + 32;
    const int t_own = t0 + col;
    if (a.x_out && own && t_own < a.T) {
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) a.x_out[((long)b * C + 16 * w + i) * a.T + t_own] = X[(16 * w + i) * W + col];
    }
    if (!a.proj_w) return;
    // ---- proj: B operand = x * mask (all window columns are written; only the owned ones are used)
    {
        float v[NPT];
        const bool in = t_own >= 0 && t_own < tend;
        MI355_UNROLL
        for (int i = 0; i < NPT; ++i) v[i] = in ? X[(16 * w + i) * W + col] : 0.0f;
        store_planes(v, col);
    }
    const int ntp = (a.proj_cout + 31) / 32;
    float pjb[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) pjb[r] = a.proj_b[rowc[r] < a.proj_cout ? rowc[r] : 0];
    float x1 = 0.0f;
    const int t_sp = t0 + OFF + (tid & 31);
    if (a.spline) x1 = a.z[((long)b * 2 + (1 - a.zch)) * a.T + (t_sp < a.T ? t_sp : 0)];
    __syncthreads();
    {
        const bool mine = ct == 0 && rt < ntp;
        f32x16 acc;
        if (mine) mm(a.proj_w, rt, OFF, acc);
        const int t = t0 + OFF + bcol;
        if (mine) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int co = rowc[r];
                if (co < a.proj_cout) {
                    float v = acc[r] + pjb[r];
                    if (t >= tend) v = 0.0f;
                    if (a.out && t < a.T) a.out[((long)b * a.proj_cout + co) * a.T + t] = v;
                    if (a.spline) X[co * 32 + bcol] = v;
                }
            }
        }
    }
    if (!a.spline) return;
    __syncthreads();
    if (tid < 32 && t_sp < a.T) {
        float* z0 = a.z + ((long)b * 2 + a.zch) * a.T + t_sp;
        float* z1 = a.z + ((long)b * 2 + (1 - a.zch)) * a.T + t_sp;
        const bool valid = t_sp < L;
        float outv = x1;
        if (x1 >= -a.tail && x1 <= a.tail) outv = spline_inverse_at(X + tid, 32, x1, a.nb, a.tail, a.inv_sqrt_fc);
        *z1 = valid ? outv : 0.0f;
        if (!valid) *z0 = 0.0f;
    }
}

bool dds_stack_supported(int C, int K, int n_layers, int proj_cout) {
    if (C != 192 || K != DDS_STACK_K || n_layers < 1 || n_layers > DDS_STACK_MAX_LAYERS || proj_cout > C) return false;
    int reach = 0, dil = 1;
    for (int i = 0; i < n_layers; ++i) { reach += dil * (K - 1) / 2; dil *= K; }
    return reach <= DDS_STACK_OFF;
}

void launch_dds_stack(const DdsStackArgs& a, int C, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!dds_stack_supported(C, a.K, a.n_layers, a.proj_w ? a.proj_cout : 0)) throw std::runtime_error("dds_stack: unsupported shape");
    if (a.spline && (a.nb > NB_MAX || 3 * a.nb - 1 != a.proj_cout)) throw std::runtime_error("dds_stack: spline parameter count");
    static const int ablate = lab_getenv("MI355VITS_DDS_ABLATE") ? atoi(lab_getenv("MI355VITS_DDS_ABLATE")) : 0;
    DdsStackArgs av = a;
    av.ablate = ablate;
    dim3 grid((a.T + 31) / 32, a.B);
    if (math_on_bf16(a.math)) {  // weights: layout-1 bf16 planes; twelve waves, 1x1 convs on the bf16 matrix cores
        const size_t shb = ((size_t)C * DDS_STACK_W + (size_t)3 * 12 * 2 * DDS_STACK_W * 4 + (size_t)12 * DDS_STACK_W) * sizeof(float);
        auto go = [&](auto kfn) {
#ifndef MI355_EMU
            set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
#endif
            LAUNCH_KERNEL(kfn, grid, dim3(768), shb, s, av);
        };
        if (a.math == MATH_BF16W) go(k_dds_stack_b3<true>);
        else go(k_dds_stack_b3<false>);
        return;
    }
    const size_t sh = ((size_t)2 * C * DDS_STACK_W + (size_t)(C / 16) * DDS_STACK_W) * sizeof(float);
    auto kfn = k_dds_stack<6>;
#ifndef MI355_EMU
    set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
#endif
    LAUNCH_KERNEL(kfn, grid, dim3(64 * 6), sh, s, av);
}

__global__ __launch_bounds__(64) void k_sdp_noise(float* z, const float* injected, int T, float noise_w,
                                                  unsigned long long seed, unsigned long long utt_base) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= T) return;
    for (int c = 0; c < 2; ++c) {
        const long o = ((long)b * 2 + c) * T + t;
        float n = 0.0f;
        if (noise_w != 0.0f) n = injected ? injected[o] : philox_normal(seed, utt_base + b, 0u, (unsigned)c, (unsigned)t);
        z[o] = n * noise_w;
    }
}
void launch_sdp_noise(float* z, const float* injected, int B, int T, float noise_w, unsigned long long seed,
                      unsigned long long utt_base, hipStream_t s) {
    LAUNCH_KERNEL(k_sdp_noise, dim3((T + 63) / 64, B), dim3(64), 0, s, z, injected, T, noise_w, seed, utt_base);
}

// EA^-1 (HF:703) on the logical channel 0, then K6: w = ceil(exp(logw) * mask * length_scale), inclusive scan.
__global__ __launch_bounds__(256) void k_durations(const float* z, int ch, float ea_m, float ea_logs, const int* len,
                                                   const int* forced, int T, float length_scale, float* logw,
                                                   int* w_ceil, int* cum, int* ylen) {
    const int b = blockIdx.x;
    const int L = len[b];
    const float es = expf(-ea_logs);
    for (int t = threadIdx.x; t < T; t += 256) {
        const float m = t < L ? 1.0f : 0.0f;
        const float lw = (z[((long)b * 2 + ch) * T + t] - ea_m) * es * m;
        logw[(long)b * T + t] = lw;
        float w = expf(lw) * m * length_scale;
        // inf / NaN / absurd durations (bad weights, huge length_scale) must not reach the float -> int cast (UB):
        // they become one over-the-cap count, which the host turns into ERR_INVALID before sizing anything
        int wc = (w < (float)DURATION_FRAME_CAP) ? (int)ceilf(w) : DURATION_FRAME_CAP + 1;
        if (forced) {
            wc = t < L ? forced[(long)b * T + t] : 0;
            if (wc < 0 || wc > DURATION_FRAME_CAP) wc = DURATION_FRAME_CAP + 1;
        }
        w_ceil[(long)b * T + t] = wc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int t = 0; t < T; ++t) {
            acc += w_ceil[(long)b * T + t];
            if (acc > 2 * DURATION_FRAME_CAP) acc = 2 * DURATION_FRAME_CAP;  // saturate: no int overflow over long rows
            cum[(long)b * T + t] = acc;
        }
        ylen[b] = acc < 1 ? 1 : acc;
    }
}
void launch_durations(const float* z, int ch, float ea_m, float ea_logs, const int* len, const int* forced, int B,
                      int T, float length_scale, float* logw, int* w_ceil, int* cum, int* ylen, hipStream_t s) {
    LAUNCH_KERNEL(k_durations, dim3(B), dim3(256), 0, s, z, ch, ea_m, ea_logs, len, forced, T, length_scale, logw,
                  w_ceil, cum, ylen);
}

// ------------------------------------------------------------------------------------------------
// K6 + K7: frame t takes phoneme j(t) = #{j : cum_j <= t} (binary search instead of the reference's dense
// [Ty x Tx] path matmul);  z_p = m_p[j] + N(0,1) * exp(logs_p[j]) * noise_scale, zero beyond the row's length.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand_prior(const float* stats, const int* cum, const int* ylen,
                                                      const float* injected, int injected_frames, int I, int Tx,
                                                      int Ty, float noise_scale, unsigned long long seed,
                                                      unsigned long long utt_base, float* z) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cg = threadIdx.x >> 6;
    if (t >= Ty) return;
    const int* cb = cum + (long)b * Tx;
    int lo = 0, hi = Tx;  // first j with cum[j] > t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cb[mid] <= t) lo = mid + 1; else hi = mid;
    }
    const int j = lo;
    const bool valid = t < ylen[b] && j < Tx;
    for (int c = cg; c < I; c += 4) {
        float v = 0.0f;
        if (valid) {
            const float m = stats[((long)b * 2 * I + c) * Tx + j];
            v = m;
            if (noise_scale != 0.0f) {
                const float ls = stats[((long)b * 2 * I + I + c) * Tx + j];
                const float n = injected ? injected[((long)b * I + c) * injected_frames + t]
                                         : philox_normal(seed, utt_base + b, 1u, (unsigned)c, (unsigned)t);
                v = m + n * expf(ls) * noise_scale;
            }
        }
        z[((long)b * I + c) * Ty + t] = v;
    }
}
void launch_expand_prior(const float* stats, const int* cum, const int* ylen, const float* injected,
                         int injected_frames, int B, int I, int Tx, int Ty, float noise_scale, unsigned long long seed,
                         unsigned long long utt_base, float* z, hipStream_t s) {
    LAUNCH_KERNEL(k_expand_prior, dim3((Ty + 63) / 64, B), dim3(256), 0, s, stats, cum, ylen, injected, injected_frames,
                  I, Tx, Ty, noise_scale, seed, utt_base, z);
}

// ------------------------------------------------------------------------------------------------
// A.11: per-utterance conditioning vectors  out[b,co] = W[co,:] . emb_g[sid[b],:] + bias[co]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_speaker_cond(const float* emb_g, const long long* sid, const float* w,
                                                      const float* bias, int gin, int Cout, float* out) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int co_raw = blockIdx.x * 4 + wid;
    const int co = co_raw < Cout ? co_raw : Cout - 1;
    const float* g = emb_g + (long)sid[b] * gin;
    float acc = 0.0f;
    for (int i = lane; i < gin; i += 64) acc = fmaf(w[(long)co * gin + i], g[i], acc);
    acc = wave_reduce_sum(acc);
    if (lane == 0 && co_raw < Cout) out[(long)b * Cout + co] = acc + (bias ? bias[co] : 0.0f);
}
void launch_speaker_cond(const float* emb_g, const long long* sid, const float* w, const float* bias, int B, int gin,
                         int Cout, float* out, hipStream_t s) {
    LAUNCH_KERNEL(k_speaker_cond, dim3((Cout + 3) / 4, B), dim3(256), 0, s, emb_g, sid, w, bias, gin, Cout, out);
}


// ------------------------------------------------------------------------------------------------
// Box probe (bench.py writes its three numbers into the JSON line next to the kernel table): what a lease's chip gives the access
// patterns the kernels depend on, measured in ~10 ms before the timed loop — a slow box can then be told from a slow kernel.
//   (1) L2-hit stream: every CU's workgroup reads the same 2.6 MB table (the size of a WaveNet layer's weight fragments, the
//       pattern of k_wn_layer_b3 / k_rb_conv / k_ups_pl) with buffer_load_dwordx4, eight loads in flight per lane;
//   (2) load latency: one dependent chain of vector loads, one wave per CU, over 2 MB (L2 hits), 32 MB (memory-side cache) and
//       1 GiB (HBM, and a TLB miss on most steps);
//   (3) HBM copy: 256 MiB read + 256 MiB written with 16-byte accesses.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_probe_l2_stream(const uint4* __restrict__ tbl, int n16, int reps, unsigned* sink) {
    const BufRsrc r = buf_rsrc(tbl);
    unsigned acc = 0u;
    constexpr int U = 8;
    const int per_round = 256 * U;
    const int rounds = n16 / per_round;  // (the table's tail past a whole round is not read)
    for (int rep = 0; rep < reps; ++rep) {
        for (int k = 0; k < rounds; ++k) {
            uint4 v[U];
            MI355_UNROLL
            for (int u = 0; u < U; ++u) v[u] = buf_load_u4(r, 16u * (unsigned)(k * per_round + u * 256 + (int)threadIdx.x), 0u);
            MI355_UNROLL
            for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x9e3779b9u) sink[blockIdx.x] = acc;  // (keeps the loads alive)
}
// dependent chain of vector loads over a zero-filled region of `nlines` (a power of two) 64-byte lines: the next line is a full-period
// LCG step of the current one PLUS the loaded word (zero — but only the memory knows), so every load waits for the one before it
__global__ __launch_bounds__(64) void k_probe_l2_latency(const unsigned* __restrict__ chain, int steps, unsigned nlines, unsigned* out) {
    const BufRsrc r = buf_rsrc(chain);
    unsigned idx = (unsigned)blockIdx.x * 2654435761u;
    for (int k = 0; k < steps; ++k) {
        idx = idx * 1664525u + 1013904223u;
        idx += __float_as_uint(buf_load_f32(r, 64u * (idx & (nlines - 1u)), 0u));
    }
    if (threadIdx.x == 0) out[blockIdx.x] = idx;
}
__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
// the same table stream with HBM traffic flowing through the L2 beside it, 8 : 1 in bytes (what the weight-streaming kernels look
// like to the cache: a layer's fragments read by every CU again and again while the activations pass through once): per round a
// workgroup reads 32 KB of the table, 4 KB of its own slice of `src` and writes 4 KB of `dst`
__global__ __launch_bounds__(256) void k_probe_l2_mixed(const uint4* __restrict__ tbl, int n16, int reps, const uint4* __restrict__ src,
                                                        uint4* __restrict__ dst, long slice16, unsigned* sink) {
    const BufRsrc r = buf_rsrc(tbl);
    unsigned acc = 0u;
    constexpr int U = 8;
    const int per_round = 256 * U;
    const int rounds = n16 / per_round;
    const uint4* sp = src + (long)blockIdx.x * slice16;
    uint4* dp = dst + (long)blockIdx.x * slice16;
    long pos = threadIdx.x;
    for (int rep = 0; rep < reps; ++rep) {
        for (int k = 0; k < rounds; ++k) {
            uint4 v[U];
            MI355_UNROLL
            for (int u = 0; u < U; ++u) v[u] = buf_load_u4(r, 16u * (unsigned)(k * per_round + u * 256 + (int)threadIdx.x), 0u);
            const uint4 sv = sp[pos];
            MI355_UNROLL
            for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            dp[pos] = sv;
            pos += 256;
            if (pos >= slice16) pos = threadIdx.x;
        }
    }
    if (acc == 0x9e3779b9u) sink[blockIdx.x] = acc;
}
void launch_probe_l2_mixed(const void* tbl, int n16, int reps, const void* src, void* dst, long slice16, unsigned* sink, int grid, hipStream_t s) {
    LAUNCH_KERNEL(k_probe_l2_mixed, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(tbl), n16, reps, static_cast<const uint4*>(src),
                  static_cast<uint4*>(dst), slice16, sink);
}
// the table stream with ONE load in flight per lane (a wave waits for each 1 KiB before asking for the next): what a kernel sees that
// streams its weight fragments a step or two ahead of the matrix cores — latency per fragment, not bandwidth
__global__ __launch_bounds__(256) void k_probe_l2_stream1(const uint4* __restrict__ tbl, int n16, int reps, unsigned* sink) {
    const BufRsrc r = buf_rsrc(tbl);
    unsigned acc = 0u;
    const int rounds = n16 / 256;
    for (int rep = 0; rep < reps; ++rep)
        for (int k = 0; k < rounds; ++k) {
            const uint4 v = buf_load_u4(r, 16u * (unsigned)(k * 256 + (int)threadIdx.x) + (acc & 0u), 0u);  // (address depends on the last load)
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 0x9e3779b9u) sink[blockIdx.x] = acc;
}
void launch_probe_l2_stream1(const void* tbl, int n16, int reps, unsigned* sink, int grid, hipStream_t s) {
    LAUNCH_KERNEL(k_probe_l2_stream1, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(tbl), n16, reps, sink);
}
void launch_probe_l2_stream(const void* tbl, int n16, int reps, unsigned* sink, int grid, hipStream_t s) {
    LAUNCH_KERNEL(k_probe_l2_stream, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(tbl), n16, reps, sink);
}
void launch_probe_l2_latency(const unsigned* chain, int steps, unsigned nlines, unsigned* out, int grid, hipStream_t s) {
    LAUNCH_KERNEL(k_probe_l2_latency, dim3(grid), dim3(64), 0, s, chain, steps, nlines, out);
}
void launch_probe_copy(const void* src, void* dst, long n16, int grid, hipStream_t s) {
    LAUNCH_KERNEL(k_probe_copy, dim3(grid), dim3(256), 0, s, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
}

}  // namespace m355
