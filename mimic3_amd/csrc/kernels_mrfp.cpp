// kernels_mrfp.cpp — one HiFi-GAN multi-receptive-field stage in MATH_BF16X3 with every element split ONCE.
//
//   y = (1/n) * sum_j RB_j(x),   RB_j: x1 = x + conv_{k_j,d1_j}(lrelu_0.1(x));  x2 = x1 + conv_{k_j,d2_j}(lrelu_0.1(x1))
//
// k_mrf_fused (kernels_mrf.cpp) keeps f32 tiles in LDS and splits a lane's operands into three bf16 terms again for every
// tap: 36 VALU per six MFMAs, as much issue time as the matrix work they feed (MFMA pipe 0.41 busy, 10.7 VALU per
// MFMA).  Here the VALU work is proportional to the ELEMENTS, not to elements x taps:
//   * both LDS tiles hold three bf16 PLANES (x = h + m + l exactly, 6 B per element): x (+halo) is split while it is staged,
//     x1 in conv1's epilogue; a record = the 8 consecutive channels 32 g + 8 q .. + 7 of one column (16 B), stored
//     [plane][k-group g][quarter q][column] — exactly one lane's B operand of v_mfma_f32_16x16x32_bf16 (k = 32 input channels =
//     one k-group per tap and MFMA), consecutive lanes on consecutive records (conflict-free ds_read_b128 when the row pitch
//     is a multiple of 16 columns);
//   * the WEIGHTS of the running conv live in registers: a wave owns ONE 16-row tile of output channels, K x (C / 32) x 3
//     fragments of 4 VGPRs (84 registers for 7 taps of 32 channels), loaded once per conv and wave from L2 — the matrix-core
//     loop carries no VALU and no global loads at all: three ds_read_b128 per six MFMAs, two steps ahead;
//   * 16-column tiles: eight waves = NWM row tiles x NWC column groups; conv2 owns NT2 output tiles per wave whose
//     accumulators persist over the resblocks (statically indexed), conv1 walks its share of the extended range
//     (T_B + 2 r2 columns) in a runtime loop: init (residual + bias, rebuilt exactly from the planes) -> K x C/32 steps of six
//     MFMAs on two independent accumulator chains -> epilogue (leaky-relu, mask, split, one 8-byte store per plane).  Two
//     waves per SIMD: one wave's init / epilogue VALU runs beside its partner's MFMAs;
//   * output-channel <-> (tile, row) map chosen so that a lane's four accumulator rows are four consecutive k-slots of a
//     record: row 4 q + i of tile mt = channel 32 (mt >> 1) + 8 q + 4 (mt & 1) + i (the weights are packed to match,
//     pack_conv_weights_p16).
// HBM traffic = read x (+halo) + write y, as before.
#include <mutex>
#include <set>
#include <type_traits>

#include "kernels.h"

namespace m355 {

namespace {
constexpr int MRFP_KMAX = 7;  // taps whose fragments fit a wave's registers next to the accumulators
constexpr size_t MRFP_LDS_LIMIT = 160 * 1024;
}  // namespace

// the running conv's fragments of row tile `mt`: [tap][k-group][plane] x (64 lanes x 16 B); taps >= K are not touched
template <int G, int K>
__device__ __forceinline__ void mrfp_load_w(uint4 (&W)[MRFP_KMAX][G][3], const uint4* __restrict__ wp) {
    MI355_UNROLL
    for (int k = 0; k < K; ++k)
        MI355_UNROLL
        for (int g = 0; g < G; ++g)
            MI355_UNROLL
            for (int p = 0; p < 3; ++p) W[k][g][p] = wp[((k * G + g) * 3 + p) * 64];
}

// bf16 slot `i` (0..3) of the two registers of a half record -> f32
__device__ __forceinline__ float bf16_slot(const uint2& v, int i) {
    const unsigned w = (i >> 1) ? v.y : v.x;
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
}

// One 16-column tile of one conv for this wave's 16 output rows: acc = init + sum over K taps x G k-groups.  xq = (plane 0,
// k-group 0, this lane's quarter, first tap's column) in the LDS planes; PS = plane stride, LD = row pitch (uint4 units).
// B fragments travel two steps ahead (ring of three); the six products of a step alternate between two accumulator
// chains (small terms | large terms) so that consecutive MFMAs never wait on each other's result.
template <int K, int G, typename Init>
__device__ __forceinline__ f32x4 mrfp_tile(const uint4 (&W)[MRFP_KMAX][G][3], const uint4* __restrict__ xq, int PS, int LD, int dil, Init init) {
    constexpr int S = K * G;
    uint4 bf[3][3];
    auto rd = [&](int s, int slot) MI355_INLINE_LAMBDA {
        const uint4* p = xq + (s % G) * 4 * LD + (s / G) * dil;
        MI355_UNROLL
        for (int pl = 0; pl < 3; ++pl) bf[slot][pl] = p[pl * PS];
    };
    rd(0, 0);
    if (S > 1) rd(1, 1);
    f32x4 accb = init();  // residual + bias (+ the running output): VALU beside the first fragments' LDS latency
    f32x4 accs;
    MI355_UNROLL
    for (int r = 0; r < 4; ++r) accs[r] = 0.0f;
    MI355_UNROLL
    for (int s = 0; s < S; ++s) {
        if (s + 2 < S) rd(s + 2, (s + 2) % 3);
        SCHED_FENCE();
        const int k = s / G, g = s % G, c = s % 3;
        accs = MFMA_16x16x32_BF16(W[k][g][2], bf[c][0], accs);  // small terms first
        accb = MFMA_16x16x32_BF16(W[k][g][1], bf[c][0], accb);
        accs = MFMA_16x16x32_BF16(W[k][g][0], bf[c][2], accs);
        accb = MFMA_16x16x32_BF16(W[k][g][0], bf[c][1], accb);
        accs = MFMA_16x16x32_BF16(W[k][g][1], bf[c][1], accs);
        accb = MFMA_16x16x32_BF16(W[k][g][0], bf[c][0], accb);
        SCHED_FENCE();
    }
    MI355_UNROLL
    for (int r = 0; r < 4; ++r) accb[r] += accs[r];
    return accb;
}

// C channels; NWM = C / 16 row tiles x NWC column groups = 8 waves; T_B = 16 NWC NT2 output columns per workgroup.
// K0, K1, K2: the resblocks' tap counts (0 = no such resblock) — compile-time, so that every weight fragment has its own
// registers and the whole stage is straight-line code between the barriers (dilations stay run-time arguments).
template <int C, int NWM, int NWC, int NT2, int K0, int K1, int K2>
__global__ __launch_bounds__(512) void k_mrf_p(MrfArgs a) {
    static_assert(NWM * NWC == 8 && NWM * 16 == C, "eight waves: C / 16 row tiles x column groups");
    constexpr int G = C / 32, T_B = 16 * NWC * NT2;
    DYN_SMEM(float, smem);
    const int LDX = a.ldx, LD1 = a.ld1, R = a.R;
    const int PSX = G * 4 * LDX, PS1 = G * 4 * LD1;
    uint4* Xp = reinterpret_cast<uint4*>(smem);   // [3][G][4][LDX]   lrelu(x), zero outside the row
    uint4* X1p = Xp + 3 * PSX;                    // [3][G][4][LD1]   lrelu(x1) of the current resblock, zero outside the row
    float* BS = reinterpret_cast<float*>(X1p + 3 * PS1);  // [nrb][2][C] biases
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int mt = wid % NWM, cg = wid / NWM;
    const int q = lane >> 4, n = lane & 15;
    const int gq = mt >> 1, hh = mt & 1;
    const int co0 = 32 * gq + 8 * q + 4 * hh;  // this lane's four output channels co0 .. co0 + 3 = half hh of record (gq, q)
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;

    uint4 W[MRFP_KMAX][G][3];
    auto wptr = [&](int j, int c, int K) MI355_INLINE_LAMBDA { return reinterpret_cast<const uint4*>(a.w[j][c]) + (long)mt * K * (G * 3 * 64) + lane; };
    mrfp_load_w<G, K0>(W, wptr(0, 0, K0));

    for (int i = tid; i < a.nrb * 2 * C; i += 512) BS[i] = a.bias[i / (2 * C)][(i / C) & 1][i % C];

    // ---- stage x[:, t0 - R : t0 - R + LDX) as planes: a thread takes one column of one (k-group, quarter) record — eight
    // 4-byte loads (256 contiguous bytes per wave and channel), leaky-relu, split, one conflict-free 16-byte store per plane.
    // Narrow tiles: the waves form column sets that share the records.
    {
        const int wpc = (LDX + 63) >> 6;          // waves per column set
        const int nparts = 8 / wpc > 0 ? 8 / wpc : 1;
        const int part = wid / wpc, col = tid - part * wpc * 64;
        const float* xb = a.x + (long)b * a.x_bs;
        const int last = len > 0 ? len - 1 : 0;
        if (part < nparts && col < LDX && !(LAB_ABLATE(a) & 2)) {
            const int tt = t0 - R + col;
            const bool in = tt >= 0 && tt < len;
            const int tc = tt < 0 ? 0 : (tt > last ? last : tt);  // every load unconditional (clamped), masked afterwards
            constexpr int RB = 4;  // records per batch: 32 loads in flight per thread
            for (int r0 = part * RB; r0 < 4 * G; r0 += nparts * RB) {
                float v[RB][8];
                MI355_UNROLL
                for (int u = 0; u < RB; ++u)
                    MI355_UNROLL
                    for (int e = 0; e < 8; ++e) v[u][e] = xb[(long)(8 * (r0 + u) + e) * a.x_ld + tc];
                SCHED_FENCE();
                MI355_UNROLL
                for (int u = 0; u < RB; ++u) {
                    MI355_UNROLL
                    for (int e = 0; e < 8; ++e) v[u][e] = in ? lrelu_f(v[u][e], 0.1f) : 0.0f;
                    uint4 h, m, l;
                    split3_pk(v[u][0], v[u][1], h.x, m.x, l.x);
                    split3_pk(v[u][2], v[u][3], h.y, m.y, l.y);
                    split3_pk(v[u][4], v[u][5], h.z, m.z, l.z);
                    split3_pk(v[u][6], v[u][7], h.w, m.w, l.w);
                    const int o = (r0 + u) * LDX + col;
                    Xp[o] = h;
                    Xp[PSX + o] = m;
                    Xp[2 * PSX + o] = l;
                }
            }
        }
    }
    __syncthreads();

    f32x4 out[NT2];
    MI355_UNROLL
    for (int i = 0; i < NT2; ++i)
        MI355_UNROLL
        for (int r = 0; r < 4; ++r) out[i][r] = 0.0f;

    // residual (x or x1 at the tile's own column, rebuilt exactly from its planes) + bias for this lane's four rows
    auto resid_bias = [&](const uint4* P, int PS, int LD, int col, const float* bs) MI355_INLINE_LAMBDA {
        const uint2* p2 = reinterpret_cast<const uint2*>(P + (gq * 4 + q) * LD + col) + hh;
        const uint2 vh = p2[0], vm = p2[2 * PS], vl = p2[4 * PS];
        const float4 bv = *reinterpret_cast<const float4*>(bs + co0);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
        f32x4 r;
        MI355_UNROLL
        for (int i = 0; i < 4; ++i) {
            const float y = (bf16_slot(vh, i) + bf16_slot(vm, i)) + bf16_slot(vl, i);  // exact: the three terms of one f32
            r[i] = (y >= 0.0f ? y : y * 10.0f) + bb[i];  // the tiles keep leaky-relu'd values; the residual is the raw one
        }
        return r;
    };

    auto resblock = [&](auto KC, auto KN, auto JC) MI355_INLINE_LAMBDA {
        constexpr int K = decltype(KC)::value, KNEXT = decltype(KN)::value, j = decltype(JC)::value;
        const int d1 = a.d1[j], d2 = a.d2[j];
        const int r1 = (K - 1) / 2 * d1, r2 = (K - 1) / 2 * d2;
        // ---- conv1 over the extended range: column e of tile q1 <-> t = t0 - r2 + e; this wave: a contiguous run of tiles
        const int n1 = (T_B + 2 * r2 + 15) >> 4;
        const int base = n1 / NWC, rem = n1 % NWC;
        const int cnt = base + (cg < rem ? 1 : 0), start = cg * base + (cg < rem ? cg : rem);
        const float* bs1 = BS + (j * 2 + 0) * C;
        for (int i = 0; i < cnt; ++i) {
            const int e = (start + i) * 16 + n;
            f32x4 acc;
            if (!(LAB_ABLATE(a) & 1)) {
                acc = mrfp_tile<K, G>(W, Xp + q * LDX + (R - r2 - r1) + e, PSX, LDX, d1,
                                      [&]() MI355_INLINE_LAMBDA { return resid_bias(Xp, PSX, LDX, (R - r2) + e, bs1); });
            } else {
                acc = resid_bias(Xp, PSX, LDX, (R - r2) + e, bs1);
            }
            if (i == cnt - 1) mrfp_load_w<G, K>(W, wptr(j, 1, K));  // conv2's fragments travel under the last epilogue and the barrier
            if (j > 0 && i == 0) __syncthreads();  // every wave is done reading the previous resblock's x1
            const int t = t0 - r2 + e;
            const bool live = t >= 0 && t < len;
            float v[4];
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) v[r] = live ? fmaxf(acc[r], 0.1f * acc[r]) : 0.0f;
            uint2 h, m, l;
            split3_pk(v[0], v[1], h.x, m.x, l.x);
            split3_pk(v[2], v[3], h.y, m.y, l.y);
            uint2* p2 = reinterpret_cast<uint2*>(X1p + (gq * 4 + q) * LD1 + e) + hh;
            p2[0] = h;
            p2[2 * PS1] = m;
            p2[4 * PS1] = l;
        }
        if (cnt == 0) {  // (never with the supported shapes: every column group owns at least one conv1 tile)
            mrfp_load_w<G, K>(W, wptr(j, 1, K));
            if (j > 0) __syncthreads();
        }
        __syncthreads();
        // ---- conv2 into the output registers: out += x1 + bias + conv(lrelu(x1)); this wave: tiles cg NT2 .. + NT2 - 1
        const float* bs2 = BS + (j * 2 + 1) * C;
        MI355_UNROLL
        for (int i = 0; i < NT2; ++i) {
            const int c = (cg * NT2 + i) * 16 + n;
            auto init = [&]() MI355_INLINE_LAMBDA {
                f32x4 r = resid_bias(X1p, PS1, LD1, c + r2, bs2);
                MI355_UNROLL
                for (int rr = 0; rr < 4; ++rr) r[rr] += out[i][rr];
                return r;
            };
            if (!(LAB_ABLATE(a) & 1)) out[i] = mrfp_tile<K, G>(W, X1p + q * LD1 + c, PS1, LD1, d2, init);
            else out[i] = init();
        }
        if constexpr (KNEXT > 0) mrfp_load_w<G, KNEXT>(W, wptr(j + 1, 0, KNEXT));  // the next resblock's first conv
    };

    resblock(std::integral_constant<int, K0>{}, std::integral_constant<int, K1>{}, std::integral_constant<int, 0>{});
    if constexpr (K1 > 0) resblock(std::integral_constant<int, K1>{}, std::integral_constant<int, K2>{}, std::integral_constant<int, 1>{});
    if constexpr (K2 > 0) resblock(std::integral_constant<int, K2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});

    const float n_rb = (float)a.nrb;
    auto store_all = [&](auto MEAN) MI355_INLINE_LAMBDA {  // the mean / scale choice once, outside the loops
        MI355_UNROLL
        for (int i = 0; i < NT2; ++i) {
            const int t = t0 + (cg * NT2 + i) * 16 + n;
            if (t < a.T && !(LAB_ABLATE(a) & 4)) {
                float* yp = a.y + (long)b * a.y_bs + (long)co0 * a.y_ld + t;
                MI355_UNROLL
                for (int r = 0; r < 4; ++r) yp[(long)r * a.y_ld] = decltype(MEAN)::value ? out[i][r] / n_rb : out[i][r] * a.out_scale;
            }
        }
    };
    if (a.out_scale > 0.0f) store_all(std::false_type{});
    else store_all(std::true_type{});
}

// ------------------------------------------------------------------------------------------------ host side
size_t p16_packed_words(int Cout, int Cin, int K) { return (size_t)(Cout / 16) * K * (Cin / 32) * 3 * 64 * 4; }

static inline uint32_t p16_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
static inline float p16_bf16_float(uint32_t b) {
    const uint32_t u = b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// [row tile mt][tap][k-group g][plane h, m, l][64 lanes][8 bf16]: lane = (quarter q = l >> 4, row m = l & 15) holds the
// input channels 32 g + 8 q + 0..7 of output channel 32 (mt >> 1) + 8 (m >> 2) + 4 (mt & 1) + (m & 3); w = h + m + l exactly,
// each term rounded to nearest on the host.
void pack_conv_weights_p16(const float* w, int Cout, int Cin, int K, uint32_t* out) {
    const int G = Cin / 32;
    for (int mt = 0; mt < Cout / 16; ++mt)
        for (int k = 0; k < K; ++k)
            for (int g = 0; g < G; ++g)
                for (int l = 0; l < 64; ++l) {
                    const int q = l >> 4, m = l & 15;
                    const int co = 32 * (mt >> 1) + 8 * (m >> 2) + 4 * (mt & 1) + (m & 3);
                    uint32_t plane[3][8];
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 32 * g + 8 * q + e;
                        const float v = w[((size_t)co * Cin + ci) * K + k];
                        const uint32_t h = p16_bf16_rne(v);
                        const float r1 = v - p16_bf16_float(h);
                        const uint32_t mm = p16_bf16_rne(r1);
                        const float r2 = r1 - p16_bf16_float(mm);
                        plane[0][e] = h; plane[1][e] = mm; plane[2][e] = p16_bf16_rne(r2);
                    }
                    for (int p = 0; p < 3; ++p) {
                        uint32_t* o = out + (((((size_t)mt * K + k) * G + g) * 3 + p) * 64 + l) * 4;
                        for (int jj = 0; jj < 4; ++jj) o[jj] = plane[p][2 * jj] | (plane[p][2 * jj + 1] << 16);
                    }
                }
}

namespace {
struct GeoP { int T_B, NWC; };
inline bool geometry_p(int C, GeoP* g) {
    if (C == 32) { *g = {320, 4}; return true; }
    return false;
}
// halo, row pitches (multiples of 16 columns: conflict-free 16-byte fragment reads) and LDS bytes of a stage
inline bool shape_p(int C, int nrb, const int* k, const int* d1, const int* d2, const GeoP& g, int* R, int* ldx, int* ld1, size_t* lds) {
    int Rm = 0, r2max = 0;
    for (int j = 0; j < nrb; ++j) {
        if (!(k[j] == 3 || k[j] == 5 || k[j] == 7) || d1[j] < 1 || d2[j] < 1) return false;
        const int r1 = (k[j] - 1) / 2 * d1[j], r2 = (k[j] - 1) / 2 * d2[j];
        Rm = Rm > r1 + r2 ? Rm : r1 + r2;
        r2max = r2max > r2 ? r2max : r2;
        if ((g.T_B + 2 * r2 + 15) / 16 < g.NWC) return false;
    }
    *R = Rm;
    *ldx = (g.T_B + 2 * Rm + 15 + 15) & ~15;  // + 15: conv1's last (rounded-up) tile reads inside its row
    *ld1 = ((g.T_B + 2 * r2max + 15) / 16) * 16;
    *lds = (size_t)(C / 32) * 4 * 3 * 16 * (size_t)(*ldx + *ld1) + (size_t)nrb * 2 * C * sizeof(float);
    return *lds <= MRFP_LDS_LIMIT && *ldx <= 512;  // (the staging loop: one column per thread)
}
}  // namespace

bool mrf_p_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    GeoP g;
    int R, ldx, ld1;
    size_t lds;
    // instantiated tap sequences: (3, 5, 7) — the "_low" voices' resblock_kernel_sizes
    if (!(nrb == 3 && k[0] == 3 && k[1] == 5 && k[2] == 7)) return false;
    return geometry_p(C, &g) && shape_p(C, nrb, k, d1, d2, g, &R, &ldx, &ld1, &lds);
}

// the > 64 KiB dynamic-LDS opt-in is a per-device function attribute: once per (kernel, device)
void set_max_dynamic_lds(const void* fn, int bytes) {
#ifndef MI355_EMU
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return;
    HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, dev});
#else
    (void)fn; (void)bytes;
#endif
}

void launch_mrf_p(MrfArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    GeoP g;
    size_t shmem = 0;
    if (!geometry_p(a.C, &g) || a.nrb < 1 || a.nrb > MRF_MAX_RB || !shape_p(a.C, a.nrb, a.k, a.d1, a.d2, g, &a.R, &a.ldx, &a.ld1, &shmem))
        throw std::runtime_error("mrf_p: unsupported stage shape");
    dim3 grid((a.T + g.T_B - 1) / g.T_B, a.B);
    auto go = [&](auto kfn) {
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)MRFP_LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(512), shmem, s, a);
    };
    const int k1 = a.nrb > 1 ? a.k[1] : 0, k2 = a.nrb > 2 ? a.k[2] : 0;
    if (a.k[0] == 3 && k1 == 5 && k2 == 7) go(k_mrf_p<32, 2, 4, 5, 3, 5, 7>);
    else throw std::runtime_error("mrf_p: unsupported tap counts");
}

}  // namespace m355
