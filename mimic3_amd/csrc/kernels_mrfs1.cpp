// kernels_mrfs1.cpp — the 32-channel HiFi-GAN MRF stage in MATH_BF16X3 as a SINGLE-PASS row sweep: all three
// resblocks in one left-to-right pass over a (row, segment) work item, x staged once, y written once — the row-sweep idea of
// kernels_mrfs.cpp without its price (three passes = 4 x the HBM bytes, which is what kept the 32-channel stage on k_mrf_p: at
// 32 channels a byte of x / y carries half the matrix work it carries at 64).
//
// What makes one pass possible at 32 channels: a conv's fragments are K x 12 registers per 16-row tile, so the eight waves can be
// specialised by (row tile, conv, RESBLOCK GROUP) and still hold everything they need for the whole segment:
//
//      wave = (mt = 2 row tiles) x (group 0 = resblocks 0 + 1: 3 + 5 taps = 96 VGPRs | group 1 = resblock 2: 7 taps = 84) x (conv1 | conv2)
//
//      staging (waves 6, 7)   x[s0 + u TS ..)       -> planes of lrelu(x) in ONE ring, read by all three conv1s        iteration u
//      conv1_j                x1_j[q0_j + p TS ..)  -> planes of lrelu(x1_j) in ring j, raw x1_j (f32) in raw ring j    iteration p + 3
//      conv2_0, conv2_1       out = (0 + rb_0) + rb_1 of block m -> an f32 ring                                         iteration m + 5
//      conv2_2                y[c0 + m TS ..) = (out + rb_2) * scale                                                     iteration m + 7
//
// in steps of TS = 32 columns (two 16-column tiles per wave, resblock and step), one workgroup barrier per step; conv1_j runs
// W1_j steps ahead of conv2_j (r2_j columns + slack), the partial sum passes from group 0 to group 1 through LDS two steps later.
// The arithmetic of an output element is k_mrf_p's, operation for operation (same MFMA sequences, out = ((0 + rb0) + rb1) + rb2
// with each resblock entering as `out + (x1 + b2)` ahead of its accumulator chain): bit-identical, so the launcher may choose by
// grid size.  Building blocks (rings addressed by lane masks, the tile, buffer addressing, the branch-free rules): mrfs.h,
// kernels_mrfs.cpp, hipx.h.
#include <type_traits>

#include "mrfs.h"

namespace m355 {

namespace {
constexpr size_t MRFS1_LDS_LIMIT = 160 * 1024;
constexpr int S1_C = 32, S1_TS = 32, S1_NT = 2, S1_TAP = 3 * 64;
constexpr int S1_LAG = 2;  // iterations between group 0's partial sum of a block and group 1 picking it up (at least)
}  // namespace

// geometry of a stage (host and device): per resblock the radii, conv1's lead, the origin of its x1 blocks relative to the
// segment's first column, the ring lengths (multiples of 16 columns) and the byte offsets of the rings in the LDS window.
// Raw rings first: a plane address minus one ring length must stay a non-negative offset (mrfs_rd).
struct MrfS1Geo {
    int r1[3], r2[3], W1[3], q0[3];  // q0: first column of conv1_j's block 0, relative to c0
    int s0;                          // first staged column, relative to c0
    int OFF1, A0, A2;                // iteration offsets: conv1_j's block p at p + OFF1, conv2 of group 0's block m at m + A0, group 1's at m + A2
    int XR, X1R[3], RR[3], OR;
    unsigned raw_off[3], out_off, bias_off, x_off, x1_off[3], total;
    bool ok;
};
__host__ __device__ inline MrfS1Geo mrfs1_geo(const int* k, const int* d1, const int* d2) {
    MrfS1Geo g;
    g.ok = true;
    // conv1_j must run W1_j iterations ahead of conv2_j: (W1 - 1) TS >= 2 r2 (conv2's reach to both sides).  Group 0's two
    // resblocks enter a block in the same iteration (one lead for both), group 1 picks the partial sum up at least LAG later.
    int wmin[3];
    for (int j = 0; j < 3; ++j) {
        g.r1[j] = (k[j] - 1) / 2 * d1[j];
        g.r2[j] = (k[j] - 1) / 2 * d2[j];
        wmin[j] = (2 * g.r2[j] + S1_TS - 1) / S1_TS + 1;
    }
    g.W1[0] = g.W1[1] = wmin[0] > wmin[1] ? wmin[0] : wmin[1];
    g.W1[2] = wmin[2] > g.W1[0] + S1_LAG ? wmin[2] : g.W1[0] + S1_LAG;
    int smax = -(1 << 30), omin = 1 << 30;
    for (int j = 0; j < 3; ++j) {
        g.q0[j] = g.r2[j] - (g.W1[j] - 1) * S1_TS;
        const int lead = g.q0[j] + g.r1[j];   // newest x column conv1_j's block 0 touches, minus TS
        const int old = g.q0[j] - g.r1[j];    // oldest
        smax = lead > smax ? lead : smax;
        omin = old < omin ? old : omin;
    }
    // staging starts at the oldest column any conv1 reads; the conv1s start once it has passed the newest column their first
    // blocks read: OFF1 staged blocks (three for the "_low" dilations: together the conv1s reach over 85 columns)
    g.s0 = omin;
    g.OFF1 = (smax + S1_TS - omin + S1_TS - 1) / S1_TS;
    g.A0 = g.OFF1 + g.W1[0];
    g.A2 = g.OFF1 + g.W1[2];
    // x ring: newest staged column of an iteration (s0 + (it + 1) TS) minus the oldest any conv1 still reads (omin + (it - OFF1) TS)
    g.XR = (((g.OFF1 + 1) * S1_TS) + 15) & ~15;
    for (int j = 0; j < 3; ++j) {
        g.X1R[j] = (2 * S1_TS + 2 * g.r2[j] + 15) & ~15;  // newest column written in an iteration minus the oldest conv2_j reads
        g.RR[j] = (2 * S1_TS + g.r2[j] + 15) & ~15;
        if (15 + 2 * g.r2[j] >= g.X1R[j] || 15 + 2 * g.r1[j] >= g.XR) g.ok = false;  // a tile's reach stays below a ring (mrfs_rd)
    }
    g.OR = (g.A2 - g.A0 + 1) * S1_TS;  // blocks between group 0 writing a partial sum and group 1 reading it, + the one being written
    unsigned off = 0;
    for (int j = 0; j < 3; ++j) { g.raw_off[j] = off; off += (unsigned)(S1_C / 4) * 16u * (unsigned)g.RR[j]; }
    g.out_off = off; off += (unsigned)(S1_C / 4) * 16u * (unsigned)g.OR;
    g.bias_off = off; off += (unsigned)(MRF_MAX_RB * 2 * S1_C * sizeof(float));
    g.x_off = off; off += 3u * 4u * 16u * (unsigned)g.XR;
    for (int j = 0; j < 3; ++j) { g.x1_off[j] = off; off += 3u * 4u * 16u * (unsigned)g.X1R[j]; }
    g.total = off;
    if (g.total > MRFS1_LDS_LIMIT) g.ok = false;
    return g;
}

template <bool LOW>  // LOW: the "_low" voices' dilations as compile-time constants; else from the arguments (CPU model: other sets)
__global__ __launch_bounds__(512) void k_mrf_s1(MrfArgs a) {
    constexpr int C = S1_C, TS = S1_TS, NT = S1_NT, TAP = S1_TAP, AH = 2;
    constexpr int K0 = 3, K1 = 5, K2 = 7;
    DYN_SMEM(float, smem);
    const int kk[3] = {K0, K1, K2};
    const int dd1[3] = {LOW ? 1 : a.d1[0], LOW ? 2 : a.d1[1], LOW ? 3 : a.d1[2]};
    const int dd2[3] = {LOW ? 2 : a.d2[0], LOW ? 6 : a.d2[1], LOW ? 12 : a.d2[2]};
    const MrfS1Geo g = mrfs1_geo(kk, dd1, dd2);
    char* L0 = reinterpret_cast<char*>(smem);
    float* BS = reinterpret_cast<float*>(L0 + g.bias_off);
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int mt = wid & 1, grp = (wid >> 1) & 1, role = wid >> 2;  // waves w and w + 4 (one SIMD): the two convs of one (row tile, group)
    const int q = lane >> 4, n = lane & 15;
    const int co0 = 8 * q + 4 * mt;  // this lane's four output channels = half mt of record (k-group 0, quarter q)
    const unsigned n16 = 16u * n;
    const unsigned XR16 = 16u * g.XR, PSX16 = 4u * XR16;

    for (int i = tid; i < a.nrb * 2 * C; i += 512) BS[i] = a.bias[i / (2 * C)][(i / C) & 1][i % C];
    __syncthreads();

    const int nseg = (a.T + a.seg - 1) / a.seg;
    const int nitems = nseg * a.B;
    const float out_mul = a.out_scale > 0.0f ? a.out_scale : 1.0f / (float)a.nrb;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = WAVE_UNIFORM(item / nseg), c0 = WAVE_UNIFORM((item - b * nseg) * a.seg);
        int len = a.len ? a.len[b] : a.T;
        if (len > a.T) len = a.T;
        len = WAVE_UNIFORM(len);
        const int last = len > 0 ? len - 1 : 0;
        const int c1 = c0 + a.seg < a.T ? c0 + a.seg : a.T;
        const int N = WAVE_UNIFORM((c1 - c0 + TS - 1) / TS);
        const int NIT = g.A2 + N;
        const BufRsrc xbuf = buf_rsrc(a.x + (long)b * a.x_bs), ybuf = buf_rsrc(a.y + (long)b * a.y_bs);
        const unsigned xrow = 4u * (unsigned)a.x_ld, yrow = 4u * (unsigned)a.y_ld;

        // this wave's fragments: conv `role` of resblock j, row tile mt, every tap
        auto load_w = [&](auto KC, int j, uint4 (&W)[1][decltype(KC)::value][3]) MI355_INLINE_LAMBDA {
            constexpr int K = decltype(KC)::value;
            const uint4* wp = reinterpret_cast<const uint4*>(a.w[j][role]) + (long)mt * K * TAP + lane;
            MI355_UNROLL
            for (int k = 0; k < K; ++k)
                MI355_UNROLL
                for (int p = 0; p < 3; ++p) W[0][k][p] = wp[k * TAP + p * 64];
        };
        auto bias4 = [&](int j, float (&bv)[4]) MI355_INLINE_LAMBDA {
            const float4 v = *reinterpret_cast<const float4*>(BS + (j * 2 + role) * C + co0);
            bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
        };

        // ------------------------------------------------------------------------------------------ conv1 of one resblock, one tile
        // x residual travels one iteration ahead (xq); results: planes of lrelu(x1) + raw x1 into resblock j's rings
        auto conv1_tile = [&](auto KC, const uint4 (&W)[1][decltype(KC)::value][3], const float (&bia)[4], int j, int d1, int e0,
                              unsigned xslot, unsigned wslot_p, unsigned wslot_r, const float (&res)[4], uint4 (&bfirst)[3], unsigned xslot_next,
                              unsigned ringq) MI355_INLINE_LAMBDA {
            constexpr int K = decltype(KC)::value;
            f32x4 acc;
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) acc[r] = res[r] + bia[r];
            if (!(LAB_ABLATE(a) & 1)) mrfs_tile<1, K, AH>(acc, W, L0, ringq, PSX16, (unsigned)g.XR, xslot, d1, lane, bfirst, xslot_next);
            const int t = e0 + n;
            const bool live = t >= 0 && t < len;
            float v[4], rr[4], s4[4];
            unsigned u[4], ur[4];
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) v[r] = live ? fmaxf(acc[r], 0.1f * acc[r]) : 0.0f;
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) u[r] = __float_as_uint(v[r]);
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) rr[r] = v[r] - __uint_as_float(u[r] & 0xffff0000u);
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) ur[r] = __float_as_uint(rr[r]);
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) s4[r] = rr[r] - __uint_as_float(ur[r] & 0xffff0000u);
            uint2 ph, pm, pl;
            ph.x = pack_hi16(u[0], u[1]); ph.y = pack_hi16(u[2], u[3]);
            pm.x = pack_hi16(ur[0], ur[1]); pm.y = pack_hi16(ur[2], ur[3]);
            pl.x = pack_hi16(__float_as_uint(s4[0]), __float_as_uint(s4[1]));
            pl.y = pack_hi16(__float_as_uint(s4[2]), __float_as_uint(s4[3]));
            const unsigned X1R16 = 16u * (unsigned)g.X1R[j], PS116 = 4u * X1R16;
            const unsigned ws = mrfs_wrap(wslot_p + (unsigned)n, (unsigned)g.X1R[j]);
            char* p1 = L0 + g.x1_off[j] + ((unsigned)q * X1R16 + 16u * ws + 8u * (unsigned)mt);
            *reinterpret_cast<uint2*>(p1) = ph;
            *reinterpret_cast<uint2*>(p1 + PS116) = pm;
            *reinterpret_cast<uint2*>(p1 + 2u * PS116) = pl;
            const unsigned wr = mrfs_wrap(wslot_r + (unsigned)n, (unsigned)g.RR[j]);
            *reinterpret_cast<float4*>(L0 + g.raw_off[j] + ((unsigned)(co0 >> 2) * 16u * (unsigned)g.RR[j] + 16u * wr)) =
                make_float4(acc[0], acc[1], acc[2], acc[3]);
        };
        auto load_x = [&](int e0, float (&v)[4]) MI355_INLINE_LAMBDA {
            const int t = e0 + n;
            const int tc = t < 0 ? 0 : (t > last ? last : t);
            const unsigned o = 4u * (unsigned)(co0 * a.x_ld + tc);
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) v[r] = buf_load_f32(xbuf, o, (unsigned)r * xrow);
            SCHED_FENCE();
        };
        // ------------------------------------------------------------------------------------------ conv2 of one resblock, one tile
        // acc carries the running sum of the resblocks (k_mrf_p: out + (x1 + b2), then the accumulator chain on the x1 planes)
        auto conv2_tile = [&](auto KC, const uint4 (&W)[1][decltype(KC)::value][3], const float (&bia)[4], int j, int d2, f32x4& acc, unsigned rslot,
                              unsigned pslot, uint4 (&bfirst)[3], unsigned pslot_next) MI355_INLINE_LAMBDA {
            constexpr int K = decltype(KC)::value;
            const unsigned rs = mrfs_wrap(rslot + (unsigned)n, (unsigned)g.RR[j]);
            const float4 x1v = *reinterpret_cast<const float4*>(L0 + g.raw_off[j] + ((unsigned)(co0 >> 2) * 16u * (unsigned)g.RR[j] + 16u * rs));
            acc[0] = acc[0] + (x1v.x + bia[0]);
            acc[1] = acc[1] + (x1v.y + bia[1]);
            acc[2] = acc[2] + (x1v.z + bia[2]);
            acc[3] = acc[3] + (x1v.w + bia[3]);
            unsigned ringq = g.x1_off[j] + (unsigned)q * 16u * (unsigned)g.X1R[j] + n16;
            OPAQUE_V(ringq);
            if (!(LAB_ABLATE(a) & 1))
                mrfs_tile<1, K, AH>(acc, W, L0, ringq, 4u * 16u * (unsigned)g.X1R[j], (unsigned)g.X1R[j], pslot, d2, lane, bfirst, pslot_next);
        };
        auto prime = [&](uint4 (&bfirst)[3], unsigned base_off, unsigned ring, unsigned slot, int d) MI355_INLINE_LAMBDA {
            const unsigned f0 = base_off + (unsigned)q * 16u * ring + n16 + 16u * slot;
            mrfs_rd<1, 1>(bfirst, 0, L0, f0, f0 - 16u * ring, WAVE_UNIFORM((int)ring - (int)slot), lane, 4u * 16u * ring, 16u * ring, d);
        };

        if (role == 0 && grp == 0) {
            // ============================================================ conv1 of resblocks 0 and 1
            uint4 Wa[1][K0][3], Wb[1][K1][3];
            load_w(std::integral_constant<int, K0>{}, 0, Wa);
            load_w(std::integral_constant<int, K1>{}, 1, Wb);
            float ba[4], bb[4];
            bias4(0, ba);
            bias4(1, bb);
            const int N1 = N + g.W1[0] - 1;  // blocks from q0 up to the last conv2 block's right halo
            unsigned xra = (unsigned)(g.q0[0] - g.r1[0] - g.s0), xrb = (unsigned)(g.q0[1] - g.r1[1] - g.s0);  // x ring slots of block 0, tap 0
            unsigned wpa = 0, wpb = 0, wra = 0, wrb = 0;
            float xqa[NT][4], xqb[NT][4];
            MI355_UNROLL
            for (int i = 0; i < NT; ++i) {
                load_x(c0 + g.q0[0] + 16 * i, xqa[i]);
                load_x(c0 + g.q0[1] + 16 * i, xqb[i]);
            }
            unsigned ringq = g.x_off + (unsigned)q * XR16 + n16;
            OPAQUE_V(ringq);
            auto iter = [&](int it, auto ACT) MI355_INLINE_LAMBDA {
                if constexpr (decltype(ACT)::value) {
                    const int p = it - g.OFF1;
                    uint4 bfirst[3];
                    prime(bfirst, g.x_off, (unsigned)g.XR, mrfs_wrap(xra, (unsigned)g.XR), dd1[0]);
                    MI355_UNROLL
                    for (int i = 0; i < NT; ++i) {
                        const int e0 = c0 + g.q0[0] + p * TS + 16 * i;
                        float res[4];
                        MI355_UNROLL
                        for (int r = 0; r < 4; ++r) res[r] = xqa[i][r];
                        load_x(e0 + TS, xqa[i]);
                        // the next tile in this wave's order: resblock 0's second tile, then resblock 1's first (its own base)
                        const unsigned xs = mrfs_wrap(xra + 16u * i, (unsigned)g.XR);
                        const unsigned xn = i + 1 < NT ? mrfs_wrap(xra + 16u * (i + 1), (unsigned)g.XR) : xs;
                        conv1_tile(std::integral_constant<int, K0>{}, Wa, ba, 0, dd1[0], e0, xs, mrfs_wrap(wpa + 16u * i, (unsigned)g.X1R[0]),
                                   mrfs_wrap(wra + 16u * i, (unsigned)g.RR[0]), res, bfirst, xn, ringq);
                    }
                    prime(bfirst, g.x_off, (unsigned)g.XR, mrfs_wrap(xrb, (unsigned)g.XR), dd1[1]);
                    MI355_UNROLL
                    for (int i = 0; i < NT; ++i) {
                        const int e0 = c0 + g.q0[1] + p * TS + 16 * i;
                        float res[4];
                        MI355_UNROLL
                        for (int r = 0; r < 4; ++r) res[r] = xqb[i][r];
                        load_x(e0 + TS, xqb[i]);
                        const unsigned xs = mrfs_wrap(xrb + 16u * i, (unsigned)g.XR);
                        const unsigned xn = i + 1 < NT ? mrfs_wrap(xrb + 16u * (i + 1), (unsigned)g.XR) : xs;
                        conv1_tile(std::integral_constant<int, K1>{}, Wb, bb, 1, dd1[1], e0, xs, mrfs_wrap(wpb + 16u * i, (unsigned)g.X1R[1]),
                                   mrfs_wrap(wrb + 16u * i, (unsigned)g.RR[1]), res, bfirst, xn, ringq);
                    }
                    xra = mrfs_wrap(xra + TS, (unsigned)g.XR);
                    xrb = mrfs_wrap(xrb + TS, (unsigned)g.XR);
                    wpa = mrfs_wrap(wpa + TS, (unsigned)g.X1R[0]);
                    wpb = mrfs_wrap(wpb + TS, (unsigned)g.X1R[1]);
                    wra = mrfs_wrap(wra + TS, (unsigned)g.RR[0]);
                    wrb = mrfs_wrap(wrb + TS, (unsigned)g.RR[1]);
                }
                __syncthreads();
            };
            MI355_NOUNROLL
            for (int it = 0; it < g.OFF1; ++it) iter(it, std::false_type{});
            MI355_NOUNROLL
            for (int it = g.OFF1; it < g.OFF1 + N1; ++it) iter(it, std::true_type{});
            MI355_NOUNROLL
            for (int it = g.OFF1 + N1; it < NIT; ++it) iter(it, std::false_type{});
        } else if (role == 0) {
            // ============================================================ conv1 of resblock 2
            uint4 Wc[1][K2][3];
            load_w(std::integral_constant<int, K2>{}, 2, Wc);
            float bc[4];
            bias4(2, bc);
            const int N1 = N + g.W1[2] - 1;
            unsigned xrc = (unsigned)(g.q0[2] - g.r1[2] - g.s0);
            unsigned wpc = 0, wrc = 0;
            float xqc[NT][4];
            MI355_UNROLL
            for (int i = 0; i < NT; ++i) load_x(c0 + g.q0[2] + 16 * i, xqc[i]);
            unsigned ringq = g.x_off + (unsigned)q * XR16 + n16;
            OPAQUE_V(ringq);
            auto iter = [&](int it, auto ACT) MI355_INLINE_LAMBDA {
                if constexpr (decltype(ACT)::value) {
                    const int p = it - g.OFF1;
                    uint4 bfirst[3];
                    prime(bfirst, g.x_off, (unsigned)g.XR, mrfs_wrap(xrc, (unsigned)g.XR), dd1[2]);
                    MI355_UNROLL
                    for (int i = 0; i < NT; ++i) {
                        const int e0 = c0 + g.q0[2] + p * TS + 16 * i;
                        float res[4];
                        MI355_UNROLL
                        for (int r = 0; r < 4; ++r) res[r] = xqc[i][r];
                        load_x(e0 + TS, xqc[i]);
                        const unsigned xs = mrfs_wrap(xrc + 16u * i, (unsigned)g.XR);
                        const unsigned xn = i + 1 < NT ? mrfs_wrap(xrc + 16u * (i + 1), (unsigned)g.XR) : xs;
                        conv1_tile(std::integral_constant<int, K2>{}, Wc, bc, 2, dd1[2], e0, xs, mrfs_wrap(wpc + 16u * i, (unsigned)g.X1R[2]),
                                   mrfs_wrap(wrc + 16u * i, (unsigned)g.RR[2]), res, bfirst, xn, ringq);
                    }
                    xrc = mrfs_wrap(xrc + TS, (unsigned)g.XR);
                    wpc = mrfs_wrap(wpc + TS, (unsigned)g.X1R[2]);
                    wrc = mrfs_wrap(wrc + TS, (unsigned)g.RR[2]);
                }
                __syncthreads();
            };
            MI355_NOUNROLL
            for (int it = 0; it < g.OFF1; ++it) iter(it, std::false_type{});
            MI355_NOUNROLL
            for (int it = g.OFF1; it < g.OFF1 + N1; ++it) iter(it, std::true_type{});
            MI355_NOUNROLL
            for (int it = g.OFF1 + N1; it < NIT; ++it) iter(it, std::false_type{});
        } else if (grp == 0) {
            // ============================================================ conv2 of resblocks 0 and 1: out = (0 + rb0) + rb1 -> LDS
            uint4 Wa[1][K0][3], Wb[1][K1][3];
            load_w(std::integral_constant<int, K0>{}, 0, Wa);
            load_w(std::integral_constant<int, K1>{}, 1, Wb);
            float ba[4], bb[4];
            bias4(0, ba);
            bias4(1, bb);
            // block 0 of conv2_j sits at column c0 = q0_j + (-q0_j): ring slots relative to the resblock's origin
            unsigned pra = (unsigned)(-g.q0[0] - g.r2[0]), prb = (unsigned)(-g.q0[1] - g.r2[1]);  // x1 plane slots of column c0 - r2 (tap 0)
            unsigned rra = (unsigned)(-g.q0[0]), rrb = (unsigned)(-g.q0[1]);                       // raw slots of column c0
            unsigned ow = 0;
            auto iter = [&](int it, auto ACT) MI355_INLINE_LAMBDA {
                if constexpr (decltype(ACT)::value) {
                    f32x4 acc[NT];
                    uint4 bfirst[3];
                    prime(bfirst, g.x1_off[0], (unsigned)g.X1R[0], mrfs_wrap(pra, (unsigned)g.X1R[0]), dd2[0]);
                    MI355_UNROLL
                    for (int i = 0; i < NT; ++i) {
                        MI355_UNROLL
                        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
                        const unsigned ps = mrfs_wrap(pra + 16u * i, (unsigned)g.X1R[0]);
                        const unsigned pn = i + 1 < NT ? mrfs_wrap(pra + 16u * (i + 1), (unsigned)g.X1R[0]) : ps;
                        conv2_tile(std::integral_constant<int, K0>{}, Wa, ba, 0, dd2[0], acc[i], mrfs_wrap(rra + 16u * i, (unsigned)g.RR[0]), ps, bfirst, pn);
                    }
                    prime(bfirst, g.x1_off[1], (unsigned)g.X1R[1], mrfs_wrap(prb, (unsigned)g.X1R[1]), dd2[1]);
                    MI355_UNROLL
                    for (int i = 0; i < NT; ++i) {
                        const unsigned ps = mrfs_wrap(prb + 16u * i, (unsigned)g.X1R[1]);
                        const unsigned pn = i + 1 < NT ? mrfs_wrap(prb + 16u * (i + 1), (unsigned)g.X1R[1]) : ps;
                        conv2_tile(std::integral_constant<int, K1>{}, Wb, bb, 1, dd2[1], acc[i], mrfs_wrap(rrb + 16u * i, (unsigned)g.RR[1]), ps, bfirst, pn);
                        const unsigned os = mrfs_wrap(ow + 16u * i + (unsigned)n, (unsigned)g.OR);
                        *reinterpret_cast<float4*>(L0 + g.out_off + ((unsigned)(co0 >> 2) * 16u * (unsigned)g.OR + 16u * os)) =
                            make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                    }
                    pra = mrfs_wrap(pra + TS, (unsigned)g.X1R[0]);
                    prb = mrfs_wrap(prb + TS, (unsigned)g.X1R[1]);
                    rra = mrfs_wrap(rra + TS, (unsigned)g.RR[0]);
                    rrb = mrfs_wrap(rrb + TS, (unsigned)g.RR[1]);
                    ow = mrfs_wrap(ow + TS, (unsigned)g.OR);
                }
                __syncthreads();
            };
            MI355_NOUNROLL
            for (int it = 0; it < g.A0; ++it) iter(it, std::false_type{});
            MI355_NOUNROLL
            for (int it = g.A0; it < g.A0 + N; ++it) iter(it, std::true_type{});
            MI355_NOUNROLL
            for (int it = g.A0 + N; it < NIT; ++it) iter(it, std::false_type{});
        } else {
            // ============================================================ conv2 of resblock 2 (+ the staging of x): y = (out + rb2) * scale
            uint4 Wc[1][K2][3];
            load_w(std::integral_constant<int, K2>{}, 2, Wc);
            float bc[4];
            bias4(2, bc);
            unsigned prc = (unsigned)(-g.q0[2] - g.r2[2]), rrc = (unsigned)(-g.q0[2]);
            unsigned orr = 0, xsw = 0;
            // staging: the 128 threads of these two waves take one record (eight channels of one column) of the block each
            const int st = tid & 127;
            const int srec = st / TS, scol = st - srec * TS;
            auto iter = [&](int it, auto ACT) MI355_INLINE_LAMBDA {
                float sv[8];
                const int ts = c0 + g.s0 + it * TS + scol;
                if (!(LAB_ABLATE(a) & 2)) {
                    const int tc = ts < 0 ? 0 : (ts > last ? last : ts);
                    const unsigned o = 4u * (unsigned)(8 * srec * a.x_ld + tc);
                    MI355_UNROLL
                    for (int e = 0; e < 8; ++e) sv[e] = buf_load_f32(xbuf, o, (unsigned)e * xrow);
                    SCHED_FENCE();  // issued here, an iteration's tiles ahead of their use
                }
                if constexpr (decltype(ACT)::value) {
                    const int m = it - g.A2;
                    uint4 bfirst[3];
                    prime(bfirst, g.x1_off[2], (unsigned)g.X1R[2], mrfs_wrap(prc, (unsigned)g.X1R[2]), dd2[2]);
                    MI355_UNROLL
                    for (int i = 0; i < NT; ++i) {
                        const unsigned os = mrfs_wrap(orr + 16u * i + (unsigned)n, (unsigned)g.OR);
                        const float4 ov = *reinterpret_cast<const float4*>(L0 + g.out_off + ((unsigned)(co0 >> 2) * 16u * (unsigned)g.OR + 16u * os));
                        f32x4 acc;
                        acc[0] = ov.x; acc[1] = ov.y; acc[2] = ov.z; acc[3] = ov.w;
                        const unsigned ps = mrfs_wrap(prc + 16u * i, (unsigned)g.X1R[2]);
                        const unsigned pn = i + 1 < NT ? mrfs_wrap(prc + 16u * (i + 1), (unsigned)g.X1R[2]) : ps;
                        conv2_tile(std::integral_constant<int, K2>{}, Wc, bc, 2, dd2[2], acc, mrfs_wrap(rrc + 16u * i, (unsigned)g.RR[2]), ps, bfirst, pn);
                        const int t = c0 + m * TS + 16 * i + n;
                        const unsigned o = (t < a.T && !(LAB_ABLATE(a) & 4)) ? 4u * (unsigned)(co0 * a.y_ld + t) : BUF_OOB;
                        MI355_UNROLL
                        for (int r = 0; r < 4; ++r) buf_store_f32(ybuf, o, (unsigned)r * yrow, acc[r] * out_mul);
                    }
                    prc = mrfs_wrap(prc + TS, (unsigned)g.X1R[2]);
                    rrc = mrfs_wrap(rrc + TS, (unsigned)g.RR[2]);
                    orr = mrfs_wrap(orr + TS, (unsigned)g.OR);
                }
                if (!(LAB_ABLATE(a) & 2)) {
                    const bool s_in = ts >= 0 && ts < len;
                    float v[8];
                    MI355_UNROLL
                    for (int e = 0; e < 8; ++e) v[e] = s_in ? fmaxf(sv[e], 0.1f * sv[e]) : 0.0f;
                    uint4 h, mm, l;
                    split3_pk(v[0], v[1], h.x, mm.x, l.x);
                    split3_pk(v[2], v[3], h.y, mm.y, l.y);
                    split3_pk(v[4], v[5], h.z, mm.z, l.z);
                    split3_pk(v[6], v[7], h.w, mm.w, l.w);
                    const unsigned ws = mrfs_wrap(xsw + (unsigned)scol, (unsigned)g.XR);
                    char* px = L0 + g.x_off + ((unsigned)srec * XR16 + 16u * ws);
                    *reinterpret_cast<uint4*>(px) = h;
                    *reinterpret_cast<uint4*>(px + PSX16) = mm;
                    *reinterpret_cast<uint4*>(px + 2u * PSX16) = l;
                    xsw = mrfs_wrap(xsw + TS, (unsigned)g.XR);
                }
                __syncthreads();
            };
            MI355_NOUNROLL
            for (int it = 0; it < g.A2; ++it) iter(it, std::false_type{});
            MI355_NOUNROLL
            for (int it = g.A2; it < NIT; ++it) iter(it, std::true_type{});
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool mrf_s1_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    if (C != S1_C || !(nrb == 3 && k[0] == 3 && k[1] == 5 && k[2] == 7)) return false;  // instantiated tap sequence: the "_low" voices'
    for (int j = 0; j < 3; ++j)
        if (d1[j] < 1 || d2[j] < 1) return false;
#ifndef MI355_EMU
    if (!(d1[0] == 1 && d2[0] == 2 && d1[1] == 2 && d2[1] == 6 && d1[2] == 3 && d2[2] == 12)) return false;  // compile-time shapes only
#endif
    return mrfs1_geo(k, d1, d2).ok;
}

// segment length for a grid (see mrf_s_segment): steps of 32 columns, at least 48 of them per segment (the pipeline fill is
// A0 + LAG = 7 iterations); 0 = the stage is too small, the caller runs k_mrf_p (same bits)
int mrf_s1_segment(int B, int T, int cus) {
    if ((long)S1_C * T * 4 >= 0x7fffffffL) return 0;
    const int min_blocks = 48;
    const int row_blocks = (T + S1_TS - 1) / S1_TS;
    if ((long)B * row_blocks < (long)cus * min_blocks) return 0;
    int best_pr = 0;
    double best = 0.0;
    for (int pr = 1; pr <= 4 * cus && row_blocks / pr >= min_blocks; ++pr) {
        const int n = (row_blocks + pr - 1) / pr;
        const int segs = (row_blocks + n - 1) / n;
        const long items = (long)B * segs;
        const long rounds = (items + cus - 1) / cus;
        const double eff = (double)items / (double)(rounds * cus) * (double)n / (double)(n + 7);  // (7 iterations of pipeline fill at the "_low" dilations)
        if (eff > best + 1e-9) {
            best = eff;
            best_pr = pr;
        }
    }
    if (best_pr == 0) return 0;
    return (row_blocks + best_pr - 1) / best_pr * S1_TS;
}

void launch_mrf_s1(MrfArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!mrf_s1_supported(a.C, a.nrb, a.k, a.d1, a.d2) || a.seg <= 0 || a.seg % S1_TS != 0) throw std::runtime_error("mrf_s1: unsupported stage shape");
    const MrfS1Geo g = mrfs1_geo(a.k, a.d1, a.d2);
    const long nitems = (long)((a.T + a.seg - 1) / a.seg) * a.B;
    const int cus = current_device_cu_count();
    dim3 grid((unsigned)(nitems < cus ? nitems : cus));  // persistent: one workgroup per CU
#ifdef MI355_LAB
    {
        const char* ab = lab_getenv("MI355VITS_MRF_ABLATE");
        a.ablate = ab ? (int)strtol(ab, nullptr, 0) : 0;
    }
#endif
    auto go = [&](auto kfn) {
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)MRFS1_LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(512), g.total, s, a);
    };
    const bool low = a.d1[0] == 1 && a.d2[0] == 2 && a.d1[1] == 2 && a.d2[1] == 6 && a.d1[2] == 3 && a.d2[2] == 12;  // the "_low" voices
    if (low) { go(k_mrf_s1<true>); return; }
#ifdef MI355_EMU
    go(k_mrf_s1<false>);
#else
    throw std::runtime_error("mrf_s1: unsupported stage shape");
#endif
}

}  // namespace m355
