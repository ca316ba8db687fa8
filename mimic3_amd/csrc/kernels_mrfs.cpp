// kernels_mrfs.cpp — the 64- and 32-channel HiFi-GAN MRF stages in MATH_BF16X3 as a ROW SWEEP: the same arithmetic
// as k_mrf_p (kernels_mrfp.cpp: planes split once, v_mfma_f32_16x16x32_bf16 tiles, bit-identical results), organised so that
//   * no halo column is ever computed twice, and
//   * a conv's weight fragments are loaded into registers ONCE per (row segment, resblock), not once per 96 / 320 columns.
//
// What k_mrf_p pays for its (row, column block) work items (profiles/r03_pmc_derived.txt): conv1 of a resblock
// must cover the block plus r2 columns on both sides, in 16-column tiles — 12 tiles for 6 kept at 64 channels and k = 7
// (1.32 x the stage's MFMAs over the three resblocks, 1.10 x at 32 channels) —, and the six convs' fragments are re-streamed
// through the L1 for every item (0.77 / 0.5 ms of a 2.4 ms launch: the L1 delivers 64 B per clock and CU).
//
// Here a work item is (row, segment of `seg` columns), and the workgroup sweeps it left to right once per resblock in steps
// of TS columns with the waves SPECIALISED by conv:
//
//      staging  (all threads)  x[s0 + u TS ..)      -> three bf16 planes of lrelu(x) in an LDS ring        iteration u
//      conv1    (waves 0-3)    x1[q0 + p TS ..)     -> planes of lrelu(x1) in a second ring, raw x1 (f32)   iteration p + 2
//      conv2    (waves 4-7)    y [c0 + m TS ..)     <- y + x1 + b2 + conv2(lrelu(x1))   (global, in place)  iteration m + 2 + W1
//
// one workgroup barrier per iteration.  conv1 runs r2 columns (+ the pipeline slack) ahead of conv2, staging r1 ahead of conv1:
// every x1 column is computed exactly once, and each wave keeps the K x C/32 x 3 fragments of ITS conv and ITS 16 output
// channels in registers for the whole sweep (168 VGPRs at k = 7 and 64 channels).  The rings are addressed modulo their
// length (a wave-uniform lane mask + one v_cndmask per B fragment, mrfs_rd), so a sweep is a plain loop: no copies, no halo
// recompute except the (r1 + r2) columns of pipeline fill at a segment's start.  Everything between a load's issue and its
// use is branch-free and the tile loops are unrolled: see the note at stage_load (hipcc's wait-count pass).
// The sum over the resblocks lives in y: resblock 0 writes its raw result, 1 adds, the last adds and scales — in exactly
// k_mrf_p's order of additions (out = ((0 + rb0) + rb1) + rb2, each as `old + (x1 + b2)` feeding the accumulator chain), so the
// two kernels agree bit for bit and the launcher may choose by grid size.  The price is HBM traffic: x is read once per
// resblock and y is read / written once more per resblock after the first (4 x the stage's algorithmic bytes).  That pays
// at 64 channels (2.42 -> 1.8 ms per launch at the bench shape) and not at 32 (half the MFMAs per byte and per tile: 2.6 - 2.85 ms
// against k_mrf_p's 2.31): mrf_s_segment keeps the 32-channel stage on k_mrf_p.  Experiments: profiles/r04_mrf_sweep.txt.
#include <type_traits>

#include "mrfs.h"

namespace m355 {

// phase clocks of one workgroup's waves 0 (conv1) and 4 (conv2), lab build, MI355VITS_MRF_ABLATE bit 128: shader cycles spent in an
// iteration's tiles, in its staging store and waiting at its barrier, summed over a sweep
#if defined(MI355_LAB) && !defined(MI355_EMU)
#define MRFS_CLK_DECL() long long ck_t = 0, ck_s = 0, ck_b = 0, ck_0 = 0, ck_1 = 0, ck_2 = 0, ck_3 = 0
#define MRFS_CLK(v) do { if (a.ablate & 128) v = __builtin_readcyclecounter(); } while (0)
#define MRFS_CLK_ACC() do { ck_t += ck_1 - ck_0; ck_s += ck_2 - ck_1; ck_b += ck_3 - ck_2; } while (0)
#define MRFS_CLK_PRINT(K, its)                                                                                                     \
    do {                                                                                                                           \
        if ((a.ablate & 128) && blockIdx.x == 7 && (threadIdx.x == 0 || threadIdx.x == 256))                                       \
            printf("mrf_s clocks k=%d role %d: %d iterations, per iteration: tiles %lld staging-store %lld barrier %lld\n", K, role, its, \
                   ck_t / (its), ck_s / (its), ck_b / (its));                                                                       \
    } while (0)
#else
#define MRFS_CLK_DECL() ((void)0)
#define MRFS_CLK(v) ((void)0)
#define MRFS_CLK_ACC() ((void)0)
#define MRFS_CLK_PRINT(K, its) ((void)0)
#endif

namespace {
constexpr size_t MRFS_LDS_LIMIT = 160 * 1024;
constexpr int MRFS_NT = 3;  // 16-column tiles per wave and iteration
constexpr int MRFS_CONV2_PRIO = 4;  // dynamic: the conv2 waves lead for the first two of an iteration's three tiles (round 5: - 3 %)
}  // namespace

// compile-time ring lengths (columns) and dilations of the "_low" voices' stages, or MrfSDyn = take them from the arguments
template <int XR_, int X1R_, int RR_, int D10, int D20, int D11, int D21, int D12, int D22>
struct MrfSShape {
    static constexpr int XR = XR_, X1R = X1R_, RR = RR_;
    static constexpr int d1(int j) { return j == 0 ? D10 : (j == 1 ? D11 : D12); }
    static constexpr int d2(int j) { return j == 0 ? D20 : (j == 1 ? D21 : D22); }
};
using MrfSDyn = MrfSShape<0, 0, 0, 0, 0, 0, 0, 0, 0>;

template <int C, int K0, int K1, int K2, typename SH>
__global__ __launch_bounds__(512) void k_mrf_s(MrfArgs a) {
    static_assert(C == 64 || C == 32, "four or two 16-row tiles");
    constexpr int G = C / 32, NMT = C / 16, NCH = 4 / NMT, NT = MRFS_NT, TS = 16 * NT * NCH, TAP = G * 3 * 64;
    constexpr int NREC = TS * (C / 8);         // 8-channel records of one staged block: one per thread, threads 0 .. NREC - 1
    static_assert(NREC <= 512 && 2 * NREC >= 512, "one record per thread; threads past NREC repeat records 0 .. 511 - NREC");
    DYN_SMEM(float, smem);
    const int XR = SH::XR ? SH::XR : a.ldx, X1R = SH::X1R ? SH::X1R : a.ld1, RR = SH::RR ? SH::RR : a.R;
    const unsigned XR16 = 16u * XR, X1R16 = 16u * X1R, RR16 = 16u * RR;
    const unsigned PSX16 = (unsigned)G * 4u * XR16, PS116 = (unsigned)G * 4u * X1R16;
    // LDS window: raw x1 first, so that every plane address minus one ring length (the wrapped form of a lane's base, see
    // mrfs_rd) is still a non-negative offset from the window's base
    char* L0 = reinterpret_cast<char*>(smem);
    char* Rw = L0;                                        // [C / 4][RR] x 16 B       raw x1 (f32), four channels per slot
    float* BS = reinterpret_cast<float*>(Rw + (unsigned)(C / 4) * RR16);  // [MRF_MAX_RB][2][C] biases
    const unsigned XOFF = (unsigned)(C / 4) * RR16 + (unsigned)(MRF_MAX_RB * 2 * C * sizeof(float));
    char* Xp = L0 + XOFF;                                 // [3][G * 4][XR] x 16 B    planes of lrelu(x), zero outside the row
    const unsigned X1OFF = XOFF + 3u * PSX16;
    char* X1p = L0 + X1OFF;                               // [3][G * 4][X1R] x 16 B   planes of lrelu(x1), zero outside the row
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int role = wid >> 2;                 // 0: conv1, 1: conv2 (+ staging)
    const int mt = (wid & 3) % NMT, chh = (wid & 3) / NMT;
    const int q = lane >> 4, n = lane & 15;
    const int gq = mt >> 1, hh = mt & 1;
    const int co0 = 32 * gq + 8 * q + 4 * hh;  // this lane's four output channels co0 .. co0 + 3 = half hh of record (gq, q)
    const unsigned n16 = 16u * n;

    for (int i = tid; i < a.nrb * 2 * C; i += 512) BS[i] = a.bias[i / (2 * C)][(i / C) & 1][i % C];
    __syncthreads();
#if !defined(MI355_EMU)
    // The SIMD's arbiter favours its older wave: waves 0 - 3 (conv1) ran their tiles 15 - 20 % faster than their SIMD partners
    // 4 - 7 (conv2) and then sat at the barrier while the partner finished alone (phase clocks, profiles/r04_mrf_sweep.txt; with the
    // roles swapped the slow side swapped too).  A higher user priority for the younger waves evens the two out.
    if (role == 1) {
        if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    }
#endif

    const int nseg = (a.T + a.seg - 1) / a.seg;
    const int nitems = nseg * a.B;
    const float out_mul = a.out_scale > 0.0f ? a.out_scale : 1.0f / (float)a.nrb;  // k_mrf_p: out * inv_rb / out * out_scale

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = WAVE_UNIFORM(item / nseg), c0 = WAVE_UNIFORM((item - b * nseg) * a.seg);
        int len = a.len ? a.len[b] : a.T;
        if (len > a.T) len = a.T;
        len = WAVE_UNIFORM(len);
        const int last = len > 0 ? len - 1 : 0;
        // ragged batches (round 6): a segment past the row's length is skipped, one that straddles it ends at the length — columns
        // past a row's end are never read unmasked by any consumer (kernels_mrfp.cpp next_item)
        if (c0 >= len) continue;
        const int cend = c0 + a.seg < a.T ? c0 + a.seg : a.T;
        const int c1 = cend < len ? cend : len;
        const int N = WAVE_UNIFORM((c1 - c0 + TS - 1) / TS);  // output blocks of this segment
        const float* xb = a.x + (long)b * a.x_bs;
        float* yb = a.y + (long)b * a.y_bs;
        // the row as a buffer: a lane's byte offset in one VGPR, the channel's row offset in an SGPR (buffer_load ... offen)
        const BufRsrc xbuf = buf_rsrc(xb), ybuf = buf_rsrc(yb);
        const unsigned xrow = 4u * (unsigned)a.x_ld, yrow = 4u * (unsigned)a.y_ld;

        auto sweep = [&](auto KC, auto JC, auto FIRSTC, auto LASTC) MI355_INLINE_LAMBDA {
            constexpr int K = decltype(KC)::value, j = decltype(JC)::value;
            constexpr bool FIRST = decltype(FIRSTC)::value, LAST = decltype(LASTC)::value;
            // B fragments ahead of their MFMAs (MRFS_AH overrides for experiments)
#ifndef MRFS_AH_SMALL
#define MRFS_AH_SMALL 1
#endif
            constexpr int AH = G == 1 ? 2 : (G * K * 12 <= 120 ? MRFS_AH_SMALL : 1);  // (k = 7 at 64 channels: no registers beyond one step ahead)
            constexpr bool DEEP = false;  // residual / y prefetch two blocks ahead instead of one: measured no gain (profiles/r04_mrf_sweep.txt)
            const int d1 = SH::d1(j) ? SH::d1(j) : a.d1[j], d2 = SH::d2(j) ? SH::d2(j) : a.d2[j];
            const int r1 = (K - 1) / 2 * d1, r2 = (K - 1) / 2 * d2;
            const int W1 = (2 * r2 + TS - 1) / TS + 1;  // iterations conv1 runs ahead of conv2
            const int N1 = N + W1 - 1;                  // conv1 blocks: x1 on [q0, q0 + N1 TS) covers [c0 - r2, c0 + N TS + r2)
            const int NS = N1 + 1;                      // staged blocks: x on [s0, s0 + NS TS)
            const int NIT = 2 + W1 + N;
            const int q0 = c0 + r2 - (W1 - 1) * TS, s0 = q0 + r1 - TS;

            // ---- this wave's fragments: its conv, its row tile, every k-group and tap
            uint4 W[G][K][3];
            {
                const uint4* wp = reinterpret_cast<const uint4*>(a.w[j][role]) + (long)mt * K * TAP + lane;
                MI355_UNROLL
                for (int g = 0; g < G; ++g)
                    MI355_UNROLL
                    for (int k = 0; k < K; ++k)
                        MI355_UNROLL
                        for (int p = 0; p < 3; ++p) W[g][k][p] = wp[k * TAP + g * (3 * 64) + p * 64];
            }
            const float4 bv = *reinterpret_cast<const float4*>(BS + (j * 2 + role) * C + co0);
            const float bia[4] = {bv.x, bv.y, bv.z, bv.w};

            // ---- staging of x block u = it as planes, by threads 0 .. NREC - 1 (both roles): a thread takes one record = eight
            // channels of one column: eight 4-byte loads issued at the top of the iteration (in flight behind the tiles), then
            // leaky-relu, split and one conflict-free 16-byte store per plane before the iteration's barrier
            // Branch-free on purpose (as are the tile loops below): a conditional block between a load's issue and its use makes
            // hipcc's wait-count pass give up on counting and emit s_waitcnt vmcnt(0) — a full drain of every outstanding load
            // and store, i.e. one exposed memory round trip per tile.  So: every thread stages (threads past NREC repeat the
            // first records: same bytes to the same slots), in every iteration (blocks past the last needed one land in ring
            // slots nobody reads any more), loads through clamped indices, stores past the row dropped by the buffer range check.
            auto stage_load = [&](int it, float (&sv)[8]) MI355_INLINE_LAMBDA {
                if (LAB_ABLATE(a) & 2) return;
                int t2 = tid;
                OPAQUE_V(t2);  // everything derived from it is recomputed per iteration (hoisted, the row offsets spill)
                t2 = t2 < NREC ? t2 : t2 - NREC;
                const int rec = t2 / TS, col = t2 - rec * TS;
                const int t = s0 + it * TS + col;
                const int tc = t < 0 ? 0 : (t > last ? last : t);  // every load unconditional (clamped), masked afterwards
                const unsigned o = 4u * (unsigned)(8 * rec * a.x_ld + tc);  // one lane offset, eight uniform row offsets
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) sv[e] = buf_load_f32(xbuf, o, (unsigned)e * xrow);
                SCHED_FENCE();  // issued HERE, a whole iteration ahead of stage_store (unfenced, hipcc sinks the loads to their use)
            };
            auto stage_store = [&](int it, const float (&sv)[8], unsigned& xsw) MI355_INLINE_LAMBDA {
                if (LAB_ABLATE(a) & 2) return;
                int t2 = tid;
                OPAQUE_V(t2);
                t2 = t2 < NREC ? t2 : t2 - NREC;
                const int rec = t2 / TS, col = t2 - rec * TS;
                const int t = s0 + it * TS + col;
                const bool s_in = t >= 0 && t < len;
                float v[8];
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) v[e] = s_in ? fmaxf(sv[e], 0.1f * sv[e]) : 0.0f;  // = leaky-relu(0.1), one compare fewer per element
                uint4 h, mm, l;
                split3_pk(v[0], v[1], h.x, mm.x, l.x);
                split3_pk(v[2], v[3], h.y, mm.y, l.y);
                split3_pk(v[4], v[5], h.z, mm.z, l.z);
                split3_pk(v[6], v[7], h.w, mm.w, l.w);
                const unsigned ws = mrfs_wrap(xsw + (unsigned)col, (unsigned)XR);
                char* px = Xp + ((unsigned)rec * XR16 + 16u * ws);
                *reinterpret_cast<uint4*>(px) = h;
                *reinterpret_cast<uint4*>(px + PSX16) = mm;
                *reinterpret_cast<uint4*>(px + 2u * PSX16) = l;
                xsw = mrfs_wrap(xsw + TS, (unsigned)XR);
            };

            if (role == 0) {
                // ================================================================== conv1 waves
                // the tile's residual (raw x at its own columns, clamped: columns outside the row are masked below) travels one
                // tile ahead of its use
                auto load_x = [&](int e0, float (&v)[4]) MI355_INLINE_LAMBDA {
                    const int t = e0 + n;
                    const int tc = t < 0 ? 0 : (t > last ? last : t);
                    const unsigned o = 4u * (unsigned)(co0 * a.x_ld + tc);  // one lane offset; the four rows are uniform base pointers
                    MI355_UNROLL
                    for (int r = 0; r < 4; ++r) v[r] = buf_load_f32(xbuf, o, (unsigned)r * xrow);
                    SCHED_FENCE();
                };
                // residuals in flight per tile position: this block's (rq) and, DEEP, the next block's (rq2) — loaded one / two blocks ahead
                float rq[NT][4], rq2[NT][4];
                MI355_UNROLL
                for (int i = 0; i < NT; ++i) {
                    load_x(q0 + chh * (16 * NT) + 16 * i, rq[i]);
                    if constexpr (DEEP) load_x(q0 + TS + chh * (16 * NT) + 16 * i, rq2[i]);
                }
                unsigned xrd = (unsigned)((TS - 2 * r1) % XR);  // x ring slot of column q0 - r1 (block 0, tap 0): (p + 1) TS - 2 r1
                unsigned x1w = 0, rww = 0;                       // x1 / raw ring slots of column q0 + p TS
                unsigned xsw = 0;                                // x ring slot of column s0 + u TS (staging)
                uint4 bfirst[3];
                // one iteration: staging loads, (ACT: this block's tiles,) staging stores, barrier.  The phases — staging only,
                // then tiles — are separate loops: a run-time `if (active)` around the tiles would leave the wait-count pass with two
                // paths of different load counts and it would drain the tiles' prefetches in front of every staging store
                MRFS_CLK_DECL();
                auto iter = [&](int it, auto ACT) MI355_INLINE_LAMBDA {
                    const int p = it - 2;
                    float sv[8];
                    stage_load(it, sv);
                    MRFS_CLK(ck_0);
                    if constexpr (decltype(ACT)::value) {
                        unsigned ringq = XOFF + (unsigned)q * XR16 + n16;  // from the window's base: the plane offsets fit the immediates
                        OPAQUE_V(ringq);
                        // a block's first tile reads its own first fragments (the columns may have been staged in the iteration
                        // that just ended: nothing of this block can be read before the barrier); the others' travel from tile to tile
                        {
                            const unsigned sb = mrfs_wrap(xrd + (unsigned)(chh * (16 * NT)), (unsigned)XR), f0 = ringq + 16u * sb;
                            mrfs_rd<G, K>(bfirst, 0, L0, f0, f0 - XR16, WAVE_UNIFORM(XR - (int)sb), lane, PSX16, XR16, d1);
                        }
                        MI355_UNROLL  // straight-line iteration bodies: the wait-count pass then counts the younger loads / stores exactly
                        for (int i = 0; i < NT; ++i) {
                            const int e0 = q0 + p * TS + chh * (16 * NT) + 16 * i;  // absolute column of lane n = 0
                            f32x4 acc;
                            MI355_UNROLL
                            for (int r = 0; r < 4; ++r) acc[r] = rq[i][r] + bia[r];
                            if constexpr (DEEP) {
                                MI355_UNROLL
                                for (int r = 0; r < 4; ++r) rq[i][r] = rq2[i][r];
                                load_x(e0 + 2 * TS, rq2[i]);  // the same tile two blocks on
                            } else {
                                load_x(e0 + TS, rq[i]);  // the same tile of the next block
                            }
                            const unsigned off = (unsigned)(chh * (16 * NT) + 16 * i);
                            const unsigned sb = mrfs_wrap(xrd + off, (unsigned)XR);
                            const unsigned sbn = i + 1 < NT ? mrfs_wrap(xrd + off + 16u, (unsigned)XR) : sb;  // (the last tile re-reads its own: discarded)
                            if (!(LAB_ABLATE(a) & 1)) mrfs_tile<G, K, AH>(acc, W, L0, ringq, PSX16, (unsigned)XR, sb, d1, lane, bfirst, sbn);
                            // epilogue: x1 (zero outside the row) -> leaky-relu -> three bf16 planes (truncation split, v = h + m + l
                            // exactly), one 8-byte store per plane; raw x1 (unmasked: conv2's residual, as in k_mrf_p) as one float4
                            const int t = e0 + n;
                            const bool live = t >= 0 && t < len;
                            float v[4];
                            MI355_UNROLL
                            for (int r = 0; r < 4; ++r) v[r] = live ? fmaxf(acc[r], 0.1f * acc[r]) : 0.0f;
                            unsigned u[4], ur[4];
                            float rr[4], s4[4];
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
                            const unsigned ws = mrfs_wrap(x1w + off + (unsigned)n, (unsigned)X1R);
                            char* p1 = X1p + ((unsigned)(gq * 4 + q) * X1R16 + 16u * ws + 8u * (unsigned)hh);
                            *reinterpret_cast<uint2*>(p1) = ph;
                            *reinterpret_cast<uint2*>(p1 + PS116) = pm;
                            *reinterpret_cast<uint2*>(p1 + 2u * PS116) = pl;
                            const unsigned wr = mrfs_wrap(rww + off + (unsigned)n, (unsigned)RR);
                            *reinterpret_cast<float4*>(Rw + ((unsigned)(co0 >> 2) * RR16 + 16u * wr)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                        }
                        xrd = mrfs_wrap(xrd + TS, (unsigned)XR);
                        x1w = mrfs_wrap(x1w + TS, (unsigned)X1R);
                        rww = mrfs_wrap(rww + TS, (unsigned)RR);
                    }
                    MRFS_CLK(ck_1);
                    stage_store(it, sv, xsw);
                    MRFS_CLK(ck_2);
                    __syncthreads();
                    MRFS_CLK(ck_3);
                    if constexpr (decltype(ACT)::value) MRFS_CLK_ACC();
                };
                MI355_NOUNROLL
                for (int it = 0; it < 2; ++it) iter(it, std::false_type{});
                MI355_NOUNROLL
                for (int it = 2; it < 2 + N1; ++it) iter(it, std::true_type{});
                MI355_NOUNROLL
                for (int it = 2 + N1; it < NIT; ++it) iter(it, std::false_type{});
                MRFS_CLK_PRINT(K, N1);
            } else {
                // ================================================================== conv2 waves (+ staging of x)
                auto load_y = [&](int t0, float (&v)[4]) MI355_INLINE_LAMBDA {
                    if (FIRST || (LAB_ABLATE(a) & 16)) return;
                    const int t = t0 + n;
                    const int tc = t < a.T ? t : a.T - 1;
                    const unsigned o = 4u * (unsigned)(co0 * a.y_ld + tc);
                    MI355_UNROLL
                    for (int r = 0; r < 4; ++r) v[r] = buf_load_f32(ybuf, o, (unsigned)r * yrow);
                    SCHED_FENCE();
                };
                float yq[NT][4], yq2[NT][4];  // y per tile position, one / two (DEEP) blocks ahead (zeros in the first resblock: nothing to read)
                MI355_UNROLL
                for (int i = 0; i < NT; ++i) {
                    MI355_UNROLL
                    for (int r = 0; r < 4; ++r) yq[i][r] = yq2[i][r] = 0.0f;
                    load_y(c0 + chh * (16 * NT) + 16 * i, yq[i]);
                    if constexpr (DEEP) load_y(c0 + TS + chh * (16 * NT) + 16 * i, yq2[i]);
                }
                unsigned x1r = (unsigned)(((W1 - 1) * TS - 2 * r2) % X1R);  // x1 ring slot of column c0 - r2 (block 0, tap 0)
                unsigned rwr = (unsigned)(((W1 - 1) * TS - r2) % RR);       // raw ring slot of column c0
                unsigned xsw = 0;                                            // x ring slot of column s0 + u TS
                uint4 bfirst[3];
                MRFS_CLK_DECL();
                auto iter = [&](int it, auto ACT) MI355_INLINE_LAMBDA {
                    const int m = it - 2 - W1;
                    float sv[8];
                    stage_load(it, sv);
                    MRFS_CLK(ck_0);
                    if constexpr (decltype(ACT)::value) {
                        unsigned ringq = X1OFF + (unsigned)q * X1R16 + n16;
                        OPAQUE_V(ringq);
                        const char* rawq = Rw + (unsigned)(co0 >> 2) * RR16;
                        {
                            const unsigned sb = mrfs_wrap(x1r + (unsigned)(chh * (16 * NT)), (unsigned)X1R), f0 = ringq + 16u * sb;
                            mrfs_rd<G, K>(bfirst, 0, L0, f0, f0 - X1R16, WAVE_UNIFORM(X1R - (int)sb), lane, PS116, X1R16, d2);
                        }
                        float4 x1n = *reinterpret_cast<const float4*>(rawq + 16u * mrfs_wrap(rwr + (unsigned)(chh * (16 * NT)) + (unsigned)n, (unsigned)RR));
                        MI355_UNROLL  // straight-line iteration bodies: the wait-count pass then counts the younger loads / stores exactly
                        for (int i = 0; i < NT; ++i) {
#if !defined(MI355_EMU)
                            // a.prio == 4 (the product default, MRFS_CONV2_PRIO; MI355VITS_MRF_PRIO overrides it in the lab build): the younger wave of the SIMD pair leads for the first
                            // two of an iteration's three tiles, the older one for the last — instead of the older one leading throughout
                            // and idling at the barrier.  (Test and branch inside one asm statement: kernels_rbc.cpp RBC_SETPRIO_YOUNG.)
                            // (a.prio: 4 = prio 1 for tiles 0, 1; 5 = for tile 0 only; 6 = for the whole iteration's tiles, 0 for its staging store)
                            if (i == 0) asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 1f\n\ts_setprio 1\n1:" ::"s"(a.prio) : "scc");
                            if (i == 1) asm volatile("s_cmp_lg_u32 %0, 5\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(a.prio) : "scc");
                            if (i == NT - 1) asm volatile("s_cmp_lg_u32 %0, 4\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(a.prio) : "scc");
#endif
                            const int t0 = c0 + m * TS + chh * (16 * NT) + 16 * i;
                            const unsigned off = (unsigned)(chh * (16 * NT) + 16 * i);
                            const float x1a[4] = {x1n.x, x1n.y, x1n.z, x1n.w};
                            f32x4 acc;
                            MI355_UNROLL
                            for (int r = 0; r < 4; ++r) acc[r] = (FIRST ? 0.0f : yq[i][r]) + (x1a[r] + bia[r]);  // k_mrf_p: out + (x1 + b2)
                            if constexpr (DEEP && !FIRST) {
                                MI355_UNROLL
                                for (int r = 0; r < 4; ++r) yq[i][r] = yq2[i][r];
                                load_y(t0 + 2 * TS, yq2[i]);  // the same tile two blocks on
                            } else {
                                load_y(t0 + TS, yq[i]);
                            }
                            // the next tile's residual and first fragments (same block only: the next block's may still be in the making)
                            const unsigned offn = i + 1 < NT ? off + 16u : off;
                            x1n = *reinterpret_cast<const float4*>(rawq + 16u * mrfs_wrap(rwr + offn + (unsigned)n, (unsigned)RR));
                            const unsigned sb = mrfs_wrap(x1r + off, (unsigned)X1R);
                            const unsigned sbn = mrfs_wrap(x1r + offn, (unsigned)X1R);
                            if (!(LAB_ABLATE(a) & 1)) mrfs_tile<G, K, AH>(acc, W, L0, ringq, PS116, (unsigned)X1R, sb, d2, lane, bfirst, sbn);
                            const int t = t0 + n;
                            // columns past the tensor: the lane's offset is moved out of the buffer's range and the hardware drops the
                            // store (no branch around the stores: see stage_load)
                            const unsigned o = (t < a.T && !(LAB_ABLATE(a) & 4)) ? 4u * (unsigned)(co0 * a.y_ld + t) : BUF_OOB;
                            MI355_UNROLL
                            for (int r = 0; r < 4; ++r) buf_store_f32(ybuf, o, (unsigned)r * yrow, LAST ? acc[r] * out_mul : acc[r]);
                        }
                        x1r = mrfs_wrap(x1r + TS, (unsigned)X1R);
                        rwr = mrfs_wrap(rwr + TS, (unsigned)RR);
#if !defined(MI355_EMU)
                        asm volatile("s_cmp_lg_u32 %0, 6\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(a.prio) : "scc");
#endif
                    }
                    MRFS_CLK(ck_1);
                    stage_store(it, sv, xsw);
                    MRFS_CLK(ck_2);
                    __syncthreads();
                    MRFS_CLK(ck_3);
                    if constexpr (decltype(ACT)::value) MRFS_CLK_ACC();
                };
                MI355_NOUNROLL
                for (int it = 0; it < 2 + W1; ++it) iter(it, std::false_type{});
                MI355_NOUNROLL
                for (int it = 2 + W1; it < NIT; ++it) iter(it, std::true_type{});
                MRFS_CLK_PRINT(K, N);
            }
        };

        sweep(std::integral_constant<int, K0>{}, std::integral_constant<int, 0>{}, std::true_type{}, std::integral_constant<bool, K1 == 0>{});
        if constexpr (K1 > 0)
            sweep(std::integral_constant<int, K1>{}, std::integral_constant<int, 1>{}, std::false_type{}, std::integral_constant<bool, K2 == 0>{});
        if constexpr (K2 > 0) sweep(std::integral_constant<int, K2>{}, std::integral_constant<int, 2>{}, std::false_type{}, std::true_type{});
    }
}

// ------------------------------------------------------------------------------------------------ host side
namespace {
struct GeoS { int TS, XR, X1R, RR; size_t lds; };
// ring lengths: multiples of 16 columns (a 16-lane fragment read then touches 16 consecutive 16-byte slots modulo the ring:
// conflict-free), at least the span between a ring's oldest column still read and its newest column written in an iteration
inline bool geometry_s(int C, int nrb, const int* k, const int* d1, const int* d2, GeoS* g) {
    if (C != 64) return false;  // (the 32-channel stage stays on k_mrf_p: its sweep forms lost or tied)
    const int TS = 16 * MRFS_NT * (4 / (C / 16));
    int r1m = 0, r2m = 0;
    for (int j = 0; j < nrb; ++j) {
        if (!(k[j] == 3 || k[j] == 5 || k[j] == 7) || d1[j] < 1 || d2[j] < 1) return false;
        const int r1 = (k[j] - 1) / 2 * d1[j], r2 = (k[j] - 1) / 2 * d2[j];
        if (2 * r1 > TS) return false;  // the staging lead of one block
        r1m = r1m > r1 ? r1m : r1;
        r2m = r2m > r2 ? r2m : r2;
    }
    g->TS = TS;
    g->XR = (2 * TS + 2 * r1m + 15) & ~15;
    g->X1R = (2 * TS + 2 * r2m + 15) & ~15;
    g->RR = (2 * TS + r2m + 15) & ~15;
    // a tile's reach inside a ring must stay below the ring's length (mrfs_wrap takes offsets < 2 x ring)
    if (15 + 2 * r1m >= g->XR || 15 + 2 * r2m >= g->X1R) return false;
    g->lds = (size_t)(C / 32) * 4 * 3 * 16 * (size_t)(g->XR + g->X1R) + (size_t)(C / 4) * 16 * g->RR + (size_t)MRF_MAX_RB * 2 * C * sizeof(float);
    return g->lds <= MRFS_LDS_LIMIT;
}
}  // namespace

bool mrf_s_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    GeoS g;
    if (!(nrb == 3 && k[0] == 3 && k[1] == 5 && k[2] == 7)) return false;  // instantiated tap sequence: the "_low" voices'
#ifndef MI355_EMU
    if (!(d1[0] == 1 && d2[0] == 2 && d1[1] == 2 && d2[1] == 6 && d1[2] == 3 && d2[2] == 12)) return false;  // compile-time shapes only
#endif
    return geometry_s(C, nrb, k, d1, d2, &g);
}

// Segment length for a grid: the sweep pays (r1 + r2) columns + three to five iterations of pipeline fill per (segment,
// resblock), so segments should be long; the chip wants at least one item per CU and an even number of them per CU.
// Returns 0 when the stage is too small for the sweep to pay (the caller runs k_mrf_p: same bits).
int mrf_s_segment(int C, int B, int T, int cus, const int* len_host) {
    // measured on the MI355X at the bench shape (profiles/r04_mrf_sweep.txt): 64 channels 2.42 -> 2.1 ms per launch; 32 channels
    // 2.36 -> 3.1 ms (half the matrix work per tile and per byte of y / x traffic: the three passes' 4 x HBM bytes and the
    // per-tile costs outweigh what the sweep saves there) — the 32-channel stage stays on k_mrf_p
    if (C != 64) return 0;
    if ((long)C * T * 4 >= 0x7fffffffL) return 0;  // a row's bytes must fit the buffer range (32-bit lane offsets): k_mrf_p otherwise
    const int TS = 16 * MRFS_NT * (4 / (C / 16));
    const int min_blocks = 24;  // fill of <= 5 iterations: <= 20 % even at the shortest segment
    // ragged batches (round 6): a row's segments past its length are not computed, so the segment length follows the blocks that
    // HAVE work — len_host = the host's copy of the rows' lengths (nullptr: every row T)
    std::vector<int> rb((size_t)B);
    long total_blocks = 0;
    for (int b = 0; b < B; ++b) {
        const int ln = len_host ? (len_host[b] > T ? T : (len_host[b] < 0 ? 0 : len_host[b])) : T;
        rb[(size_t)b] = (ln + TS - 1) / TS;
        total_blocks += rb[(size_t)b];
    }
    if (total_blocks < (long)cus * min_blocks) return 0;
    bool uniform = true;
    for (int b = 0; b < B; ++b) uniform = uniform && rb[(size_t)b] == (T + TS - 1) / TS;
    // segments per row: the count that keeps the chip busiest — items / (rounds x CUs) of the persistent loop — times the
    // share of a sweep that is not pipeline fill (N of N + 5 iterations); ties go to the fewer, longer segments
    const int row_blocks = (T + TS - 1) / TS;
    int best_pr = 0;
    double best = 0.0;
    for (int pr = 1; pr <= 4 * cus && row_blocks / pr >= min_blocks; ++pr) {
        const int n = (row_blocks + pr - 1) / pr;  // blocks per segment
        long items = 0;
        for (int b = 0; b < B; ++b) items += (rb[(size_t)b] + n - 1) / n;
        const long rounds = (items + cus - 1) / cus;
        // uniform rows: round 4's measure (fill of the persistent rounds x share of a sweep that is not pipeline fill); ragged rows:
        // the chip's work (blocks with work, spread evenly) over what the busiest CU does (its rounds of n + 5 iterations)
        const double eff = uniform ? (double)items / (double)(rounds * cus) * (double)n / (double)(n + 5)
                                   : ((double)total_blocks / (double)cus) / ((double)rounds * (double)(n + 5));
        if (eff > best + 1e-9) {
            best = eff;
            best_pr = pr;
        }
    }
    if (best_pr == 0) return 0;
    return (row_blocks + best_pr - 1) / best_pr * TS;
}

void launch_mrf_s(MrfArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    GeoS g;
    if (!geometry_s(a.C, a.nrb, a.k, a.d1, a.d2, &g) || a.seg <= 0 || a.seg % g.TS != 0) throw std::runtime_error("mrf_s: unsupported stage shape");
    a.ldx = g.XR;
    a.ld1 = g.X1R;
    a.R = g.RR;
    a.prio = MRFS_CONV2_PRIO;  // s_setprio of the conv2 waves (see the kernel)
    if (const char* pr = lab_getenv("MI355VITS_MRF_PRIO")) a.prio = atoi(pr);
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
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)MRFS_LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(512), g.lds, s, a);
    };
    const int k1 = a.nrb > 1 ? a.k[1] : 0, k2 = a.nrb > 2 ? a.k[2] : 0;
    if (!(a.k[0] == 3 && k1 == 5 && k2 == 7)) throw std::runtime_error("mrf_s: unsupported tap counts");
    const bool low = a.d1[0] == 1 && a.d2[0] == 2 && a.d1[1] == 2 && a.d2[1] == 6 && a.d1[2] == 3 && a.d2[2] == 12;  // the "_low" voices
    if (a.C != 64) throw std::runtime_error("mrf_s: 64 channels only");
    if (low && g.XR == 128 && g.X1R == 176 && g.RR == 144) { go(k_mrf_s<64, 3, 5, 7, MrfSShape<128, 176, 144, 1, 2, 2, 6, 3, 12>>); return; }
#ifdef MI355_EMU
    go(k_mrf_s<64, 3, 5, 7, MrfSDyn>);
#else
    throw std::runtime_error("mrf_s: unsupported stage shape");
#endif
}

}  // namespace m355
