// kernels_rbc.cpp — the decoder's convs whose whole input fits LDS, in MATH_BF16X3: k_rb_conv (below), k_ups_pl and k_ups64 (the
// polyphase upsamplers, further down).  k_rb_conv = one dense Conv1d of a 128-channel ResBlock2 (stage 0 of the HiFi-GAN decoder):
//
//      y[b, co, t] (+)= (res[b, co, t] + bias[co] + sum_ci sum_k W[co, ci, k] * lrelu(x[b, ci, t - pad + k dil] * mask)) * out_scale
//
// Why this stage has its own kernel.  At 128 channels neither MRF kernel's idea fits: a conv's weight fragments are 96 KiB per tap
// (k_mrf_p / k_mrf_s keep them in registers: 688 KiB at seven taps against a 512 KiB register file), and a resblock's two
// activation tiles with their halos (k = 7, dilations 3 and 12: 72 + 90 columns) leave no room in LDS for output columns.  So
// the stage runs conv by conv, as it always did — but the staged split kernel it ran on (k_conv1d_b3: 32-channel chunks, stage /
// barrier / compute / barrier per chunk, 0.34 of the matrix-core roof) pays two barriers and one exposed staging round trip per
// 32 input channels.  Here a work item is (row, 128 output columns) with ALL 128 input channels resident:
//
//   * LDS holds the item's input as three bf16 planes (x = h + m + l exactly, hipx.h) of lrelu(x), zero outside the row:
//     [plane][16 records of 8 channels][N + halo columns] x 16 B = 768 B per column, at most 208 columns = 156 KiB;
//   * eight waves = the eight 16-row tiles of output channels; a wave walks k-groups x taps ("steps") and, per step, its eight
//     16-column tiles: one weight fragment (3 x 16 B per lane, streamed from L2 four steps ahead, buffer addressing) feeds
//     48 v_mfma_f32_16x16x32_bf16, the activation fragments come from LDS one tile pair ahead (conflict-free ds_read_b128);
//   * the planes are TWO half-buffers (k-groups 0, 1 | 2, 3).  While the matrix cores work through one half, the other half is
//     refilled: phase 0 (k-groups 0, 1) stages the item's own k-groups 2, 3, phase 1 stages k-groups 0, 1 of the workgroup's NEXT
//     item — loads issued one round (eight per thread) per step at the top of the phase, leaky-relu + split + LDS stores dealt
//     out between the MFMAs of later steps (sched_group_barrier).  No staging phase with idle matrix cores, two barriers per item;
//   * the residual of a tile is prefetched during phase 1; the epilogue is bias + residual + scale (+ old y) and 64-byte row
//     stores (lanes past the tensor switched off by the buffer range check: no branch between a load's issue and its use,
//     kernels_mrfs.cpp).
// The sum order of an output element — k-group, tap, six products small terms first — does not depend on the item width, so the
// 32-column form used for small grids (one utterance) gives the same bits as the 128-column form.
// Weights: pack_conv_weights_p16 fragments (the k_mrf_p order: [row tile][tap][k-group][plane][lane]).
#include <algorithm>
#include <vector>
#include <type_traits>

#include "kernels.h"

namespace m355 {

namespace {
constexpr size_t RBC_LDS_LIMIT = 160 * 1024;
constexpr int RBC_C = 128, RBC_G = RBC_C / 32, RBC_REC = RBC_C / 8;
constexpr int RBC_PW_DEFAULT = 1;  // wide items on the producer-wave form (k_rb_conv_pw)
constexpr int RBC_WR = 4;  // weight-fragment ring: the running step and three ahead (the step count is a multiple of 4)
}  // namespace

// Item order of the persistent workgroups.  Workgroup w runs on XCD w % 8 and walks the items w, w + W, w + 2 W, ... (W = the grid);
// consecutive items of a row — which share their halo columns, or (k_ups64: 127-position items) the 128-byte lines their rows' ends
// fall into — would so land on eight different XCDs and be fetched from HBM by each of them.  XCD-major: logical item it = 8 q + x
// becomes the q-th item of XCD x's contiguous eighth of the items (a bijection for every item count), so that the 32 CUs of an XCD work
// on neighbouring items at the same time and the shared bytes come out of that XCD's L2 (k_mrf_p has walked its items like this since round 3).
__device__ __forceinline__ int rbc_item(int it, int n, int order) {
    if (!order) return it;
    const int x = it & 7, q = it >> 3, f = n >> 3, rm = n & 7;
    return x * f + (x < rm ? x : rm) + q;
}
constexpr int RBC_ITEM_ORDER = 1;

template <int K, int DIL, int NCT>
struct RbcGeo {
    static constexpr int N = 16 * NCT, PAD = (K - 1) / 2 * DIL, LD = N + 2 * PAD, LDP = (LD + 15) & ~15;
    static constexpr unsigned REC16 = 16u * LDP, PS16 = RBC_REC * REC16;  // bytes per record row / per plane
    static constexpr int HALF = (RBC_REC / 2) * LD;                        // (record, column) pairs of one half-buffer
    static constexpr int ROUNDS = (HALF + 511) / 512;                      // eight-load rounds per thread and half
    static constexpr size_t LDS = 3 * (size_t)PS16;
};

template <int K, int DIL, int NCT>
__global__ __launch_bounds__(512) void k_rb_conv(ConvArgs a) {
    using GE = RbcGeo<K, DIL, NCT>;
    constexpr int G = RBC_G, N = GE::N, PAD = GE::PAD, LD = GE::LD, LDP = GE::LDP, S = G * K, SH = S / 2;
    constexpr int ROUNDS = GE::ROUNDS, HALF = GE::HALF, NP = NCT / 2, WR = RBC_WR;
    constexpr unsigned REC16 = GE::REC16, PS16 = GE::PS16;
    constexpr int SD = (SH - ROUNDS) < 4 ? (SH - ROUNDS) : 4;  // steps between a staging round's loads and its stores
    static_assert(NCT % 2 == 0 && S % WR == 0 && SD >= 2 && GE::LDS <= RBC_LDS_LIMIT, "shape");
    DYN_SMEM(float, smem);
    char* L0 = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, mt = WAVE_UNIFORM(tid >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int co0 = 32 * (mt >> 1) + 8 * q + 4 * (mt & 1);  // this lane's four output channels (pack_conv_weights_p16's row order)
    const BufRsrc wbuf = buf_rsrc(a.w);
    const unsigned wl = 16u * (unsigned)lane;
    const unsigned wmt = (unsigned)mt * (unsigned)(K * G * 3 * 64 * 16);
    // (plane p, record q, column n): one base register per plane, so that every fragment read is base + a 16-bit immediate
    unsigned lq[3];
    MI355_UNROLL
    for (int p = 0; p < 3; ++p) {
        lq[p] = (unsigned)p * PS16 + (unsigned)(q * LDP + n) * 16u;
        OPAQUE_V(lq[p]);
    }
    float bia[4];
    MI355_UNROLL
    for (int r = 0; r < 4; ++r) bia[r] = a.bias ? a.bias[co0 + r] : 0.0f;
    // work items = the (row, column block) pairs that HAVE work, a.nvalid of them, numbered row by row (ragged batches, round 6:
    // kernels_mrfp.cpp has the argument): row b holds ceil(len_b / N); a cursor turns a valid index into (row, block) — the
    // indices a workgroup decodes only grow
    const int nitems = a.nvalid;
    const unsigned xrow = 4u * (unsigned)a.x_ld, yrow = 4u * (unsigned)a.y_ld, rrow = 4u * (unsigned)a.res_ld;

    struct Item { int b, t0, len, last; };
    // a row's length through the scalar cache (s_load_dword: the constant address space tells hipcc that the table is read-only;
    // as a plain global load it costs the item loop a vmcnt(0) — every prefetch drained — and one exposed round trip per item)
    auto row_len = [&](int b) MI355_INLINE_LAMBDA {
#ifdef MI355_EMU
        return a.in_len[b];
#else
        typedef const int __attribute__((address_space(4))) * cptr_t;
        return ((cptr_t)(a.in_len))[b];
#endif
    };
    struct Cursor { int r, base, nv; };
    auto row_nv = [&](int b) MI355_INLINE_LAMBDA {
        int len = row_len(b);
        if (len > a.T) len = a.T;
        return len > 0 ? (len + N - 1) / N : 0;
    };
    auto decode = [&](int it, Cursor& c) MI355_INLINE_LAMBDA {
        Item o;
        it = rbc_item(it, nitems, a.item_order);
        while (it >= c.base + c.nv && c.r + 1 < a.B) {  // (it < a.nvalid: ends inside the batch; the row bound guards a stale count)
            c.base += c.nv;
            ++c.r;
            c.nv = row_nv(c.r);
        }
        c.r = WAVE_UNIFORM(c.r);
        c.base = WAVE_UNIFORM(c.base);
        c.nv = WAVE_UNIFORM(c.nv);
        o.b = c.r;
        o.t0 = WAVE_UNIFORM((it - c.base) * N);
        int len = row_len(o.b);
        if (len > a.T) len = a.T;
        o.len = len;
        o.last = o.len > 0 ? o.len - 1 : 0;
        return o;
    };

    // ---- staging of one half-buffer (records 8 half .. 8 half + 7) of an item, round by round: a thread takes (record, column)
    // pairs tid + 512 round (past the end: the last pair again — same bytes to the same slot, no branch): eight 4-byte loads = the
    // record's eight channels at that column (256 contiguous bytes per wave and row), clamped into the row and masked afterwards
    auto stage_load = [&](const Item& im, const BufRsrc& xb, int half, int round, float (&sv)[8]) MI355_INLINE_LAMBDA {
        int t2 = tid;
        OPAQUE_V(t2);  // everything derived from it is recomputed per round (hoisted, the offsets of all rounds occupy registers)
        int idx = t2 + 512 * round;
        if (512 * (round + 1) > HALF) idx = idx < HALF ? idx : HALF - 1;
        const int rec = idx / LD, col = idx - rec * LD;
        const int t = im.t0 - PAD + col;
        const int tc = t < 0 ? 0 : (t > im.last ? im.last : t);
        const unsigned o = 4u * (unsigned)(8 * (8 * half + rec) * a.x_ld + tc);
        MI355_UNROLL
        for (int e = 0; e < 8; ++e) sv[e] = buf_load_f32(xb, o, (unsigned)e * xrow);
    };
    // one quarter of a round's leaky-relu + split (pieces 0 .. 3 = channel pairs), the three 16-byte stores behind the last piece
    struct StagePos { unsigned addr; bool in; };
    auto stage_pos = [&](const Item& im, int half, int round) MI355_INLINE_LAMBDA {  // where a round's record goes, and whether it is inside the row
        int t2 = tid;
        OPAQUE_V(t2);
        int idx = t2 + 512 * round;
        if (512 * (round + 1) > HALF) idx = idx < HALF ? idx : HALF - 1;
        const int rec = idx / LD, col = idx - rec * LD;
        const int t = im.t0 - PAD + col;
        StagePos sp;
        sp.in = t >= 0 && t < im.len;
        sp.addr = (unsigned)(8 * half + rec) * REC16 + 16u * (unsigned)col;
        return sp;
    };
    auto stage_piece = [&](int piece, const float (&sv)[8], uint4 (&ph)[3], const StagePos& sp) MI355_INLINE_LAMBDA {
        const float v0 = sp.in ? lrelu_f(sv[2 * piece], a.in_slope) : 0.0f, v1 = sp.in ? lrelu_f(sv[2 * piece + 1], a.in_slope) : 0.0f;
        unsigned h, m, l;
        split3_sc(v0, v1, h, m, l);
        if (piece == 0) { ph[0].x = h; ph[1].x = m; ph[2].x = l; }
        if (piece == 1) { ph[0].y = h; ph[1].y = m; ph[2].y = l; }
        if (piece == 2) { ph[0].z = h; ph[1].z = m; ph[2].z = l; }
        if (piece == 3) {
            ph[0].w = h; ph[1].w = m; ph[2].w = l;
            char* px = L0 + sp.addr;
            MI355_UNROLL
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(px + (unsigned)p * PS16) = ph[p];
        }
    };
    auto w_load = [&](int s, uint4 (&w)[3]) MI355_INLINE_LAMBDA {  // fragments of step s = (k-group s / K, tap s % K)
        const int g = s / K, k = s % K;
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) w[p] = buf_load_u4(wbuf, wl, wmt + (unsigned)(((k * G + g) * 3 + p) * 64 * 16));
    };
    auto b_read = [&](int s, int j, uint4 (&bf)[3]) MI355_INLINE_LAMBDA {  // activation fragments of tile j at step s
        const int g = s / K, k = s % K;
        MI355_UNROLL
        for (int p = 0; p < 3; ++p)
            bf[p] = *reinterpret_cast<const uint4*>(L0 + lq[p] + (unsigned)((4 * g) * LDP + 16 * j + k * DIL) * 16u);
    };

    const int it_first = (int)blockIdx.x;
    if (it_first >= nitems) return;
    Cursor cur;
    cur.r = 0;
    cur.base = 0;
    cur.nv = row_nv(0);
    // ---- prologue: half 0 of the first item, the first steps' weight fragments
    {
        const Item im = decode(it_first, cur);
        const BufRsrc xb = buf_rsrc(a.x + (long)im.b * a.x_bs);
        float sv[ROUNDS][8];
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) stage_load(im, xb, 0, r, sv[r]);
        SCHED_FENCE();
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) {
            uint4 ph[3];
            const StagePos sp = stage_pos(im, 0, r);
            MI355_UNROLL
            for (int pc = 0; pc < 4; ++pc) stage_piece(pc, sv[r], ph, sp);
        }
    }
    uint4 Wr[WR][3];
    MI355_UNROLL
    for (int s = 0; s < WR - 1; ++s) w_load(s, Wr[s]);
    __syncthreads();

    MI355_NOUNROLL
    for (int it = it_first, itv = it_first; it < nitems; it = itv) {
        itv = it + (int)gridDim.x;
        const Item im = decode(it, cur);
        const int itn = itv < nitems ? itv : it;  // (no next item: this one's half 0 again, unread)
        const Item imn = decode(itn, cur);
        const BufRsrc xb = buf_rsrc(a.x + (long)im.b * a.x_bs), xbn = buf_rsrc(a.x + (long)imn.b * a.x_bs);
        const BufRsrc ybuf = buf_rsrc(a.y + (long)im.b * a.y_bs);
        const BufRsrc rbuf = buf_rsrc(a.res + (long)im.b * a.res_bs);
        f32x4 acc[NCT];
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.0f;
        float rq[NCT][4];  // residuals (phase 1)
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) rq[j][r] = 0.0f;
        auto load_res = [&](int j) MI355_INLINE_LAMBDA {
            const int t = im.t0 + 16 * j + n;
            const int tc = t < a.T ? t : a.T - 1;
            const unsigned o = 4u * (unsigned)(co0 * a.res_ld + tc);
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) rq[j][r] = buf_load_f32(rbuf, o, (unsigned)r * rrow);
        };

        MI355_UNROLL
        for (int h = 0; h < 2; ++h) {
            // this phase computes on half h and refills half 1 - h: the item's own second half, or the next item's first
            const Item& ims = h == 0 ? im : imn;
            const BufRsrc& xbs = h == 0 ? xb : xbn;
            const int hs = 1 - h;
            float sv[ROUNDS][8];
            uint4 ph[3];
            uint4 Bf[2][2][3];
            b_read(h * SH, 0, Bf[0][0]);
            b_read(h * SH, 1, Bf[0][1]);
            MI355_UNROLL
            for (int sl = 0; sl < SH; ++sl) {
                const int s = h * SH + sl;
                // (1) this step's loads: the weight fragments three steps on, one staging round, (phase 1) residual tiles
                w_load((s + WR - 1) % S, Wr[(s + WR - 1) % WR]);
                if (sl < ROUNDS) stage_load(ims, xbs, hs, sl, sv[sl]);
                if (h == 1) {
                    MI355_UNROLL
                    for (int j = 0; j < NCT; ++j)
                        if (j * (SH - 1) / NCT == sl) load_res(j);
                }
                const int sr = sl - SD;  // round stored in this step
                StagePos sp = {0u, false};
                if (sr >= 0 && sr < ROUNDS) sp = stage_pos(ims, hs, sr);
                SCHED_FENCE();
                // (2) the tile pairs; between their MFMAs the staging round whose loads were issued SD steps ago
                MI355_UNROLL
                for (int jp = 0; jp < NP; ++jp) {
                    const int cur = (NP % 2 == 0) ? (jp & 1) : ((s * NP + jp) & 1);
                    if (jp + 1 < NP) {
                        b_read(s, 2 * jp + 2, Bf[cur ^ 1][0]);
                        b_read(s, 2 * jp + 3, Bf[cur ^ 1][1]);
                    } else if (sl + 1 < SH) {
                        b_read(s + 1, 0, Bf[cur ^ 1][0]);
                        b_read(s + 1, 1, Bf[cur ^ 1][1]);
                    }
                    SCHED_FENCE();
                    {
                        const uint4(&W)[3] = Wr[s % WR];
                        f32x4 c0 = acc[2 * jp], c1 = acc[2 * jp + 1];
                        const uint4(&B0)[3] = Bf[cur][0];
                        const uint4(&B1)[3] = Bf[cur][1];
                        c0 = MFMA_16x16x32_BF16(W[2], B0[0], c0);  // small terms first
                        c1 = MFMA_16x16x32_BF16(W[2], B1[0], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[2], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[2], c1);
                        c0 = MFMA_16x16x32_BF16(W[1], B0[1], c0);
                        c1 = MFMA_16x16x32_BF16(W[1], B1[1], c1);
                        c0 = MFMA_16x16x32_BF16(W[1], B0[0], c0);
                        c1 = MFMA_16x16x32_BF16(W[1], B1[0], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[1], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[1], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[0], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[0], c1);
                        acc[2 * jp] = c0;
                        acc[2 * jp + 1] = c1;
                    }
                    if (sr >= 0 && sr < ROUNDS) {
                        if (NP >= 4) {
                            if (jp < 4) stage_piece(jp, sv[sr], ph, sp);
                        } else {  // fewer pairs than pieces: the pieces share the pairs
                            MI355_UNROLL
                            for (int pc = 0; pc < 4; ++pc)
                                if (pc * NP / 4 == jp) stage_piece(pc, sv[sr], ph, sp);
                        }
                        MI355_UNROLL
                        for (int u = 0; u < 12; ++u) {
                            SCHED_GROUP(0x8, 1);  // one MFMA ...
                            SCHED_GROUP(0x2, 3);  // ... then up to three VALU of the staging piece
                        }
                    }
                    SCHED_FENCE();
                }
            }
            __syncthreads();
        }

        // ---- epilogue: (res + (acc + bias)) * scale (+ old y); columns past the tensor dropped by the range check.  The old
        // values of an accumulating conv are loaded for all tiles before the first store (a load behind a store cannot be
        // waited for without waiting for the store's acknowledgement: the memory counter retires in order)
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) acc[j][r] = (rq[j][r] + (acc[j][r] + bia[r])) * a.out_scale;
        if (a.accumulate) {
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j) {
                const int t = im.t0 + 16 * j + n;
                const int tc = t < a.T ? t : a.T - 1;
                const unsigned oc = 4u * (unsigned)(co0 * a.y_ld + tc);
                MI355_UNROLL
                for (int r = 0; r < 4; ++r) rq[j][r] = buf_load_f32(ybuf, oc, (unsigned)r * yrow);
            }
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j)
                MI355_UNROLL
                for (int r = 0; r < 4; ++r) acc[j][r] += rq[j][r];
        }
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j) {
            const int t = im.t0 + 16 * j + n;
            const unsigned o = t < a.T ? 4u * (unsigned)(co0 * a.y_ld + t) : BUF_OOB;
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) buf_store_f32(ybuf, o, (unsigned)r * yrow, acc[j][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_rb_conv_pw — the same conv, the same bits, with PRODUCER WAVES (round 5).  profiles/r04_rbc_step_clocks.txt: a step of k_rb_conv
// whose instruction stream carries no staging work runs at the matrix pipe's full rate for both waves of a SIMD; a step that issues a
// staging round's loads, or its leaky-relu / split / ds_write_b128, takes 2 - 6 x — a wave that waits for a slot in the vector-memory or
// LDS-store queue issues no MFMA meanwhile.  So the staging leaves the matrix waves: twelve waves, three per SIMD (168 registers each);
// waves 0 .. 7 are the eight 16-row tiles as before and their streams hold nothing but ds_read_b128, MFMA and the weight-fragment
// loads (+ the residual loads at the end of phase 1); waves 8 .. 11 (one per SIMD) only stage: global -> leaky-relu -> truncation
// split -> the three planes.  Same two half-buffers, same two barriers per item: during phase h the producers write half 1 - h from
// registers they LOADED one phase earlier (the data has long arrived: no wait in front of the LDS stores) and then issue the loads
// of the half after that in one burst — a CU's vector-memory path returns in issue order across waves, so the matrix waves' weight
// loads wait behind whatever the producers have in flight: one burst per phase costs that once, a trickle would cost it every step.
// Element by element the arithmetic is k_rb_conv's (k-group, tap, six products small terms first, the same epilogue): bit-identical.
// PRIO: the SIMD's arbiter serves the OLDER of two ready waves, so of a SIMD's two matrix waves (w and w + 4) the older one takes ~70 %
// of the matrix pipe while both have work, finishes its phase early and idles at the barrier, and the younger one runs the rest of
// its phase ALONE — at ~73 % of the pipe's rate (one wave's two accumulator chains and one-pair-ahead LDS reads do not fill it).
// With PRIO the younger wave runs the first ~3/4 of a phase's steps at s_setprio 1 (it leads), then drops to 0 (the older one
// leads): both reach the barrier together, and the pipe has two ready waves for the whole phase.
// (s_setprio takes an immediate and only waves 4 .. 7 want it: the test and the branch live inside ONE asm statement — as a C++ `if`
// they split the unrolled phase into basic blocks, and hipcc's wait-count pass answers that with vmcnt(0) drains and 92 bytes of spills)
#ifdef MI355_EMU
#define RBC_SETPRIO_YOUNG(mt, p) ((void)0)
#else
#define RBC_SETPRIO_YOUNG(mt, p) asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 1f\n\ts_setprio " #p "\n1:" ::"s"(mt) : "scc")
#endif
// CLK (lab build, MI355VITS_RBC_CLOCKS=1): shader-clock stamps of waves 0, 4 (the two matrix waves of one SIMD) and 8 (its producer)
// at the phase boundaries of every item, written by lane 0 to a.part ([workgroup][wave slot][item][8 stamps], low 32 bits)
template <int K, int DIL, int WD, int PRIO, bool CLK = false>  // PRIO: 0 off, 1 = 3/4 of a phase, 2 = every other step, 3 = every other tile pair
__global__ __launch_bounds__(768) void k_rb_conv_pw(ConvArgs a) {
    constexpr int NCT = 8;
    using GE = RbcGeo<K, DIL, NCT>;
    constexpr int G = RBC_G, N = GE::N, PAD = GE::PAD, LD = GE::LD, LDP = GE::LDP, S = G * K, SH = S / 2;
    constexpr int HALF = GE::HALF, NP = NCT / 2, WR = RBC_WR;
    constexpr int RP = (HALF + 255) / 256;  // staging rounds of a half-buffer for the 256 producer threads
    constexpr unsigned REC16 = GE::REC16, PS16 = GE::PS16;
    static_assert(S % WR == 0 && WD >= 1 && WD < WR && GE::LDS <= RBC_LDS_LIMIT, "shape");
    DYN_SMEM(float, smem);
    char* L0 = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = WAVE_UNIFORM(tid >> 6);
    // work items = the (row, column block) pairs that HAVE work, a.nvalid of them, numbered row by row (ragged batches, round 6:
    // kernels_mrfp.cpp has the argument): row b holds ceil(len_b / N); a cursor turns a valid index into (row, block) — the
    // indices a workgroup decodes only grow
    const int nitems = a.nvalid;
    const unsigned xrow = 4u * (unsigned)a.x_ld, yrow = 4u * (unsigned)a.y_ld, rrow = 4u * (unsigned)a.res_ld;

    struct Item { int b, t0, len, last; };
    auto row_len = [&](int b) MI355_INLINE_LAMBDA {
#ifdef MI355_EMU
        return a.in_len[b];
#else
        typedef const int __attribute__((address_space(4))) * cptr_t;
        return ((cptr_t)(a.in_len))[b];
#endif
    };
    struct Cursor { int r, base, nv; };
    auto row_nv = [&](int b) MI355_INLINE_LAMBDA {
        int len = row_len(b);
        if (len > a.T) len = a.T;
        return len > 0 ? (len + N - 1) / N : 0;
    };
    auto decode = [&](int it, Cursor& c) MI355_INLINE_LAMBDA {
        Item o;
        it = rbc_item(it, nitems, a.item_order);
        while (it >= c.base + c.nv && c.r + 1 < a.B) {  // (it < a.nvalid: ends inside the batch; the row bound guards a stale count)
            c.base += c.nv;
            ++c.r;
            c.nv = row_nv(c.r);
        }
        c.r = WAVE_UNIFORM(c.r);
        c.base = WAVE_UNIFORM(c.base);
        c.nv = WAVE_UNIFORM(c.nv);
        o.b = c.r;
        o.t0 = WAVE_UNIFORM((it - c.base) * N);
        int len = row_len(o.b);
        if (len > a.T) len = a.T;
        o.len = len;
        o.last = o.len > 0 ? o.len - 1 : 0;
        return o;
    };
    const int it_first = (int)blockIdx.x;
    if (it_first >= nitems) return;
    Cursor cur;
    cur.r = 0;
    cur.base = 0;
    cur.nv = row_nv(0);
    [[maybe_unused]] int clk_item = 0;
    auto stamp = [&](int slot, int k) MI355_INLINE_LAMBDA {
#ifndef MI355_EMU
        if constexpr (CLK) {
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            const BufRsrc dbg = buf_rsrc(a.part);
            const unsigned o = (lane == 0 && slot >= 0 && clk_item < 8) ? 4u * (unsigned)((((int)blockIdx.x * 3 + slot) * 8 + clk_item) * 8 + k) : BUF_OOB;
            buf_store_f32(dbg, o, 0u, __uint_as_float(t));
        }
#endif
    };

    if (wv >= 8) {
        // ================================================================ producer waves: (record, column) pairs pt + 256 round
        const int cslot = wv == 8 ? 2 : -1;
        const int pt = tid - 512;
        auto p_load = [&](const Item& im, int half, int round, float (&sv)[8]) MI355_INLINE_LAMBDA {
            const BufRsrc xb = buf_rsrc(a.x + (long)im.b * a.x_bs);
            int t2 = pt;
            OPAQUE_V(t2);
            int idx = t2 + 256 * round;
            if (256 * (round + 1) > HALF) idx = idx < HALF ? idx : HALF - 1;  // past the end: the last pair again (same bytes, same slot)
            const int rec = idx / LD, col = idx - rec * LD;
            const int t = im.t0 - PAD + col;
            const int tc = t < 0 ? 0 : (t > im.last ? im.last : t);
            const unsigned o = 4u * (unsigned)(8 * (8 * half + rec) * a.x_ld + tc);
            MI355_UNROLL
            for (int e = 0; e < 8; ++e) sv[e] = buf_load_f32(xb, o, (unsigned)e * xrow);
        };
        auto p_store = [&](const Item& im, int half, int round, const float (&sv)[8]) MI355_INLINE_LAMBDA {
            int t2 = pt;
            OPAQUE_V(t2);
            int idx = t2 + 256 * round;
            if (256 * (round + 1) > HALF) idx = idx < HALF ? idx : HALF - 1;
            const int rec = idx / LD, col = idx - rec * LD;
            const int t = im.t0 - PAD + col;
            const bool in = t >= 0 && t < im.len;
            uint4 ph[3];
            unsigned h, m, l;
            split3_sc(in ? lrelu_f(sv[0], a.in_slope) : 0.0f, in ? lrelu_f(sv[1], a.in_slope) : 0.0f, h, m, l);
            ph[0].x = h; ph[1].x = m; ph[2].x = l;
            split3_sc(in ? lrelu_f(sv[2], a.in_slope) : 0.0f, in ? lrelu_f(sv[3], a.in_slope) : 0.0f, h, m, l);
            ph[0].y = h; ph[1].y = m; ph[2].y = l;
            split3_sc(in ? lrelu_f(sv[4], a.in_slope) : 0.0f, in ? lrelu_f(sv[5], a.in_slope) : 0.0f, h, m, l);
            ph[0].z = h; ph[1].z = m; ph[2].z = l;
            split3_sc(in ? lrelu_f(sv[6], a.in_slope) : 0.0f, in ? lrelu_f(sv[7], a.in_slope) : 0.0f, h, m, l);
            ph[0].w = h; ph[1].w = m; ph[2].w = l;
            char* px = L0 + (unsigned)(8 * half + rec) * REC16 + 16u * (unsigned)col;
            MI355_UNROLL
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(px + (unsigned)p * PS16) = ph[p];
        };
        float sv[RP][8];
        {
            const Item im = decode(it_first, cur);
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_load(im, 0, r, sv[r]);
            SCHED_FENCE();
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_store(im, 0, r, sv[r]);
            SCHED_FENCE();
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_load(im, 1, r, sv[r]);
        }
        __syncthreads();
        MI355_NOUNROLL
        for (int it = it_first, itv = it_first; it < nitems; it = itv) {
            itv = it + (int)gridDim.x;
            const Item im = decode(it, cur);
            const int itn = itv < nitems ? itv : it;  // (no next item: this one again, unread)
            const Item imn = decode(itn, cur);
            // phase 0 (the matrix waves read half 0): this item's half 1 from the registers, then the loads of the next item's half 0
            stamp(cslot, 0);
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_store(im, 1, r, sv[r]);
            SCHED_FENCE();
            stamp(cslot, 1);
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_load(imn, 0, r, sv[r]);
            stamp(cslot, 2);
            __syncthreads();
            stamp(cslot, 3);
            // phase 1 (they read half 1): the next item's half 0, then the loads of its half 1
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_store(imn, 0, r, sv[r]);
            SCHED_FENCE();
            stamp(cslot, 4);
            MI355_UNROLL
            for (int r = 0; r < RP; ++r) p_load(imn, 1, r, sv[r]);
            stamp(cslot, 5);
            __syncthreads();
            stamp(cslot, 6);
            ++clk_item;
        }
        return;
    }

    // ==================================================================== matrix waves: wave = 16-row tile of output channels
    const int mt = wv;
    const int cslot = mt == 0 ? 0 : (mt == 4 ? 1 : -1);
    const int q = lane >> 4, n = lane & 15;
    const int co0 = 32 * (mt >> 1) + 8 * q + 4 * (mt & 1);
    const BufRsrc wbuf = buf_rsrc(a.w);
    const unsigned wl = 16u * (unsigned)lane;
    const unsigned wmt = (unsigned)mt * (unsigned)(K * G * 3 * 64 * 16);
    unsigned lq[3];
    MI355_UNROLL
    for (int p = 0; p < 3; ++p) {
        lq[p] = (unsigned)p * PS16 + (unsigned)(q * LDP + n) * 16u;
        OPAQUE_V(lq[p]);
    }
    auto w_load = [&](int s, uint4 (&w)[3]) MI355_INLINE_LAMBDA {
        const int g = s / K, k = s % K;
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) w[p] = buf_load_u4(wbuf, wl, wmt + (unsigned)(((k * G + g) * 3 + p) * 64 * 16));
    };
    auto b_read = [&](int s, int j, uint4 (&bf)[3]) MI355_INLINE_LAMBDA {
        const int g = s / K, k = s % K;
        MI355_UNROLL
        for (int p = 0; p < 3; ++p)
            bf[p] = *reinterpret_cast<const uint4*>(L0 + lq[p] + (unsigned)((4 * g) * LDP + 16 * j + k * DIL) * 16u);
    };
    float bia[4];
    MI355_UNROLL
    for (int r = 0; r < 4; ++r) bia[r] = a.bias ? a.bias[co0 + r] : 0.0f;
    uint4 Wr[WR][3];
    MI355_UNROLL
    for (int s = 0; s < WD; ++s) w_load(s, Wr[s]);
    __syncthreads();

    MI355_NOUNROLL
    for (int it = it_first, itv = it_first; it < nitems; it = itv) {
        itv = it + (int)gridDim.x;
        const Item im = decode(it, cur);
        const BufRsrc ybuf = buf_rsrc(a.y + (long)im.b * a.y_bs);
        const BufRsrc rbuf = buf_rsrc(a.res + (long)im.b * a.res_bs);
        f32x4 acc[NCT];
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.0f;
        float rq[NCT][4];
        auto load_res = [&](int j) MI355_INLINE_LAMBDA {
            const int t = im.t0 + 16 * j + n;
            const int tc = t < a.T ? t : a.T - 1;
            const unsigned o = 4u * (unsigned)(co0 * a.res_ld + tc);
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) rq[j][r] = buf_load_f32(rbuf, o, (unsigned)r * rrow);
        };
        stamp(cslot, 0);
        MI355_UNROLL
        for (int h = 0; h < 2; ++h) {
            uint4 Bf[2][2][3];
            b_read(h * SH, 0, Bf[0][0]);
            b_read(h * SH, 1, Bf[0][1]);
            if (PRIO == 1) RBC_SETPRIO_YOUNG(mt, 1);
            MI355_UNROLL
            for (int sl = 0; sl < SH; ++sl) {
                const int s = h * SH + sl;
                if (PRIO == 1 && sl == (3 * SH + 2) / 4) RBC_SETPRIO_YOUNG(mt, 0);
                if (PRIO == 2) {
                    if (sl & 1) RBC_SETPRIO_YOUNG(mt, 0);
                    else RBC_SETPRIO_YOUNG(mt, 1);
                }
                w_load((s + WD) % S, Wr[(s + WD) % WR]);
                if (h == 1 && sl >= SH - 2) {  // the residual tiles, four per step in the phase's last two steps
                    MI355_UNROLL
                    for (int j = 0; j < 4; ++j) load_res(4 * (sl - (SH - 2)) + j);
                }
                SCHED_FENCE();
                MI355_UNROLL
                for (int jp = 0; jp < NP; ++jp) {
                    const int cur = jp & 1;
                    if (PRIO == 3) {
                        if (jp & 1) RBC_SETPRIO_YOUNG(mt, 0);
                        else RBC_SETPRIO_YOUNG(mt, 1);
                    }
                    if (jp + 1 < NP) {
                        b_read(s, 2 * jp + 2, Bf[cur ^ 1][0]);
                        b_read(s, 2 * jp + 3, Bf[cur ^ 1][1]);
                    } else if (sl + 1 < SH) {
                        b_read(s + 1, 0, Bf[cur ^ 1][0]);
                        b_read(s + 1, 1, Bf[cur ^ 1][1]);
                    }
                    SCHED_FENCE();
                    {
                        const uint4(&W)[3] = Wr[s % WR];
                        f32x4 c0 = acc[2 * jp], c1 = acc[2 * jp + 1];
                        const uint4(&B0)[3] = Bf[cur][0];
                        const uint4(&B1)[3] = Bf[cur][1];
                        c0 = MFMA_16x16x32_BF16(W[2], B0[0], c0);  // small terms first
                        c1 = MFMA_16x16x32_BF16(W[2], B1[0], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[2], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[2], c1);
                        c0 = MFMA_16x16x32_BF16(W[1], B0[1], c0);
                        c1 = MFMA_16x16x32_BF16(W[1], B1[1], c1);
                        c0 = MFMA_16x16x32_BF16(W[1], B0[0], c0);
                        c1 = MFMA_16x16x32_BF16(W[1], B1[0], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[1], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[1], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[0], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[0], c1);
                        acc[2 * jp] = c0;
                        acc[2 * jp + 1] = c1;
                    }
                    SCHED_FENCE();
                }
            }
            if (PRIO >= 2) RBC_SETPRIO_YOUNG(mt, 0);
            stamp(cslot, 1 + 2 * h);
            __syncthreads();
            stamp(cslot, 2 + 2 * h);
        }
        // ---- epilogue (k_rb_conv's, operation by operation)
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) acc[j][r] = (rq[j][r] + (acc[j][r] + bia[r])) * a.out_scale;
        if (a.accumulate) {
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j) {
                const int t = im.t0 + 16 * j + n;
                const int tc = t < a.T ? t : a.T - 1;
                const unsigned oc = 4u * (unsigned)(co0 * a.y_ld + tc);
                MI355_UNROLL
                for (int r = 0; r < 4; ++r) rq[j][r] = buf_load_f32(ybuf, oc, (unsigned)r * yrow);
            }
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j)
                MI355_UNROLL
                for (int r = 0; r < 4; ++r) acc[j][r] += rq[j][r];
        }
        MI355_UNROLL
        for (int j = 0; j < NCT; ++j) {
            const int t = im.t0 + 16 * j + n;
            const unsigned o = t < a.T ? 4u * (unsigned)(co0 * a.y_ld + t) : BUF_OOB;
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) buf_store_f32(ybuf, o, (unsigned)r * yrow, acc[j][r]);
        }
        stamp(cslot, 5);
        ++clk_item;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The polyphase upsamplers 256 -> 128 and 128 -> 64 (x 8, k = 16) in the same form.  ConvTranspose1d(k = 2 s, stride s)
// of lrelu(x) = a two-tap stride-1 conv with s * Cout "phase rows" (kernels.h: shuf_*; row m = c s + phase): output position i,
// row m = w'[m, :, 0] . x[:, i - 1] + w'[m, :, 1] . x[:, i], landing at y[c][i s + phase - s / 2].  A work item is (row, N output
// positions) with all CIN input channels resident as planes (N + 1 columns; 128 channels: N = 128, 256 channels: N = 64); the
// phase rows run in blocks of 128 (8 waves x one 16-row tile; four / eight blocks per item), every block through the two
// half-buffers' k-groups.  The halves are refilled as in k_rb_conv, except that a half's loads are issued one phase before its
// stores (a phase is only four or eight steps here): half 1 of an item is stored during its first phase (loaded during the
// previous item's last one), half 0 of the NEXT item during its last phase (loaded during the one before), with a barrier in
// front of that phase as well — every wave must be done with the item's last pass over half 0.  Weights in natural row order
// (pack_conv_weights_p16n), so a wave's 16 rows are two channels x 8 phases: lanes q = 0, 1 (r = 0..3) write adjacent 16 bytes and a
// store instruction covers whole 512-byte runs of y.  (The 64 -> 32 upsampler, x 4, has its own kernel below: k_ups64.)
template <int I, int NN, typename F>
__device__ __forceinline__ void rbc_static_for(F&& f) {
    if constexpr (I < NN) {
        f(std::integral_constant<int, I>{});
        rbc_static_for<I + 1, NN>(f);
    }
}
// staging rounds of a phase of SH steps: round r belongs to step r SH / ROUNDS; first / one-past-last round of step sl
constexpr int rbc_round_lo(int sl, int SH, int ROUNDS) {
    int r = 0;
    while (r < ROUNDS && r * SH / ROUNDS < sl) ++r;
    return r;
}
constexpr int rbc_round_hi(int sl, int SH, int ROUNDS) {
    int r = 0;
    while (r < ROUNDS && r * SH / ROUNDS <= sl) ++r;
    return r;
}

template <int CIN, int STR, int NCT>
struct UpsGeo {
    static constexpr int G = CIN / 32, REC = CIN / 8, N = 16 * NCT, LD = N + 1, LDP = (LD + 15) & ~15;
    static constexpr int M = STR * (CIN / 2), RB = M / 128;
    static constexpr unsigned REC16 = 16u * LDP, PS16 = REC * REC16;
    static constexpr int HALF = (REC / 2) * LD, ROUNDS = (HALF + 511) / 512;
    static constexpr size_t LDS = 3 * (size_t)PS16;
};

template <int CIN, int STR, int NCT>
__global__ __launch_bounds__(512) void k_ups_pl(ConvArgs a) {
    using GE = UpsGeo<CIN, STR, NCT>;
    constexpr int G = GE::G, REC = GE::REC, N = GE::N, LD = GE::LD, LDP = GE::LDP, RB = GE::RB, K = 2;
    constexpr int S = G * K, SH = S / 2, NPH = 2 * RB, ROUNDS = GE::ROUNDS, HALF = GE::HALF, NP = NCT / 2, WR = RBC_WR;
    constexpr unsigned REC16 = GE::REC16, PS16 = GE::PS16;
    static_assert(NCT % 2 == 0 && (RB * S) % WR == 0 && G % 2 == 0 && GE::M % 128 == 0 && (STR == 4 || STR == 8) && GE::LDS <= RBC_LDS_LIMIT, "shape");
    DYN_SMEM(float, smem);
    char* L0 = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, mt = WAVE_UNIFORM(tid >> 6);
    const int q = lane >> 4, n = lane & 15;
    const BufRsrc wbuf = buf_rsrc(a.w);
    const unsigned wl = 16u * (unsigned)lane;
    unsigned lq[3];
    MI355_UNROLL
    for (int p = 0; p < 3; ++p) {
        lq[p] = (unsigned)p * PS16 + (unsigned)(q * LDP + n) * 16u;
        OPAQUE_V(lq[p]);
    }
    // work items = the (row, block of N output positions) pairs that have work: row b holds ceil((len_b + 1) / N) of them (a.T = Tin + 1
    // positions for a full row); a.nvalid of them, numbered row by row, decoded with a forward-only cursor (ragged batches: k_rb_conv)
    const int nitems = a.nvalid;
    const unsigned xrow = 4u * (unsigned)a.x_ld;

    struct Item { int b, t0, len, last; };
    auto row_len = [&](int b) MI355_INLINE_LAMBDA {
#ifdef MI355_EMU
        return a.in_len[b];
#else
        typedef const int __attribute__((address_space(4))) * cptr_t;
        return ((cptr_t)(a.in_len))[b];
#endif
    };
    struct Cursor { int r, base, nv; };
    auto row_nv = [&](int b) MI355_INLINE_LAMBDA {
        int len = row_len(b);
        if (len > a.Tin) len = a.Tin;
        return len > 0 ? (len + 1 + N - 1) / N : 0;
    };
    auto decode = [&](int it, Cursor& c) MI355_INLINE_LAMBDA {
        Item o;
        it = rbc_item(it, nitems, a.item_order);
        while (it >= c.base + c.nv && c.r + 1 < a.B) {
            c.base += c.nv;
            ++c.r;
            c.nv = row_nv(c.r);
        }
        c.r = WAVE_UNIFORM(c.r);
        c.base = WAVE_UNIFORM(c.base);
        c.nv = WAVE_UNIFORM(c.nv);
        o.b = c.r;
        o.t0 = WAVE_UNIFORM((it - c.base) * N);
        int len = row_len(o.b);
        if (len > a.Tin) len = a.Tin;
        o.len = len;
        o.last = o.len > 0 ? o.len - 1 : 0;
        return o;
    };
    auto stage_load = [&](const Item& im, const BufRsrc& xb, int half, int round, float (&sv)[8]) MI355_INLINE_LAMBDA {
        int t2 = tid;
        OPAQUE_V(t2);
        int idx = t2 + 512 * round;
        if (512 * (round + 1) > HALF) idx = idx < HALF ? idx : HALF - 1;
        const int rec = idx / LD, col = idx - rec * LD;
        const int t = im.t0 - 1 + col;
        const int tc = t < 0 ? 0 : (t > im.last ? im.last : t);
        const unsigned o = 4u * (unsigned)(8 * ((REC / 2) * half + rec) * a.x_ld + tc);
        MI355_UNROLL
        for (int e = 0; e < 8; ++e) sv[e] = buf_load_f32(xb, o, (unsigned)e * xrow);
    };
    struct StagePos { unsigned addr; bool in; };
    auto stage_pos = [&](const Item& im, int half, int round) MI355_INLINE_LAMBDA {
        int t2 = tid;
        OPAQUE_V(t2);
        int idx = t2 + 512 * round;
        if (512 * (round + 1) > HALF) idx = idx < HALF ? idx : HALF - 1;
        const int rec = idx / LD, col = idx - rec * LD;
        const int t = im.t0 - 1 + col;
        StagePos sp;
        sp.in = t >= 0 && t < im.len;
        sp.addr = (unsigned)((REC / 2) * half + rec) * REC16 + 16u * (unsigned)col;
        return sp;
    };
    auto stage_piece = [&](int piece, const float (&sv)[8], uint4 (&ph)[3], const StagePos& sp) MI355_INLINE_LAMBDA {
        const float v0 = sp.in ? lrelu_f(sv[2 * piece], a.in_slope) : 0.0f, v1 = sp.in ? lrelu_f(sv[2 * piece + 1], a.in_slope) : 0.0f;
        unsigned h, m, l;
        split3_sc(v0, v1, h, m, l);
        if (piece == 0) { ph[0].x = h; ph[1].x = m; ph[2].x = l; }
        if (piece == 1) { ph[0].y = h; ph[1].y = m; ph[2].y = l; }
        if (piece == 2) { ph[0].z = h; ph[1].z = m; ph[2].z = l; }
        if (piece == 3) {
            ph[0].w = h; ph[1].w = m; ph[2].w = l;
            char* px = L0 + sp.addr;
            MI355_UNROLL
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(px + (unsigned)p * PS16) = ph[p];
        }
    };
    // step u of an item = (row block u / S, k-group (u % S) / K, tap u % K)
    auto w_load = [&](int u, uint4 (&w)[3]) MI355_INLINE_LAMBDA {
        const int rb = u / S, g = (u % S) / K, k = u % K;
        MI355_UNROLL
        for (int p = 0; p < 3; ++p)
            w[p] = buf_load_u4(wbuf, wl, (unsigned)(rb * 8 * K * G * 3 * 1024) + (unsigned)mt * (unsigned)(K * G * 3 * 1024) + (unsigned)(((k * G + g) * 3 + p) * 1024));
    };
    auto b_read = [&](int u, int j, uint4 (&bf)[3]) MI355_INLINE_LAMBDA {
        const int g = (u % S) / K, k = u % K;
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const uint4*>(L0 + lq[p] + (unsigned)((4 * g) * LDP + 16 * j + k) * 16u);
    };

    if ((int)blockIdx.x >= nitems) return;
    Cursor cur;
    cur.r = 0;
    cur.base = 0;
    cur.nv = row_nv(0);
    float sv[ROUNDS][8];  // the half being staged: loaded one phase ahead of its stores
    {
        // prologue: half 0 of the first item (loaded and stored), the loads of its half 1, the first steps' weight fragments
        const Item im = decode(blockIdx.x, cur);
        const BufRsrc xb = buf_rsrc(a.x + (long)im.b * a.x_bs);
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) stage_load(im, xb, 0, r, sv[r]);
        SCHED_FENCE();
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) {
            uint4 ph[3];
            const StagePos sp = stage_pos(im, 0, r);
            MI355_UNROLL
            for (int pc = 0; pc < 4; ++pc) stage_piece(pc, sv[r], ph, sp);
        }
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) stage_load(im, xb, 1, r, sv[r]);
    }
    uint4 Wr[WR][3];
    MI355_UNROLL
    for (int u = 0; u < WR - 1; ++u) w_load(u, Wr[u]);
    __syncthreads();

    MI355_NOUNROLL
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const Item im = decode(it, cur);
        const int itn = it + (int)gridDim.x < nitems ? it + (int)gridDim.x : it;
        const Item imn = decode(itn, cur);
        const BufRsrc xbn = buf_rsrc(a.x + (long)imn.b * a.x_bs);
        const BufRsrc ybuf = buf_rsrc(a.y + (long)im.b * a.y_bs);
        f32x4 acc[NCT];
        rbc_static_for<0, NPH>([&](auto PHI) MI355_INLINE_LAMBDA {
            constexpr int phi = decltype(PHI)::value;
            constexpr int rb = phi / 2, h = phi & 1;
            if constexpr (h == 0) {
                MI355_UNROLL
                for (int j = 0; j < NCT; ++j)
                    MI355_UNROLL
                    for (int r = 0; r < 4; ++r) acc[j][r] = 0.0f;
            }
            // what this phase stores (loaded one phase earlier) and what it loads (stored one phase later)
            constexpr bool st1 = phi == 0, st0 = phi == NPH - 1;          // half 1 of this item | half 0 of the next
            constexpr bool ld0 = phi == NPH - 2, ld1 = phi == NPH - 1;    // half 0 of the next item | half 1 of the next
            uint4 ph[3];
            uint4 Bf[2][2][3];
            b_read(phi * SH, 0, Bf[0][0]);
            b_read(phi * SH, 1, Bf[0][1]);
            rbc_static_for<0, SH>([&](auto SL) MI355_INLINE_LAMBDA {
                constexpr int sl = decltype(SL)::value;
                constexpr int u = phi * SH + sl;
                constexpr int r0 = rbc_round_lo(sl, SH, ROUNDS), r1 = rbc_round_hi(sl, SH, ROUNDS), nr = r1 - r0;  // this step's rounds
                static_assert(nr <= NP, "a round needs a tile pair of its own");
                w_load((u + WR - 1) % (RB * S), Wr[(u + WR - 1) % WR]);
                if constexpr (ld0 && !st1) {  // (one row block: the loads follow their registers' stores, below)
                    rbc_static_for<r0, r1>([&](auto R) MI355_INLINE_LAMBDA { stage_load(imn, xbn, 0, decltype(R)::value, sv[decltype(R)::value]); });
                }
                SCHED_FENCE();
                rbc_static_for<0, NP>([&](auto JP) MI355_INLINE_LAMBDA {
                    constexpr int jp = decltype(JP)::value;
                    constexpr int cur = (NP % 2 == 0) ? (jp & 1) : ((u * NP + jp) & 1);
                    if constexpr (jp + 1 < NP) {
                        b_read(u, 2 * jp + 2, Bf[cur ^ 1][0]);
                        b_read(u, 2 * jp + 3, Bf[cur ^ 1][1]);
                    } else if constexpr (sl + 1 < SH) {
                        b_read(u + 1, 0, Bf[cur ^ 1][0]);
                        b_read(u + 1, 1, Bf[cur ^ 1][1]);
                    }
                    SCHED_FENCE();
                    {
                        const uint4(&W)[3] = Wr[u % WR];
                        f32x4 c0 = acc[2 * jp], c1 = acc[2 * jp + 1];
                        const uint4(&B0)[3] = Bf[cur][0];
                        const uint4(&B1)[3] = Bf[cur][1];
                        c0 = MFMA_16x16x32_BF16(W[2], B0[0], c0);  // small terms first
                        c1 = MFMA_16x16x32_BF16(W[2], B1[0], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[2], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[2], c1);
                        c0 = MFMA_16x16x32_BF16(W[1], B0[1], c0);
                        c1 = MFMA_16x16x32_BF16(W[1], B1[1], c1);
                        c0 = MFMA_16x16x32_BF16(W[1], B0[0], c0);
                        c1 = MFMA_16x16x32_BF16(W[1], B1[0], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[1], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[1], c1);
                        c0 = MFMA_16x16x32_BF16(W[0], B0[0], c0);
                        c1 = MFMA_16x16x32_BF16(W[0], B1[0], c1);
                        acc[2 * jp] = c0;
                        acc[2 * jp + 1] = c1;
                    }
                    if constexpr ((st1 || st0) && nr > 0) {
                        // this step's rounds share its tile pairs: round r0 + x on pairs [x NP / nr, (x + 1) NP / nr), four pieces each
                        constexpr int x = jp * nr / NP, p0 = (x * NP + nr - 1) / nr, p1 = ((x + 1) * NP + nr - 1) / nr, np = p1 - p0;
                        constexpr int r = r0 + x;
                        const StagePos sp = stage_pos(st1 ? im : imn, st1 ? 1 : 0, r);
                        rbc_static_for<0, 4>([&](auto PC) MI355_INLINE_LAMBDA {
                            constexpr int pc = decltype(PC)::value;
                            if constexpr (pc * np / 4 == jp - p0) stage_piece(pc, sv[r], ph, sp);
                        });
                        MI355_UNROLL
                        for (int xx = 0; xx < 12; ++xx) {
                            SCHED_GROUP(0x8, 1);
                            SCHED_GROUP(0x2, 3);
                        }
                    }
                    SCHED_FENCE();
                });
                // the registers of the rounds stored in this step take the next half's loads
                if constexpr (ld1 || (ld0 && st1)) {
                    rbc_static_for<r0, r1>([&](auto R) MI355_INLINE_LAMBDA { stage_load(imn, xbn, ld1 ? 1 : 0, decltype(R)::value, sv[decltype(R)::value]); });
                    SCHED_FENCE();
                }
            });
            // barriers: behind a phase that stored a half (its readers come next), and in front of the phase that overwrites half 0
            // with the next item's columns (every wave must be done with this item's last pass over half 0)
            if constexpr (st1 || st0 || phi == NPH - 2) __syncthreads();
            if constexpr (h == 1) {
                // ---- epilogue of row block rb: bias, then the phases of a channel side by side
                const int m0 = 128 * rb + 16 * mt + 4 * q;  // this lane's four phase rows
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + m0);
                const float bia[4] = {bv.x, bv.y, bv.z, bv.w};
                MI355_UNROLL
                for (int j = 0; j < NCT; ++j) {
                    const int i = im.t0 + 16 * j + n;
                    float v[4];
                    MI355_UNROLL
                    for (int r = 0; r < 4; ++r) v[r] = acc[j][r] + bia[r];
                    if (STR == 8) {
                        const int c = m0 >> 3, n0 = 8 * i + (m0 & 7) - 4;
                        const unsigned o = (i < a.T && n0 >= 0 && n0 + 3 < a.shuf_T) ? 4u * (unsigned)(c * a.y_ld + n0) : BUF_OOB;
                        buf_store_f4(ybuf, o, 0u, v[0], v[1], v[2], v[3]);
                    } else {
                        const int c = m0 >> 2, n0 = 4 * i - 2;
                        const unsigned o0 = (i < a.T && i >= 1) ? 4u * (unsigned)(c * a.y_ld + n0) : BUF_OOB;
                        const unsigned o1 = (i < a.T && n0 + 3 < a.shuf_T) ? 4u * (unsigned)(c * a.y_ld + n0 + 2) : BUF_OOB;
                        buf_store_f2(ybuf, o0, 0u, v[0], v[1]);
                        buf_store_f2(ybuf, o1, 0u, v[2], v[3]);
                    }
                }
            }
        });
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The 64 -> 32 upsampler (x 4, k = 8): 1.2 GB of HBM traffic per launch at the bench shape against 0.15 ms of matrix work — a
// memory-bound kernel, and in the half-buffer form above it ran no faster than the staged kernel (0.39 ms): every workgroup ends
// its item with 32 stores per lane, all CUs at once, and the loads of the next phase sit behind them in the in-order memory
// counter.  What this shape offers instead: its four steps' weight fragments are 48 registers — RESIDENT, no weight stream —, so
// the loop can be tile-major: a tile pair runs all four steps and is stored at once while the next pair computes (stores spread
// evenly over the item), and with only 384 bytes per column the input fits twice: item i + 1's planes are written to the other
// buffer while item i computes (loads issued one whole item earlier still: two items ahead of their use), one barrier per item.
// Items are 127 output positions = 128 input columns: exactly two staging rounds for the 512 threads.
template <int NCT>
struct Ups64Geo {
    static constexpr int CIN = 64, G = 2, REC = 8, STR = 4, K = 2, S = G * K;
    static constexpr int NPOS = 16 * NCT - 1, LD = 16 * NCT, LDP = LD;        // positions per item; staged columns = pitch
    static constexpr unsigned REC16 = 16u * LDP, PS16 = REC * REC16, BUF = 3 * PS16;
    static constexpr int FULL = REC * LD, ROUNDS = (FULL + 511) / 512;
    static constexpr size_t LDS = 2 * (size_t)BUF;
};

template <int NCT, bool ST16>
__global__ __launch_bounds__(512) void k_ups64(ConvArgs a) {
    using GE = Ups64Geo<NCT>;
    constexpr int G = GE::G, K = GE::K, S = GE::S, NPOS = GE::NPOS, LD = GE::LD, LDP = GE::LDP, ROUNDS = GE::ROUNDS, FULL = GE::FULL, NP = NCT / 2;
    constexpr unsigned REC16 = GE::REC16, PS16 = GE::PS16, BUF = GE::BUF;
    constexpr int NSLOT = NP * S;  // (tile pair, step) slots of an item
    static_assert(NCT % 2 == 0 && GE::LDS <= RBC_LDS_LIMIT && ROUNDS * 4 <= NSLOT, "shape");
    DYN_SMEM(float, smem);
    char* L0 = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, mt = WAVE_UNIFORM(tid >> 6);
    const int q = lane >> 4, n = lane & 15;
    const BufRsrc wbuf = buf_rsrc(a.w);
    const int nitems = a.nvalid;  // (row, block of NPOS positions) pairs with work, numbered row by row (k_ups_pl has the argument)
    const unsigned xrow = 4u * (unsigned)a.x_ld;
    if ((int)blockIdx.x >= nitems) return;

    struct Item { int b, t0, len, last; };
    auto row_len = [&](int b) MI355_INLINE_LAMBDA {
#ifdef MI355_EMU
        return a.in_len[b];
#else
        typedef const int __attribute__((address_space(4))) * cptr_t;
        return ((cptr_t)(a.in_len))[b];
#endif
    };
    struct Cursor { int r, base, nv; };
    auto row_nv = [&](int b) MI355_INLINE_LAMBDA {
        int len = row_len(b);
        if (len > a.Tin) len = a.Tin;
        return len > 0 ? (len + 1 + NPOS - 1) / NPOS : 0;
    };
    auto decode = [&](int it, Cursor& c) MI355_INLINE_LAMBDA {
        Item o;
        it = rbc_item(it, nitems, a.item_order);
        while (it >= c.base + c.nv && c.r + 1 < a.B) {
            c.base += c.nv;
            ++c.r;
            c.nv = row_nv(c.r);
        }
        c.r = WAVE_UNIFORM(c.r);
        c.base = WAVE_UNIFORM(c.base);
        c.nv = WAVE_UNIFORM(c.nv);
        o.b = c.r;
        o.t0 = WAVE_UNIFORM((it - c.base) * NPOS);
        int len = row_len(o.b);
        if (len > a.Tin) len = a.Tin;
        o.len = len;
        o.last = o.len > 0 ? o.len - 1 : 0;
        return o;
    };
    auto stage_load = [&](const Item& im, int round, float (&sv)[8]) MI355_INLINE_LAMBDA {
        const BufRsrc xb = buf_rsrc(a.x + (long)im.b * a.x_bs);
        int t2 = tid;
        OPAQUE_V(t2);
        int idx = t2 + 512 * round;
        if (512 * (round + 1) > FULL) idx = idx < FULL ? idx : FULL - 1;
        const int rec = idx / LD, col = idx - rec * LD;
        const int t = im.t0 - 1 + col;
        const int tc = t < 0 ? 0 : (t > im.last ? im.last : t);
        const unsigned o = 4u * (unsigned)(8 * rec * a.x_ld + tc);
        MI355_UNROLL
        for (int e = 0; e < 8; ++e) sv[e] = buf_load_f32(xb, o, (unsigned)e * xrow);
    };
    struct StagePos { unsigned addr; bool in; };
    auto stage_pos = [&](const Item& im, unsigned buf, int round) MI355_INLINE_LAMBDA {
        int t2 = tid;
        OPAQUE_V(t2);
        int idx = t2 + 512 * round;
        if (512 * (round + 1) > FULL) idx = idx < FULL ? idx : FULL - 1;
        const int rec = idx / LD, col = idx - rec * LD;
        const int t = im.t0 - 1 + col;
        StagePos sp;
        sp.in = t >= 0 && t < im.len;
        sp.addr = buf + (unsigned)rec * REC16 + 16u * (unsigned)col;
        return sp;
    };
    auto stage_piece = [&](int piece, const float (&sv)[8], uint4 (&ph)[3], const StagePos& sp) MI355_INLINE_LAMBDA {
        const float v0 = sp.in ? lrelu_f(sv[2 * piece], a.in_slope) : 0.0f, v1 = sp.in ? lrelu_f(sv[2 * piece + 1], a.in_slope) : 0.0f;
        unsigned h, m, l;
        split3_sc(v0, v1, h, m, l);
        if (piece == 0) { ph[0].x = h; ph[1].x = m; ph[2].x = l; }
        if (piece == 1) { ph[0].y = h; ph[1].y = m; ph[2].y = l; }
        if (piece == 2) { ph[0].z = h; ph[1].z = m; ph[2].z = l; }
        if (piece == 3) {
            ph[0].w = h; ph[1].w = m; ph[2].w = l;
            char* px = L0 + sp.addr;
            MI355_UNROLL
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(px + (unsigned)p * PS16) = ph[p];
        }
    };

    // the four steps' fragments of this wave's row tile: resident (step u = k-group u / K, tap u % K)
    uint4 Wr[S][3];
    MI355_UNROLL
    for (int u = 0; u < S; ++u)
        MI355_UNROLL
        for (int p = 0; p < 3; ++p)
            Wr[u][p] = buf_load_u4(wbuf, 16u * (unsigned)lane, (unsigned)mt * (unsigned)(K * G * 3 * 1024) + (unsigned)((((u % K) * G + u / K) * 3 + p) * 1024));
    const int m0 = 16 * mt + 4 * q;  // this lane's channel (m0 >> 2), phases 0..3
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + m0);
    const float bia[4] = {bv.x, bv.y, bv.z, bv.w};

    // prologue: the first item's planes into buffer 0, the second item's loads in flight
    float sv[ROUNDS][8];
    Cursor csr;
    csr.r = 0;
    csr.base = 0;
    csr.nv = row_nv(0);
    // the item being computed, the one staged into the other buffer and the one whose loads are in flight: decoded ONCE each, in the
    // order the workgroup meets them (the cursor only moves forward); past the last item: the previous one again, unread
    Item im = decode(blockIdx.x, csr), im1 = im;
    {
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) stage_load(im, r, sv[r]);
        SCHED_FENCE();
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) {
            uint4 ph[3];
            const StagePos sp = stage_pos(im, 0u, r);
            MI355_UNROLL
            for (int pc = 0; pc < 4; ++pc) stage_piece(pc, sv[r], ph, sp);
        }
        if ((int)(blockIdx.x + gridDim.x) < nitems) im1 = decode((int)(blockIdx.x + gridDim.x), csr);
        MI355_UNROLL
        for (int r = 0; r < ROUNDS; ++r) stage_load(im1, r, sv[r]);
    }
    __syncthreads();

    unsigned cur = 0;  // byte offset of the buffer this item computes from
    MI355_NOUNROLL
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        Item im2 = im1;
        if (it + 2 * (int)gridDim.x < nitems) im2 = decode(it + 2 * (int)gridDim.x, csr);
        const BufRsrc ybuf = buf_rsrc(a.y + (long)im.b * a.y_bs);
        const unsigned other = cur == 0u ? BUF : 0u;  // the buffer item i + 1 is staged into
        const bool edge = im.t0 == 0 || 4 * (im.t0 + NPOS) + 1 >= a.shuf_T;  // position 0 or the row's last position belongs to this item
        unsigned lq[3];
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) {
            lq[p] = cur + (unsigned)p * PS16 + (unsigned)(q * LDP + n) * 16u;
            OPAQUE_V(lq[p]);
        }
        auto b_read = [&](int u, int j, uint4 (&bf)[3]) MI355_INLINE_LAMBDA {
            const int g = u / K, k = u % K;
            MI355_UNROLL
            for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const uint4*>(L0 + lq[p] + (unsigned)((4 * g) * LDP + 16 * j + k) * 16u);
        };
        uint4 ph[3];
        StagePos sp = {0u, false};
        uint4 Bf[2][2][3];
        b_read(0, 0, Bf[0][0]);
        b_read(0, 1, Bf[0][1]);
        rbc_static_for<0, NP>([&](auto JP) MI355_INLINE_LAMBDA {
            constexpr int jp = decltype(JP)::value;
            f32x4 c0, c1;
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) c0[r] = c1[r] = 0.0f;
            rbc_static_for<0, S>([&](auto U) MI355_INLINE_LAMBDA {
                constexpr int u = decltype(U)::value;
                constexpr int slot = jp * S + u, cb = slot & 1;
                // staging of item i + 1 into the other buffer: round slot / 4, piece slot % 4 (slots 0 .. 4 ROUNDS - 1); behind a
                // round's last piece its registers take the loads of item i + 2
                constexpr int sr = slot / 4, pc = slot % 4;
                if constexpr (sr < ROUNDS && pc == 0) sp = stage_pos(im1, other, sr);
                if constexpr (u + 1 < S) {
                    b_read(u + 1, 2 * jp, Bf[cb ^ 1][0]);
                    b_read(u + 1, 2 * jp + 1, Bf[cb ^ 1][1]);
                } else if constexpr (jp + 1 < NP) {
                    b_read(0, 2 * jp + 2, Bf[cb ^ 1][0]);
                    b_read(0, 2 * jp + 3, Bf[cb ^ 1][1]);
                }
                SCHED_FENCE();
                {
                    const uint4(&W)[3] = Wr[u];
                    const uint4(&B0)[3] = Bf[cb][0];
                    const uint4(&B1)[3] = Bf[cb][1];
                    c0 = MFMA_16x16x32_BF16(W[2], B0[0], c0);  // small terms first
                    c1 = MFMA_16x16x32_BF16(W[2], B1[0], c1);
                    c0 = MFMA_16x16x32_BF16(W[0], B0[2], c0);
                    c1 = MFMA_16x16x32_BF16(W[0], B1[2], c1);
                    c0 = MFMA_16x16x32_BF16(W[1], B0[1], c0);
                    c1 = MFMA_16x16x32_BF16(W[1], B1[1], c1);
                    c0 = MFMA_16x16x32_BF16(W[1], B0[0], c0);
                    c1 = MFMA_16x16x32_BF16(W[1], B1[0], c1);
                    c0 = MFMA_16x16x32_BF16(W[0], B0[1], c0);
                    c1 = MFMA_16x16x32_BF16(W[0], B1[1], c1);
                    c0 = MFMA_16x16x32_BF16(W[0], B0[0], c0);
                    c1 = MFMA_16x16x32_BF16(W[0], B1[0], c1);
                }
                if constexpr (sr < ROUNDS) {
                    stage_piece(pc, sv[sr], ph, sp);
                    MI355_UNROLL
                    for (int x = 0; x < 12; ++x) {
                        SCHED_GROUP(0x8, 1);
                        SCHED_GROUP(0x2, 3);
                    }
                    if constexpr (pc == 3) {
                        SCHED_FENCE();
                        stage_load(im2, sr, sv[sr]);
                    }
                }
                SCHED_FENCE();
            });
            // this pair is complete: bias, the four phases of a channel side by side (n = 4 i - 2 + phase): ONE 16-byte store per tile
            // (8-byte aligned: the memory pipe is paid per store instruction and per 64-byte segment touched, and two 8-byte stores
            // 16 bytes apart touch every segment twice); the first and the last item of a row, where half a position falls outside
            // the row, take the two 8-byte stores with their own range tests (a wave-uniform choice)
            MI355_UNROLL
            for (int e = 0; e < 2; ++e) {
                const f32x4& c = e == 0 ? c0 : c1;
                const int il = 16 * (2 * jp + e) + n;  // position inside the item
                const int i = im.t0 + il;
                const bool on = il < NPOS && i < a.T;
                const int ch = m0 >> 2, n0 = 4 * i - 2;
                if (ST16 && !edge) {
                    const unsigned o = on ? 4u * (unsigned)(ch * a.y_ld + n0) : BUF_OOB;
                    buf_store_f4(ybuf, o, 0u, c[0] + bia[0], c[1] + bia[1], c[2] + bia[2], c[3] + bia[3]);
                } else {
                    const unsigned o0 = (on && i >= 1) ? 4u * (unsigned)(ch * a.y_ld + n0) : BUF_OOB;
                    const unsigned o1 = (on && n0 + 3 < a.shuf_T) ? 4u * (unsigned)(ch * a.y_ld + n0 + 2) : BUF_OOB;
                    buf_store_f2(ybuf, o0, 0u, c[0] + bia[0], c[1] + bia[1]);
                    buf_store_f2(ybuf, o1, 0u, c[2] + bia[2], c[3] + bia[3]);
                }
            }
        });
        __syncthreads();
        cur = other;
        im = im1;
        im1 = im2;
    }
}

// ------------------------------------------------------------------------------------------------ host side
namespace {
inline bool rbc_shape(int K, int dil) {  // the "_low" voices' stage-0 convs (instantiated tap / dilation pairs)
    return (K == 3 && (dil == 1 || dil == 2)) || (K == 5 && (dil == 2 || dil == 6)) || (K == 7 && (dil == 3 || dil == 12));
}
}  // namespace

bool rb_conv_supported(const ConvArgs& a) {
    // the kernels address x, res and y through 32-bit buffer offsets `row * ld + column` (bytes): every leading dimension must cover
    // the row (ld >= T) and the farthest element of a batch row must stay inside the 2 GiB range of a buffer resource — a caller
    // with a larger pitch falls back to the staged kernels instead of reading zeros / dropping stores through the range check
    const long ld = std::max(std::max((long)a.x_ld, (long)a.res_ld), (long)a.y_ld);
    return a.Cin == RBC_C && a.Cout == RBC_C && rbc_shape(a.K, a.dil) && a.epi == EPI_STD && a.res && a.in_len && !a.cond && !a.relu && !a.res_sub && !a.mask_before_res &&
           !a.out_len && !a.shuf_s && a.Tin < 0 && a.pad == (a.K - 1) / 2 * a.dil && a.ksplit == 1 && a.x_ld >= a.T && a.res_ld >= a.T && a.y_ld >= a.T &&
           (long)RBC_C * ld * 4 < 0x7fffffffL;
}

// a.w = the conv's pack_conv_weights_p16 fragments.  wide = 128-column items (grids that fill the chip), else 32-column items:
// the same bits either way
void launch_rb_conv(ConvArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!rb_conv_supported(a)) throw std::runtime_error("rb_conv: unsupported shape");
    const int cus = current_device_cu_count();
    a.item_order = RBC_ITEM_ORDER;
    if (const char* f = lab_getenv("MI355VITS_RBC_ITEM_ORDER")) a.item_order = atoi(f);  // lab / tests: 0 = items w, w + W, ... as in round 4
    // (the form follows the items that HAVE work: a ragged batch is as large as the sum of its rows)
    bool wide = mrf_valid_items(a.in_len_host, a.in_len, a.B, a.T, 128) >= cus;
    if (const char* f = lab_getenv("MI355VITS_RBC_WIDE")) wide = atoi(f) != 0;  // lab / tests
    auto go = [&](auto kfn, size_t lds, int ncols) {
        a.nvalid = mrf_valid_items(a.in_len_host, a.in_len, a.B, a.T, ncols);
        const long nitems = a.nvalid;
        if (nitems <= 0) return;
        dim3 grid((unsigned)(nitems < cus ? nitems : cus));  // persistent: one workgroup per CU
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)RBC_LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(512), lds, s, a);
    };
    // wide items: the producer-wave form (twelve waves), MI355VITS_RBC_PW=0 (lab / tests): staging inside the matrix waves' streams
    int pw = RBC_PW_DEFAULT;
    if (const char* f = lab_getenv("MI355VITS_RBC_PW")) pw = atoi(f);
    auto go_pw = [&](auto kfn, size_t lds) {
        a.nvalid = mrf_valid_items(a.in_len_host, a.in_len, a.B, a.T, 128);
        const long nitems = a.nvalid;
        if (nitems <= 0) return;
        dim3 grid((unsigned)(nitems < cus ? nitems : cus));
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)RBC_LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(768), lds, s, a);
    };
#if defined(MI355_LAB) && !defined(MI355_EMU)
    if (wide && lab_getenv("MI355VITS_RBC_CLOCKS") && ((a.K == 7 && a.dil == 3) || (a.K == 3 && a.dil == 1))) {
        static int shots = 0;
        if (shots < 4) {  // (two launches of each shape: the first warms the caches)
            ++shots;
            a.nvalid = mrf_valid_items(a.in_len_host, a.in_len, a.B, a.T, 128);
            const long nitems = a.nvalid;
            const int grid = (int)(nitems < cus ? nitems : cus);
            float* dbg = nullptr;
            const size_t nb = (size_t)grid * 3 * 8 * 8 * 4;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbg), nb));
            HIP_CHECK(hipMemsetAsync(dbg, 0, nb, s));
            ConvArgs c = a;
            c.part = dbg;
            const int prio = pw == 2 ? 1 : 0;
            auto launch_c = [&](auto kfn, size_t lds) {
                set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)RBC_LDS_LIMIT);
                LAUNCH_KERNEL(kfn, dim3(grid), dim3(768), lds, s, c);
            };
            if (a.K == 7) { if (prio) launch_c(k_rb_conv_pw<7, 3, 3, 1, true>, RbcGeo<7, 3, 8>::LDS); else launch_c(k_rb_conv_pw<7, 3, 3, 0, true>, RbcGeo<7, 3, 8>::LDS); }
            else { if (prio) launch_c(k_rb_conv_pw<3, 1, 3, 1, true>, RbcGeo<3, 1, 8>::LDS); else launch_c(k_rb_conv_pw<3, 1, 3, 0, true>, RbcGeo<3, 1, 8>::LDS); }
            HIP_CHECK(hipStreamSynchronize(s));
            std::vector<unsigned> h((size_t)grid * 3 * 8 * 8);
            HIP_CHECK(hipMemcpy(h.data(), dbg, nb, hipMemcpyDeviceToHost));
            (void)hipFree(dbg);
            const int wg = grid > 7 ? 7 : 0;
            const char* names[3] = {"matrix wave 0", "matrix wave 4", "producer wave 8"};
            for (int slot = 0; slot < 3; ++slot) {
                fprintf(stderr, "rb_conv_pw<%d,%d> prio=%d clocks wg %d %s:", a.K, a.dil, (int)prio, wg, names[slot]);
                for (int it = 0; it < 6; ++it) {
                    const unsigned* t = &h[(((size_t)wg * 3 + slot) * 8 + it) * 8];
                    if (slot < 2) fprintf(stderr, " | item %d: phase0 %u barrier %u phase1 %u barrier %u epilogue %u", it, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]);
                    else fprintf(stderr, " | item %d: stores %u loads %u barrier %u stores %u loads %u barrier %u", it, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
                }
                const unsigned* t0 = &h[(((size_t)wg * 3 + slot) * 8 + 0) * 8];
                const unsigned* t5 = &h[(((size_t)wg * 3 + slot) * 8 + 5) * 8];
                fprintf(stderr, " | items 0..5 span %u\n", t5[slot < 2 ? 5 : 6] - t0[0]);
            }
            return;
        }
    }
#endif
#if defined(MI355_LAB) && !defined(MI355_EMU)
#define RBC_LAB_PRIO(KK, DD)                                                                     \
    if (wide && pw == 3) { go_pw(k_rb_conv_pw<KK, DD, 3, 2>, RbcGeo<KK, DD, 8>::LDS); return; }  \
    if (wide && pw == 4) { go_pw(k_rb_conv_pw<KK, DD, 3, 3>, RbcGeo<KK, DD, 8>::LDS); return; }
#else
#define RBC_LAB_PRIO(KK, DD)
#endif
#define RBC_CASE(KK, DD)                                                             \
    if (a.K == KK && a.dil == DD) {                                                  \
        RBC_LAB_PRIO(KK, DD)                                                         \
        if (wide && pw == 2) go_pw(k_rb_conv_pw<KK, DD, 3, 1>, RbcGeo<KK, DD, 8>::LDS);   \
        else if (wide && pw) go_pw(k_rb_conv_pw<KK, DD, 3, 0>, RbcGeo<KK, DD, 8>::LDS);   \
        else if (wide) go(k_rb_conv<KK, DD, 8>, RbcGeo<KK, DD, 8>::LDS, 128);        \
        else go(k_rb_conv<KK, DD, 2>, RbcGeo<KK, DD, 2>::LDS, 32);                   \
        return;                                                                      \
    }
    RBC_CASE(3, 1)
    RBC_CASE(3, 2)
    RBC_CASE(5, 2)
    RBC_CASE(5, 6)
    RBC_CASE(7, 3)
    RBC_CASE(7, 12)
#undef RBC_CASE
    throw std::runtime_error("rb_conv: unsupported shape");
}


// natural row order (row tile t = rows 16 t .. 16 t + 15): [row tile][tap][k-group][plane][lane][8 bf16], lane = (quarter q, row l & 15)
// holds the input channels 32 g + 8 q + 0..7; w = h + m + l exactly, each term rounded to nearest on the host
void pack_conv_weights_p16n(const float* w, int Cout, int Cin, int K, uint32_t* out) {
    auto rne = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
    auto tof = [](uint32_t b) {
        const uint32_t u = b << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    const int G = Cin / 32;
    for (int t = 0; t < Cout / 16; ++t)
        for (int k = 0; k < K; ++k)
            for (int g = 0; g < G; ++g)
                for (int l = 0; l < 64; ++l) {
                    const int q = l >> 4, co = 16 * t + (l & 15);
                    uint32_t plane[3][8];
                    for (int e = 0; e < 8; ++e) {
                        const float v = w[((size_t)co * Cin + 32 * g + 8 * q + e) * K + k];
                        const uint32_t h = rne(v);
                        const float r1 = v - tof(h);
                        const uint32_t mm = rne(r1);
                        const float r2 = r1 - tof(mm);
                        plane[0][e] = h; plane[1][e] = mm; plane[2][e] = rne(r2);
                    }
                    for (int p = 0; p < 3; ++p) {
                        uint32_t* o = out + (((((size_t)t * K + k) * G + g) * 3 + p) * 64 + l) * 4;
                        for (int jj = 0; jj < 4; ++jj) o[jj] = plane[p][2 * jj] | (plane[p][2 * jj + 1] << 16);
                    }
                }
}

bool ups_pl_supported(const ConvArgs& a) {
    bool s4 = a.Cin == 64 && a.shuf_s == 4;
    if (lab_getenv("MI355VITS_NO_UPS64")) s4 = false;  // lab / tests: the 64 -> 32 upsampler on the staged polyphase kernel
    const bool s8 = (a.Cin == 128 || a.Cin == 256) && a.shuf_s == 8;
    return (s8 || s4) && a.K == 2 && a.dil == 1 && a.pad == 1 && a.Cout == a.shuf_s * (a.Cin / 2) && a.shuf_cout == a.Cin / 2 && a.shuf_p == a.shuf_s / 2 &&
           a.epi == EPI_STD && a.bias && a.in_len && a.Tin >= 0 && a.T == a.Tin + 1 && a.shuf_T == a.Tin * a.shuf_s && !a.res && !a.cond && !a.relu &&
           !a.accumulate && a.out_scale == 1.0f && !a.out_len && a.ksplit == 1 && a.y_ld % 4 == 0 && a.y_bs % 4 == 0 &&
           reinterpret_cast<uintptr_t>(a.y) % 16 == 0 && reinterpret_cast<uintptr_t>(a.bias) % 16 == 0 &&  // (16-byte stores of y, float4 loads of the bias)
           a.x_ld >= a.Tin && a.y_ld >= a.shuf_T && (long)a.Cin * a.x_ld * 4 < 0x7fffffffL && (long)(a.Cin / 2) * a.y_ld * 4 < 0x7fffffffL;
}

// a.w = the polyphase filter's pack_conv_weights_p16n fragments
void launch_ups_pl(ConvArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!ups_pl_supported(a)) throw std::runtime_error("ups_pl: unsupported shape");
    const int cus = current_device_cu_count();
    a.item_order = RBC_ITEM_ORDER;
    if (const char* f = lab_getenv("MI355VITS_RBC_ITEM_ORDER")) a.item_order = atoi(f);  // lab / tests: 0 = items w, w + W, ... as in round 4
    auto go = [&](auto kfn, size_t lds, int ncols) {
        a.nvalid = mrf_valid_items(a.in_len_host, a.in_len, a.B, a.T, ncols, 1);  // a row of len input positions has len + 1 output positions
        const long nitems = a.nvalid;
        if (nitems <= 0) return;
        dim3 grid((unsigned)(nitems < cus ? nitems : cus));
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)RBC_LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(512), lds, s, a);
    };
    if (a.Cin == 256) {  // 256 -> 128: eight row blocks over 64-column items (32 records of planes: 123 KiB)
        bool wide = (long)a.B * ((a.T + 63) / 64) >= cus;
        if (const char* f = lab_getenv("MI355VITS_RBC_WIDE")) wide = atoi(f) != 0;  // lab / tests
        if (wide) go(k_ups_pl<256, 8, 4>, UpsGeo<256, 8, 4>::LDS, 64);
        else go(k_ups_pl<256, 8, 2>, UpsGeo<256, 8, 2>::LDS, 32);
    } else if (a.Cin == 128) {
        bool wide = (long)a.B * ((a.T + 127) / 128) >= cus;
        if (const char* f = lab_getenv("MI355VITS_RBC_WIDE")) wide = atoi(f) != 0;  // lab / tests
        if (wide) go(k_ups_pl<128, 8, 8>, UpsGeo<128, 8, 8>::LDS, 128);
        else go(k_ups_pl<128, 8, 2>, UpsGeo<128, 8, 2>::LDS, 32);
    } else {
        const long n127 = (long)a.B * ((a.T + 126) / 127);
        bool wide = n127 >= cus;
        if (const char* f = lab_getenv("MI355VITS_RBC_WIDE")) wide = atoi(f) != 0;
        const bool st8 = lab_getenv("MI355VITS_UPS64_ST8") != nullptr;  // lab / tests: two 8-byte stores per tile (round 4)
        if (wide && st8) go(k_ups64<8, false>, Ups64Geo<8>::LDS, 127);
        else if (wide) go(k_ups64<8, true>, Ups64Geo<8>::LDS, 127);
        else if (st8) go(k_ups64<2, false>, Ups64Geo<2>::LDS, 31);
        else go(k_ups64<2, true>, Ups64Geo<2>::LDS, 31);
    }
}

}  // namespace m355
