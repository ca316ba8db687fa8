// kernels_wn.cpp — one WaveNet layer of the residual-coupling flow as ONE kernel.
//
//   a  = in_l(h) [+ cond_l(g)]              k=5 conv, H -> 2H
//   u  = tanh(a[:H]) * sigmoid(a[H:])
//   rs = res_skip_l(u)                      1x1 conv, H -> 2H (last layer: H -> H)
//   h' = (h + rs[:H]) * mask ;  skip += rs[H:]        (last layer: skip += rs)
//
// Layer-at-a-time this is two launches and moves u (H x T) through HBM; here a workgroup owns 32 time columns of ALL
// channels: the h tile (+halo) is staged once into LDS, wave p of the H/32 waves computes gate pair p (rows c and
// H + c land in the same lane, so the gate is pure register math), u goes to a second LDS tile, and the same waves
// run the 1x1 conv from it: wave p produces rows {c, H + c} again, i.e. exactly its own slice of h' and of skip.
// Both MFMA operand streams follow the pipeline of kernels_mrf.cpp (A: 8-register rings four steps ahead from
// L2; B: LDS, one step ahead; taps unrolled).  h is read from one buffer and written to another (neighbouring
// workgroups read each other's halo columns).
#include "kernels.h"
#include "b3.h"

namespace m355 {

// two output tiles (A streams wp0 / wp1), one 32-column tile, K taps x CP channel pairs, accumulate into acc0/acc1
template <int CP>
__device__ __forceinline__ void wn_mfma2(f32x16& acc0, f32x16& acc1, const float* __restrict__ wp0,
                                         const float* __restrict__ wp1, const float* __restrict__ xw, int LD, int K, int dil) {
    static_assert(CP % 8 == 0, "channel pairs per tap must be a multiple of 8");
    const int ld2 = 2 * LD;
    float r0[8], r1[8];
    MI355_UNROLL
    for (int u = 0; u < 4; ++u) { r0[u] = wp0[u * 64]; r1[u] = wp1[u * 64]; }
    float bb[2];
    bb[0] = xw[0];
    const float* g0 = wp0;
    const float* g1 = wp1;
    for (int k = 0; k < K; ++k) {
        MI355_UNROLL
        for (int cp0 = 0; cp0 < CP; cp0 += 8) {
            const float* base = xw + k * dil + cp0 * ld2;
            const bool last = (k == K - 1) && (cp0 + 8 == CP);
            const float* nbase = last ? base : ((cp0 + 8 < CP) ? base + 8 * ld2 : xw + (k + 1) * dil);
            const float* h0 = last ? g0 - 8 * 64 : g0;  // the last group prefetches harmlessly from itself
            const float* h1 = last ? g1 - 8 * 64 : g1;
            MI355_UNROLL
            for (int u = 0; u < 8; ++u) {
                const float a0 = r0[u], a1 = r1[u];
                r0[(u + 4) & 7] = (u < 4) ? g0[(u + 4) * 64] : h0[(u + 4) * 64];
                r1[(u + 4) & 7] = (u < 4) ? g1[(u + 4) * 64] : h1[(u + 4) * 64];
                bb[(u + 1) & 1] = (u < 7) ? base[(u + 1) * ld2] : nbase[0];
                acc0 = MFMA_32x32x2_F32(a0, bb[u & 1], acc0);
                acc1 = MFMA_32x32x2_F32(a1, bb[u & 1], acc1);
                SCHED_FENCE();
            }
            g0 += 8 * 64;
            g1 += 8 * 64;
        }
    }
}

template <int NP>  // H = 32 * NP hidden channels, NP waves
__global__ __launch_bounds__(64 * NP) MIN_WAVES_PER_SIMD(NP > 1 ? 5 : 1) void k_wn_layer(WnArgs a) {
    constexpr int H = 32 * NP;
    constexpr int CP = H / 2;
    constexpr int T_B = 32;
    DYN_SMEM(float, smem);
    const int LDX = a.ldx;
    float* X = smem;  // [H][LDX] h tile (+halo), zero outside the row
    float* U = smem;  // [H][32]  gated activations; reuses the h tile's space once the in-layer conv is done
    const int tid = threadIdx.x, lane = tid & 63, p = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;
    const int pad = (a.K - 1) / 2 * a.dil;
    const int tlo = t0 - pad;
    const int ts = tlo >= 0 ? (tlo & ~3) : -(((-tlo) + 3) & ~3);
    const int toff = tlo - ts;

    if (!(LAB_ABLATE(a) & 2)) {   // stage h[b, :, ts : ts + LDX)
        const float* hb = a.h_in + (long)b * a.h_bs;
        if (a.vec) {
            const int ld4 = LDX >> 2;
            for (int idx = tid; idx < H * ld4; idx += 64 * NP) {
                const int r = idx / ld4, c4 = idx - r * ld4;
                const int t = ts + 4 * c4;
                const float* row = hb + (long)r * a.h_ld;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (t >= 0 && t + 3 < len) {
                    v = *reinterpret_cast<const float4*>(row + t);
                } else {
                    if (t >= 0 && t < len) v.x = row[t];
                    if (t + 1 >= 0 && t + 1 < len) v.y = row[t + 1];
                    if (t + 2 >= 0 && t + 2 < len) v.z = row[t + 2];
                    if (t + 3 >= 0 && t + 3 < len) v.w = row[t + 3];
                }
                *reinterpret_cast<float4*>(X + r * LDX + 4 * c4) = v;
            }
        } else {
            for (int idx = tid; idx < H * LDX; idx += 64 * NP) {
                const int r = idx / LDX, c = idx - r * LDX;
                const int t = ts + c;
                X[idx] = (t >= 0 && t < len) ? hb[(long)r * a.h_ld + t] : 0.0f;
            }
        }
    }
    __syncthreads();

    const int t = t0 + bcol;
    // ---- in-layer conv (gate pair p) + gate
    f32x16 gate, hres;
    {
        f32x16 acc0, acc1;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * p + (r & 3) + 8 * (r >> 2) + 4 * brow;
            float v0 = a.b_in[c], v1 = a.b_in[H + c];
            if (a.cond) { v0 += a.cond[(long)b * a.cond_bs + c]; v1 += a.cond[(long)b * a.cond_bs + H + c]; }
            acc0[r] = v0;
            acc1[r] = v1;
        }
        const float* wp0 = a.w_in + (long)(2 * p) * a.K * CP * 64 + lane;
        const float* wp1 = wp0 + (long)a.K * CP * 64;
        if (!(LAB_ABLATE(a) & 1)) wn_mfma2<CP>(acc0, acc1, wp0, wp1, X + brow * LDX + toff + bcol, LDX, a.K, a.dil);
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            gate[r] = wn_gate_f(acc0[r], acc1[r]);
            const int c = 32 * p + (r & 3) + 8 * (r >> 2) + 4 * brow;
            hres[r] = X[c * LDX + toff + pad + bcol];
        }
    }
    __syncthreads();  // every wave is done with the h tile: U takes its place
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int c = 32 * p + (r & 3) + 8 * (r >> 2) + 4 * brow;
        U[c * 32 + bcol] = gate[r];
    }
    __syncthreads();
    // ---- res/skip 1x1 conv: rows c (-> h') and H + c (-> skip); last layer: rows c -> skip
    {
        const bool two = a.Crs == 2 * H;
        f32x16 acc0, acc1;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * p + (r & 3) + 8 * (r >> 2) + 4 * brow;
            acc0[r] = a.b_rs[c];
            acc1[r] = two ? a.b_rs[H + c] : 0.0f;
        }
        const float* wp0 = a.w_rs + (long)p * CP * 64 + lane;
        const float* wp1 = two ? a.w_rs + (long)(NP + p) * CP * 64 + lane : wp0;
        if (!(LAB_ABLATE(a) & 1)) wn_mfma2<CP>(acc0, acc1, wp0, wp1, U + brow * 32 + bcol, 32, 1, 0);
        if (t < a.T && !((LAB_ABLATE(a) & 4) && acc0[0] != 1.2345f)) {
            const bool live = t < len;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * p + (r & 3) + 8 * (r >> 2) + 4 * brow;
                float* sp = a.skip + (long)b * a.s_bs + (long)c * a.s_ld + t;
                if (two) {
                    a.h_out[(long)b * a.h_bs + (long)c * a.h_ld + t] = live ? hres[r] + acc0[r] : 0.0f;  // same order as k_wn_layer_h192
                    *sp = a.skip_init ? acc1[r] : *sp + acc1[r];
                } else {
                    *sp = a.skip_init ? acc0[r] : *sp + acc0[r];
                }
            }
        }
    }
}


// MT output tiles (A streams wp[m]), one 32-column tile, K taps x CP channel pairs
template <int MT, int CP>
__device__ __forceinline__ void wn_mfma(f32x16 (&acc)[MT], const float* const (&wp)[MT], const float* __restrict__ xw, int LD,
                                        int K, int dil) {
    static_assert(CP % 8 == 0, "channel pairs per tap must be a multiple of 8");
    const int ld2 = 2 * LD;
    float ra[MT][8];
    const float* g[MT];
    MI355_UNROLL
    for (int m = 0; m < MT; ++m) {
        g[m] = wp[m];
        MI355_UNROLL
        for (int u = 0; u < 4; ++u) ra[m][u] = wp[m][u * 64];
    }
    float bb[2];
    bb[0] = xw[0];
    for (int k = 0; k < K; ++k) {
        MI355_UNROLL
        for (int cp0 = 0; cp0 < CP; cp0 += 8) {
            const float* base = xw + k * dil + cp0 * ld2;
            const bool last = (k == K - 1) && (cp0 + 8 == CP);
            const float* nbase = last ? base : ((cp0 + 8 < CP) ? base + 8 * ld2 : xw + (k + 1) * dil);
            const int back = last ? 8 * 64 : 0;  // the last group prefetches harmlessly from itself
            MI355_UNROLL
            for (int u = 0; u < 8; ++u) {
                float av[MT];
                MI355_UNROLL
                for (int m = 0; m < MT; ++m) {
                    av[m] = ra[m][u];
                    ra[m][(u + 4) & 7] = (u < 4) ? g[m][(u + 4) * 64] : (g[m] - back)[(u + 4) * 64];
                }
                bb[(u + 1) & 1] = (u < 7) ? base[(u + 1) * ld2] : nbase[0];
                MI355_UNROLL
                for (int m = 0; m < MT; ++m) acc[m] = MFMA_32x32x2_F32(av[m], bb[u & 1], acc[m]);
                SCHED_FENCE();
            }
            MI355_UNROLL
            for (int m = 0; m < MT; ++m) g[m] += 8 * 64;
        }
    }
}

// Same product with the B operand packed for 16-byte LDS reads: the tile is stored [channel-pair group of 4][brow][column][4]
// (element (c, col) at (((c >> 3) * 2 + (c & 1)) * LD + col) * 4 + ((c >> 1) & 3)), so one ds_read_b128 delivers a
// lane's B fragments for four consecutive k-steps — a quarter of the LDS instructions of the b32 version, worth ~10 % of
// the loop for this tile shape (tools/mfma_ceiling.hip, "B4").  x4 points at (group 0, this lane's brow, its column).
template <int MT, int CP>
__device__ __forceinline__ void wn_mfma_b4(f32x16 (&acc)[MT], const float* const (&wp)[MT], const float4* __restrict__ x4, int LD,
                                           int K, int dil) {
    static_assert(CP % 8 == 0, "channel pairs per tap must be a multiple of 8");
    float ra[MT][8];
    const float* g[MT];
    MI355_UNROLL
    for (int m = 0; m < MT; ++m) {
        g[m] = wp[m];
        MI355_UNROLL
        for (int u = 0; u < 4; ++u) ra[m][u] = wp[m][u * 64];
    }
    float4 b0 = x4[0], b1 = b0;
    for (int k = 0; k < K; ++k) {
        MI355_UNROLL
        for (int cp0 = 0; cp0 < CP; cp0 += 8) {
            const float4* base = x4 + k * dil + (cp0 >> 2) * 2 * LD;
            const bool last = (k == K - 1) && (cp0 + 8 == CP);
            const float4* nbase = last ? base : ((cp0 + 8 < CP) ? base + 4 * LD : x4 + (k + 1) * dil);
            const int back = last ? 8 * 64 : 0;  // the last group prefetches harmlessly from itself
            MI355_UNROLL
            for (int u = 0; u < 8; ++u) {
                float av[MT];
                MI355_UNROLL
                for (int m = 0; m < MT; ++m) {
                    av[m] = ra[m][u];
                    ra[m][(u + 4) & 7] = (u < 4) ? g[m][(u + 4) * 64] : (g[m] - back)[(u + 4) * 64];
                }
                if (u == 0) b1 = base[2 * LD];  // second group of four of this step group
                const float4 bq = u < 4 ? b0 : b1;
                const float bv = (u & 3) == 0 ? bq.x : ((u & 3) == 1 ? bq.y : ((u & 3) == 2 ? bq.z : bq.w));
                if (u == 4) b0 = nbase[0];      // first group of the next step group (b0 is free from here on)
                MI355_UNROLL
                for (int m = 0; m < MT; ++m) acc[m] = MFMA_32x32x2_F32(av[m], bv, acc[m]);
                SCHED_FENCE();
            }
            MI355_UNROLL
            for (int m = 0; m < MT; ++m) g[m] += 8 * 64;
        }
    }
}

// H = 192: four waves, each three of the twelve 32-row tiles of either conv, 32 time columns per workgroup.  With
// three MFMA tiles per wave and three workgroups per CU every SIMD carries exactly three waves (six waves of two
// tiles leave the SIMDs 5/5/4/4), and fewer, fatter waves run the operand streams closer to the matrix-core rate
// (tools/mfma_ceiling.hip).  The price: tanh and sigmoid rows of a gate pair now sit in different waves, so the raw
// in-layer result takes a round trip through LDS (it reuses the h tile's space) before gating.
template <int NW>  // waves per workgroup: 4 (3 tiles each) or 12 (1 tile each: shortest MFMA chain, for tiny grids)
__global__ __launch_bounds__(64 * NW) MIN_WAVES_PER_SIMD(3) void k_wn_layer_h192(WnArgs a) {
    constexpr int H = 192, CP = 96, NTILE = 12, MT = NTILE / NW, NTH = 64 * NW;
    static_assert(NW == 4 || NW == 12, "wave count");
    DYN_SMEM(float, smem);
    const int LDX = a.ldx;
    float* X = smem;  // [H][LDX] h tile (+halo), zero outside the row; later A [2H][32] raw in-layer result, U = A[:H]
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 32;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;
    const int pad = (a.K - 1) / 2 * a.dil;
    const int tlo = t0 - pad;
    const int ts = tlo >= 0 ? (tlo & ~3) : -(((-tlo) + 3) & ~3);
    const int toff = tlo - ts;
    const float* hb = a.h_in + (long)b * a.h_bs;
    // LDS index of element (channel c, column col) of a tile with `ld` columns: see wn_mfma_b4
    auto pk = [](int c, int col, int ld) { return ((((c >> 3) * 2 + (c & 1)) * ld + col) << 2) + ((c >> 1) & 3); };
    if (!(LAB_ABLATE(a) & 2)) {
        // one float4 of LDS = four channels (8g + 2q + brow, q = 0..3) at one column.  A thread moves a 4 x 4 block: four
        // 16-byte loads along time (one per channel), transposed in registers, four 16-byte LDS stores (one per column)
        const int ld4 = LDX >> 2;
        for (int idx = tid; idx < (H / 4) * ld4; idx += NTH) {
            const int gb = idx / ld4, c4 = idx - gb * ld4;  // gb = g * 2 + brow
            const int c0 = (gb >> 1) * 8 + (gb & 1);
            const int tt = ts + 4 * c4;
            float v[4][4];
            MI355_UNROLL
            for (int q = 0; q < 4; ++q) {
                const float* row = hb + (long)(c0 + 2 * q) * a.h_ld;
                if (a.vec && tt >= 0 && tt + 3 < len) {
                    const float4 r4 = *reinterpret_cast<const float4*>(row + tt);
                    v[q][0] = r4.x; v[q][1] = r4.y; v[q][2] = r4.z; v[q][3] = r4.w;
                } else {
                    MI355_UNROLL
                    for (int j = 0; j < 4; ++j) v[q][j] = (tt + j >= 0 && tt + j < len) ? row[tt + j] : 0.0f;
                }
            }
            MI355_UNROLL
            for (int j = 0; j < 4; ++j)
                reinterpret_cast<float4*>(X)[gb * LDX + 4 * c4 + j] = make_float4(v[0][j], v[1][j], v[2][j], v[3][j]);
        }
    }
    __syncthreads();

    const int t = t0 + bcol;
    const bool two = a.Crs == 2 * H;
    f32x16 hres[MT];
    const float* condp = a.cond ? a.cond + (long)b * a.cond_bs : a.b_in;
    const float cond_on = a.cond ? 1.0f : 0.0f;
    {   // ---- in-layer conv: packed tiles 3w .. 3w+2 (tile 2p = rows 32p.. of the tanh half, 2p+1 = same rows, sigmoid half)
        f32x16 acc[MT];
        const float* wp[MT];
        MI355_UNROLL
        for (int m = 0; m < MT; ++m) {
            const int q = MT * w + m;
            const int row0 = ((q & 1) ? H : 0) + 32 * (q >> 1);
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int c = row0 + (r & 3) + 8 * (r >> 2) + 4 * brow;
                // unconditional loads (a test per element serialises them); without conditioning the second term is
                // +0.0f * bias: the sum is unchanged bit for bit
                acc[m][r] = a.b_in[c] + cond_on * condp[c];
            }
            wp[m] = a.w_in + (long)q * a.K * CP * 64 + lane;
        }
        if (!(LAB_ABLATE(a) & 1))
            wn_mfma_b4<MT, CP>(acc, wp, reinterpret_cast<const float4*>(X) + brow * LDX + toff + bcol, LDX, a.K, a.dil);
        if (two && MT * w < NTILE / 2) {  // the residual input of this wave's h' tiles, while the h tile is still there
            MI355_UNROLL
            for (int m = 0; m < MT; ++m)
                MI355_UNROLL
                for (int r = 0; r < 16; ++r)
                    hres[m][r] = X[pk(32 * (MT * w + m) + (r & 3) + 8 * (r >> 2) + 4 * brow, toff + pad + bcol, LDX)];
        }
        __syncthreads();  // every wave is done with the h tile
        MI355_UNROLL
        for (int m = 0; m < MT; ++m) {
            const int q = MT * w + m;
            const int row0 = ((q & 1) ? H : 0) + 32 * (q >> 1);
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {  // both halves in the packed layout of U, so the gate below works in place
                const int cc = 32 * (q >> 1) + (r & 3) + 8 * (r >> 2) + 4 * brow;
                X[((q & 1) ? H * 32 : 0) + pk(cc, bcol, 32)] = acc[m][r];
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < H * 32; idx += NTH) {  // gate in place: U[c][t] = tanh(A[c][t]) * sigmoid(A[H + c][t])
        const float at = X[idx], as = X[H * 32 + idx];
        const float gate = wn_gate_f(at, as);
        X[idx] = gate;
    }
    __syncthreads();
    // ---- res/skip 1x1 conv: row tiles q < 6 -> h', q >= 6 -> skip (last layer: 6 tiles, all -> skip)
    const float4* U = reinterpret_cast<const float4*>(X) + brow * 32 + bcol;
    const bool live = t < len;
    auto finish = [&](const f32x16& acc, int q, const f32x16& hr) {
        if (t >= a.T || ((LAB_ABLATE(a) & 4) && acc[0] != 1.2345f)) return;
        if (two && 32 * q < H) {  // wave-uniform: a tile lies entirely in the h' half or in the skip half
            MI355_UNROLL
            for (int r = 0; r < 16; ++r)
                a.h_out[(long)b * a.h_bs + (long)(32 * q + (r & 3) + 8 * (r >> 2) + 4 * brow) * a.h_ld + t] = live ? hr[r] + acc[r] : 0.0f;
            return;
        }
        // skip: the 16 old values under one test, in flight together (a test per element serialises the loads)
        float* sp = a.skip + (long)b * a.s_bs + (long)(32 * q - (two ? H : 0) + 4 * brow) * a.s_ld + t;
        float old[16];
        if (!a.skip_init) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) old[r] = sp[(long)((r & 3) + 8 * (r >> 2)) * a.s_ld];
        }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) sp[(long)((r & 3) + 8 * (r >> 2)) * a.s_ld] = a.skip_init ? acc[r] : old[r] + acc[r];
    };
    if (two) {
        f32x16 acc[MT];
        const float* wp[MT];
        MI355_UNROLL
        for (int m = 0; m < MT; ++m) {
            const int q = MT * w + m;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[m][r] = a.b_rs[32 * q + (r & 3) + 8 * (r >> 2) + 4 * brow];
            wp[m] = a.w_rs + (long)q * CP * 64 + lane;
        }
        if (!(LAB_ABLATE(a) & 1)) wn_mfma_b4<MT, CP>(acc, wp, U, 32, 1, 0);
        MI355_UNROLL
        for (int m = 0; m < MT; ++m) finish(acc[m], MT * w + m, hres[m]);
    } else if (NW == 12) {  // 6 tiles, one per wave
        if (w < 6) {
            f32x16 a1[1];
            const float* w1[1] = {a.w_rs + (long)w * CP * 64 + lane};
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) a1[0][r] = a.b_rs[32 * w + (r & 3) + 8 * (r >> 2) + 4 * brow];
            if (!(LAB_ABLATE(a) & 1)) wn_mfma_b4<1, CP>(a1, w1, U, 32, 1, 0);
            finish(a1[0], w, hres[0]);
        }
    } else {  // 6 tiles: waves 0, 1 take two, waves 2, 3 one
        const int nq = w < 2 ? 2 : 1;
        f32x16 acc[2];
        const float* wp[2];
        MI355_UNROLL
        for (int m = 0; m < 2; ++m) {
            const int q = m == 0 ? w : w + 4;
            const int qq = q < 6 ? q : w;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[m][r] = a.b_rs[32 * qq + (r & 3) + 8 * (r >> 2) + 4 * brow];
            wp[m] = a.w_rs + (long)qq * CP * 64 + lane;
        }
        if (!(LAB_ABLATE(a) & 1)) {
            if (nq == 2) {
                wn_mfma_b4<2, CP>(acc, wp, U, 32, 1, 0);
            } else {
                f32x16 a1[1] = {acc[0]};
                const float* w1[1] = {wp[0]};
                wn_mfma_b4<1, CP>(a1, w1, U, 32, 1, 0);
                acc[0] = a1[0];
            }
        }
        finish(acc[0], w, hres[0]);
        if (nq == 2) finish(acc[1], w + 4, hres[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// The same WaveNet layer in MATH_BF16X3 (f32 operands split 3 x bf16, six products on v_mfma_f32_32x32x16_bf16).
// At 16x the f32 MFMA rate the 32-column geometry above would be bound by streaming the layer's weights (every
// workgroup reads all of them; 32 columns of reuse), so the tile is turned around:
//   * a workgroup owns 96 time columns of ALL 2H = 384 in-layer rows: B x T / 96 workgroups (256 for the batch-32
//     bench shape: one per CU), four waves = one per SIMD with the whole 512-register file each;
//   * a wave computes 3 row tiles x 3 column tiles (9 accumulators): every weight fragment feeds 3 MFMA groups, every
//     activation fragment 3 — 18 operand fetches per 54 MFMAs;
//   * h (+halo) is split into three bf16 planes once while it is staged (b3.h), the raw in-layer result goes through
//     LDS (the dead h planes' space), is gated by all threads, split, and becomes the B operand of the res/skip conv;
//     u never leaves the CU.
// LDS: max(3 planes x 192 ch x 104 col x 2 B = 117 KiB, raw 384 x 96 x 4 B = 144 KiB).
// ------------------------------------------------------------------------------------------------
constexpr int WNB_H = 192, WNB_NG = WNB_H / 16;
constexpr int WNB_DEFAULT_WAVES = 4;
constexpr int WNB_RING = 4;  // weight-fragment buffers of the four-wave form
constexpr int WNB_EPI = 1;   // epilogue form of the four-wave, 96-column kernel (template parameter EP; 1 since round 5: profiles/r05_wn_epilogue_ab.txt)
// phase clocks of one workgroup (lab build, MI355VITS_WN_ABLATE bit 64): shader-clock deltas printed by workgroup (3, 5), wave 0
#if defined(MI355_LAB) && !defined(MI355_EMU)
#define WN_TS_DECL() long long wn_ts[7] = {0, 0, 0, 0, 0, 0, 0}
#define WN_TS(i) do { if (a.ablate & 64) wn_ts[i] = __builtin_readcyclecounter(); } while (0)
#define WN_TS_END()                                                                                                                  \
    do {                                                                                                                             \
        if ((a.ablate & 64) && blockIdx.x == 3 && blockIdx.y == 5 && threadIdx.x == 0) {                                            \
            __builtin_amdgcn_s_waitcnt(0);                                                                                           \
            wn_ts[6] = __builtin_readcyclecounter();                                                                                 \
            printf("wn phases (cycles): stage %lld inconv %lld rawstore %lld gate+planes %lld rs %lld epilogue %lld total %lld\n", wn_ts[1] - wn_ts[0], \
                   wn_ts[2] - wn_ts[1], wn_ts[3] - wn_ts[2], wn_ts[4] - wn_ts[3], wn_ts[5] - wn_ts[4], wn_ts[6] - wn_ts[5], wn_ts[6] - wn_ts[0]);   \
        }                                                                                                                            \
    } while (0)
#else
#define WN_TS_DECL() ((void)0)
#define WN_TS(i) ((void)0)
#define WN_TS_END() ((void)0)
#endif  // waves of the 96-column form: see launch_wn_layer_b3

// NT column tiles per workgroup: 3 (96 columns) when the grid fills the chip, 1 (32 columns) for small grids (one
// utterance: a third of the dependent MFMA chain per wave, three times the workgroups).  The arithmetic of an output
// element — chunk, tap, group and product order — is the same in both, so the results are bit-identical and the choice
// may depend on the grid size.

// H2 (MATH_F16X2): operands as two fp16 planes (activations x 2^4 while they are split, weights x 2^13 when packed), three
// products per multiply-add; the accumulators run scaled by 2^17 (bias and conditioning enter scaled) and are unscaled,
// exactly, where they leave the matrix cores — the gate and the residual / skip updates see the same values as before.
// MW row tiles (of the 12) per wave: 3 = four waves, one per SIMD with the whole register file each; 1 = twelve waves, three per
// SIMD with 170 registers each — the same fragments fetched once per workgroup, the same products in the same order per
// accumulator (bit-identical), but three instruction streams per SIMD to cover each other's L2 / LDS waits in the matrix loops
// (the four-wave form loses 40 % on a box whose fabric answers slower).
// RA: weight-fragment ring of the four-wave form (b3.h b3_chunk_ra): 2 = one group ahead (rounds 2 - 4), 4 = three groups ahead
// EP: the epilogue's old-value loads (h residual: L2 hits, skip accumulator: HBM) — 0 = one tile ahead of the stores (rounds 2 - 4), 1 = three
// tiles ahead (48 loads in flight per lane: the nine tiles of a wave were nine dependent round trips), 2 = three ahead AND the first three
// tiles' loads issued in front of the barrier behind the gate phase (they land under the res/skip conv's first steps; lab form)
// TW ("two workgroups per CU"): 64-column tiles whose LDS fits twice into a CU (h planes 68 columns = 76.5 KiB, u planes
// 72 KiB) with <= 256 registers per wave, so that one workgroup's
// HBM phases (staging, the h / skip read-modify-write) run under the other's matrix loops — at the price of streaming the layer's
// fragments once per 64 instead of once per 96 columns.  The same products in the same order per output element: bit-identical.
template <bool W1, int NT, bool H2, int MW, int RA, int EP, bool TW>
__device__ __forceinline__ void wn_layer_b3_body(const WnArgs& a) {
    constexpr int NWV = 12 / MW, NTH = 64 * NWV;
    static_assert(!(W1 && H2), "one reduced-operand variant at a time");
    static_assert(!TW || (NT == 2 && MW == 3 && !H2), "the two-per-CU form: four waves, 64 columns");
    constexpr int H = WNB_H, T_B = 32 * NT, NG = WNB_NG;
    constexpr int NP = H2 ? 2 : 3, GW = H2 ? 128 : 192;  // planes per operand; uint4 per weight-fragment group
    constexpr float ACC = H2 ? F16X2_ACC_SCALE : 1.0f, UNACC = H2 ? 1.0f / F16X2_ACC_SCALE : 1.0f;
    DYN_SMEM(float, smem);
    uint4* planes = reinterpret_cast<uint4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;
    if (t0 >= len) return;  // ragged batches (round 6): a tile past its row's length is never computed (kernels_mrfp.cpp next_item)
    const int pad = (a.K - 1) / 2 * a.dil;
    const int tlo = t0 - pad;
    const int ts = TW ? tlo : (tlo >= 0 ? (tlo & ~3) : -(((-tlo) + 3) & ~3));  // (TW: no alignment slack — 68 staged columns, 76.5 KiB)
    const int toff = tlo - ts;
    const int LD = a.ldx;          // staged columns of the h tile
    WN_TS_DECL();
    WN_TS(0);
    const int PS = NG * 2 * LD;    // uint4 per plane
    const bool two = a.Crs == 2 * H;

    if (!(LAB_ABLATE(a) & 2)) stage_planes<NG, (TW ? 8 : (MW == 3 ? NG : 4)), H2>(a.h_in + (long)b * a.h_bs, a.h_ld, LD, ts, len, 1.0f, planes, PS, tid, NTH);  // column sets x row batches: one round trip
    __syncthreads();
    WN_TS(1);

    // ---- in-layer conv: wave w owns row tiles w, w + 4, w + 8 (rows 32 q .. 32 q + 31 of the 2H), all 3 column tiles
    // (speaker conditioning without a branch per element — a test per load makes hipcc wait for each one in turn)
    const float* condp = a.cond ? a.cond + (long)b * a.cond_bs : a.b_in;
    const float cond_on = a.cond ? 1.0f : 0.0f;
    {
        f32x16 acc[MW][NT];
        const uint4* wp[MW];
        MI355_UNROLL
        for (int i = 0; i < MW; ++i) {
            const int q = w + NWV * i;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                // tile q = the tanh rows (tile rows 0 - 15) and the sigmoid rows (16 - 31) of channels 16 q .. 16 q + 15: see the gate below
                const int c = (r < 8 ? 0 : H) + 16 * q + (r & 3) + 8 * ((r >> 2) & 1) + 4 * brow;
                const float v = (a.b_in[c] + cond_on * condp[c]) * ACC;  // unconditional loads: all in flight together
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) acc[i][j][r] = v;
            }
            wp[i] = reinterpret_cast<const uint4*>(a.w_in) + (long)q * a.K * NG * GW + lane;
        }
        if (!(LAB_ABLATE(a) & 1)) {
            // (twelve waves: 170 registers each — the form with the B fragments single-buffered, same products in the same order)
            if constexpr (H2) h2_chunk<MW, NT, NG, NT>(acc, wp, planes + brow * LD + bcol + toff, PS, LD, a.K, NG, a.dil);
            else if constexpr (MW == 1 || TW) b3_chunk_lean<MW, NT, NG, W1>(acc, wp, planes + brow * LD + bcol + toff, PS, LD, a.K, NG, a.dil);  // (TW: 256 registers)
            else if constexpr (RA > 2 && NT >= 4) b3_chunk_lean_ra<MW, NT, NG, W1, RA>(acc, wp, planes + brow * LD + bcol + toff, PS, LD, a.K, NG, a.dil);  // (four column tiles: B single-buffered)
            else if constexpr (RA > 2) b3_chunk_ra<MW, NT, NG, NT, W1, RA>(acc, wp, planes + brow * LD + bcol + toff, PS, LD, a.K, NG, a.dil);
            else b3_chunk<MW, NT, NG, NT, W1>(acc, wp, planes + brow * LD + bcol + toff, PS, LD, a.K, NG, a.dil);
        }
        __syncthreads();  // every wave is done with the h planes: u's planes take their place (column pitch T_B)
        WN_TS(2);
        WN_TS(3);
        // ---- gate in registers (round 6): the in-layer conv's rows are stored permuted (Engine::add_conv_data, packed_b3w / packed_h2s of a gate
        // conv) so that 32-row tile q holds tanh rows of channels 16 q .. 16 q + 15 in its rows 0 - 15 and their sigmoid rows in 16 - 31: accumulator
        // r < 8 and r + 8 of a lane are the two halves of ONE channel, and a lane's eight gated values of a (row tile, column tile) are exactly
        // the eight k-slots of one B-operand record (16-channel group q, half brow, its column) of the res/skip conv — the raw result never
        // goes through LDS (rounds 2 - 5: 144 ds_write_b32 + a barrier + 144 ds_read_b32 per lane).  Same values, same gate: same bits.
        constexpr int PSU0 = NG * 2 * T_B;
        MI355_UNROLL
        for (int i = 0; i < MW; ++i)
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) {
                float u[8];
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) u[e] = wn_gate_f(acc[i][j][e] * UNACC, acc[i][j][e + 8] * UNACC);
                const int idx = (2 * (w + NWV * i) + brow) * T_B + 32 * j + bcol;
                if constexpr (H2) {
                    uint4 h4, m4;
                    split2_pk(u[0] * F16X2_X_SCALE, u[1] * F16X2_X_SCALE, h4.x, m4.x);
                    split2_pk(u[2] * F16X2_X_SCALE, u[3] * F16X2_X_SCALE, h4.y, m4.y);
                    split2_pk(u[4] * F16X2_X_SCALE, u[5] * F16X2_X_SCALE, h4.z, m4.z);
                    split2_pk(u[6] * F16X2_X_SCALE, u[7] * F16X2_X_SCALE, h4.w, m4.w);
                    planes[idx] = h4;
                    planes[PSU0 + idx] = m4;
                } else {
                    uint4 h4, m4, l4;
                    split3_pk(u[0], u[1], h4.x, m4.x, l4.x);
                    split3_pk(u[2], u[3], h4.y, m4.y, l4.y);
                    split3_pk(u[4], u[5], h4.z, m4.z, l4.z);
                    split3_pk(u[6], u[7], h4.w, m4.w, l4.w);
                    planes[idx] = h4;
                    planes[PSU0 + idx] = m4;
                    planes[2 * PSU0 + idx] = l4;
                }
            }
    }
    const int ntr = a.Crs / 32;
    const float* src[MW];
    float* dst[MW];
    long ldr[MW];
    bool valid[MW], use_old[MW], to_h[MW];
    MI355_UNROLL
    for (int i = 0; i < MW; ++i) {
        const int q = w + NWV * i;
        valid[i] = q < ntr;
        const int qc = valid[i] ? q : ntr - 1;
        to_h[i] = two && 32 * qc < H;  // wave-uniform
        if (to_h[i]) {
            src[i] = a.h_in + (long)b * a.h_bs + (long)(32 * qc) * a.h_ld;
            dst[i] = a.h_out + (long)b * a.h_bs + (long)(32 * qc) * a.h_ld;
            ldr[i] = a.h_ld;
            use_old[i] = true;
        } else {
            const int sub = two ? H : 0;
            src[i] = a.skip + (long)b * a.s_bs + (long)(32 * qc - sub) * a.s_ld;
            dst[i] = a.skip + (long)b * a.s_bs + (long)(32 * qc - sub) * a.s_ld;
            ldr[i] = a.s_ld;
            use_old[i] = !a.skip_init;
        }
    }
    auto load_tile = [&](int i, int j, float (&old)[16]) {
        const int t = t0 + j * 32 + bcol;
        const int tc = t < a.T ? t : a.T - 1;
        if (valid[i] && use_old[i]) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) old[r] = src[i][(long)((r & 3) + 8 * (r >> 2) + 4 * brow) * ldr[i] + tc];
        } else {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) old[r] = 0.0f;
        }
    };
    constexpr int EPA = EP == 0 ? 1 : 3, EPR = EPA + 1;  // tiles ahead; ring of old-value tiles
    float old[EPR][16];
    // EP >= 1: the same loads / stores as buffer instructions — a tile's base in the resource, ONE lane offset per tile, the row offsets in
    // SGPRs, and the wave-uniform / per-lane conditions folded into the offset (out of range: the load returns 0, the store is dropped), so
    // nothing between a load's issue and its use is conditional (hipcc's wait-count pass: DESIGN 4.6 item 1)
    auto load_tile_b = [&](int i, int j, float (&o)[16]) MI355_INLINE_LAMBDA {
        const int t = t0 + j * 32 + bcol;
        const int tc = t < a.T ? t : a.T - 1;
        const BufRsrc rs = buf_rsrc(src[i]);
        const unsigned row4 = 4u * (unsigned)ldr[i];
        const unsigned vo = (valid[i] && use_old[i]) ? (unsigned)(4 * brow) * row4 + 4u * (unsigned)tc : BUF_OOB;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) o[r] = buf_load_f32(rs, vo, (unsigned)((r & 3) + 8 * (r >> 2)) * row4);
    };
    if constexpr (EP == 2) {
        MI355_UNROLL
        for (int k = 0; k < EPA; ++k) load_tile_b(k / NT, k % NT, old[k]);
        SCHED_FENCE();
    }
    constexpr int PSU = NG * 2 * T_B;
    __syncthreads();
    WN_TS(4);
    // ---- res/skip 1x1 conv: Crs / 32 row tiles (12, last layer 6), tile q on wave q % 4
    f32x16 acc[MW][NT];
    {
        const uint4* wp[MW];
        MI355_UNROLL
        for (int i = 0; i < MW; ++i) {
            int q = w + NWV * i;
            if (q >= ntr) q = ntr - 1;  // beyond the last tile: recompute it, discarded below
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const float v = a.b_rs[32 * q + (r & 3) + 8 * (r >> 2) + 4 * brow] * ACC;
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) acc[i][j][r] = v;
            }
            wp[i] = reinterpret_cast<const uint4*>(a.w_rs) + (long)q * NG * GW + lane;
        }
        if (!(LAB_ABLATE(a) & 1)) {
            if (two || MW == 1) {  // (twelve waves, six tiles: waves 6 .. 11 recompute the last tile, discarded below)
                if constexpr (H2) h2_chunk<MW, NT, NG, NT>(acc, wp, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
                else if constexpr (MW == 1 || TW) b3_chunk_lean<MW, NT, NG, W1>(acc, wp, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
                else if constexpr (RA > 2 && NT >= 4) b3_chunk_lean_ra<MW, NT, NG, W1, RA>(acc, wp, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
                else if constexpr (RA > 2) b3_chunk_ra<MW, NT, NG, NT, W1, RA>(acc, wp, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
                else b3_chunk<MW, NT, NG, NT, W1>(acc, wp, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
            } else if constexpr (MW == 3) {  // 6 tiles: waves 0, 1 two tiles, waves 2, 3 one (second index clamped)
                f32x16 a2[2][NT];
                const uint4* w2[2] = {wp[0], wp[1]};
                MI355_UNROLL
                for (int i = 0; i < 2; ++i)
                    MI355_UNROLL
                    for (int j = 0; j < NT; ++j) a2[i][j] = acc[i][j];
                if constexpr (H2) h2_chunk<2, NT, NG, NT>(a2, w2, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
                else b3_chunk<2, NT, NG, NT, W1>(a2, w2, planes + brow * T_B + bcol, PSU, T_B, 1, NG, 0);
                MI355_UNROLL
                for (int i = 0; i < 2; ++i)
                    MI355_UNROLL
                    for (int j = 0; j < NT; ++j) acc[i][j] = a2[i][j];
            }
        }
        if constexpr (H2) {  // back to the unscaled domain (exact)
            MI355_UNROLL
            for (int i = 0; i < MW; ++i)
                MI355_UNROLL
                for (int j = 0; j < NT; ++j)
                    MI355_UNROLL
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= UNACC;
        }
    }
    WN_TS(5);
    if ((LAB_ABLATE(a) & 4) && acc[0][0][0] != 1.2345f) return;
    // ---- epilogue: rows < H (two-output layers): h' = (h + rs) * mask; the others: skip (+)= rs.  The 16 old values a
    // lane needs per 32 x 32 tile are loaded unconditionally (clamped column) and one tile ahead of the stores: the
    // memory counter retires in order, so a load issued after a store cannot be waited for without waiting for that
    // store's acknowledgement too — with tile k + 1's loads in front of tile k's stores no wait includes a fresh store.
    auto store_tile = [&](int i, int j, const float (&old)[16]) {
        const int t = t0 + j * 32 + bcol;
        if (!valid[i] || t >= a.T) return;
        const bool live = t < len || !to_h[i];  // h' is masked, skip is not (its consumer masks)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) dst[i][(long)((r & 3) + 8 * (r >> 2) + 4 * brow) * ldr[i] + t] = live ? old[r] + acc[i][j][r] : 0.0f;
    };
    if constexpr (EP == 0) {
        load_tile(0, 0, old[0]);
        MI355_UNROLL
        for (int k = 0; k < MW * NT; ++k) {
            if (k + 1 < MW * NT) load_tile((k + 1) / NT, (k + 1) % NT, old[(k + 1) & 1]);
            SCHED_FENCE();
            store_tile(k / NT, k % NT, old[k & 1]);
            SCHED_FENCE();
        }
    } else {
        auto store_tile_b = [&](int i, int j, const float (&o)[16]) MI355_INLINE_LAMBDA {
            const int t = t0 + j * 32 + bcol;
            const bool live = t < len || !to_h[i];
            const BufRsrc rs = buf_rsrc(dst[i]);
            const unsigned row4 = 4u * (unsigned)ldr[i];
            const unsigned vo = (valid[i] && t < a.T) ? (unsigned)(4 * brow) * row4 + 4u * (unsigned)t : BUF_OOB;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) buf_store_f32(rs, vo, (unsigned)((r & 3) + 8 * (r >> 2)) * row4, live ? o[r] + acc[i][j][r] : 0.0f);
        };
        if constexpr (EP == 1) {
            MI355_UNROLL
            for (int k = 0; k < EPA && k < MW * NT; ++k) load_tile_b(k / NT, k % NT, old[k]);
        }
        MI355_UNROLL
        for (int k = 0; k < MW * NT; ++k) {
            if (k + EPA < MW * NT) load_tile_b((k + EPA) / NT, (k + EPA) % NT, old[(k + EPA) % EPR]);
            SCHED_FENCE();
            store_tile_b(k / NT, k % NT, old[k % EPR]);
            SCHED_FENCE();
        }
    }
    WN_TS_END();
}

template <bool W1, int NT, bool H2 = false, int MW = 3, int RA = 2, int EP = 0>
__global__ __launch_bounds__(64 * (12 / MW)) void k_wn_layer_b3(WnArgs a) {
    wn_layer_b3_body<W1, NT, H2, MW, RA, EP, false>(a);
}
// the two-per-CU form: <= 256 registers per wave (two waves per SIMD), 76.5 KiB of LDS
template <bool W1>
__global__ __launch_bounds__(256, 2) void k_wn_layer_b3_tw(WnArgs a) {
    wn_layer_b3_body<W1, 2, false, 3, 2, 1, true>(a);
}

bool wn_layer_b3_supported(int H, int K, int dil) {
    return H == WNB_H && (K % 2) == 1 && K >= 1 && dil >= 1 && (K - 1) * dil <= 24;
}

void launch_wn_layer_b3(WnArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!wn_layer_b3_supported(a.H, a.K, a.dil)) throw std::runtime_error("wn_layer_b3: unsupported shape");
    static const int ablate = lab_getenv("MI355VITS_WN_ABLATE") ? atoi(lab_getenv("MI355VITS_WN_ABLATE")) : 0;
    a.ablate = ablate;
    a.vec = (a.h_ld % 4 == 0) && (a.h_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.h_in) % 16 == 0);
    // 96-column tiles when they fill the chip, 32-column tiles for small grids (same bits, see k_wn_layer_b3)
    const char* nt_s = lab_getenv("MI355VITS_WN_B3_NT");  // read per launch: tests flip it inside one process
    const long nwg3 = (long)((a.T + 95) / 96) * a.B;
    int nt = nt_s ? atoi(nt_s) : (nwg3 < 128 ? 1 : 3);
    // 128-column tiles (round 6: they fit since the gate left LDS — 153 KiB of h planes; a fragment group feeds 72 instead of 54 MFMAs, the
    // per-workgroup phases are paid once per 128 columns: 0.96 of the 96-column form's time per column, profiles/r06_wn_nt4_ab.txt) when the
    // persistent rounds of the grid come out shorter: rounds x tile cost (4/3 x 0.96 = 1.28 of a 96-column tile).  Same bits.
    if (!nt_s && nt == 3 && a.math != MATH_F16X2) {
        const long cus = current_device_cu_count();
        const long nwg4 = (long)((a.T + 127) / 128) * a.B;
        if ((double)((nwg4 + cus - 1) / cus) * 1.28 < (double)((nwg3 + cus - 1) / cus)) nt = 4;
    }
    if (nt == 4 && (a.math == MATH_F16X2 || ((long)a.h_ld * 128 >= 0x7fffffffL || (long)a.s_ld * 128 >= 0x7fffffffL || (a.K - 1) * a.dil > 8))) nt = 3;  // (buffer-addressed epilogue; 160 KiB of LDS)
    // the two-workgroups-per-CU form (64-column tiles, k_wn_layer_b3_tw): large grids of the default / bf16-weights math
    // Measured on the MI355X (profiles/r05_wn_two_per_cu_ab.txt): 8.5 % SLOWER than the 96-column form at batch 256 (14.3 vs 13.2 ms per 16
    // layers x 256 rows), 31 % slower at batch 32 (384 workgroups on 512 slots) — what the overlap of the HBM phases gains, the 1.5 x
    // weight stream through the L1 and the leaner loop (2 x 3 instead of 3 x 3 tiles per fragment pair) lose again.  Lab build and CPU
    // model only, as the A/B of that statement (MI355VITS_WN_TW=1).
    bool tw = false;
#if defined(MI355_LAB) || defined(MI355_EMU)
    if (const char* f = lab_getenv("MI355VITS_WN_TW")) tw = atoi(f) != 0 && nt == 3 && a.math != MATH_F16X2;
#endif
    if ((long)a.h_ld * 128 >= 0x7fffffffL || (long)a.s_ld * 128 >= 0x7fffffffL) tw = false;  // (its epilogue: a 32-row tile within the buffer range)
    if (tw && (size_t)3 * WNB_NG * 2 * (64 + (a.K - 1) * a.dil) * 16 > 80 * 1024) tw = false;  // two of them must fit a CU's 160 KiB
    if (tw) nt = 2;
    const int tb = 32 * nt;
    a.ldx = tw ? tb + (a.K - 1) * a.dil : ((tb + (a.K - 1) * a.dil + 3 + 3) & ~3);
    const size_t shmem = (size_t)3 * WNB_NG * 2 * a.ldx * 16;  // the h planes; u's planes (column pitch tb <= ldx) take their place
    dim3 grid((a.T + tb - 1) / tb, a.B);
    // four waves (one per SIMD, the whole register file each) is the product's form.  The twelve-wave form (three per SIMD; MW = 1)
    // is bit-identical and measured EQUAL on the MI355X (16 layers 1.77 vs 1.76 ms, profiles/r04_wn_experiments.txt): the matrix
    // loops do not wait on latencies that more waves could cover, what the layer loses it loses in its serial phases — so it is
    // compiled into the lab build and the CPU model only (MI355VITS_WN_WAVES=12), as the A/B of that statement
    const char* nw_s = lab_getenv("MI355VITS_WN_WAVES");
    const int nw = nw_s ? atoi(nw_s) : WNB_DEFAULT_WAVES;
    auto go = [&](auto kfn, int threads) {
#ifndef MI355_EMU
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
#endif
        LAUNCH_KERNEL(kfn, grid, dim3(threads), shmem, s, a);
    };
#if defined(MI355_LAB) || defined(MI355_EMU)
    if (tw) {
        if (a.math == MATH_BF16W) go(k_wn_layer_b3_tw<true>, 256);
        else go(k_wn_layer_b3_tw<false>, 256);
    } else
#endif
    if (nt == 4 && a.math != MATH_F16X2) {
        // weight fragments three groups ahead as in the 96-column form, the B fragments single-buffered (b3_chunk_lean_ra: with both
        // double-buffered the loop spills 148 B at four column tiles); MI355VITS_WN_RING=2: one group ahead, B double-buffered
        int ring = WNB_RING;
        if (const char* f = lab_getenv("MI355VITS_WN_RING")) ring = atoi(f);
        if (a.math == MATH_BF16W) go(k_wn_layer_b3<true, 4, false, 3, 4, 1>, 256);
        else if (ring > 2) go(k_wn_layer_b3<false, 4, false, 3, 4, 1>, 256);
        else go(k_wn_layer_b3<false, 4, false, 3, 2, 1>, 256);
    } else if (nt == 1) {
        if (a.math == MATH_F16X2) go(k_wn_layer_b3<false, 1, true>, 256);
        else if (a.math == MATH_BF16W) go(k_wn_layer_b3<true, 1>, 256);
        else go(k_wn_layer_b3<false, 1>, 256);
#if defined(MI355_LAB) || defined(MI355_EMU)
    } else if (nw == 12 && a.math != MATH_F16X2) {
        if (a.math == MATH_BF16W) go(k_wn_layer_b3<true, 3, false, 1>, 768);
        else go(k_wn_layer_b3<false, 3, false, 1>, 768);
#endif
    } else {
        // default math, large grids: the weight fragments three groups ahead (WNB_RING; MI355VITS_WN_RING=2: one group ahead)
        int ring = WNB_RING;
        if (const char* f = lab_getenv("MI355VITS_WN_RING")) ring = atoi(f);
        int epi = WNB_EPI;
        if (const char* f = lab_getenv("MI355VITS_WN_EPI")) epi = atoi(f);
        if ((long)a.h_ld * 128 >= 0x7fffffffL || (long)a.s_ld * 128 >= 0x7fffffffL) epi = 0;  // a 32-row tile must fit the buffer range
        if (a.math == MATH_F16X2) go(k_wn_layer_b3<false, 3, true>, 256);
        else if (a.math == MATH_BF16W) go(k_wn_layer_b3<true, 3>, 256);
        else if (ring > 2 && epi == 2) go(k_wn_layer_b3<false, 3, false, 3, 4, 2>, 256);
        else if (ring > 2 && epi == 1) go(k_wn_layer_b3<false, 3, false, 3, 4, 1>, 256);
        else if (ring > 2) go(k_wn_layer_b3<false, 3, false, 3, 4>, 256);
        else go(k_wn_layer_b3<false, 3>, 256);
    }
}

bool wn_layer_fused_supported(int H, int K, int dil) {
    return (H == 32 || H == 192) && (K % 2) == 1 && K >= 1 && dil >= 1 && (size_t)H * (32 + (K - 1) * dil + 8) * 4 <= 48 * 1024;
}

void launch_wn_layer(WnArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (!wn_layer_fused_supported(a.H, a.K, a.dil)) throw std::runtime_error("wn_layer: unsupported shape");
    a.ldx = (32 + (a.K - 1) * a.dil + 3 + 3) & ~3;
    static const int ablate = lab_getenv("MI355VITS_WN_ABLATE") ? atoi(lab_getenv("MI355VITS_WN_ABLATE")) : 0;
    a.ablate = ablate;
    a.vec = (a.h_ld % 4 == 0) && (a.h_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.h_in) % 16 == 0);
    const size_t shmem = (size_t)a.H * a.ldx * sizeof(float);  // ldx >= 32: U fits in the h tile
    dim3 grid((a.T + 31) / 32, a.B);
    // Three geometries with identical arithmetic (same bits): 4 waves x 3 tiles keeps the SIMDs evenly loaded when the
    // grid fills the chip; 6 x 2 and 12 x 1 have ever shorter dependent MFMA chains per wave, which is what matters when
    // only a few dozen workgroups exist (one utterance: 31 workgroups per layer).
    const char* six_s = lab_getenv("MI355VITS_WN_SIX_WAVES");  // read per launch: tests flip it inside one process
    const int six_env = six_s ? atoi(six_s) : -1;
    const long nwg = (long)grid.x * grid.y;
    const int geom = six_env >= 0 ? six_env : (nwg < 128 ? 2 : (nwg < 512 ? 1 : 0));  // 0: 4x3, 1: 6x2, 2: 12x1
    const size_t sh4 = (size_t)a.H * (a.ldx > 64 ? a.ldx : 64) * sizeof(float);  // h tile, then [2H][32] raw result
    if (a.H == 192 && geom == 0) {
        LAUNCH_KERNEL(k_wn_layer_h192<4>, grid, dim3(256), sh4, s, a);
    } else if (a.H == 192 && geom == 2) {
        LAUNCH_KERNEL(k_wn_layer_h192<12>, grid, dim3(768), sh4, s, a);
    } else if (a.H == 192) {
        auto k = k_wn_layer<6>;
        LAUNCH_KERNEL(k, grid, dim3(384), shmem, s, a);
    } else {
        auto k = k_wn_layer<1>;
        LAUNCH_KERNEL(k, grid, dim3(64), shmem, s, a);
    }
}

}  // namespace m355
