// kernels_mrf.cpp — one HiFi-GAN multi-receptive-field stage as ONE kernel.
//
//   y = (1/n) * sum_j RB_j(x),   RB_j: x1 = x + conv_{k_j,d1_j}(lrelu_0.1(x));  x2 = x1 + conv_{k_j,d2_j}(lrelu_0.1(x1))
//
// Conv-by-conv this stage moves ~7 activation tensors per resblock through HBM and its k=3 convs sit at
// 24 FLOP/B — HBM-bound on MI355X (ridge 19.7 FLOP/B fp32).  Here a workgroup owns a [C, T_B] output tile:
//   * x[C, T_B + 2R] (R = largest receptive radius of the n resblocks) is staged ONCE into LDS, already leaky-relu'd
//     (the MFMA B operand is then a bare LDS read); the residual path recovers the raw value (v < 0 ? 10 v : v);
//   * conv1 of a resblock runs on the fp32 matrix cores over T_B + 2*r2 columns and leaves x1 (masked to the row's
//     own length) in a second LDS tile; conv2 consumes it and accumulates x2 into per-lane registers that persist
//     across the n resblocks;
//   * A fragments stream from L2/L1 in fragment order (same packing as k_conv1d_mfma) with a one-step register
//     prefetch; B fragments are conflict-free ds_read_b32 (32 consecutive columns per half-wave);
//   * y is written once.  HBM traffic = read x + write y = 8*C bytes per sample (layer-at-a-time: ~50*C).
// Wave layout: WM waves over the C/32 output-channel tiles x WT waves over time; 4 waves, one per SIMD; LDS ~155 KiB
// (one workgroup per CU).  Columns are processed in 32-wide MFMA tiles; conv1's ceil((T_B+2*r2)/32) tiles are dealt
// round-robin to the WT time waves.
#include <cstdlib>

#include "kernels.h"
#include <type_traits>
#include "b3.h"

namespace m355 {

// One wave's share of a conv on the matrix cores: NTL column tiles (32 columns each, `tstride` floats apart in the
// LDS tile), all of the wave's 32 output channels, K taps x CP channel pairs, accumulated INTO acc (callers preload
// bias / residual there).  The LDS tiles hold leaky-relu'd activations, so a B fragment is a bare ds_read_b32.
// The instruction stream per k-step is kept to the MFMAs plus one A load, the B reads and an address add:
//   * steps are walked in groups of 8 channel pairs of one tap, so every address in the unrolled body is the group
//     base plus a compile-time multiple of the row pitch (no per-step index arithmetic);
//   * A fragments: 8-register ring, four steps ahead (L2 latency); B fragments: two statically indexed buffers,
//     one step ahead — no register copies anywhere in the loop.
// LDS index of element (channel c, column col) of a tile with `ld` columns: four channels of the same MFMA half
// (c = 8g + 2q + brow, q = 0..3) sit side by side, so a lane's B fragments of four consecutive k-steps are ONE
// ds_read_b128.  The A fragments are packed the same way ([tile][tap][group of 4 pairs][lane][4]): one
// global_load_dwordx4 per four k-steps.  16-byte operand fetches are worth 5-9 % of the loop for these tile shapes
// (tools/mfma_ceiling.hip, "x4" rows).
// (pk, f4c, stage_tile_pk: hipx.h)

template <int NTL, int NA, int CP>
__device__ __forceinline__ void mfma_conv_tiles(f32x16 (&acc)[NA], const float4* __restrict__ wp4, const float4* __restrict__ x4,
                                                int tstride, int LD, int K, int dil, int ablate) {
    static_assert(NTL <= NA, "tile count");
    static_assert(CP % 16 == 0, "channel pairs per tap must be a multiple of 16");
    if (ablate & 1) return;
    constexpr int NG = CP / 4;  // groups of four channel pairs per tap
    // A: ring of four 16-byte registers, two groups (8 k-steps) ahead; B: two buffers, one group ahead
    float4 ra[4];
    ra[0] = wp4[0];
    ra[1] = wp4[64];
    float4 bb[2][NTL];
    MI355_UNROLL
    for (int i = 0; i < NTL; ++i) bb[0][i] = x4[i * tstride];
    for (int k = 0; k < K; ++k) {
        const float4* wk = wp4 + (long)k * NG * 64;
        const float4* xk = x4 + k * dil;
        const bool last_tap = k == K - 1;
        // all groups of a tap unrolled (hipcc drains the vector-memory counter at loop headers it cannot see through);
        // every load is unconditional — at the very end the prefetches read the last group again
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const float4* wa = (last_tap && g + 2 >= NG) ? wk + (NG - 1) * 64 : wk + (g + 2) * 64;
            ra[(g + 2) & 3] = *wa;
            const float4* xb = (g + 1 < NG) ? xk + (g + 1) * 2 * LD : (last_tap ? xk + g * 2 * LD : xk + dil);
            MI355_UNROLL
            for (int i = 0; i < NTL; ++i) bb[(g + 1) & 1][i] = xb[i * tstride];
            MI355_UNROLL
            for (int q = 0; q < 4; ++q) {
                MI355_UNROLL
                for (int i = 0; i < NTL; ++i) acc[i] = MFMA_32x32x2_F32(f4c(ra[g & 3], q), f4c(bb[g & 1][i], q), acc[i]);
            }
            SCHED_FENCE();
        }
    }
}

// The same share of a conv with the f32 operands split into three bf16 planes each (hipx.h: split3) and the six leading
// partial products on the bf16 matrix cores — f32-grade results at 6/16 of the f32 MFMA's time.  The LDS tiles are the
// same packed f32 tiles: a lane's eight k-slots of a 16-channel group are two ds_read_b128 (channels 16G + brow + 2e and
// 16G + 8 + brow + 2e), split in registers; the weights come pre-split ([tile][tap][16-ch group][plane][lane] x 16 B,
// pack_conv_weights_bf16x3).  A planes and the raw B floats are fetched one group ahead.
template <int NTL, int NA, int CP, bool W1 = false>
__device__ __forceinline__ void mfma_conv_tiles_b3(f32x16 (&acc)[NA], const uint4* __restrict__ wp, const float4* __restrict__ x4,
                                                   int tstride, int LD, int K, int dil, int ablate) {
    static_assert(NTL <= NA, "tile count");
    static_assert(CP % 16 == 0, "channel pairs per tap must be a multiple of 16");
    if (ablate & 1) return;
    constexpr int NG = CP / 8;  // 16-channel groups per tap (even)
    constexpr int NPA = W1 ? 1 : 3;  // weight planes fetched ("bf16 weights": the leading term only)
    uint4 ra[2][3];
    MI355_UNROLL
    for (int p = 0; p < NPA; ++p) ra[0][p] = wp[p * 64];
    float4 xb[2][NTL][2];
    MI355_UNROLL
    for (int i = 0; i < NTL; ++i) {
        xb[0][i][0] = x4[i * tstride];
        xb[0][i][1] = x4[i * tstride + 2 * LD];
    }
    for (int k = 0; k < K; ++k) {
        const uint4* wk = wp + (long)k * NG * 192;
        const float4* xk = x4 + k * dil;
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            // next group's operands (the very last iteration re-reads its own: every load stays unconditional)
            const uint4* wa = (last_tap && g + 1 >= NG) ? wk + g * 192 : wk + (g + 1) * 192;
            MI355_UNROLL
            for (int p = 0; p < NPA; ++p) ra[(g + 1) & 1][p] = wa[p * 64];
            const float4* xn = (g + 1 < NG) ? xk + (g + 1) * 4 * LD : (last_tap ? xk + g * 4 * LD : xk + dil);
            MI355_UNROLL
            for (int i = 0; i < NTL; ++i) {
                xb[(g + 1) & 1][i][0] = xn[i * tstride];
                xb[(g + 1) & 1][i][1] = xn[i * tstride + 2 * LD];
            }
            SCHED_FENCE();  // the prefetches stay ahead of this group's arithmetic (hipcc would sink them to their use)
            const uint4 ah = ra[g & 1][0], am = ra[g & 1][W1 ? 0 : 1], al = ra[g & 1][W1 ? 0 : 2];
            MI355_UNROLL
            for (int i = 0; i < NTL; ++i) {
                uint4 bh, bm, bl;
                split3_x8(xb[g & 1][i][0], xb[g & 1][i][1], bh, bm, bl);
                // small terms first
                if constexpr (!W1) acc[i] = MFMA_32x32x16_BF16(al, bh, acc[i]);
                acc[i] = MFMA_32x32x16_BF16(ah, bl, acc[i]);
                if constexpr (!W1) acc[i] = MFMA_32x32x16_BF16(am, bm, acc[i]);
                if constexpr (!W1) acc[i] = MFMA_32x32x16_BF16(am, bh, acc[i]);
                acc[i] = MFMA_32x32x16_BF16(ah, bm, acc[i]);
                acc[i] = MFMA_32x32x16_BF16(ah, bh, acc[i]);
            }
            SCHED_FENCE();
        }
    }
}

// MATH_F16X2: the same share of a conv with both operands as two fp16 terms (hipx.h: split2) and three products on
// v_mfma_f32_32x32x16_f16.  Weight fragments: [tile][tap][16-ch group][plane h | m][lane] x 16 B (pack_conv_weights_f16x2),
// pre-scaled by 2^13; the activation tiles hold values pre-scaled by 2^4 — the accumulators live scaled by 2^17.
template <int NTL, int NA, int CP>
__device__ __forceinline__ void mfma_conv_tiles_h2(f32x16 (&acc)[NA], const uint4* __restrict__ wp, const float4* __restrict__ x4,
                                                   int tstride, int LD, int K, int dil, int ablate) {
    static_assert(NTL <= NA, "tile count");
    static_assert(CP % 16 == 0, "channel pairs per tap must be a multiple of 16");
    if (ablate & 1) return;
    constexpr int NG = CP / 8;  // 16-channel groups per tap (even)
    uint4 ra[2][2];
    MI355_UNROLL
    for (int p = 0; p < 2; ++p) ra[0][p] = wp[p * 64];
    float4 xb[2][NTL][2];
    MI355_UNROLL
    for (int i = 0; i < NTL; ++i) {
        xb[0][i][0] = x4[i * tstride];
        xb[0][i][1] = x4[i * tstride + 2 * LD];
    }
    for (int k = 0; k < K; ++k) {
        const uint4* wk = wp + (long)k * NG * 128;
        const float4* xk = x4 + k * dil;
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const uint4* wa = (last_tap && g + 1 >= NG) ? wk + g * 128 : wk + (g + 1) * 128;
            MI355_UNROLL
            for (int p = 0; p < 2; ++p) ra[(g + 1) & 1][p] = wa[p * 64];
            const float4* xn = (g + 1 < NG) ? xk + (g + 1) * 4 * LD : (last_tap ? xk + g * 4 * LD : xk + dil);
            MI355_UNROLL
            for (int i = 0; i < NTL; ++i) {
                xb[(g + 1) & 1][i][0] = xn[i * tstride];
                xb[(g + 1) & 1][i][1] = xn[i * tstride + 2 * LD];
            }
            SCHED_FENCE();
            const uint4 ah = ra[g & 1][0], am = ra[g & 1][1];
            MI355_UNROLL
            for (int i = 0; i < NTL; ++i) {
                uint4 bh, bm;
                split2_x8(xb[g & 1][i][0], xb[g & 1][i][1], bh, bm);
                acc[i] = MFMA_32x32x16_F16(am, bh, acc[i]);  // small terms first
                acc[i] = MFMA_32x32x16_F16(ah, bm, acc[i]);
                acc[i] = MFMA_32x32x16_F16(ah, bh, acc[i]);
            }
            SCHED_FENCE();
        }
    }
}

// weight fragments of row tile `wm` of a conv, lane offset included, and the matching inner loop
template <int MATH, int NTL, int NA, int CP>
__device__ __forceinline__ void conv_tiles(f32x16 (&acc)[NA], const float* __restrict__ w, int wm, int lane, const float4* __restrict__ x4,
                                           int tstride, int LD, int K, int dil, int ablate) {
    if constexpr (MATH == 3) {
        const uint4* wp = reinterpret_cast<const uint4*>(w) + (long)wm * K * (CP / 8) * 128 + lane;
        mfma_conv_tiles_h2<NTL, NA, CP>(acc, wp, x4, tstride, LD, K, dil, ablate);
    } else if constexpr (MATH == 1 || MATH == 2) {
        const uint4* wp = reinterpret_cast<const uint4*>(w) + (long)wm * K * (CP / 8) * 192 + lane;
        mfma_conv_tiles_b3<NTL, NA, CP, MATH == 2>(acc, wp, x4, tstride, LD, K, dil, ablate);
    } else {
        const float4* wp = reinterpret_cast<const float4*>(w + (long)wm * K * CP * 64) + lane;
        mfma_conv_tiles<NTL, NA, CP>(acc, wp, x4, tstride, LD, K, dil, ablate);
    }
}

// inverse of leaky-relu(0.1) up to one rounding: the LDS tiles keep activated values, the residual needs the raw one
__device__ __forceinline__ float unlrelu(float v) { return v >= 0.0f ? v : v * 10.0f; }

// conv1 of a resblock, MFMA part, for a wave that owns NTL column tiles q = wt + WT*i of the extended range:
// acc = x + bias + conv(lrelu(x)).  Reads only the X tile, so it may run before the barrier that releases X1.
template <int MATH, int NTL, int NA, int CP, int WT>
__device__ __forceinline__ void mrf_conv1_compute(f32x16 (&acc)[NA], const float* __restrict__ w, const float* bs /*LDS*/,
                                                  const float* X, int LDX, int R, int r1, int r2, int K, int d1, int wm,
                                                  int wt, int brow, int bcol, int ablate) {
    MI355_UNROLL
    for (int i = 0; i < NTL; ++i) {
        const int e = (wt + WT * i) * 32 + bcol;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
            if constexpr (MATH == 3) acc[i][r] = (unlrelu(X[pk(co, (R - r2) + e, LDX)]) * (1.0f / F16X2_X_SCALE) + bs[co]) * F16X2_ACC_SCALE;
            else acc[i][r] = unlrelu(X[pk(co, (R - r2) + e, LDX)]) + bs[co];
        }
    }
    const float4* xw = reinterpret_cast<const float4*>(X) + brow * LDX + (R - r2 - r1) + bcol + wt * 32;
    conv_tiles<MATH, NTL, NA, CP>(acc, w, wm, brow * 32 + bcol, xw, WT * 32, LDX, K, d1, ablate);
    if constexpr (MATH == 3) {  // back to the unscaled domain (exact): the epilogue is the same for every mode
        MI355_UNROLL
        for (int i = 0; i < NTL; ++i)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][r] *= 1.0f / F16X2_ACC_SCALE;
    }
}

// conv2 for a wave that owns NTL output tiles p = wt + WT*i:  out += x1 + bias + conv(lrelu(x1)), accumulated
// straight into the wave's persistent output registers (no epilogue).
template <int MATH, int NTL, int NA, int CP, int WT>
__device__ __forceinline__ void mrf_conv2(f32x16 (&out)[NA], const float* __restrict__ w, const float* bs /*LDS*/,
                                          const float* X1, int LD1, int r2, int K, int d2, int wm, int wt, int brow, int bcol, int ablate) {
    MI355_UNROLL
    for (int i = 0; i < NTL; ++i) {
        const int c0 = (wt + WT * i) * 32 + bcol;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
            if constexpr (MATH == 3) out[i][r] += (unlrelu(X1[pk(co, c0 + r2, LD1)]) * (1.0f / F16X2_X_SCALE) + bs[co]) * F16X2_ACC_SCALE;
            else out[i][r] += unlrelu(X1[pk(co, c0 + r2, LD1)]) + bs[co];
        }
    }
    const float4* xw = reinterpret_cast<const float4*>(X1) + brow * LD1 + bcol + wt * 32;
    conv_tiles<MATH, NTL, NA, CP>(out, w, wm, brow * 32 + bcol, xw, WT * 32, LD1, K, d2, ablate);
}

// WM x WT = 8 waves (two per SIMD).  Output tile T_B = 32 * N2 columns; column tiles of both convs are dealt
// round-robin to the WT time waves.  Barrier placement: conv1 of resblock j+1 only reads the X tile, so it is issued
// right after conv2 of resblock j and the barrier that releases X1 sits before conv1's *epilogue* — a wave that
// finishes its conv2 share early flows straight into the next MFMA stream instead of idling at a barrier.
// LDXC / LD1C: row pitches of the two LDS tiles when known at compile time (the "_low" voices' stage shapes): every
// ds_read of an unrolled k-step group then carries its offset as an immediate; 0 = take them from the arguments.
// MATH: 0 = f32 MFMA (v_mfma_f32_32x32x2_f32), 1 = f32 operands split 3 x bf16, six products on the bf16 MFMA,
// 2 = the same with the weights' leading bf16 term only (three products; "bf16 weights").
template <int WM, int WT, int N2, int NT1MAX, int NT2MAX, int LDXC = 0, int LD1C = 0, int MATH = 0>
__global__ __launch_bounds__(512) void k_mrf_fused(MrfArgs a) {
    static_assert(WM * WT == 8, "8 waves per workgroup");
    constexpr float XS = MATH == 3 ? F16X2_X_SCALE : 1.0f;          // scale of the values kept in the LDS tiles
    constexpr float OUT_UNSCALE = MATH == 3 ? 1.0f / F16X2_ACC_SCALE : 1.0f;
    static_assert(NT1MAX == 3 && NT2MAX == 2, "static dispatch below");
    constexpr int C = 32 * WM;
    constexpr int T_B = 32 * N2;
    constexpr int CP = C / 2;
    DYN_SMEM(float, smem);
    const int LDX = LDXC ? LDXC : a.ldx, LD1 = LD1C ? LD1C : a.ld1, R = a.R;
    float* X = smem;               // [C][LDX]  lrelu(x), zero outside the row
    float* X1 = smem + C * LDX;    // [C][LD1]  lrelu(x1) of the current resblock, zero outside the row
    float* BS = X1 + C * LD1;      // [nrb][2][C] biases
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    // waves w and w+4 share a SIMD: give them complementary tile counts (wt < WT/2 gets the extra tile)
    const int wm = (WM == 1) ? 0 : (WM == 2 ? ((wid >> 1) & 1) : (((wid & 3) >> 1) + 2 * (wid >> 2)));
    const int wt = (WM == 1) ? wid : (WM == 2 ? ((wid >> 2) * 2 + (wid & 1)) : ((wid & 1) ^ (wid >> 2)));
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;

    for (int i = tid; i < a.nrb * 2 * C; i += 512) {
        const int j = i / (2 * C), q = (i / C) & 1, c = i % C;
        BS[i] = a.bias[j][q][c];
    }
    if (!(LAB_ABLATE(a) & 2)) stage_tile_pk<512>(a.x + (long)b * a.x_bs, a.x_ld, C, LDX, t0 - R, len, 0.1f, X, a.vec, XS);
    __syncthreads();

    f32x16 out[NT2MAX];
    MI355_UNROLL
    for (int i = 0; i < NT2MAX; ++i)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) out[i][r] = 0.0f;
    const int nt2 = (N2 - wt + WT - 1) / WT;  // output tiles of this wave (wave-uniform)
    f32x16 acc1[NT1MAX];
    int nt1 = 0;

    auto conv1_compute = [&](int j) {
        const int K = a.k[j], d1 = a.d1[j];
        const int r1 = (K - 1) / 2 * d1, r2 = (K - 1) / 2 * a.d2[j];
        const int n1 = (T_B + 2 * r2 + 31) / 32;  // conv1 column tiles: extended column e <-> t = t0 - r2 + e
        nt1 = n1 > wt ? (n1 - wt + WT - 1) / WT : 0;
        const float* wp = a.w[j][0];
        const float* bs = BS + (j * 2 + 0) * C;
        if (nt1 >= 3) mrf_conv1_compute<MATH, 3, NT1MAX, CP, WT>(acc1, wp, bs, X, LDX, R, r1, r2, K, d1, wm, wt, brow, bcol, LAB_ABLATE(a));
        else if (nt1 == 2) mrf_conv1_compute<MATH, 2, NT1MAX, CP, WT>(acc1, wp, bs, X, LDX, R, r1, r2, K, d1, wm, wt, brow, bcol, LAB_ABLATE(a));
        else if (nt1 == 1) mrf_conv1_compute<MATH, 1, NT1MAX, CP, WT>(acc1, wp, bs, X, LDX, R, r1, r2, K, d1, wm, wt, brow, bcol, LAB_ABLATE(a));
    };

    conv1_compute(0);
    for (int j = 0; j < a.nrb; ++j) {
        const int K = a.k[j], d2 = a.d2[j];
        const int r2 = (K - 1) / 2 * d2;
        if (j > 0) __syncthreads();  // every wave is done reading the previous resblock's x1
        // ---- conv1 epilogue: x1 (zero outside the row) -> LDS
        MI355_UNROLL
        for (int i = 0; i < NT1MAX; ++i) {
            if (i < nt1) {
                const int e = (wt + WT * i) * 32 + bcol;
                const int t = t0 - r2 + e;
                const bool live = t >= 0 && t < len;
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
                    X1[pk(co, e, LD1)] = live ? fmaxf(acc1[i][r], 0.1f * acc1[i][r]) * XS : 0.0f;
                }
            }
        }
        __syncthreads();
        // ---- conv2 into the output registers, then straight on to the next resblock's conv1
        {
            const float* wp = a.w[j][1];
            const float* bs = BS + (j * 2 + 1) * C;
            if (nt2 >= 2) mrf_conv2<MATH, 2, NT2MAX, CP, WT>(out, wp, bs, X1, LD1, r2, K, d2, wm, wt, brow, bcol, LAB_ABLATE(a));
            else if (nt2 == 1) mrf_conv2<MATH, 1, NT2MAX, CP, WT>(out, wp, bs, X1, LD1, r2, K, d2, wm, wt, brow, bcol, LAB_ABLATE(a));
        }
        if (j + 1 < a.nrb) conv1_compute(j + 1);
    }

    const float n = (float)a.nrb;
    auto store_all = [&](auto MEAN) {  // the mean / scale choice once, outside the loops
        MI355_UNROLL
        for (int i = 0; i < NT2MAX; ++i) {
            const int t = t0 + (wt + WT * i) * 32 + bcol;
            if (i < nt2 && t < a.T && !(LAB_ABLATE(a) & 4)) {
                float* yp = a.y + (long)b * a.y_bs + (long)(wm * 32 + 4 * brow) * a.y_ld + t;
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const float o = out[i][r] * OUT_UNSCALE;  // exact
                    yp[(long)((r & 3) + 8 * (r >> 2)) * a.y_ld] = decltype(MEAN)::value ? o / n : o * a.out_scale;
                }
            }
        }
    };
    if (a.out_scale > 0.0f) store_all(std::false_type{});
    else store_all(std::true_type{});
}

namespace {
// geometry per channel count:  C = 32: 1 x 8 waves, T_B = 512 (16 tiles, 2 per wave);  C = 64: 2 x 4 waves, T_B = 192;
// C = 128: 4 x 2 waves, T_B = 96
struct Geo { int C, T_B, WT, NT1MAX; };
inline bool geometry(int C, Geo* g) {
    if (C == 32) { *g = {32, 512, 8, 3}; return true; }
    if (C == 64) { *g = {64, 192, 4, 3}; return true; }
    if (C == 128) { *g = {128, 96, 2, 3}; return true; }  // fits only the narrow resblocks (halo <= 16): see the engine
    return false;
}
constexpr size_t LDS_LIMIT = 160 * 1024;
}  // namespace

bool mrf_fused_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    Geo g;
    if (!geometry(C, &g) || nrb < 1 || nrb > MRF_MAX_RB) return false;
    int R = 0, r2max = 0;
    for (int j = 0; j < nrb; ++j) {
        if (k[j] < 1 || (k[j] % 2) == 0 || d1[j] < 1 || d2[j] < 1) return false;
        const int r1 = (k[j] - 1) / 2 * d1[j], r2 = (k[j] - 1) / 2 * d2[j];
        R = R > r1 + r2 ? R : r1 + r2;
        r2max = r2max > r2 ? r2max : r2;
        const int n1 = (g.T_B + 2 * r2 + 31) / 32;
        if ((n1 + g.WT - 1) / g.WT > g.NT1MAX) return false;
    }
    const int Rp = (R + 3) & ~3;
    const size_t ldx = (size_t)g.T_B + 2 * Rp + 32;
    const size_t ld1 = (size_t)((g.T_B + 2 * r2max + 31) / 32) * 32;
    return ((size_t)C * (ldx + ld1) + (size_t)nrb * 2 * C) * sizeof(float) <= LDS_LIMIT;
}

void launch_mrf_fused(MrfArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    Geo g;
    if (!geometry(a.C, &g) || !mrf_fused_supported(a.C, a.nrb, a.k, a.d1, a.d2))
        throw std::runtime_error("mrf_fused: unsupported stage shape");
    int R = 0, r2max = 0;
    for (int j = 0; j < a.nrb; ++j) {
        const int r1 = (a.k[j] - 1) / 2 * a.d1[j], r2 = (a.k[j] - 1) / 2 * a.d2[j];
        R = R > r1 + r2 ? R : r1 + r2;
        r2max = r2max > r2 ? r2max : r2;
    }
    a.R = (R + 3) & ~3;               // staging halo rounded up to 4: the tile starts on a 16-byte boundary
    a.ldx = g.T_B + 2 * a.R + 32;     // +32: conv1's last (rounded-up) column tile stays inside its row
    a.ld1 = ((g.T_B + 2 * r2max + 31) / 32) * 32;
    a.vec = (a.x_ld % 4 == 0) && (a.x_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0);
    {
        const char* ab = lab_getenv("MI355VITS_MRF_ABLATE");
        a.ablate = ab ? atoi(ab) : 0;
    }
    const size_t shmem = ((size_t)a.C * (a.ldx + a.ld1) + (size_t)a.nrb * 2 * a.C) * sizeof(float);
    dim3 grid((a.T + g.T_B - 1) / g.T_B, a.B);
    auto go = [&](auto kfn) {
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), (int)LDS_LIMIT);
        LAUNCH_KERNEL(kfn, grid, dim3(512), shmem, s, a);
    };
    if (a.math == MATH_F16X2) {
        if (a.C == 32) {
            if (a.ldx == 640 && a.ld1 == 608) go(k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 3>);
            else go(k_mrf_fused<1, 8, 16, 3, 2, 0, 0, 3>);
        } else if (a.C == 128) {
            if (a.ldx == 160 && a.ld1 == 128) go(k_mrf_fused<4, 2, 3, 3, 2, 160, 128, 3>);
            else go(k_mrf_fused<4, 2, 3, 3, 2, 0, 0, 3>);
        } else {
            if (a.ldx == 320 && a.ld1 == 288) go(k_mrf_fused<2, 4, 6, 3, 2, 320, 288, 3>);
            else go(k_mrf_fused<2, 4, 6, 3, 2, 0, 0, 3>);
        }
        return;
    }
    if (a.math == MATH_BF16X3) {
        if (a.C == 32) {
            if (a.ldx == 640 && a.ld1 == 608) go(k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 1>);
            else go(k_mrf_fused<1, 8, 16, 3, 2, 0, 0, 1>);
        } else if (a.C == 128) {
            if (a.ldx == 160 && a.ld1 == 128) go(k_mrf_fused<4, 2, 3, 3, 2, 160, 128, 1>);
            else go(k_mrf_fused<4, 2, 3, 3, 2, 0, 0, 1>);
        } else {
            if (a.ldx == 320 && a.ld1 == 288) go(k_mrf_fused<2, 4, 6, 3, 2, 320, 288, 1>);
            else go(k_mrf_fused<2, 4, 6, 3, 2, 0, 0, 1>);
        }
        return;
    }
    if (a.math == MATH_BF16W) {  // generic row pitches only: this mode is a measured variant, not the product default
        if (a.C == 32) go(k_mrf_fused<1, 8, 16, 3, 2, 0, 0, 2>);
        else if (a.C == 128) go(k_mrf_fused<4, 2, 3, 3, 2, 0, 0, 2>);
        else go(k_mrf_fused<2, 4, 6, 3, 2, 0, 0, 2>);
        return;
    }
    if (a.C == 32) {
        if (a.ldx == 640 && a.ld1 == 608) go(k_mrf_fused<1, 8, 16, 3, 2, 640, 608>);
        else go(k_mrf_fused<1, 8, 16, 3, 2>);
    } else if (a.C == 128) {
        if (a.ldx == 160 && a.ld1 == 128) go(k_mrf_fused<4, 2, 3, 3, 2, 160, 128>);
        else go(k_mrf_fused<4, 2, 3, 3, 2>);
    } else {
        if (a.ldx == 320 && a.ld1 == 288) go(k_mrf_fused<2, 4, 6, 3, 2, 320, 288>);
        else go(k_mrf_fused<2, 4, 6, 3, 2>);
    }
}

}  // namespace m355
