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
// POST (32 channels, the last stage): the workgroup's 512 columns of mean(resblocks) go to LDS (the x tile's space)
// instead of global memory, and 504 of them — the tile minus conv_post's halo of 3 (4, to keep the staging aligned) per
// side — become audio samples: leaky-relu(0.01), 7-tap 32 -> 1 conv in the order of k_conv_post_tanh_vec (same bits),
// tanh, per-row peak.  Saves the stage output's write, its re-read and a launch.
template <int WM, int WT, int N2, int NT1MAX, int NT2MAX, int LDXC = 0, int LD1C = 0, int MATH = 0, bool POST = false>
__global__ __launch_bounds__(512) void k_mrf_fused(MrfArgs a) {
    static_assert(WM * WT == 8, "8 waves per workgroup");
    static_assert(!POST || WM == 1, "conv_post is fused behind the 32-channel stage only");
    static_assert(!POST || MATH != 3, "MATH_F16X2 keeps its output accumulators scaled: no conv_post fusion");
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
    const int t0 = POST ? blockIdx.x * (T_B - 8) - 4 : blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;

    for (int i = tid; i < a.nrb * 2 * C; i += 512) {
        const int j = i / (2 * C), q = (i / C) & 1, c = i % C;
        BS[i] = a.bias[j][q][c];
    }
    // POST: conv_post's weights as [C][8] (7 taps + a zero; 32-byte rows, 16-byte aligned: two broadcast ds_read_b128 per channel)
    float* PW = BS + ((a.nrb * 2 * C + 3) & ~3);
    if constexpr (POST) {
        if (tid < C * 8) PW[tid] = (tid & 7) < MRF_POST_K ? a.post_w[(tid >> 3) * MRF_POST_K + (tid & 7)] : 0.0f;
    }
    if (!(a.ablate & 2)) stage_tile_pk<512>(a.x + (long)b * a.x_bs, a.x_ld, C, LDX, t0 - R, len, 0.1f, X, a.vec, XS);
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
        if (nt1 >= 3) mrf_conv1_compute<MATH, 3, NT1MAX, CP, WT>(acc1, wp, bs, X, LDX, R, r1, r2, K, d1, wm, wt, brow, bcol, a.ablate);
        else if (nt1 == 2) mrf_conv1_compute<MATH, 2, NT1MAX, CP, WT>(acc1, wp, bs, X, LDX, R, r1, r2, K, d1, wm, wt, brow, bcol, a.ablate);
        else if (nt1 == 1) mrf_conv1_compute<MATH, 1, NT1MAX, CP, WT>(acc1, wp, bs, X, LDX, R, r1, r2, K, d1, wm, wt, brow, bcol, a.ablate);
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
            if (nt2 >= 2) mrf_conv2<MATH, 2, NT2MAX, CP, WT>(out, wp, bs, X1, LD1, r2, K, d2, wm, wt, brow, bcol, a.ablate);
            else if (nt2 == 1) mrf_conv2<MATH, 1, NT2MAX, CP, WT>(out, wp, bs, X1, LD1, r2, K, d2, wm, wt, brow, bcol, a.ablate);
        }
        if (j + 1 < a.nrb) conv1_compute(j + 1);
    }

    const float n = (float)a.nrb;
    if constexpr (POST) {
        // X is free: its last readers (conv1 of the last resblock) passed the barrier in front of the last conv2
        constexpr int YLD = T_B + 8;  // +8: the two half-waves (rows 4 apart) land on different banks
        float* Y = X;
        int vl = a.audio_len ? a.audio_len[b] : a.T;
        if (vl > a.T) vl = a.T;
        auto put = [&](auto MEAN) {  // the mean / scale choice once, outside the loops
            MI355_UNROLL
            for (int i = 0; i < NT2MAX; ++i) {
                if (i < nt2) {
                    const int c0 = (wt + WT * i) * 32 + bcol;
                    const int t = t0 + c0;
                    const bool live = t >= 0 && t < vl;
                    MI355_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int co = (r & 3) + 8 * (r >> 2) + 4 * brow;
                        const float v = decltype(MEAN)::value ? out[i][r] / n : out[i][r] * a.out_scale;
                        Y[co * YLD + c0 + 1] = live ? (v >= 0.0f ? v : v * 0.01f) : 0.0f;
                    }
                }
            }
        };
        if (a.out_scale > 0.0f) put(std::false_type{});
        else put(std::true_type{});
        __syncthreads();
        // two consecutive output samples per thread, t = t0 + 4 + q and the next (q even, q < T_B - 8): their eight input
        // columns q + 1 .. q + 8 are four 8-byte LDS reads per channel (Y is stored one column to the right: 8-byte aligned)
        const int q = 2 * tid;
        const int t = t0 + 4 + q;
        float pk = 0.0f;
        if (q < T_B - 8 && t < a.T && !(a.ablate & 4)) {
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const float2* yr = reinterpret_cast<const float2*>(Y + c * YLD + q + 2);
                const float2 p0 = yr[0], p1 = yr[1], p2 = yr[2], p3 = yr[3];
                const float yv[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
                const float4 w0 = reinterpret_cast<const float4*>(PW)[2 * c], w1 = reinterpret_cast<const float4*>(PW)[2 * c + 1];
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                MI355_UNROLL
                for (int k = 0; k < MRF_POST_K; ++k) {
                    acc0 = fmaf(wv[k], yv[k], acc0);
                    acc1 = fmaf(wv[k], yv[k + 1], acc1);
                }
            }
            const float y0 = tanhf(acc0), y1 = tanhf(acc1);
            if (t < vl) pk = fabsf(y0);
            if (t + 1 < vl) pk = fmaxf(pk, fabsf(y1));
            float* ap = a.audio + (long)b * a.audio_bs + t;
            if (t + 1 < a.T && (a.audio_bs & 1) == 0) {
                *reinterpret_cast<float2*>(ap) = make_float2(y0, y1);
            } else {
                ap[0] = y0;
                if (t + 1 < a.T) ap[1] = y1;
            }
        }
        pk = wave_reduce_max(pk);
        if (lane == 0) BS[wid] = pk;  // the biases are no longer needed
        __syncthreads();
        if (tid == 0) {
            float m = BS[0];
            MI355_UNROLL
            for (int w8 = 1; w8 < 8; ++w8) m = fmaxf(m, BS[w8]);
            atomicMax(a.peak_bits + b, __float_as_uint(m));
        }
        return;
    }
    auto store_all = [&](auto MEAN) {  // the mean / scale choice once, outside the loops
        MI355_UNROLL
        for (int i = 0; i < NT2MAX; ++i) {
            const int t = t0 + (wt + WT * i) * 32 + bcol;
            if (i < nt2 && t < a.T && !(a.ablate & 4)) {
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

bool mrf_fused_post_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    return C == 32 && mrf_fused_supported(C, nrb, k, d1, d2);
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
        const char* ab = getenv("MI355VITS_MRF_ABLATE");
        a.ablate = ab ? atoi(ab) : 0;
    }
    const bool post = a.post_w != nullptr;
    const size_t shmem = ((size_t)a.C * (a.ldx + a.ld1) + (size_t)a.nrb * 2 * a.C + (post ? a.C * 8 + 4 : 0)) * sizeof(float);
    if (post && !(a.C == 32 && a.audio && a.peak_bits && (size_t)a.C * (g.T_B + 8) <= (size_t)a.C * a.ldx && shmem <= LDS_LIMIT))
        throw std::runtime_error("mrf_fused: conv_post fusion needs the 32-channel stage");
    dim3 grid(post ? (a.T + g.T_B - 9) / (g.T_B - 8) : (a.T + g.T_B - 1) / g.T_B, a.B);
    auto go = [&](auto kfn) {
#ifndef MI355_EMU
        static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        (void)once;
#endif
        LAUNCH_KERNEL(kfn, grid, dim3(512), shmem, s, a);
    };
    if (post) {  // generic row pitches except for the default math on the "_low" shapes
        if (a.math == MATH_BF16X3) {
            if (a.ldx == 640 && a.ld1 == 608) go(k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 1, true>);
            else go(k_mrf_fused<1, 8, 16, 3, 2, 0, 0, 1, true>);
        } else if (a.math == MATH_BF16W) {
            go(k_mrf_fused<1, 8, 16, 3, 2, 0, 0, 2, true>);
        } else {
            if (a.ldx == 640 && a.ld1 == 608) go(k_mrf_fused<1, 8, 16, 3, 2, 640, 608, 0, true>);
            else go(k_mrf_fused<1, 8, 16, 3, 2, 0, 0, 0, true>);
        }
        return;
    }
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

// ------------------------------------------------------------------------------------------------
// The fused stage in MATH_BF16X3 with the split done once per element: both LDS tiles hold three bf16 PLANES
// ([plane][16-channel group][half][column] x 16 B, b3.h) instead of f32 values, so the matrix-core loops carry no VALU
// work at all (on the fly the split costs as much issue time as the MFMAs it feeds: tools/bf16x3_probe, V3 vs V5).
//   * x (+halo) is split while it is staged (stage_planes), x1 in conv1's epilogue — a lane's 16 rows of a 32 x 32 tile
//     are exactly two 16-byte plane records (eight channels each), so the epilogue writes 6 ds_write_b128 per tile and
//     the residual / accumulator preload reads 6 ds_read_b128 (x = h + m + l exactly);
//   * four waves, one per SIMD with the whole register file: a wave owns a contiguous run of up to NT1MAX column tiles
//     of one 32-row tile, every weight fragment feeds all of them (b3_chunk);
//   * 6 B per element instead of 4: T_B = 256 columns for 32 channels, 128 for 64 (halo 45 + 36 per side).
// ------------------------------------------------------------------------------------------------
// a lane's rows r = 0..7 (a = 0, 1) of row tile wm are the eight k-slots of record [group 2 wm][half brow], rows 8..15 of
// [group 2 wm + 1][half brow]: value of row r = plane sums of slot (r & 3) + 4 ((r >> 2) & 1)
__device__ __forceinline__ void planes_to_rows(const uint4* __restrict__ P, int PS, int LD, int wm, int brow, int col, float (&v)[16]) {
    MI355_UNROLL
    for (int hg = 0; hg < 2; ++hg) {
        const int o = ((2 * wm + hg) * 2 + brow) * LD + col;
        const uint4 h = P[o], m = P[PS + o], l = P[2 * PS + o];
        const unsigned hh[4] = {h.x, h.y, h.z, h.w}, mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {l.x, l.y, l.z, l.w};
        MI355_UNROLL
        for (int q = 0; q < 4; ++q) {  // slots 2q, 2q + 1
            v[8 * hg + 2 * q] = (__uint_as_float(hh[q] << 16) + __uint_as_float(mm[q] << 16)) + __uint_as_float(ll[q] << 16);
            v[8 * hg + 2 * q + 1] = (__uint_as_float(hh[q] & 0xffff0000u) + __uint_as_float(mm[q] & 0xffff0000u)) + __uint_as_float(ll[q] & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ void rows_to_planes(uint4* __restrict__ P, int PS, int LD, int wm, int brow, int col, const float (&v)[16]) {
    MI355_UNROLL
    for (int hg = 0; hg < 2; ++hg) {
        uint4 h, m, l;
        split3_pk(v[8 * hg + 0], v[8 * hg + 1], h.x, m.x, l.x);
        split3_pk(v[8 * hg + 2], v[8 * hg + 3], h.y, m.y, l.y);
        split3_pk(v[8 * hg + 4], v[8 * hg + 5], h.z, m.z, l.z);
        split3_pk(v[8 * hg + 6], v[8 * hg + 7], h.w, m.w, l.w);
        const int o = ((2 * wm + hg) * 2 + brow) * LD + col;
        P[o] = h;
        P[PS + o] = m;
        P[2 * PS + o] = l;
    }
}
// accumulator register r of a lane <-> slot order of the two records: r = 4 a + m  ->  record a >> 1, slot 4 (a & 1) + m
__device__ __forceinline__ int rec_index(int r) { return 8 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3); }

template <int C, int WM, int WT, int N2, int NT1MAX>
__global__ __launch_bounds__(256) void k_mrf_b3(MrfArgs a) {
    static_assert(WM * WT == 4 && C == 32 * WM, "4 waves: WM row tiles x WT column runs");
    static_assert(N2 % WT == 0, "output column tiles divide evenly");
    constexpr int NG = C / 16, T_B = 32 * N2, NT2 = N2 / WT;
    DYN_SMEM(float, smem);
    const int LDX = a.ldx, LD1 = a.ld1, R = a.R;
    const int PSX = NG * 2 * LDX, PS1 = NG * 2 * LD1;
    uint4* Xp = reinterpret_cast<uint4*>(smem);
    uint4* X1p = Xp + 3 * PSX;
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int wm = wid / WT, wt = wid % WT;
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;

    if (!(a.ablate & 2)) stage_planes<NG>(a.x + (long)b * a.x_bs, a.x_ld, LDX, t0 - R, len, 0.1f, Xp, PSX, tid, 256);
    __syncthreads();

    f32x16 out[1][NT2];
    MI355_UNROLL
    for (int i = 0; i < NT2; ++i)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) out[0][i][r] = 0.0f;
    // biases travel one conv ahead of their use (16 loads in flight under the previous conv's matrix-core loop): a wave
    // alone on its SIMD has nothing else to hide an L2 round trip per conv behind
    float bias_n[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) bias_n[r] = a.bias[0][0][32 * wm + (r & 3) + 8 * (r >> 2) + 4 * brow];

    for (int j = 0; j < a.nrb; ++j) {
        const int K = a.k[j], d1 = a.d1[j], d2 = a.d2[j];
        const int r1 = (K - 1) / 2 * d1, r2 = (K - 1) / 2 * d2;
        // ---- conv1 over the extended range e in [0, T_B + 2 r2): column tile q <-> t = t0 - r2 + 32 q + bcol; this wave: a
        // contiguous run of cnt tiles from `start`
        const int n1 = (T_B + 2 * r2 + 31) / 32;
        const int base = n1 / WT, rem = n1 % WT;
        const int cnt = base + (wt < rem ? 1 : 0), start = wt * base + (wt < rem ? wt : rem);
        f32x16 acc1[1][NT1MAX];
        float bias[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bias[r] = bias_n[r];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bias_n[r] = a.bias[j][1][32 * wm + (r & 3) + 8 * (r >> 2) + 4 * brow];  // conv2's, for later
        MI355_UNROLL
        for (int i = 0; i < NT1MAX; ++i) {
            if (i < cnt) {
                float v[16];
                planes_to_rows(Xp, PSX, LDX, wm, brow, (R - r2) + (start + i) * 32 + bcol, v);
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) acc1[0][i][r] = unlrelu(v[rec_index(r)]) + bias[r];
            }
        }
        {
            const uint4* wp[1] = {reinterpret_cast<const uint4*>(a.w[j][0]) + (long)wm * K * NG * 192 + lane};
            const uint4* xq = Xp + brow * LDX + bcol + (R - r2 - r1) + start * 32;
            if (!(a.ablate & 1)) {
                if (cnt >= 4 && NT1MAX >= 4) b3_chunk<1, (NT1MAX >= 4 ? 4 : NT1MAX), NG, NT1MAX>(acc1, wp, xq, PSX, LDX, K, NG, d1);
                else if (cnt == 3 && NT1MAX >= 3) b3_chunk<1, (NT1MAX >= 3 ? 3 : NT1MAX), NG, NT1MAX>(acc1, wp, xq, PSX, LDX, K, NG, d1);
                else if (cnt == 2) b3_chunk<1, 2, NG, NT1MAX>(acc1, wp, xq, PSX, LDX, K, NG, d1);
                else if (cnt == 1) b3_chunk<1, 1, NG, NT1MAX>(acc1, wp, xq, PSX, LDX, K, NG, d1);
            }
        }
        if (j > 0) __syncthreads();  // every wave is done reading the previous resblock's x1
        // ---- conv1 epilogue: x1 (zero outside the row), leaky-relu, split -> planes
        MI355_UNROLL
        for (int i = 0; i < NT1MAX; ++i) {
            if (i < cnt) {
                const int e = (start + i) * 32 + bcol;
                const int t = t0 - r2 + e;
                const bool live = t >= 0 && t < len;
                float v[16];
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const float x1 = acc1[0][i][r];
                    v[rec_index(r)] = live ? fmaxf(x1, 0.1f * x1) : 0.0f;
                }
                if (e < LD1) rows_to_planes(X1p, PS1, LD1, wm, brow, e, v);
            }
        }
        __syncthreads();
        // ---- conv2 into the output registers: out += x1 + bias + conv(lrelu(x1)); this wave: tiles wt * NT2 ..
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bias[r] = bias_n[r];
        {
            const int jn = j + 1 < a.nrb ? j + 1 : j;  // next resblock's conv1 bias (the last one re-reads its own)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) bias_n[r] = a.bias[jn][0][32 * wm + (r & 3) + 8 * (r >> 2) + 4 * brow];
        }
        MI355_UNROLL
        for (int i = 0; i < NT2; ++i) {
            float v[16];
            planes_to_rows(X1p, PS1, LD1, wm, brow, r2 + (wt * NT2 + i) * 32 + bcol, v);
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) out[0][i][r] += unlrelu(v[rec_index(r)]) + bias[r];
        }
        {
            const uint4* wp[1] = {reinterpret_cast<const uint4*>(a.w[j][1]) + (long)wm * K * NG * 192 + lane};
            if (!(a.ablate & 1)) b3_chunk<1, NT2, NG>(out, wp, X1p + brow * LD1 + bcol + wt * NT2 * 32, PS1, LD1, K, NG, d2);
        }
    }

    const float n = (float)a.nrb;
    MI355_UNROLL
    for (int i = 0; i < NT2; ++i) {
        const int t = t0 + (wt * NT2 + i) * 32 + bcol;
        if (t < a.T && !(a.ablate & 4)) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
                a.y[(long)b * a.y_bs + (long)co * a.y_ld + t] = a.out_scale > 0.0f ? out[0][i][r] * a.out_scale : out[0][i][r] / n;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 32 channels (the last stage: one 32-row tile, so a weight fragment feeds only the column tiles of its own wave, and
// eight waves fetching every fragment from L2 for every workgroup is more traffic than the CU's vector-memory path
// carries next to the matrix cores).  Here the weights of the running conv sit in LDS as well:
//   * LDS: x planes + x1 planes (pre-split, as k_mrf_b3) + a segment (<= kp taps) of the running conv's bf16x3 fragments
//     (2 groups x 3 KiB per tap); both MFMA operands are ds_read_b128, the loops carry no VALU and no global loads at
//     all, so a wave alone on its SIMD (whole register file) runs them back to back with one group of look-ahead;
//   * the next segment and the next bias travel into registers under the running loop (nothing in the loop waits on
//     the vector-memory counter) and are written to LDS between the barriers that separate two segments;
//   * four waves; conv2 computes 8 column tiles (2 per wave), conv1 the extended range tb + 2 r2 in 8..12 tiles dealt in
//     contiguous runs of 2 or 3; tb <= 256 is chosen so that the widest conv1 is a whole number of tiles
//     (the "_low" voices, r2 = 2 / 12 / 36: tb = 248, conv1 tiles 8 / 9 / 10).
// ------------------------------------------------------------------------------------------------
constexpr int MW_KMAX = 11, MW_N1MAX = 10, MW_WREGS = 8;  // taps per conv; conv1 column tiles; uint4 per thread and segment (kp <= 5)

__global__ __launch_bounds__(256) void k_mrf_b3w(MrfArgs a) {
    constexpr int NG = 2, NT2 = 2, NT1 = 3;
    DYN_SMEM(float, smem);
    const int LDX = a.ldx, LD1 = a.ld1, R = a.R, T_B = a.tb, KP = a.kp;
    const int PSX = NG * 2 * LDX, PS1 = NG * 2 * LD1;
    uint4* Xp = reinterpret_cast<uint4*>(smem);
    uint4* X1p = Xp + 3 * PSX;
    uint4* Wl = X1p + 3 * PS1;  // [taps of the segment][NG][3 planes][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wt = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * T_B;
    int len = a.len ? a.len[b] : a.T;
    if (len > a.T) len = a.T;

    // a conv of K taps runs in ceil(K / KP) segments of (nearly) equal length
    auto seg_taps = [&](int K) { const int np = (K + KP - 1) / KP; return (K + np - 1) / np; };
    uint4 wr[MW_WREGS];
    auto w_fetch = [&](const float* wsrc, int k0, int taps) {  // clamped: every load unconditional
        const uint4* src = reinterpret_cast<const uint4*>(wsrc) + (long)k0 * NG * 192;
        const int n = taps * NG * 192;
        MI355_UNROLL
        for (int i = 0; i < MW_WREGS; ++i) {
            const int idx = tid + 256 * i;
            wr[i] = src[idx < n ? idx : n - 1];
        }
    };
    auto w_store = [&](int taps) {
        const int n = taps * NG * 192;
        MI355_UNROLL
        for (int i = 0; i < MW_WREGS; ++i) {
            const int idx = tid + 256 * i;
            if (idx < n) Wl[idx] = wr[i];
        }
    };
    auto bias_load = [&](const float* bp, float (&bv)[16]) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bv[r] = bp[(r & 3) + 8 * (r >> 2) + 4 * brow];
    };
    // fetch the segment that follows (j, q, k0 .. k0 + taps): the same conv's next taps, the other conv, the next resblock
    auto fetch_next = [&](int j, int q, int k0, int taps) {
        const int K = a.k[j];
        if (k0 + taps < K) {
            const int st = seg_taps(K);
            w_fetch(a.w[j][q], k0 + taps, K - (k0 + taps) < st ? K - (k0 + taps) : st);
        } else if (q == 0) {
            w_fetch(a.w[j][1], 0, seg_taps(K));
        } else {
            const int jn = j + 1 < a.nrb ? j + 1 : j;  // after the last conv: a harmless re-read
            w_fetch(a.w[jn][0], 0, seg_taps(a.k[jn]));
        }
    };
    auto next_taps = [&](int j, int q, int k0, int taps) {
        const int K = a.k[j];
        if (k0 + taps < K) { const int st = seg_taps(K); return K - (k0 + taps) < st ? K - (k0 + taps) : st; }
        if (q == 0) return seg_taps(K);
        return seg_taps(a.k[j + 1 < a.nrb ? j + 1 : j]);
    };

    w_fetch(a.w[0][0], 0, seg_taps(a.k[0]));
    if (!(a.ablate & 2)) stage_planes<NG>(a.x + (long)b * a.x_bs, a.x_ld, LDX, t0 - R, len, 0.1f, Xp, PSX, tid, 256);
    w_store(seg_taps(a.k[0]));
    float bias_n[16];
    bias_load(a.bias[0][0], bias_n);
    __syncthreads();

    f32x16 out[1][NT2];
    MI355_UNROLL
    for (int i = 0; i < NT2; ++i)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) out[0][i][r] = 0.0f;
    const uint4* wl[1] = {Wl + lane};

    for (int j = 0; j < a.nrb; ++j) {
        const int K = a.k[j], d1 = a.d1[j], d2 = a.d2[j];
        const int r1 = (K - 1) / 2 * d1, r2 = (K - 1) / 2 * d2;
        const int st = seg_taps(K);
        // ---- conv1 over the extended range e in [0, tb + 2 r2): column tile q <-> t = t0 - r2 + 32 q + bcol; this wave: a
        // contiguous run of cnt (2 or 3) tiles from `start`
        const int n1 = (T_B + 2 * r2 + 31) / 32;
        const int base = n1 >> 2, rem = n1 & 3;
        const int cnt = base + (wt < rem ? 1 : 0), start = wt * base + (wt < rem ? wt : rem);
        float bias[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bias[r] = bias_n[r];
        bias_load(a.bias[j][1], bias_n);
        f32x16 acc1[1][NT1];
        MI355_UNROLL
        for (int i = 0; i < NT1; ++i) {
            if (i < cnt) {
                float v[16];
                planes_to_rows(Xp, PSX, LDX, 0, brow, (R - r2) + (start + i) * 32 + bcol, v);
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) acc1[0][i][r] = unlrelu(v[rec_index(r)]) + bias[r];
            }
        }
        for (int k0 = 0; k0 < K; k0 += st) {
            const int taps = K - k0 < st ? K - k0 : st;
            fetch_next(j, 0, k0, taps);
            const uint4* xq = Xp + brow * LDX + bcol + (R - r2 - r1) + start * 32 + k0 * d1;
            if (!(a.ablate & 1)) {
                if (cnt >= 3) b3_chunk<1, 3, NG, NT1>(acc1, wl, xq, PSX, LDX, taps, NG, d1);
                else if (cnt == 2) b3_chunk<1, 2, NG, NT1>(acc1, wl, xq, PSX, LDX, taps, NG, d1);
                else if (cnt == 1) b3_chunk<1, 1, NG, NT1>(acc1, wl, xq, PSX, LDX, taps, NG, d1);
            }
            __syncthreads();  // every wave is done with this segment (last one: and with the previous resblock's x1)
            if (k0 + taps >= K) {
                // conv1 epilogue: x1 (zero outside the row), leaky-relu, split -> planes
                MI355_UNROLL
                for (int i = 0; i < NT1; ++i) {
                    if (i < cnt) {
                        const int e = (start + i) * 32 + bcol;
                        const int t = t0 - r2 + e;
                        const bool live = t >= 0 && t < len;
                        float v[16];
                        MI355_UNROLL
                        for (int r = 0; r < 16; ++r) {
                            const float x1 = acc1[0][i][r];
                            v[rec_index(r)] = live ? fmaxf(x1, 0.1f * x1) : 0.0f;
                        }
                        if (e < LD1) rows_to_planes(X1p, PS1, LD1, 0, brow, e, v);
                    }
                }
            }
            w_store(next_taps(j, 0, k0, taps));
            __syncthreads();
        }
        // ---- conv2 into the output registers: out += x1 + bias + conv(lrelu(x1)); this wave: output tiles 2 wt, 2 wt + 1
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bias[r] = bias_n[r];
        bias_load(a.bias[j + 1 < a.nrb ? j + 1 : j][0], bias_n);
        MI355_UNROLL
        for (int i = 0; i < NT2; ++i) {
            float v[16];
            planes_to_rows(X1p, PS1, LD1, 0, brow, r2 + (wt * NT2 + i) * 32 + bcol, v);
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) out[0][i][r] += unlrelu(v[rec_index(r)]) + bias[r];
        }
        for (int k0 = 0; k0 < K; k0 += st) {
            const int taps = K - k0 < st ? K - k0 : st;
            const bool last_seg = j + 1 == a.nrb && k0 + taps >= K;
            if (!last_seg) fetch_next(j, 1, k0, taps);
            if (!(a.ablate & 1)) b3_chunk<1, NT2, NG>(out, wl, X1p + brow * LD1 + bcol + wt * NT2 * 32 + k0 * d2, PS1, LD1, taps, NG, d2);
            if (!last_seg) {
                __syncthreads();
                w_store(next_taps(j, 1, k0, taps));
                __syncthreads();
            }
        }
    }

    const bool mean = !(a.out_scale > 0.0f);
    const float n = (float)a.nrb;
    MI355_UNROLL
    for (int i = 0; i < NT2; ++i) {
        const int c0 = (wt * NT2 + i) * 32 + bcol;
        const int t = t0 + c0;
        if (c0 < T_B && t < a.T && !(a.ablate & 4)) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * brow;
                a.y[(long)b * a.y_bs + (long)co * a.y_ld + t] = mean ? out[0][i][r] / n : out[0][i][r] * a.out_scale;
            }
        }
    }
}

namespace {
struct GeoB3 { int C, T_B, WT, NT1MAX; };
inline bool geometry_b3(int C, GeoB3* g) {
    if (C == 32) { *g = {32, 256, 4, 3}; return true; }
    if (C == 64) { *g = {64, 128, 2, 4}; return true; }
    return false;
}
inline void shape_b3(const GeoB3& g, int nrb, const int* k, const int* d1, const int* d2, int* R, int* ldx, int* ld1, bool* ok) {
    int Rm = 0, r2max = 0;
    *ok = true;
    for (int j = 0; j < nrb; ++j) {
        if (k[j] < 1 || (k[j] % 2) == 0 || d1[j] < 1 || d2[j] < 1) { *ok = false; return; }
        const int r1 = (k[j] - 1) / 2 * d1[j], r2 = (k[j] - 1) / 2 * d2[j];
        Rm = Rm > r1 + r2 ? Rm : r1 + r2;
        r2max = r2max > r2 ? r2max : r2;
        const int n1 = (g.T_B + 2 * r2 + 31) / 32;
        if ((n1 + g.WT - 1) / g.WT > g.NT1MAX) *ok = false;
    }
    *R = (Rm + 3) & ~3;  // the staged window starts on a 16-byte boundary
    *ldx = (g.T_B + *R + Rm + 3) & ~3;
    *ld1 = (g.T_B + 2 * r2max + 3) & ~3;
}
}  // namespace

namespace {
// geometry of k_mrf_b3w: halo R, row pitches, output columns per workgroup, LDS bytes
inline bool shape_b3w(int nrb, const int* k, const int* d1, const int* d2, int* R, int* ldx, int* ld1, int* tb, int* kp, size_t* lds) {
    int Rm = 0, r2max = 0, over = 0, kmax = 1;
    for (int j = 0; j < nrb; ++j) {
        if (k[j] < 1 || (k[j] % 2) == 0 || k[j] > MW_KMAX || d1[j] < 1 || d2[j] < 1) return false;
        const int r1 = (k[j] - 1) / 2 * d1[j], r2 = (k[j] - 1) / 2 * d2[j];
        Rm = Rm > r1 + r2 ? Rm : r1 + r2;
        r2max = r2max > r2 ? r2max : r2;
        over = over > r1 - r2 ? over : r1 - r2;
        kmax = kmax > k[j] ? kmax : k[j];
    }
    // output columns: at most 8 tiles (2 per wave), and the widest conv1 (tb + 2 max r2) a whole number of tiles <= 10
    int t = 32 * MW_N1MAX - 2 * r2max;
    if (t > 256) t = 256;
    if (t < 128) return false;
    const int n1max = (t + 2 * r2max + 31) / 32;
    *R = Rm;
    *tb = t;
    *ldx = (Rm + 32 * n1max + over + 3) & ~3;
    *ld1 = (32 * ((t + 31) / 32) + 2 * r2max + 3) & ~3;
    const size_t planes = (size_t)192 * (*ldx + *ld1);
    if (planes + 2 * 6144 > LDS_LIMIT) return false;
    int p = (int)((LDS_LIMIT - planes) / 6144);  // taps of weight fragments that fit next to the planes
    if (p > 5) p = 5;                            // MW_WREGS uint4 per thread = 5 taps
    if (p > kmax) p = kmax;
    *kp = p;
    *lds = planes + (size_t)p * 6144;
    return true;
}
// opt-in (MI355VITS_MRF_B3W=1): measured 5.4 ms against the on-the-fly kernel's 2.9 ms on the bench workload — with one
// 32-row tile every MFMA needs 0.67 KiB of LDS operands (the loops run at the LDS port's pace, not the matrix cores'),
// the 248-column tile makes ten short weight segments whose L2 fetches are exposed, and the halo costs 11 % more tiles
inline bool no_b3w() { return getenv("MI355VITS_MRF_B3W") == nullptr; }  // read per call: tests flip it inside one process
}  // namespace

bool mrf_b3w_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    int R, ldx, ld1, tb, kp;
    size_t lds;
    return C == 32 && nrb >= 1 && nrb <= MRF_MAX_RB && !no_b3w() && shape_b3w(nrb, k, d1, d2, &R, &ldx, &ld1, &tb, &kp, &lds);
}

bool mrf_b3_supported(int C, int nrb, const int* k, const int* d1, const int* d2) {
    if (mrf_b3w_supported(C, nrb, k, d1, d2)) return true;
    GeoB3 g;
    if (!geometry_b3(C, &g) || nrb < 1 || nrb > MRF_MAX_RB) return false;
    int R, ldx, ld1;
    bool ok;
    shape_b3(g, nrb, k, d1, d2, &R, &ldx, &ld1, &ok);
    return ok && (size_t)96 * (C / 16) * (ldx + ld1) <= LDS_LIMIT;
}

void launch_mrf_b3(MrfArgs a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    if (mrf_b3w_supported(a.C, a.nrb, a.k, a.d1, a.d2)) {
        size_t shmem;
        shape_b3w(a.nrb, a.k, a.d1, a.d2, &a.R, &a.ldx, &a.ld1, &a.tb, &a.kp, &shmem);
        const char* ab = getenv("MI355VITS_MRF_ABLATE");
        a.ablate = ab ? atoi(ab) : 0;
        dim3 grid((a.T + a.tb - 1) / a.tb, a.B);
#ifndef MI355_EMU
        static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mrf_b3w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        (void)once;
#endif
        LAUNCH_KERNEL(k_mrf_b3w, grid, dim3(256), shmem, s, a);
        return;
    }
    GeoB3 g;
    if (!geometry_b3(a.C, &g) || !mrf_b3_supported(a.C, a.nrb, a.k, a.d1, a.d2)) throw std::runtime_error("mrf_b3: unsupported stage shape");
    bool ok;
    shape_b3(g, a.nrb, a.k, a.d1, a.d2, &a.R, &a.ldx, &a.ld1, &ok);
    a.vec = (a.x_ld % 4 == 0) && (a.x_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0);
    {
        const char* ab = getenv("MI355VITS_MRF_ABLATE");
        a.ablate = ab ? atoi(ab) : 0;
    }
    const size_t shmem = (size_t)96 * (a.C / 16) * (a.ldx + a.ld1);
    dim3 grid((a.T + g.T_B - 1) / g.T_B, a.B);
    auto go = [&](auto kfn) {
#ifndef MI355_EMU
        static hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        (void)once;
#endif
        LAUNCH_KERNEL(kfn, grid, dim3(256), shmem, s, a);
    };
    if (a.C == 32) go(k_mrf_b3<32, 1, 4, 8, 3>);
    else go(k_mrf_b3<64, 2, 2, 4, 4>);
}

}  // namespace m355
