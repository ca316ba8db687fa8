// b3.h — device building blocks of the split-bf16 (MATH_BF16X3) kernels: staging an activation chunk as three bf16
// planes, and one wave's share of the implicit GEMM over such planes on v_mfma_f32_32x32x16_bf16.
// (the split itself: hipx.h, split3_pk; the weight fragment order: kernels.h, pack_conv_weights_bf16x3_mode layout 1)
#pragma once
#include "kernels.h"

namespace m355 {

// Fault injection of the CPU model only (tests/test_emu_engine.py::test_tight_bound_catches_a_dropped_partial_product): with
// MI355VITS_EMU_DROP = 1 / 2 / 3 the b3 loops leave out the l_w x h_x / h_w x l_x / m_w x m_x product (the three smallest of the six).
#ifdef MI355_EMU
static inline int b3_drop() {  // 0 none, 1 = l_w x h_x, 2 = h_w x l_x, 3 = m_w x m_x
    const char* e = getenv("MI355VITS_EMU_DROP");
    return e ? atoi(e) : 0;
}
#else
__device__ __forceinline__ constexpr int b3_drop() { return 0; }
#endif

// W1: the weights contribute their leading bf16 term only ("bf16 weights", MATH_BF16W): three products per multiply-add
template <int MT, int NT, int NG, int NA = NT, bool W1 = false>  // NG 16-channel groups per chunk; NA >= NT: accumulator array extent
__device__ __forceinline__ void b3_chunk(f32x16 (&acc)[MT][NA], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS /*plane stride*/,
                                         int LD, int K, int groups_per_tap, int dil) {
    // wp[i]: (tap 0, first group of this chunk, plane 0) of row tile i, lane offset included; a group is 192 uint4,
    // consecutive taps are groups_per_tap * 192 apart.  xq: (plane 0, group 0, this lane's half and column).
    uint4 ra[2][MT][3];
    uint4 rb[2][NT][3];
    const int drop = b3_drop();
    constexpr int NPA = W1 ? 1 : 3;  // weight planes fetched
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int p = 0; p < NPA; ++p) ra[0][i][p] = wp[i][p * 64];
    MI355_UNROLL
    for (int j = 0; j < NT; ++j)
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) rb[0][j][p] = xq[p * PS + j * 32];
    for (int k = 0; k < K; ++k) {
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;  // NG is even: the parity carries over from tap to tap
            // next group's operands (the very last one re-reads itself: every load stays unconditional)
            const bool wrap = g + 1 == NG;
            const long woff = (wrap ? (last_tap ? (long)k * groups_per_tap + g : (long)(k + 1) * groups_per_tap) : (long)k * groups_per_tap + g + 1) * 192;
            const int xoff = wrap ? (last_tap ? k * dil + g * 2 * LD : (k + 1) * dil) : k * dil + (g + 1) * 2 * LD;
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int p = 0; p < NPA; ++p) ra[nxt][i][p] = wp[i][woff + p * 64];
            MI355_UNROLL
            for (int j = 0; j < NT; ++j)
                MI355_UNROLL
                for (int p = 0; p < 3; ++p) rb[nxt][j][p] = xq[p * PS + xoff + j * 32];
            SCHED_FENCE();
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) {
                    f32x16 c = acc[i][j];
                    if constexpr (!W1) if (drop != 1) c = MFMA_32x32x16_BF16(ra[cur][i][2], rb[cur][j][0], c);  // small terms first
                    if (drop != 2) c = MFMA_32x32x16_BF16(ra[cur][i][0], rb[cur][j][2], c);
                    if constexpr (!W1) if (drop != 3) c = MFMA_32x32x16_BF16(ra[cur][i][1], rb[cur][j][1], c);
                    if constexpr (!W1) c = MFMA_32x32x16_BF16(ra[cur][i][1], rb[cur][j][0], c);
                    c = MFMA_32x32x16_BF16(ra[cur][i][0], rb[cur][j][1], c);
                    c = MFMA_32x32x16_BF16(ra[cur][i][0], rb[cur][j][0], c);
                    acc[i][j] = c;
                }
            SCHED_FENCE();
        }
    }
}

// b3_chunk with the WEIGHT fragments RA - 1 groups ahead instead of one (a ring of RA buffers, NG % RA == 0 so that the ring
// position of a group is the same in every tap): a group of the 3 x 3-tile WaveNet loop is 54 MFMAs = 0.85 us, and one group of
// prefetch covers an L2 hit but not a fragment that the layer's activation stream has pushed out of the XCD's L2 (refetched from
// the memory-side cache / HBM: ~1 - 2 us) — on the boxes where that happens often every weight-streaming kernel runs 20 - 50 %
// slower (BENCH_r03 / r04, profiles/r05_*).  The activation fragments (LDS) stay one group ahead.  Same products in the same order
// per accumulator as b3_chunk: bit-identical.
template <int MT, int NT, int NG, int NA, bool W1, int RA>
__device__ __forceinline__ void b3_chunk_ra(f32x16 (&acc)[MT][NA], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS, int LD,
                                            int K, int groups_per_tap, int dil) {
    static_assert(RA >= 2 && NG % RA == 0 && RA - 1 <= NG, "ring of RA weight-fragment buffers");
    uint4 ra[RA][MT][3];
    uint4 rb[2][NT][3];
    const int drop = b3_drop();
    constexpr int NPA = W1 ? 1 : 3;
    constexpr int D = RA - 1;  // groups ahead
    // linear group index n = k * NG + g of the K * NG groups; past the last one: the last again (every load unconditional)
    const int ntot = K * NG;
    auto load_a = [&](int n, uint4 (&dst)[MT][3]) MI355_INLINE_LAMBDA {
        const int nc = n < ntot ? n : ntot - 1;
        const int k = nc / NG, g = nc - k * NG;
        const long woff = ((long)k * groups_per_tap + g) * 192;
        MI355_UNROLL
        for (int i = 0; i < MT; ++i)
            MI355_UNROLL
            for (int p = 0; p < NPA; ++p) dst[i][p] = wp[i][woff + p * 64];
    };
    MI355_UNROLL
    for (int d = 0; d < D; ++d) load_a(d, ra[d]);
    MI355_UNROLL
    for (int j = 0; j < NT; ++j)
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) rb[0][j][p] = xq[p * PS + j * 32];
    for (int k = 0; k < K; ++k) {
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            const bool wrap = g + 1 == NG;
            const int xoff = wrap ? (last_tap ? k * dil + g * 2 * LD : (k + 1) * dil) : k * dil + (g + 1) * 2 * LD;
            load_a(k * NG + g + D, ra[(g + D) % RA]);
            MI355_UNROLL
            for (int j = 0; j < NT; ++j)
                MI355_UNROLL
                for (int p = 0; p < 3; ++p) rb[nxt][j][p] = xq[p * PS + xoff + j * 32];
            SCHED_FENCE();
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) {
                    f32x16 c = acc[i][j];
                    if constexpr (!W1) if (drop != 1) c = MFMA_32x32x16_BF16(ra[g % RA][i][2], rb[cur][j][0], c);  // small terms first
                    if (drop != 2) c = MFMA_32x32x16_BF16(ra[g % RA][i][0], rb[cur][j][2], c);
                    if constexpr (!W1) if (drop != 3) c = MFMA_32x32x16_BF16(ra[g % RA][i][1], rb[cur][j][1], c);
                    if constexpr (!W1) c = MFMA_32x32x16_BF16(ra[g % RA][i][1], rb[cur][j][0], c);
                    c = MFMA_32x32x16_BF16(ra[g % RA][i][0], rb[cur][j][1], c);
                    c = MFMA_32x32x16_BF16(ra[g % RA][i][0], rb[cur][j][0], c);
                    acc[i][j] = c;
                }
            SCHED_FENCE();
        }
    }
}

// MATH_F16X2 form of b3_chunk: two fp16 planes per operand (hipx.h: split2), three products on v_mfma_f32_32x32x16_f16.
// A group of weight fragments is 128 uint4 (two planes x 64 lanes); the activation planes are [2][group][half][column].
template <int MT, int NT, int NG, int NA = NT>
__device__ __forceinline__ void h2_chunk(f32x16 (&acc)[MT][NA], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS, int LD, int K,
                                         int groups_per_tap, int dil) {
    uint4 ra[2][MT][2];
    uint4 rb[2][NT][2];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int p = 0; p < 2; ++p) ra[0][i][p] = wp[i][p * 64];
    MI355_UNROLL
    for (int j = 0; j < NT; ++j)
        MI355_UNROLL
        for (int p = 0; p < 2; ++p) rb[0][j][p] = xq[p * PS + j * 32];
    for (int k = 0; k < K; ++k) {
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            const bool wrap = g + 1 == NG;
            const long woff = (wrap ? (last_tap ? (long)k * groups_per_tap + g : (long)(k + 1) * groups_per_tap) : (long)k * groups_per_tap + g + 1) * 128;
            const int xoff = wrap ? (last_tap ? k * dil + g * 2 * LD : (k + 1) * dil) : k * dil + (g + 1) * 2 * LD;
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int p = 0; p < 2; ++p) ra[nxt][i][p] = wp[i][woff + p * 64];
            MI355_UNROLL
            for (int j = 0; j < NT; ++j)
                MI355_UNROLL
                for (int p = 0; p < 2; ++p) rb[nxt][j][p] = xq[p * PS + xoff + j * 32];
            SCHED_FENCE();
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) {
                    f32x16 c = acc[i][j];
                    c = MFMA_32x32x16_F16(ra[cur][i][1], rb[cur][j][0], c);  // small terms first
                    c = MFMA_32x32x16_F16(ra[cur][i][0], rb[cur][j][1], c);
                    c = MFMA_32x32x16_F16(ra[cur][i][0], rb[cur][j][0], c);
                    acc[i][j] = c;
                }
            SCHED_FENCE();
        }
    }
}

// The same loop with the B operand single-buffered: column tile j's three plane fragments are refilled for the next
// group right after the MFMAs of column tile j have issued (the other column tiles' MFMAs cover the LDS latency), so a
// wave holds NT x 3 instead of 2 x NT x 3 B fragments: 36 VGPRs fewer at NT = 3, which is what lets eight waves (two
// per SIMD, 256 VGPRs each) carry 2 x 3 accumulator tiles.  Same products in the same order per accumulator.
template <int MT, int NT, int NG, bool W1 = false>
__device__ __forceinline__ void b3_chunk_lean(f32x16 (&acc)[MT][NT], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS, int LD,
                                              int K, int groups_per_tap, int dil) {
    uint4 ra[2][MT][3];
    uint4 rb[NT][3];
    constexpr int NPA = W1 ? 1 : 3;
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int p = 0; p < NPA; ++p) ra[0][i][p] = wp[i][p * 64];
    MI355_UNROLL
    for (int j = 0; j < NT; ++j)
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) rb[j][p] = xq[p * PS + j * 32];
    for (int k = 0; k < K; ++k) {
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            const bool wrap = g + 1 == NG;
            const long woff = (wrap ? (last_tap ? (long)k * groups_per_tap + g : (long)(k + 1) * groups_per_tap) : (long)k * groups_per_tap + g + 1) * 192;
            const int xoff = wrap ? (last_tap ? k * dil + g * 2 * LD : (k + 1) * dil) : k * dil + (g + 1) * 2 * LD;
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int p = 0; p < NPA; ++p) ra[nxt][i][p] = wp[i][woff + p * 64];
            SCHED_FENCE();
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) {
                // the MT accumulators of this column tile are independent chains: product by product across them
                if constexpr (!W1) {
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[cur][i][2], rb[j][0], acc[i][j]);  // small terms first
                }
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[cur][i][0], rb[j][2], acc[i][j]);
                if constexpr (!W1) {
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[cur][i][1], rb[j][1], acc[i][j]);
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[cur][i][1], rb[j][0], acc[i][j]);
                }
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[cur][i][0], rb[j][1], acc[i][j]);
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[cur][i][0], rb[j][0], acc[i][j]);
                SCHED_FENCE();
                MI355_UNROLL
                for (int p = 0; p < 3; ++p) rb[j][p] = xq[p * PS + xoff + j * 32];
                SCHED_FENCE();
            }
        }
    }
}

// b3_chunk_lean with the WEIGHT fragments RA - 1 groups ahead (b3_chunk_ra's ring): the form for FOUR column tiles per wave (k_wn_layer_b3's
// 128-column tiles, round 6) — 3 x 4 accumulator tiles are 192 registers, the double-buffered B fragments of b3_chunk_ra another 96 and the
// ring 144: more than the file.  Single-buffered B (48) + the ring fits without a spill, and the fragments keep their three groups
// (3 x 72 MFMAs = 3.6 us) of cover for the boxes whose L2 loses them to the activation stream.  Same products in the same order per accumulator.
template <int MT, int NT, int NG, bool W1, int RA>
__device__ __forceinline__ void b3_chunk_lean_ra(f32x16 (&acc)[MT][NT], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS, int LD,
                                                 int K, int groups_per_tap, int dil) {
    static_assert(RA >= 2 && NG % RA == 0 && RA - 1 <= NG, "ring of RA weight-fragment buffers");
    uint4 ra[RA][MT][3];
    uint4 rb[NT][3];
    constexpr int NPA = W1 ? 1 : 3;
    constexpr int D = RA - 1;  // groups ahead
    const int ntot = K * NG;
    auto load_a = [&](int n, uint4 (&dst)[MT][3]) MI355_INLINE_LAMBDA {
        const int nc = n < ntot ? n : ntot - 1;  // past the last group: the last again (every load unconditional)
        const int k = nc / NG, g = nc - k * NG;
        const long woff = ((long)k * groups_per_tap + g) * 192;
        MI355_UNROLL
        for (int i = 0; i < MT; ++i)
            MI355_UNROLL
            for (int p = 0; p < NPA; ++p) dst[i][p] = wp[i][woff + p * 64];
    };
    MI355_UNROLL
    for (int d = 0; d < D; ++d) load_a(d, ra[d]);
    MI355_UNROLL
    for (int j = 0; j < NT; ++j)
        MI355_UNROLL
        for (int p = 0; p < 3; ++p) rb[j][p] = xq[p * PS + j * 32];
    for (int k = 0; k < K; ++k) {
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const bool wrap = g + 1 == NG;
            const int xoff = wrap ? (last_tap ? k * dil + g * 2 * LD : (k + 1) * dil) : k * dil + (g + 1) * 2 * LD;
            load_a(k * NG + g + D, ra[(g + D) % RA]);
            SCHED_FENCE();
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) {
                if constexpr (!W1) {
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[g % RA][i][2], rb[j][0], acc[i][j]);  // small terms first
                }
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[g % RA][i][0], rb[j][2], acc[i][j]);
                if constexpr (!W1) {
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[g % RA][i][1], rb[j][1], acc[i][j]);
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[g % RA][i][1], rb[j][0], acc[i][j]);
                }
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[g % RA][i][0], rb[j][1], acc[i][j]);
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_BF16(ra[g % RA][i][0], rb[j][0], acc[i][j]);
                SCHED_FENCE();
                MI355_UNROLL
                for (int p = 0; p < 3; ++p) rb[j][p] = xq[p * PS + xoff + j * 32];
                SCHED_FENCE();
            }
        }
    }
}

// MATH_F16X2 form of b3_chunk_lean (B fragments single-buffered, two fp16 planes, three products)
template <int MT, int NT, int NG>
__device__ __forceinline__ void h2_chunk_lean(f32x16 (&acc)[MT][NT], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS, int LD,
                                              int K, int groups_per_tap, int dil) {
    uint4 ra[2][MT][2];
    uint4 rb[NT][2];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int p = 0; p < 2; ++p) ra[0][i][p] = wp[i][p * 64];
    MI355_UNROLL
    for (int j = 0; j < NT; ++j)
        MI355_UNROLL
        for (int p = 0; p < 2; ++p) rb[j][p] = xq[p * PS + j * 32];
    for (int k = 0; k < K; ++k) {
        const bool last_tap = k == K - 1;
        MI355_UNROLL
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            const bool wrap = g + 1 == NG;
            const long woff = (wrap ? (last_tap ? (long)k * groups_per_tap + g : (long)(k + 1) * groups_per_tap) : (long)k * groups_per_tap + g + 1) * 128;
            const int xoff = wrap ? (last_tap ? k * dil + g * 2 * LD : (k + 1) * dil) : k * dil + (g + 1) * 2 * LD;
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int p = 0; p < 2; ++p) ra[nxt][i][p] = wp[i][woff + p * 64];
            SCHED_FENCE();
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) {
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_F16(ra[cur][i][1], rb[j][0], acc[i][j]);  // small terms first
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_F16(ra[cur][i][0], rb[j][1], acc[i][j]);
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) acc[i][j] = MFMA_32x32x16_F16(ra[cur][i][0], rb[j][0], acc[i][j]);
                SCHED_FENCE();
                MI355_UNROLL
                for (int p = 0; p < 2; ++p) rb[j][p] = xq[p * PS + xoff + j * 32];
                SCHED_FENCE();
            }
        }
    }
}

// stage x[c0 : c0 + 16 NG, ts : ts + LD) as three bf16 planes: mask, leaky-relu and split fused.  A thread takes one
// column; per (group, half) it loads that column's eight channels (16G + 4h + 0..3 and 16G + 8 + 4h + 0..3: eight
// 4-byte loads, 256 contiguous bytes per wave and row), splits four pairs and stores one 16-byte slot per plane.
// Consecutive lanes store consecutive slots, so every ds_write_b128 is bank-conflict free (a thread owning four
// columns — 16-byte loads along time — stores 64 bytes apart from its neighbour: a four-way conflict on each of its
// twelve stores, as expensive as the loop's ds_read_b128 traffic).  No alignment demands.  GB (group, half) rows are
// loaded per batch = GB * 8 loads in flight per thread.
// F16 (MATH_F16X2): two fp16 planes of the value scaled by F16X2_X_SCALE instead of three bf16 planes.
template <int NG, int GB = 4, bool F16 = false>
__device__ __forceinline__ void stage_planes(const float* __restrict__ xb, long x_ld, int LD, int ts, int tend, float slope,
                                             uint4* __restrict__ planes, int PS, int tid, int nthreads) {
    static_assert((NG * 2) % GB == 0, "batches of GB (group, half) rows");
    const int last = tend > 0 ? tend - 1 : 0;
    // narrow tiles (LD <= nthreads / 2): the threads form nthreads / LD column sets that share the rows' batches, so
    // that all of them load (a 104-column WaveNet tile on 256 threads: 2 sets x 12 rows instead of 104 threads x 24)
    int nparts = 1, part = 0, col0 = tid;
    if (nthreads >= 2 * LD) {
        nparts = nthreads / LD;
        part = tid / LD;
        col0 = tid - part * LD;
        if (part >= nparts) return;
    }
    for (int col = col0; col < LD; col += nthreads) {
        const int tt = ts + col;
        const bool in = tt >= 0 && tt < tend;
        const int tc = tt < 0 ? 0 : (tt > last ? last : tt);  // every load unconditional (clamped), masked afterwards
        for (int g0 = part * GB; g0 < NG * 2; g0 += nparts * GB) {
            float v[GB][8];
            MI355_UNROLL
            for (int u = 0; u < GB; ++u) {
                const int gh = g0 + u, cbase = (gh >> 1) * 16 + (gh & 1) * 4;
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) v[u][e] = xb[(long)(cbase + 8 * (e >> 2) + (e & 3)) * x_ld + tc];
            }
            SCHED_FENCE();  // the GB * 8 loads stay in flight together
            MI355_UNROLL
            for (int u = 0; u < GB; ++u) {
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) v[u][e] = in ? lrelu_f(v[u][e], slope) : 0.0f;
                const int o = (g0 + u) * LD + col;
                if constexpr (F16) {
                    MI355_UNROLL
                    for (int e = 0; e < 8; ++e) v[u][e] *= F16X2_X_SCALE;
                    uint4 h, m;
                    split2_pk(v[u][0], v[u][1], h.x, m.x);
                    split2_pk(v[u][2], v[u][3], h.y, m.y);
                    split2_pk(v[u][4], v[u][5], h.z, m.z);
                    split2_pk(v[u][6], v[u][7], h.w, m.w);
                    planes[o] = h;
                    planes[PS + o] = m;
                } else {
                    uint4 h, m, l;
                    split3_pk(v[u][0], v[u][1], h.x, m.x, l.x);
                    split3_pk(v[u][2], v[u][3], h.y, m.y, l.y);
                    split3_pk(v[u][4], v[u][5], h.z, m.z, l.z);
                    split3_pk(v[u][6], v[u][7], h.w, m.w, l.w);
                    planes[o] = h;
                    planes[PS + o] = m;
                    planes[2 * PS + o] = l;
                }
            }
        }
    }
}

}  // namespace m355
