// b3.h — device building blocks of the split-bf16 (MATH_BF16X3) kernels: staging an activation chunk as three bf16
// planes, and one wave's share of the implicit GEMM over such planes on v_mfma_f32_32x32x16_bf16.
// (the split itself: hipx.h, split3_pk; the weight fragment order: kernels.h, pack_conv_weights_bf16x3_mode layout 1)
#pragma once
#include "kernels.h"

namespace m355 {

// W1: the weights contribute their leading bf16 term only ("bf16 weights", MATH_BF16W): three products per multiply-add
template <int MT, int NT, int NG, int NA = NT, bool W1 = false>  // NG 16-channel groups per chunk; NA >= NT: accumulator array extent
__device__ __forceinline__ void b3_chunk(f32x16 (&acc)[MT][NA], const uint4* const (&wp)[MT], const uint4* __restrict__ xq, int PS /*plane stride*/,
                                         int LD, int K, int groups_per_tap, int dil) {
    // wp[i]: (tap 0, first group of this chunk, plane 0) of row tile i, lane offset included; a group is 192 uint4,
    // consecutive taps are groups_per_tap * 192 apart.  xq: (plane 0, group 0, this lane's half and column).
    uint4 ra[2][MT][3];
    uint4 rb[2][NT][3];
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
                    if constexpr (!W1) c = MFMA_32x32x16_BF16(ra[cur][i][2], rb[cur][j][0], c);  // small terms first
                    c = MFMA_32x32x16_BF16(ra[cur][i][0], rb[cur][j][2], c);
                    if constexpr (!W1) c = MFMA_32x32x16_BF16(ra[cur][i][1], rb[cur][j][1], c);
                    if constexpr (!W1) c = MFMA_32x32x16_BF16(ra[cur][i][1], rb[cur][j][0], c);
                    c = MFMA_32x32x16_BF16(ra[cur][i][0], rb[cur][j][1], c);
                    c = MFMA_32x32x16_BF16(ra[cur][i][0], rb[cur][j][0], c);
                    acc[i][j] = c;
                }
            SCHED_FENCE();
        }
    }
}

// stage x[c0 : c0 + 16 NG, ts : ts + LD) as three bf16 planes: mask, leaky-relu and split fused.  A thread takes one
// (group, half) and four columns: eight 16-byte loads along time (channels 16G + 4h + 0..3 and 16G + 8 + 4h + 0..3),
// 16 pair-splits, twelve 16-byte LDS stores.
template <int NG>
__device__ __forceinline__ void stage_planes(const float* __restrict__ xb, long x_ld, int LD, int ts, int tend, float slope,
                                             uint4* __restrict__ planes, int PS, int vec, int nthreads = 256) {
    const int ld4 = LD >> 2;
    const int n_items = NG * 2 * ld4;
    // two items per thread and pass: their sixteen 16-byte loads are in flight together (one memory round trip per pass)
    for (int idx0 = threadIdx.x; idx0 < n_items; idx0 += 2 * nthreads) {
        float v[2][8][4];
        MI355_UNROLL
        for (int u = 0; u < 2; ++u) {
            const int idx = idx0 + u * nthreads;
            const int idc = idx < n_items ? idx : idx0;  // a missing second item re-reads the first (discarded)
            const int gh = idc / ld4, c4 = idc - gh * ld4;  // gh = group * 2 + half
            const int cbase = (gh >> 1) * 16 + (gh & 1) * 4;
            const int tt = ts + 4 * c4;
            // the in-range test depends on the columns only: one branch around all eight rows, so that the eight loads
            // are in flight together (a test per row makes hipcc wait for every load before it issues the next)
            if (vec && tt >= 0 && tt + 3 < tend) {
                float4 r4[8];
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) r4[e] = *reinterpret_cast<const float4*>(xb + (long)(cbase + 8 * (e >> 2) + (e & 3)) * x_ld + tt);
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) { v[u][e][0] = r4[e].x; v[u][e][1] = r4[e].y; v[u][e][2] = r4[e].z; v[u][e][3] = r4[e].w; }
            } else {
                MI355_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const float* row = xb + (long)(cbase + 8 * (e >> 2) + (e & 3)) * x_ld;
                    MI355_UNROLL
                    for (int j = 0; j < 4; ++j) v[u][e][j] = (tt + j >= 0 && tt + j < tend) ? row[tt + j] : 0.0f;
                }
            }
        }
        MI355_UNROLL
        for (int u = 0; u < 2; ++u) {
            const int idx = idx0 + u * nthreads;
            if (idx >= n_items) continue;
            const int gh = idx / ld4, c4 = idx - gh * ld4;
            MI355_UNROLL
            for (int j = 0; j < 4; ++j) {
                uint4 h, m, l;
                split3_pk(lrelu_f(v[u][0][j], slope), lrelu_f(v[u][1][j], slope), h.x, m.x, l.x);
                split3_pk(lrelu_f(v[u][2][j], slope), lrelu_f(v[u][3][j], slope), h.y, m.y, l.y);
                split3_pk(lrelu_f(v[u][4][j], slope), lrelu_f(v[u][5][j], slope), h.z, m.z, l.z);
                split3_pk(lrelu_f(v[u][6][j], slope), lrelu_f(v[u][7][j], slope), h.w, m.w, l.w);
                const int o = gh * LD + 4 * c4 + j;
                planes[o] = h;
                planes[PS + o] = m;
                planes[2 * PS + o] = l;
            }
        }
    }
}

}  // namespace m355
