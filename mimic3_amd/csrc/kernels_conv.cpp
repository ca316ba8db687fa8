// kernels_conv.cpp — Conv1d / ConvTranspose1d / decoder-tail kernels for gfx950 (MI355X).
//
// Conv1d semantics:  y[co,t] = b[co] + sum_ci sum_k W[co,ci,k] * x[ci, t - pad + k*dil]
// (zero outside [0,T)).  Two implementations share one epilogue:
//   * k_conv1d_mfma   — implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32,
//                       cdna_hip_programming.md §3): M = C_out, N = time, K = C_in x taps.  The input
//                       tile (+halo) is staged once per C_in chunk into LDS with the input-side pointwise
//                       op (mask, leaky-relu) fused; B fragments are conflict-free ds_read_b32 of 32
//                       consecutive time samples per half-wave; A fragments come pre-packed in fragment
//                       order from HBM/L2 (one coalesced 256 B load per k-step); all output-side pointwise
//                       work (bias, speaker conditioning, relu, residual, WaveNet gate, res/skip update,
//                       coupling subtract, masks, MRF mean) is fused into the epilogue.
//   * k_conv1d_generic — plain VALU/LDS tiled kernel for any shape; reference for tests and A/B.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <cstring>

#include "kernels.h"
#include <atomic>
#include <type_traits>
#include "b3.h"

namespace m355 {

// ------------------------------------------------------------------------------------------------
// shared epilogues
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void epi_std(const ConvArgs& a, int b, int co, int t, float v, int out_len) {
    if (a.bias) v += a.bias[co];
    if (a.shuf_s) {  // polyphase ConvTranspose1d (co' = c*s + r): phase r of position t lands at n = t*s + r - p
        const int c = co / a.shuf_s, r = co - c * a.shuf_s;
        const int n = t * a.shuf_s + r - a.shuf_p;
        if (n >= 0 && n < a.shuf_T) a.y[(long)b * a.y_bs + (long)c * a.y_ld + n] = v;
        return;
    }
    if (a.cond) v += a.cond[(long)b * a.cond_bs + co];
    if (a.relu) v = fmaxf(v, 0.0f);
    if (a.mask_before_res && t >= out_len) v = 0.0f;
    if (a.res) {
        const float r = a.res[(long)b * a.res_bs + (long)co * a.res_ld + t];
        v = a.res_sub ? r - v : r + v;
    }
    v *= a.out_scale;
    if (!a.mask_before_res && t >= out_len) v = 0.0f;
    float* yp = a.y + (long)b * a.y_bs + (long)co * a.y_ld + t;
    if (a.accumulate) v += *yp;
    *yp = v;
}

// epi_std on four consecutive time samples t .. t+3 (all < T, rows 16-byte aligned): 16-byte loads / stores
// what the 4-wide standard epilogue reads from global memory, apart from the conv result: loaded for a batch of items
// before the first of them is stored (a load issued after a store cannot be waited for without waiting for the store's
// acknowledgement as well — the memory counter retires in order)
struct EpiStd4In { float bias, cond; float4 r4, y4; };
__device__ __forceinline__ EpiStd4In epi_std4_load(const ConvArgs& a, int b, int co, int t) {
    EpiStd4In L;
    L.bias = a.bias ? a.bias[co] : 0.0f;
    L.cond = a.cond ? a.cond[(long)b * a.cond_bs + co] : 0.0f;
    L.r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.res) L.r4 = *reinterpret_cast<const float4*>(a.res + (long)b * a.res_bs + (long)co * a.res_ld + t);
    L.y4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.accumulate) L.y4 = *reinterpret_cast<const float4*>(a.y + (long)b * a.y_bs + (long)co * a.y_ld + t);
    return L;
}
__device__ __forceinline__ void epi_std4_finish(const ConvArgs& a, int b, int co, int t, float4 v, int out_len, const EpiStd4In& L) {
    float x[4] = {v.x, v.y, v.z, v.w};
    const float rr[4] = {L.r4.x, L.r4.y, L.r4.z, L.r4.w};
    const float yy[4] = {L.y4.x, L.y4.y, L.y4.z, L.y4.w};
    MI355_UNROLL
    for (int m = 0; m < 4; ++m) {
        float q = x[m];
        if (a.bias) q += L.bias;  // same order of additions as epi_std: bias, then cond
        if (a.cond) q += L.cond;
        if (a.relu) q = fmaxf(q, 0.0f);
        if (a.mask_before_res && t + m >= out_len) q = 0.0f;
        if (a.res) q = a.res_sub ? rr[m] - q : rr[m] + q;
        q *= a.out_scale;
        if (!a.mask_before_res && t + m >= out_len) q = 0.0f;
        if (a.accumulate) q += yy[m];
        x[m] = q;
    }
    *reinterpret_cast<float4*>(a.y + (long)b * a.y_bs + (long)co * a.y_ld + t) = make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void epi_std4(const ConvArgs& a, int b, int co, int t, float4 v, int out_len) {
    epi_std4_finish(a, b, co, t, v, out_len, epi_std4_load(a, b, co, t));
}

// WaveNet gate (A.9): u = tanh(a[:H] + cond) * sigmoid(a[H:] + cond)
__device__ __forceinline__ void epi_gate(const ConvArgs& a, int b, int c, int t, float v0, float v1) {
    if (a.bias) { v0 += a.bias[c]; v1 += a.bias[c + a.H]; }
    if (a.cond) { v0 += a.cond[(long)b * a.cond_bs + c]; v1 += a.cond[(long)b * a.cond_bs + c + a.H]; }
    // tanh(x) = 1 - 2 / (exp(2x) + 1), sigmoid(x) = 1 / (1 + exp(-x)); clamp keeps exp finite
    const float gate = wn_gate_f(v0, v1);
    a.y[(long)b * a.y_bs + (long)c * a.y_ld + t] = gate;
}

// WaveNet res/skip update (A.9): h = (h + rs[:H]) * mask ; skip += rs[H:]  (last layer: skip += rs)
__device__ __forceinline__ void epi_resskip(const ConvArgs& a, int b, int co, int t, float v, int out_len) {
    if (a.bias) v += a.bias[co];
    if (a.Cout == a.H || co >= a.H) {
        const int c2 = (a.Cout == a.H) ? co : co - a.H;
        float* sp = a.y2 + (long)b * a.y2_bs + (long)c2 * a.y2_ld + t;
        *sp = a.skip_init ? v : *sp + v;
    } else {
        float* hp = a.y + (long)b * a.y_bs + (long)co * a.y_ld + t;
        float h = *hp + v;
        if (t >= out_len) h = 0.0f;
        *hp = h;
    }
}

// ------------------------------------------------------------------------------------------------
// The same epilogues for a lane's 16 rows of one 32 x 32 accumulator tile (column t): every optional operand (bias,
// conditioning, residual, accumulate target, skip) is loaded for all 16 rows under ONE wave-uniform test, so the loads
// are in flight together.  Written element by element (`if (a.res) v += a.res[...]` per row) hipcc emits a branch and
// an s_waitcnt vmcnt(0) per optional load — up to 64 dependent memory round trips per tile.  Arithmetic and its order
// are exactly epi_std / epi_resskip / epi_gate (same bits).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_row(int r, int brow) { return (r & 3) + 8 * (r >> 2) + 4 * brow; }

__device__ __forceinline__ void epi_std_tile(const ConvArgs& a, int b, int co0, int t, int brow, const f32x16& acc, int out_len) {
    if (a.shuf_s) {  // polyphase scatter, odd phase counts: rare, element by element
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + tile_row(r, brow);
            if (co < a.Cout) epi_std(a, b, co, t, acc[r], out_len);
        }
        return;
    }
    const int cmax = a.Cout - 1;
    float v[16], rv[16], yv[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    if (a.bias) {
        float bv[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); bv[r] = a.bias[co < cmax ? co : cmax]; }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) v[r] += bv[r];
    }
    if (a.cond) {
        float cv[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); cv[r] = a.cond[(long)b * a.cond_bs + (co < cmax ? co : cmax)]; }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) v[r] += cv[r];
    }
    if (a.res) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); rv[r] = a.res[(long)b * a.res_bs + (long)(co < cmax ? co : cmax) * a.res_ld + t]; }
    }
    if (a.accumulate) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); yv[r] = a.y[(long)b * a.y_bs + (long)(co < cmax ? co : cmax) * a.y_ld + t]; }
    }
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + tile_row(r, brow);
        float q = v[r];
        if (a.relu) q = fmaxf(q, 0.0f);
        if (a.mask_before_res && t >= out_len) q = 0.0f;
        if (a.res) q = a.res_sub ? rv[r] - q : rv[r] + q;
        q *= a.out_scale;
        if (!a.mask_before_res && t >= out_len) q = 0.0f;
        if (a.accumulate) q += yv[r];
        if (co < a.Cout) a.y[(long)b * a.y_bs + (long)co * a.y_ld + t] = q;
    }
}

__device__ __forceinline__ void epi_resskip_tile(const ConvArgs& a, int b, int co0, int t, int brow, const f32x16& acc, int out_len) {
    if (a.Cout != a.H && (a.H & 31)) {  // the h / skip boundary cuts through a tile: element by element
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + tile_row(r, brow);
            if (co < a.Cout) epi_resskip(a, b, co, t, acc[r], out_len);
        }
        return;
    }
    const int cmax = a.Cout - 1;
    float v[16], old[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    if (a.bias) {
        float bv[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); bv[r] = a.bias[co < cmax ? co : cmax]; }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) v[r] += bv[r];
    }
    const bool to_skip = a.Cout == a.H || co0 >= a.H;  // wave-uniform: H is a multiple of 32 on this path (checked by the launcher)
    if (to_skip) {
        const int sub = a.Cout == a.H ? 0 : a.H;
        if (!a.skip_init) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); old[r] = a.y2[(long)b * a.y2_bs + (long)((co < cmax ? co : cmax) - sub) * a.y2_ld + t]; }
        }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + tile_row(r, brow);
            if (co < a.Cout) a.y2[(long)b * a.y2_bs + (long)(co - sub) * a.y2_ld + t] = a.skip_init ? v[r] : old[r] + v[r];
        }
    } else {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int co = co0 + tile_row(r, brow); old[r] = a.y[(long)b * a.y_bs + (long)co * a.y_ld + t]; }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + tile_row(r, brow);
            float h = old[r] + v[r];
            if (t >= out_len) h = 0.0f;
            a.y[(long)b * a.y_bs + (long)co * a.y_ld + t] = h;
        }
    }
}

// gate pair: acc0 = rows c (tanh half), acc1 = rows H + c (sigmoid half), c = c0 + row
__device__ __forceinline__ void epi_gate_tile(const ConvArgs& a, int b, int c0, int t, int brow, const f32x16& acc0, const f32x16& acc1) {
    const int cmax = a.H - 1;
    float v0[16], v1[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) { v0[r] = acc0[r]; v1[r] = acc1[r]; }
    if (a.bias) {
        float b0[16], b1[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { const int c = c0 + tile_row(r, brow), cc = c < cmax ? c : cmax; b0[r] = a.bias[cc]; b1[r] = a.bias[cc + a.H]; }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { v0[r] += b0[r]; v1[r] += b1[r]; }
    }
    if (a.cond) {
        float c0v[16], c1v[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + tile_row(r, brow), cc = c < cmax ? c : cmax;
            c0v[r] = a.cond[(long)b * a.cond_bs + cc];
            c1v[r] = a.cond[(long)b * a.cond_bs + cc + a.H];
        }
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) { v0[r] += c0v[r]; v1[r] += c1v[r]; }
    }
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int c = c0 + tile_row(r, brow);
        const float gate = wn_gate_f(v0[r], v1[r]);
        if (c < a.H) a.y[(long)b * a.y_bs + (long)c * a.y_ld + t] = gate;
    }
}

// Polyphase ConvTranspose1d scatter straight from the accumulators.  Output channels are ordered co' = c*s + r
// (channel-major / phase-minor), so the 4 consecutive rows a lane holds per register group are 4 consecutive phases of
// one channel = 4 consecutive output samples: a 16-byte store per lane, 1 KiB contiguous per store instruction.
template <int MT, int NT>
__device__ __forceinline__ void epi_polyphase_regs(const ConvArgs& a, f32x16 (&acc)[MT][NT], int b, int tcol0, int tile0,
                                                   int brow) {
    // this lane's 16 biases per row tile, loaded under one test (a test per element serialises the loads) and waited
    // for before the first store: a load the compiler sinks between the stores has to wait, through the in-order
    // vmcnt, for every store issued before it — one write round trip per register group
    float bv[MT][16];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bv[i][r] = 0.0f;
    if (a.bias) {
        MI355_UNROLL
        for (int i = 0; i < MT; ++i)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int cop = 32 * (tile0 + i) + 8 * (r >> 2) + 4 * brow + (r & 3);
                bv[i][r] = a.bias[cop < a.Cout ? cop : a.Cout - 1];
            }
    }
    // bias added in place (the accumulators are dead after the epilogue), so every load is consumed before the fence
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int j = 0; j < NT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bv[i][r];
    SCHED_FENCE();
    const int bcol = threadIdx.x & 31;
    const int tw = WAVE_UNIFORM(tcol0 - bcol);  // the wave's first column
    const int al = (-a.shuf_p) & 3;             // alignment class of every store of this launch (s % 4 == 0, rows in fours)
    // interior tile (whole wave, all NT column tiles, every row tile complete): straight-line stores, no per-lane tests
    const bool interior = a.yvec && (al == 0 || al == 2) && (a.Cout & 31) == 0 && 32 * (tile0 + MT) <= a.Cout && tw + NT * 32 <= a.T &&
                          (long)tw * a.shuf_s - a.shuf_p >= 0 && (long)(tw + NT * 32) * a.shuf_s - a.shuf_p <= a.shuf_T;
    if (interior) {
        // the alignment class is decided once, outside the loops (inside, hipcc merges the two store shapes into a
        // dword + dwordx3 pair for both)
        auto store_all = [&](auto AL) {
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) {
                MI355_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const int cop = 32 * (tile0 + i) + 8 * g + 4 * brow;
                    const int c = cop / a.shuf_s, r0 = cop - c * a.shuf_s;
                    float* yrow = a.y + (long)b * a.y_bs + (long)c * a.y_ld + (r0 - a.shuf_p) + (long)tcol0 * a.shuf_s;
                    MI355_UNROLL
                    for (int j = 0; j < NT; ++j) {
                        float* yp = yrow + (long)j * 32 * a.shuf_s;
                        const float v0 = acc[i][j][4 * g + 0], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                        if constexpr (decltype(AL)::value == 0) {
                            *reinterpret_cast<float4*>(yp) = make_float4(v0, v1, v2, v3);
                        } else {
                            *reinterpret_cast<float2*>(yp) = make_float2(v0, v1);
                            *reinterpret_cast<float2*>(yp + 2) = make_float2(v2, v3);
                        }
                    }
                }
            }
        };
        if (al == 0) store_all(std::integral_constant<int, 0>{});
        else store_all(std::integral_constant<int, 2>{});
        return;
    }
    MI355_UNROLL
    for (int j = 0; j < NT; ++j) {
        const int t = tcol0 + j * 32;
        if (t >= a.T) continue;
        MI355_UNROLL
        for (int i = 0; i < MT; ++i) {
            MI355_UNROLL
            for (int g = 0; g < 4; ++g) {
                const int cop = 32 * (tile0 + i) + 8 * g + 4 * brow;  // multiple of 4
                if (cop >= a.Cout) continue;
                const int c = cop / a.shuf_s, r0 = cop - c * a.shuf_s;
                const int n0 = t * a.shuf_s + r0 - a.shuf_p;
                float v[4];
                MI355_UNROLL
                for (int m = 0; m < 4; ++m) v[m] = acc[i][j][4 * g + m];
                float* yp = a.y + (long)b * a.y_bs + (long)c * a.y_ld + n0;
                if (n0 >= 0 && n0 + 3 < a.shuf_T && (n0 & 3) == 0 && a.yvec) {
                    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (n0 >= 0 && n0 + 3 < a.shuf_T && (n0 & 1) == 0 && a.yvec) {
                    *reinterpret_cast<float2*>(yp) = make_float2(v[0], v[1]);
                    *reinterpret_cast<float2*>(yp + 2) = make_float2(v[2], v[3]);
                } else {
                    MI355_UNROLL
                    for (int m = 0; m < 4; ++m)
                        if (n0 + m >= 0 && n0 + m < a.shuf_T) yp[m] = v[m];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// generic VALU kernel
// ------------------------------------------------------------------------------------------------
constexpr int G_CO = 32, G_T = 64, G_CI = 8;

__global__ __launch_bounds__(256) void k_conv1d_generic(ConvArgs a) {
    DYN_SMEM(float, smem);
    const int LD = G_T + (a.K - 1) * a.dil;
    float* xs = smem;              // [G_CI][LD]
    float* ws = smem + G_CI * LD;  // [G_CO][G_CI][K]
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * G_T;
    const bool gate = (a.epi == EPI_GATE);
    // output channels of this thread: STD/RESSKIP: co0 + 2*ty + i ; GATE: (c, c + H) with c = 16*blockIdx.y + ty
    int co_i[2];
    if (gate) {
        const int c = blockIdx.y * 16 + ty;
        co_i[0] = (c < a.H) ? c : -1;
        co_i[1] = (c < a.H) ? c + a.H : -1;
    } else {
        for (int i = 0; i < 2; ++i) {
            const int co = blockIdx.y * G_CO + 2 * ty + i;
            co_i[i] = (co < a.Cout) ? co : -1;
        }
    }
    const int Tin = a.Tin >= 0 ? a.Tin : a.T;
    const int in_len = a.in_len ? a.in_len[b] : Tin;
    const int out_len = a.out_len ? a.out_len[b] : a.T;
    float acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

    for (int c0 = 0; c0 < a.Cin; c0 += G_CI) {
        for (int idx = tid; idx < G_CI * LD; idx += 256) {
            const int ci = idx / LD, tt = idx - ci * LD;
            const int c = c0 + ci, t = t0 - a.pad + tt;
            float v = 0.0f;
            if (c < a.Cin && t >= 0 && t < Tin && t < in_len) {
                v = a.x[(long)b * a.x_bs + (long)c * a.x_ld + t];
                v = v >= 0.0f ? v : v * a.in_slope;
            }
            xs[idx] = v;
        }
        // weights of the 32 rows this block needs: row r = 2*ty' + i  (ty' = 0..15)
        for (int idx = tid; idx < G_CO * G_CI * a.K; idx += 256) {
            const int r = idx / (G_CI * a.K);
            const int rem = idx - r * (G_CI * a.K);
            const int ci = rem / a.K, k = rem - ci * a.K;
            const int tyy = r >> 1, i = r & 1;
            int co;
            if (gate) {
                const int c = blockIdx.y * 16 + tyy;
                co = (c < a.H) ? c + i * a.H : -1;
            } else {
                co = blockIdx.y * G_CO + r;
                if (co >= a.Cout) co = -1;
            }
            float v = 0.0f;
            if (co >= 0 && c0 + ci < a.Cin) v = a.w[((long)co * a.Cin + c0 + ci) * a.K + k];
            ws[idx] = v;
        }
        __syncthreads();
        for (int ci = 0; ci < G_CI; ++ci) {
            for (int k = 0; k < a.K; ++k) {
                const float w0 = ws[((2 * ty + 0) * G_CI + ci) * a.K + k];
                const float w1 = ws[((2 * ty + 1) * G_CI + ci) * a.K + k];
                const float* xp = xs + ci * LD + tx * 4 + k * a.dil;
                for (int j = 0; j < 4; ++j) {
                    const float xv = xp[j];
                    acc[0][j] = fmaf(w0, xv, acc[0][j]);
                    acc[1][j] = fmaf(w1, xv, acc[1][j]);
                }
            }
        }
        __syncthreads();
    }
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + tx * 4 + j;
        if (t >= a.T) continue;
        if (gate) {
            if (co_i[0] >= 0) epi_gate(a, b, co_i[0], t, acc[0][j], acc[1][j]);
        } else {
            for (int i = 0; i < 2; ++i) {
                if (co_i[i] < 0) continue;
                if (a.epi == EPI_RESSKIP) epi_resskip(a, b, co_i[i], t, acc[i][j], out_len);
                else epi_std(a, b, co_i[i], t, acc[i][j], out_len);
            }
        }
    }
}

void launch_conv1d_generic(const ConvArgs& a, hipStream_t s) {
    if (a.T <= 0 || a.B <= 0) return;
    const int LD = G_T + (a.K - 1) * a.dil;
    const size_t shmem = sizeof(float) * ((size_t)G_CI * LD + (size_t)G_CO * G_CI * a.K);
    const int ny = (a.epi == EPI_GATE) ? (a.H + 15) / 16 : (a.Cout + G_CO - 1) / G_CO;
    dim3 grid((a.T + G_T - 1) / G_T, ny, a.B);
    LAUNCH_KERNEL(k_conv1d_generic, grid, dim3(256), shmem, s, a);
}

// ------------------------------------------------------------------------------------------------
// fp32-MFMA implicit-GEMM kernel
// ------------------------------------------------------------------------------------------------
// Packed A operand: [n_tiles][K][Cin/2][64]; lane l of (tile, k, cp) holds W[row(tile, l&31)][2cp + (l>>5)][k]
// — exactly the v_mfma_f32_32x32x2_f32 A fragment (A[i = l&31][k = l>>5]).
// Tile -> output-channel map: STD/RESSKIP: co = 32*tile + r.  GATE: tiles come in pairs (2m, 2m+1) =
// rows {c, H + c}, c = 32m + r, so one wave holds both gate operands of a channel in the same lane.
static inline int tile_row_to_co(int epi, int tile, int r, int Cout, int H) {
    if (epi == EPI_GATE) {
        const int c = 32 * (tile >> 1) + r;
        return c < H ? (tile & 1) * H + c : -1;
    }
    const int co = 32 * tile + r;
    return co < Cout ? co : -1;
}
static inline int n_tiles_for(int epi, int Cout, int H) {
    return epi == EPI_GATE ? 2 * ((H + 31) / 32) : (Cout + 31) / 32;
}

size_t mfma_packed_floats(int Cout, int Cin, int K) {
    // GATE packing has the same tile count when H % 32 == 0; take the larger bound to be safe
    const int nt_std = (Cout + 31) / 32;
    const int nt_gate = 2 * ((Cout / 2 + 31) / 32);
    const int nt = nt_std > nt_gate ? nt_std : nt_gate;
    return (size_t)nt * K * (Cin / 2) * 64;
}

// epi selects the tile->channel map; for GATE, H = Cout / 2.
void pack_conv_weights_mfma_mode(const float* w, int Cout, int Cin, int K, int epi, float* out) {
    const int H = Cout / 2;
    const int nt = n_tiles_for(epi, Cout, H);
    const int cp_n = Cin / 2;
    for (int tile = 0; tile < nt; ++tile)
        for (int k = 0; k < K; ++k)
            for (int cp = 0; cp < cp_n; ++cp)
                for (int l = 0; l < 64; ++l) {
                    const int co = tile_row_to_co(epi, tile, l & 31, Cout, H);
                    const int ci = 2 * cp + (l >> 5);
                    out[(((size_t)tile * K + k) * cp_n + cp) * 64 + l] = co >= 0 ? w[((size_t)co * Cin + ci) * K + k] : 0.0f;
                }
}
// regroup packed records (pairs fastest) so that a lane's fragments of four consecutive channel pairs are contiguous:
// [tile][tap][group of 4 pairs][lane][4] (16-byte A loads of the fused MRF stage).  Needs Cin % 8 == 0.
void regroup_packed_x4(const float* packed, size_t n_floats, float* out) {
    const size_t recs = n_floats / 64;
    for (size_t r = 0; r < recs; ++r)
        for (int l = 0; l < 64; ++l) out[((r >> 2) * 64 + l) * 4 + (r & 3)] = packed[r * 64 + l];
}
size_t bf16x3_packed_words(int Cout, int Cin, int K) { return (size_t)(Cout / 32) * K * (Cin / 16) * 3 * 64 * 4; }

static inline uint32_t bf16_rne_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
static inline float bf16_bits_to_float(uint32_t b) {
    const uint32_t u = b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// layout 0 (packed activation tiles of the fused MRF stage): slot e of half h <-> channel 16G + (e < 4 ? 0 : 8) + h + 2 (e & 3)
// layout 1 (staged planes of k_conv1d_b3):                     slot e of half h <-> channel 16G + 8 (e >> 2) + 4 h + (e & 3)
static inline int b3_slot_channel(int layout, int half, int e) {
    return layout == 0 ? (e < 4 ? 0 : 8) + half + 2 * (e & 3) : 8 * (e >> 2) + 4 * half + (e & 3);
}

size_t bf16x3_packed_words_mode(int Cout, int Cin, int K, int epi) {
    return (size_t)n_tiles_for(epi, Cout, Cout / 2) * K * (Cin / 16) * 3 * 64 * 4;
}

void pack_conv_weights_bf16x3_mode(const float* w, int Cout, int Cin, int K, int epi, int layout, uint32_t* out) {
    const int ng = Cin / 16, H = Cout / 2;
    const int nt = n_tiles_for(epi, Cout, H);
    for (int tile = 0; tile < nt; ++tile)
        for (int k = 0; k < K; ++k)
            for (int g = 0; g < ng; ++g)
                for (int l = 0; l < 64; ++l) {
                    const int co = tile_row_to_co(epi, tile, l & 31, Cout, H), half = l >> 5;
                    uint32_t plane[3][8];
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * g + b3_slot_channel(layout, half, e);
                        const float v = co >= 0 ? w[((size_t)co * Cin + ci) * K + k] : 0.0f;
                        const uint32_t h = bf16_rne_bits(v);
                        const float r1 = v - bf16_bits_to_float(h);
                        const uint32_t m = bf16_rne_bits(r1);
                        const float r2 = r1 - bf16_bits_to_float(m);
                        plane[0][e] = h; plane[1][e] = m; plane[2][e] = bf16_rne_bits(r2);
                    }
                    for (int p = 0; p < 3; ++p) {
                        uint32_t* o = out + (((((size_t)tile * K + k) * ng + g) * 3 + p) * 64 + l) * 4;
                        for (int j = 0; j < 4; ++j) o[j] = plane[p][2 * j] | (plane[p][2 * j + 1] << 16);
                    }
                }
}

// IEEE half bits of a float, round to nearest even (host-side weight packing of MATH_F16X2)
static inline uint32_t f16_rne_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, ax = u & 0x7fffffffu;
    if (ax > 0x7f800000u) return sign | 0x7e00u;
    if (ax >= 0x477ff000u) return sign | 0x7c00u;  // rounds to infinity (callers keep |w| far below)
    const int e = (int)(ax >> 23) - 127;
    if (e < -25) return sign;
    if (e >= -14) {
        uint32_t r = ax + 0xfffu + ((ax >> 13) & 1u);  // round the 13 dropped bits to nearest even
        return sign | (((r >> 23) - 112) << 10) | ((r >> 13) & 0x3ffu);
    }
    // subnormal half: unit 2^-24
    const double v = (double)f < 0 ? -(double)f : (double)f;
    const double q = v * 16777216.0;
    uint32_t n = (uint32_t)q;
    const double frac = q - n;
    if (frac > 0.5 || (frac == 0.5 && (n & 1))) ++n;
    return sign | n;  // n == 1024 is the smallest normal: the bit pattern is right as it is
}
static inline float f16_bits_to_float(uint32_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    float f;
    if (e == 0) {
        f = (float)m * 5.9604644775390625e-08f;
        memcpy(&u, &f, 4);
        u |= sign;
    } else if (e == 31) {
        u = sign | 0x7f800000u | (m << 13);
    } else {
        u = sign | ((e + 112) << 23) | (m << 13);
    }
    memcpy(&f, &u, 4);
    return f;
}

size_t f16x2_packed_words(int Cout, int Cin, int K) { return (size_t)(Cout / 32) * K * (Cin / 16) * 2 * 64 * 4; }

// MATH_F16X2 fragments of the fused MRF stage: [tile][tap][16-channel group][plane h | m][lane] x 16 B, k-slot order of
// layout 0 (the packed f32 activation tiles; 1 = the staged planes), weights scaled by 2^13 first (exact), h = half(w'),
// m = half(w' - h); rows in plain order.
bool pack_conv_weights_f16x2(const float* w, int Cout, int Cin, int K, uint32_t* out, int layout) {
    const size_t n = (size_t)Cout * Cin * K;
    for (size_t i = 0; i < n; ++i)
        if (!(std::fabs(w[i]) < 7.99f)) return false;
    const int ng = Cin / 16, nt = Cout / 32;
    for (int tile = 0; tile < nt; ++tile)
        for (int k = 0; k < K; ++k)
            for (int g = 0; g < ng; ++g)
                for (int l = 0; l < 64; ++l) {
                    const int co = 32 * tile + (l & 31), half = l >> 5;
                    uint32_t plane[2][8];
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * g + b3_slot_channel(layout, half, e);
                        const float v = w[((size_t)co * Cin + ci) * K + k] * F16X2_W_SCALE;
                        const uint32_t h = f16_rne_bits(v);
                        plane[0][e] = h;
                        plane[1][e] = f16_rne_bits(v - f16_bits_to_float(h));
                    }
                    for (int p = 0; p < 2; ++p) {
                        uint32_t* o = out + (((((size_t)tile * K + k) * ng + g) * 2 + p) * 64 + l) * 4;
                        for (int j = 0; j < 4; ++j) o[j] = plane[p][2 * j] | (plane[p][2 * j + 1] << 16);
                    }
                }
    return true;
}

void pack_conv_weights_bf16x3(const float* w, int Cout, int Cin, int K, uint32_t* out) {
    pack_conv_weights_bf16x3_mode(w, Cout, Cin, K, EPI_STD, 0, out);
}

void pack_conv_weights_mfma(const float* w, int Cout, int Cin, int K, float* out) {
    pack_conv_weights_mfma_mode(w, Cout, Cin, K, EPI_STD, out);
}

bool conv1d_mfma_supported(int Cin, int Cout, int K, int dil) {
    (void)Cout; (void)K; (void)dil;
    return Cin >= 2 && (Cin % 2) == 0;
}

// One C_in chunk of the implicit GEMM for a wave: MT x NT accumulator tiles, K taps x cpn channel pairs.
// wp[i]: packed A fragments of tile i at (k = 0, first pair of the chunk), lane offset included; record (k, cp)
// sits (k * cpairs + cp) * 64 floats further.  When the step count is a multiple of 8 the A fragments run through an
// 8-register ring four steps ahead (no drain at the loop edge); B fragments (LDS) are fetched one step ahead.
// RING: -1 = plain loop (any step count); 0 = ring pipeline, runtime number of 8-step groups per tap;
// N > 0 = ring pipeline with the N channel pairs of a tap fully unrolled (N = 32 <=> 64-channel chunks): hipcc drains
// the vector-memory counter at every loop header, so the ring keeps its distance only inside straight-line code.
template <int MT, int NT, int RING>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[MT][NT], const float* const (&wp)[MT], const float* __restrict__ xw,
                                           int LD, int K, int cpn, int cpairs, int dil) {
    const int steps = K * cpn;
    float bf_n[NT];
    MI355_UNROLL
    for (int j = 0; j < NT; ++j) bf_n[j] = xw[j * 32];
    int k = 0, cp = 0;
    if (RING >= 0) {
        // cpn % 8 == 0 (launcher): steps are walked in groups of 8 channel pairs of one tap.  Per group one pointer
        // add per tile; inside the unrolled body every A address is that pointer plus an immediate and every B address
        // a running LDS address plus an immediate: the loop is MFMAs, loads and one address add per step.
        // A through an 8-register ring four steps ahead, B through two statically indexed buffers one step ahead.
        const int ld2 = 2 * LD;
        const int jump = (cpairs - cpn) * 64;  // from the end of this chunk's records of tap k to the start of tap k+1
        float ring[MT][8];
        MI355_UNROLL
        for (int u = 0; u < 4; ++u)
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) ring[i][u] = wp[i][u * 64];
        float bb[2][NT];
        MI355_UNROLL
        for (int j = 0; j < NT; ++j) bb[0][j] = bf_n[j];
        const int cpn_s = RING > 0 ? RING : cpn;  // compile-time when the tap is unrolled
        for (int kk = 0; kk < K; ++kk) {
            MI355_UNROLL
            for (int cp0 = 0; cp0 < cpn_s; cp0 += 8) {
                const float* base = xw + kk * dil + cp0 * ld2;
                const bool last_of_tap = cp0 + 8 == cpn_s;
                const bool last = (kk == K - 1) && last_of_tap;
                // (the very last group prefetches harmlessly from itself: loads stay unconditional, so the compiler's
                //  s_waitcnt counts stay exact and the ring keeps its four-step distance)
                const float* nbase = last ? base : (last_of_tap ? xw + (kk + 1) * dil : base + 8 * ld2);
                const float* wg[MT];   // this group's first record
                const float* wg2[MT];  // where records (u + 4) >= 8 live: next group of the tap, or the next tap
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) {
                    wg[i] = wp[i] + (kk * cpairs + cp0) * 64;
                    wg2[i] = last ? wg[i] - 8 * 64 : (last_of_tap ? wg[i] + jump : wg[i]);
                }
                MI355_UNROLL
                for (int u = 0; u < 8; ++u) {
                    if (u < 4) {
                        MI355_UNROLL
                        for (int i = 0; i < MT; ++i) ring[i][u + 4] = wg[i][(u + 4) * 64];
                    } else {
                        MI355_UNROLL
                        for (int i = 0; i < MT; ++i) ring[i][u - 4] = wg2[i][(u + 4) * 64];
                    }
                    if (u < 7) {
                        MI355_UNROLL
                        for (int j = 0; j < NT; ++j) bb[(u + 1) & 1][j] = base[(u + 1) * ld2 + j * 32];
                    } else {
                        MI355_UNROLL
                        for (int j = 0; j < NT; ++j) bb[0][j] = nbase[j * 32];
                    }
                    MI355_UNROLL
                    for (int i = 0; i < MT; ++i)
                        MI355_UNROLL
                        for (int j = 0; j < NT; ++j) acc[i][j] = MFMA_32x32x2_F32(ring[i][u], bb[u & 1][j], acc[i][j]);
                    SCHED_FENCE();
                }
            }
        }
    } else {
        float af_n[MT];
        MI355_UNROLL
        for (int i = 0; i < MT; ++i) af_n[i] = wp[i][0];
        for (int s = 0; s < steps; ++s) {
            float af[MT], bf[NT];
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) af[i] = af_n[i];
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) bf[j] = bf_n[j];
            if (++cp == cpn) { cp = 0; ++k; }
            if (s + 1 < steps) {
                MI355_UNROLL
                for (int i = 0; i < MT; ++i) af_n[i] = wp[i][(k * cpairs + cp) * 64];
                const float* xr = xw + (2 * cp) * LD + k * dil;
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) bf_n[j] = xr[j * 32];
            }
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) acc[i][j] = MFMA_32x32x2_F32(af[i], bf[j], acc[i][j]);
        }
    }
}

// Epilogue shared by the staged kernels (f32 MFMA and split-bf16): C/D layout col = lane&31, row = (r&3) + 8*(r>>2) +
// 4*(lane>>5).  `xs` is the workgroup's LDS (free after the last chunk; at least the size launch_cfg computed).
template <int MT, int NT, int WM, int WN, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], float* xs, int b, int t0, int tile0, int n_tiles,
                                              int wm, int wn, int brow, int bcol, int tid, int out_len) {
    constexpr int T_B = 32 * NT * WN;
    if ((LAB_ABLATE(a) & 4) && acc[0][0][0] != 1.2345f) return;
    if (EPI == EPI_STD && a.ovec) {
        // Through LDS (free after the last chunk) so that global memory sees whole rows: the C/D fragment gives a lane
        // one column of 16 rows, i.e. 128-byte pieces per store instruction; re-read row-major, every lane moves 16
        // bytes and a wave 1 KiB of one row (residual / accumulate reads likewise).  One pass per MT index: the block's
        // WM x 32 rows of that index.  Polyphase ConvTranspose1d (rows = channel-major / phase-minor): the LDS row of a
        // channel is its T_B * s consecutive output samples.
        constexpr int ROWS = 32 * WM;
        const int s_ = a.shuf_s ? a.shuf_s : 1;
        const int rows_o = ROWS / s_;          // output rows (channels) per pass
        const int cols_o = T_B * s_;           // output samples per row
        const int LDO = cols_o + 4;
        const long n_lo = (long)t0 * s_ - a.shuf_p;                              // global sample of LDS column 0
        long n_end = (long)(t0 + T_B < a.T ? t0 + T_B : a.T) * s_ - a.shuf_p;    // exclusive
        const long n_max = a.shuf_s ? a.shuf_T : a.T;
        if (n_end > n_max) n_end = n_max;
        const long n4_lo = n_lo >= 0 ? (n_lo & ~3L) : 0;
        const int quads = (int)((n_end - n4_lo + 3) / 4);
        MI355_UNROLL
        for (int i = 0; i < MT; ++i) {
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) {
                const int tcol = (wn * NT + j) * 32 + bcol;
                MI355_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const int row = wm * 32 + 8 * g + 4 * brow;  // + m, m = 0..3
                    if (a.shuf_s) {
                        const int cop = 32 * (tile0 + i) + 8 * g + 4 * brow;
                        float4 v;
                        v.x = acc[i][j][4 * g + 0] + ((a.bias && cop + 0 < a.Cout) ? a.bias[cop + 0] : 0.0f);
                        v.y = acc[i][j][4 * g + 1] + ((a.bias && cop + 1 < a.Cout) ? a.bias[cop + 1] : 0.0f);
                        v.z = acc[i][j][4 * g + 2] + ((a.bias && cop + 2 < a.Cout) ? a.bias[cop + 2] : 0.0f);
                        v.w = acc[i][j][4 * g + 3] + ((a.bias && cop + 3 < a.Cout) ? a.bias[cop + 3] : 0.0f);
                        *reinterpret_cast<float4*>(xs + (row / s_) * LDO + tcol * s_ + (row % s_)) = v;
                    } else {
                        MI355_UNROLL
                        for (int m = 0; m < 4; ++m) xs[(row + m) * LDO + tcol] = acc[i][j][4 * g + m];
                    }
                }
            }
            __syncthreads();
            if (!a.shuf_s) {
                // plain conv: four items per thread and batch — their global reads (bias, residual, accumulate) first,
                // then the four stores
                constexpr int U = 4;
                const int total = rows_o * quads;
                for (int idx0 = tid; idx0 < total; idx0 += 256 * U) {
                    EpiStd4In L[U];
                    int copv[U];
                    long nv[U];
                    int kind[U];  // 0: nothing, 1: whole quad inside, 2: edge quad
                    MI355_UNROLL
                    for (int u = 0; u < U; ++u) {
                        const int idx = idx0 + 256 * u;
                        kind[u] = 0;
                        if (idx < total) {
                            const int rr = idx / quads, q = idx - rr * quads;
                            copv[u] = 32 * ((blockIdx.y * WM + rr / 32) * MT + i) + rr % 32;
                            nv[u] = n4_lo + 4L * q;
                            if (copv[u] < a.Cout) kind[u] = (nv[u] >= n_lo && nv[u] + 3 < n_end) ? 1 : 2;
                            if (kind[u] == 1) L[u] = epi_std4_load(a, b, copv[u], (int)nv[u]);
                        }
                    }
                    SCHED_FENCE();
                    MI355_UNROLL
                    for (int u = 0; u < U; ++u) {
                        if (kind[u] == 0) continue;
                        const int idx = idx0 + 256 * u;
                        const int rr = idx / quads;
                        const long n = nv[u];
                        const float* src = xs + rr * LDO + (n - n_lo);
                        if (kind[u] == 1) {
                            float4 v;
                            if (((n - n_lo) & 3) == 0) v = *reinterpret_cast<const float4*>(src);
                            else v = make_float4(src[0], src[1], src[2], src[3]);
                            epi_std4_finish(a, b, copv[u], (int)n, v, out_len, L[u]);
                        } else {
                            for (int m = 0; m < 4; ++m) {
                                if (n + m < n_lo || n + m < 0 || n + m >= n_end) continue;
                                epi_std(a, b, copv[u], (int)(n + m), src[m], out_len);
                            }
                        }
                    }
                }
            } else
            for (int idx = tid; idx < rows_o * quads; idx += 256) {
                const int rr = idx / quads, q = idx - rr * quads;
                const int wmr = (rr * s_) / 32;                                    // which wave row this came from
                const int cop = 32 * ((blockIdx.y * WM + wmr) * MT + i) + (rr * s_) % 32;
                if (cop >= a.Cout) continue;
                const long n = n4_lo + 4L * q;
                const float* src = xs + rr * LDO + (n - n_lo);
                if (n >= n_lo && n + 3 < n_end) {
                    float4 v;
                    if (((n - n_lo) & 3) == 0) v = *reinterpret_cast<const float4*>(src);
                    else v = make_float4(src[0], src[1], src[2], src[3]);
                    if (a.shuf_s) *reinterpret_cast<float4*>(a.y + (long)b * a.y_bs + (long)(cop / s_) * a.y_ld + n) = v;
                    else epi_std4(a, b, cop, (int)n, v, out_len);
                } else {
                    for (int m = 0; m < 4; ++m) {
                        if (n + m < n_lo || n + m < 0 || n + m >= n_end) continue;
                        if (a.shuf_s) a.y[(long)b * a.y_bs + (long)(cop / s_) * a.y_ld + n + m] = src[m];
                        else epi_std(a, b, cop, (int)(n + m), src[m], out_len);
                    }
                }
            }
            if (i + 1 < MT) __syncthreads();
        }
        return;
    }
    if (EPI == EPI_STD && a.shuf_s && (a.shuf_s & 3) == 0) {
        epi_polyphase_regs<MT, NT>(a, acc, b, t0 + wn * NT * 32 + bcol, tile0, brow);
        return;
    }
    MI355_UNROLL
    for (int j = 0; j < NT; ++j) {
        const int t = t0 + (wn * NT + j) * 32 + bcol;
        if (t >= a.T) continue;
        if (EPI == EPI_GATE) {
            if (tile0 + 1 < n_tiles) epi_gate_tile(a, b, 32 * (tile0 >> 1), t, brow, acc[0][j], acc[MT - 1][j]);
        } else {
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) {
                if (32 * (tile0 + i) >= a.Cout) continue;
                if (EPI == EPI_RESSKIP) epi_resskip_tile(a, b, 32 * (tile0 + i), t, brow, acc[i][j], out_len);
                else epi_std_tile(a, b, 32 * (tile0 + i), t, brow, acc[i][j], out_len);
            }
        }
    }
}

template <int MT, int NT, int WM, int WN, int EPI, int RING>
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD(MT * NT >= 4 ? 3 : 4) void k_conv1d_mfma(ConvArgs a, int CI_C) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(EPI != EPI_GATE || MT == 2, "gate needs the tile pair in one wave");
    DYN_SMEM(float, xs);  // [CI_C][LD]
    constexpr int T_B = 32 * NT * WN;
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * T_B;
    // staged window starts at ts = floor4(t0 - pad) so that rows can move as aligned 16-byte vectors
    const int tlo = t0 - a.pad;
    const int ts = tlo >= 0 ? (tlo & ~3) : -(((-tlo) + 3) & ~3);
    const int toff = tlo - ts;  // 0..3
    const int LD = (T_B + (a.K - 1) * a.dil + 3 + 3) & ~3;
    const int n_tiles = (EPI == EPI_GATE) ? 2 * ((a.H + 31) / 32) : (a.Cout + 31) / 32;
    const int tile0 = (blockIdx.y * WM + wm) * MT;
    const int cpairs = a.Cin >> 1;
    const int Tin = a.Tin >= 0 ? a.Tin : a.T;
    const int in_len = a.in_len ? a.in_len[b] : Tin;
    const int out_len = a.out_len ? a.out_len[b] : a.T;
    const float* xb = a.x + (long)b * a.x_bs;

    f32x16 acc[MT][NT];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int j = 0; j < NT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int brow = lane >> 5, bcol = lane & 31;
    for (int c0 = 0; c0 < a.Cin; c0 += CI_C) {
        // ---- stage x[c0:c0+CI_C, ts : ts+LD) with mask + leaky-relu fused
        if (!(LAB_ABLATE(a) & 2)) stage_tile_256(xb + (long)c0 * a.x_ld, a.x_ld, CI_C, LD, ts, Tin < in_len ? Tin : in_len, a.in_slope, xs, a.vec);
        __syncthreads();
        if (!(LAB_ABLATE(a) & 1)) {
            const float* wp[MT];
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) {
                int tile = tile0 + i;
                if (tile >= n_tiles) tile = n_tiles - 1;  // out-of-range tile: recompute the last one, discarded below
                wp[i] = a.w + ((long)tile * a.K * cpairs + (c0 >> 1)) * 64 + lane;
            }
            mfma_chunk<MT, NT, RING>(acc, wp, xs + brow * LD + bcol + wn * NT * 32 + toff, LD, a.K, CI_C >> 1, cpairs, a.dil);
        }
        __syncthreads();
    }

    // ---- epilogue
    conv_epilogue<MT, NT, WM, WN, EPI>(a, acc, xs, b, t0, tile0, n_tiles, wm, wn, brow, bcol, tid, out_len);
}

// ------------------------------------------------------------------------------------------------
// The same implicit GEMM with the f32 operands split into three bf16 planes each (hipx.h: split3) and the six leading
// partial products on v_mfma_f32_32x32x16_bf16 — f32-grade results at 6/16 of the f32 MFMA's time (MATH_BF16X3).
//   * an input chunk of CI_C channels is split ONCE while it is staged: LDS holds three planes [plane][16-channel
//     group][half][column] x 16 B, a lane's eight k-slots of one plane side by side, so the loop's B operand is one
//     ds_read_b128 per (plane, column tile) and carries no VALU work at all;
//   * the weights come pre-split in the matching fragment order (pack_conv_weights_bf16x3_mode, layout 1), three
//     global_load_dwordx4 per (row tile, group), fetched one group ahead; taps are a runtime loop, the groups of a
//     chunk are unrolled;
//   * MT x NT accumulator tiles per wave share every fetch (A across NT columns tiles, B across MT row tiles);
//     the epilogues are the f32 kernel's.
// ------------------------------------------------------------------------------------------------
// H2 (MATH_F16X2): two fp16 planes per operand (a.wb3 = pack_conv_weights_f16x2 layout 1), accumulators scaled by 2^17 until
// just before the epilogue.
template <int MT, int NT, int WM, int WN, int EPI, int NG, bool W1 = false, bool H2 = false>
__global__ __launch_bounds__(256) void k_conv1d_b3(ConvArgs a) {
    static_assert(!(W1 && H2), "one reduced-operand variant at a time");
    constexpr int GW = H2 ? 128 : 192;  // uint4 per weight-fragment group
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(EPI != EPI_GATE || MT == 2, "gate needs the tile pair in one wave");
    static_assert(NG % 2 == 0, "an even number of 16-channel groups per chunk");
    DYN_SMEM(float, xs);
    constexpr int T_B = 32 * NT * WN;
    constexpr int CI_C = 16 * NG;
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * T_B;
    const int tlo = t0 - a.pad;
    const int ts = tlo >= 0 ? (tlo & ~3) : -(((-tlo) + 3) & ~3);
    const int toff = tlo - ts;  // 0..3
    const int LD = (T_B + (a.K - 1) * a.dil + 3 + 3) & ~3;
    const int PS = NG * 2 * LD;  // uint4 per plane
    const int n_tiles = (EPI == EPI_GATE) ? 2 * ((a.H + 31) / 32) : (a.Cout + 31) / 32;
    const int tile0 = (blockIdx.y * WM + wm) * MT;
    const int gpt = a.Cin >> 4;  // 16-channel groups per tap in the packed weights
    const int Tin = a.Tin >= 0 ? a.Tin : a.T;
    const int in_len = a.in_len ? a.in_len[b] : Tin;
    const int out_len = a.out_len ? a.out_len[b] : a.T;
    const float* xb = a.x + (long)b * a.x_bs;
    uint4* planes = reinterpret_cast<uint4*>(xs);

    f32x16 acc[MT][NT];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int j = 0; j < NT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int brow = lane >> 5, bcol = lane & 31;
    for (int c0 = 0; c0 < a.Cin; c0 += CI_C) {
        if (!(LAB_ABLATE(a) & 2)) stage_planes<NG, 4, H2>(xb + (long)c0 * a.x_ld, a.x_ld, LD, ts, Tin < in_len ? Tin : in_len, a.in_slope, planes, PS, tid, 256);
        __syncthreads();
        if (!(LAB_ABLATE(a) & 1)) {
            const uint4* wp[MT];
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) {
                int tile = tile0 + i;
                if (tile >= n_tiles) tile = n_tiles - 1;  // out-of-range tile: recompute the last one, discarded below
                wp[i] = reinterpret_cast<const uint4*>(a.wb3) + ((long)tile * a.K * gpt + (c0 >> 4)) * GW + lane;
            }
            if constexpr (H2) h2_chunk<MT, NT, NG, NT>(acc, wp, planes + brow * LD + bcol + wn * NT * 32 + toff, PS, LD, a.K, gpt, a.dil);
            else b3_chunk<MT, NT, NG, NT, W1>(acc, wp, planes + brow * LD + bcol + wn * NT * 32 + toff, PS, LD, a.K, gpt, a.dil);
        }
        __syncthreads();
    }
    if constexpr (H2) {  // back to the unscaled domain (exact)
        MI355_UNROLL
        for (int i = 0; i < MT; ++i)
            MI355_UNROLL
            for (int j = 0; j < NT; ++j)
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= 1.0f / F16X2_ACC_SCALE;
    }
    conv_epilogue<MT, NT, WM, WN, EPI>(a, acc, xs, b, t0, tile0, n_tiles, wm, wn, brow, bcol, tid, out_len);
}

// ------------------------------------------------------------------------------------------------
// Persistent producer / consumer form of k_conv1d_b3 for the polyphase upsamplers (few taps, so little matrix-core work
// per staged chunk: a workgroup that stages, computes and stores in turn leaves the CU idle through two memory round
// trips per tile; measured 0.25 matrix-core busy, 68 - 100 TFLOP/s).  Eight waves: waves 0-3 own the accumulators and
// run b3_chunk + the register scatter epilogue on LDS buffer s & 1 while waves 4-7 stage step s + 1 (next chunk, or the
// first chunk of this workgroup's next tile) into the other buffer; one barrier per step.  The grid is one workgroup per
// CU; tiles are dealt round-robin with the row block fastest, so the row blocks sharing an input tile run side by side.
// The chunk order of an output is the same as in k_conv1d_b3 (results identical).
// ------------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int NG, bool W1 = false, bool H2 = false>
__global__ __launch_bounds__(512) void k_conv1d_b3_pc(ConvArgs a) {
    static_assert(WM * WN == 4, "4 consumer waves per workgroup");
    static_assert(!(W1 && H2), "one reduced-operand variant at a time");
    constexpr int GW = H2 ? 128 : 192;
    DYN_SMEM(float, xs);
    constexpr int T_B = 32 * NT * WN;
    constexpr int CI_C = 16 * NG;
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const bool producer = wid >= 4;
    const int wm = (wid & 3) / WN, wn = (wid & 3) % WN;
    const int LD = (T_B + (a.K - 1) * a.dil + 3 + 3) & ~3;
    const int PS = NG * 2 * LD;  // uint4 per plane
    const int n_tiles = (a.Cout + 31) / 32;
    const int gpt = a.Cin >> 4;
    const int Tin = a.Tin >= 0 ? a.Tin : a.T;
    const int col_tiles = (a.T + T_B - 1) / T_B;
    const int row_blocks = (n_tiles + MT * WM - 1) / (MT * WM);
    const long total = (long)col_tiles * row_blocks * a.B;
    const int nchunks = a.Cin / CI_C;
    // Workgroup w is observed to run on XCD w % 8 (its own L2).  Renumber the workgroups XCD-major so that the row blocks
    // of one column tile — consecutive tile numbers, read the same input tile at the same time — share an L2 instead of
    // fetching it once per XCD (the 256 -> 128 layer has 8 row blocks per input tile).  Placement only: any mapping is correct.
    const unsigned G = gridDim.x;
    const unsigned vid = (G % 8 == 0 && !(LAB_ABLATE(a) & 16)) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const long n_mine = vid < total ? (total - vid + G - 1) / G : 0;
    const long steps = n_mine * nchunks;
    uint4* planes = reinterpret_cast<uint4*>(xs);
    const int brow = lane >> 5, bcol = lane & 31;

    auto stage = [&](long s, int stid, int nthreads) {
        const long i = s / nchunks;
        const int chunk = (int)(s - i * nchunks);
        const long tile = vid + i * G;
        const long q = tile / row_blocks;
        const int ct = (int)(q % col_tiles), b = (int)(q / col_tiles);
        const int in_len = a.in_len ? a.in_len[b] : Tin;
        // all 8 x 2 NG loads of a thread in one batch: the CU's vector-memory path returns data in issue order across
        // waves, so while the producers' HBM loads are in flight the consumers' weight-fragment loads (L2 hits) wait
        // behind them — one miss train per step costs the consumers one memory latency, two cost two
        stage_planes<NG, NG * 2, H2>(a.x + (long)b * a.x_bs + (long)chunk * CI_C * a.x_ld, a.x_ld, LD, ct * T_B - a.pad, Tin < in_len ? Tin : in_len,
                                 a.in_slope, planes + (s & 1) * 3 * PS, PS, stid, nthreads);
    };

    if (steps > 0) stage(0, tid, 512);
    __syncthreads();
    // two loops, one per role, with the same number of barriers: the accumulators exist in the consumer waves' loop only
    // (in one loop with a role branch inside they would be live through the producer branch as well)
    if (producer) {
        for (long s = 0; s < steps; ++s) {
            if (s + 1 < steps && !(LAB_ABLATE(a) & 2)) stage(s + 1, tid - 256, 256);
            __syncthreads();
        }
        return;
    }
    f32x16 acc[MT][NT];
    for (long s = 0; s < steps; ++s) {
        const long i = s / nchunks;
        const int chunk = (int)(s - i * nchunks);
        const long tile = vid + i * G;
        const int rb = (int)(tile % row_blocks);
        const int tile0 = (rb * WM + wm) * MT;
        if (chunk == 0) {
            MI355_UNROLL
            for (int ii = 0; ii < MT; ++ii)
                MI355_UNROLL
                for (int jj = 0; jj < NT; ++jj)
                    MI355_UNROLL
                    for (int r = 0; r < 16; ++r) acc[ii][jj][r] = 0.0f;
        }
        const uint4* wp[MT];
        MI355_UNROLL
        for (int ii = 0; ii < MT; ++ii) {
            int t = tile0 + ii;
            if (t >= n_tiles) t = n_tiles - 1;
            wp[ii] = reinterpret_cast<const uint4*>(a.wb3) + ((long)t * a.K * gpt + (chunk * CI_C >> 4)) * GW + lane;
        }
        if (!(LAB_ABLATE(a) & 1)) {
            if constexpr (H2) h2_chunk_lean<MT, NT, NG>(acc, wp, planes + (s & 1) * 3 * PS + brow * LD + bcol + wn * NT * 32, PS, LD, a.K, gpt, a.dil);
            else b3_chunk_lean<MT, NT, NG, W1>(acc, wp, planes + (s & 1) * 3 * PS + brow * LD + bcol + wn * NT * 32, PS, LD, a.K, gpt, a.dil);
        }
        if constexpr (H2) {
            if (chunk == nchunks - 1) {  // back to the unscaled domain (exact)
                MI355_UNROLL
                for (int ii = 0; ii < MT; ++ii)
                    MI355_UNROLL
                    for (int jj = 0; jj < NT; ++jj)
                        MI355_UNROLL
                        for (int r = 0; r < 16; ++r) acc[ii][jj][r] *= 1.0f / F16X2_ACC_SCALE;
            }
        }
        if (chunk == nchunks - 1 && (!(LAB_ABLATE(a) & 4) || acc[0][0][0] == 1.2345f)) {
            const long q = tile / row_blocks;
            const int ct = (int)(q % col_tiles), b = (int)(q / col_tiles);
            epi_polyphase_regs<MT, NT>(a, acc, b, ct * T_B + wn * NT * 32 + bcol, tile0, brow);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-free variant for pointwise convs (q/k/v/o, res_skip, 1x1 of the duration predictor, flow pre/post) and for
// short-sequence convs (encoder FFN, K = 3, T = phonemes): no halo worth staging / too few workgroups to hide the
// stage-barrier-compute cycle.  Both MFMA operands stream from global memory (A: packed weights; B: 32 consecutive
// time samples of two channels per half-wave, one 128-byte segment per tap, served by L1/L2) through 8-register
// rings four steps ahead; no barriers at all.
// ------------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void k_conv_direct_mfma(ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int T_B = 32 * NT * WN;
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * T_B;
    const int n_tiles = (EPI == EPI_GATE) ? 2 * ((a.H + 31) / 32) : (a.Cout + 31) / 32;
    const int tile0 = (blockIdx.y * WM + wm) * MT;
    const int cpairs = a.Cin >> 1;
    const int Tin = a.Tin >= 0 ? a.Tin : a.T;
    int tend = a.in_len ? a.in_len[b] : Tin;
    if (tend > Tin) tend = Tin;
    const int out_len = a.out_len ? a.out_len[b] : a.T;

    f32x16 acc[MT][NT];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i)
        MI355_UNROLL
        for (int j = 0; j < NT; ++j)
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const float* wp[MT];
    MI355_UNROLL
    for (int i = 0; i < MT; ++i) {
        int tile = tile0 + i;
        if (tile >= n_tiles) tile = n_tiles - 1;
        wp[i] = a.w + (long)tile * a.K * cpairs * 64 + lane;
    }
    const float* xrow = a.x + (long)b * a.x_bs + (long)brow * a.x_ld;
    int tj[NT];
    MI355_UNROLL
    for (int j = 0; j < NT; ++j) tj[j] = t0 - a.pad + (wn * NT + j) * 32 + bcol;
    const long xstep = 2L * a.x_ld;
    const int steps = a.K * cpairs;  // multiple of 8 (checked by the launcher)
    float ra[MT][8], rb[NT][8];
    int kp = 0, cpp = 0;  // prefetch stream position (tap, channel pair)
    auto fetch = [&](int slot) {
        const int kk = kp < a.K ? kp : a.K - 1;
        const int cc = kp < a.K ? cpp : cpairs - 1;
        MI355_UNROLL
        for (int i = 0; i < MT; ++i) ra[i][slot] = wp[i][(kk * cpairs + cc) * 64];
        MI355_UNROLL
        for (int j = 0; j < NT; ++j) {
            const int t = tj[j] + kk * a.dil;
            rb[j][slot] = (t >= 0 && t < tend) ? xrow[cc * xstep + t] : 0.0f;
        }
        if (++cpp == cpairs) { cpp = 0; ++kp; }
    };
    MI355_UNROLL
    for (int u = 0; u < 4; ++u) fetch(u);
    for (int s0 = 0; s0 < steps; s0 += 8) {
        MI355_UNROLL
        for (int u = 0; u < 8; ++u) {
            float af[MT], bf[NT];
            MI355_UNROLL
            for (int i = 0; i < MT; ++i) af[i] = ra[i][u];
            MI355_UNROLL
            for (int j = 0; j < NT; ++j) bf[j] = lrelu_f(rb[j][u], a.in_slope);
            fetch((u + 4) & 7);
            MI355_UNROLL
            for (int i = 0; i < MT; ++i)
                MI355_UNROLL
                for (int j = 0; j < NT; ++j) acc[i][j] = MFMA_32x32x2_F32(af[i], bf[j], acc[i][j]);
            SCHED_FENCE();
        }
    }

    if (EPI == EPI_STD && a.shuf_s && (a.shuf_s & 3) == 0) {
        epi_polyphase_regs<MT, NT>(a, acc, b, t0 + wn * NT * 32 + bcol, tile0, brow);
        return;
    }
    MI355_UNROLL
    for (int j = 0; j < NT; ++j) {
        const int t = t0 + (wn * NT + j) * 32 + bcol;
        if (t >= a.T) continue;
        MI355_UNROLL
        for (int i = 0; i < MT; ++i) {
            if (32 * (tile0 + i) >= a.Cout) continue;
            if (EPI == EPI_RESSKIP) epi_resskip_tile(a, b, 32 * (tile0 + i), t, brow, acc[i][j], out_len);
            else epi_std_tile(a, b, 32 * (tile0 + i), t, brow, acc[i][j], out_len);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-K variant of the LDS-free kernel for deep, short convs (encoder FFN conv_2: K x C_in = 3 x 768 = 1152
// k-steps over only 128 phonemes).  One workgroup = one 32 x 32 output tile; its four waves each take a quarter of
// the k-steps, then reduce through LDS in a fixed order (deterministic) and every wave finishes 4 of the 16 row
// groups.  Grid = all output tiles (768 at batch 32) instead of a quarter of them, with k-chains a quarter as long.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void k_conv_direct_splitk(ConvArgs a) {
    DYN_SMEM(float, red);  // [4 waves][16 regs][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32;
    const int tile = blockIdx.y;
    const int cpairs = a.Cin >> 1;
    const int Tin = a.Tin >= 0 ? a.Tin : a.T;
    int tend = a.in_len ? a.in_len[b] : Tin;
    if (tend > Tin) tend = Tin;
    const int out_len = a.out_len ? a.out_len[b] : a.T;

    f32x16 acc;
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* wp = a.w + (long)tile * a.K * cpairs * 64 + lane;
    const float* xrow = a.x + (long)b * a.x_bs + (long)brow * a.x_ld;
    const int tj = t0 - a.pad + bcol;
    const long xstep = 2L * a.x_ld;
    const int steps = (a.K * cpairs) >> 2;  // per wave; multiple of 8 (launcher)
    const int s_begin = wid * steps;
    int kp = s_begin / cpairs, cpp = s_begin - kp * cpairs;  // prefetch stream position
    const int k_end = a.K;
    float ra[8], rb[8];
    auto fetch = [&](int slot) {
        const int kk = kp < k_end ? kp : k_end - 1;
        const int cc = kp < k_end ? cpp : cpairs - 1;
        ra[slot] = wp[(kk * cpairs + cc) * 64];
        const int t = tj + kk * a.dil;
        rb[slot] = (t >= 0 && t < tend) ? xrow[cc * xstep + t] : 0.0f;
        if (++cpp == cpairs) { cpp = 0; ++kp; }
    };
    MI355_UNROLL
    for (int u = 0; u < 4; ++u) fetch(u);
    for (int s0 = 0; s0 < steps; s0 += 8) {
        MI355_UNROLL
        for (int u = 0; u < 8; ++u) {
            const float af = ra[u];
            const float bf = lrelu_f(rb[u], a.in_slope);
            fetch((u + 4) & 7);
            acc = MFMA_32x32x2_F32(af, bf, acc);
            SCHED_FENCE();
        }
    }
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) red[(wid * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    const int t = t0 + bcol;
    if (t < a.T) {
        MI355_UNROLL
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wid + q;
            const float v = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) +
                            red[(3 * 16 + r) * 64 + lane];
            const int co = 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * brow;
            if (co < a.Cout) {
                if (EPI == EPI_RESSKIP) epi_resskip(a, b, co, t, v, out_len);
                else epi_std(a, b, co, t, v, out_len);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Short-sequence dense convs of the text encoder (q/k/v, o, FFN conv_1 / conv_2: T = phonemes, 128 x 32 columns at the
// bench shape) in MATH_BF16X3 / BF16W.  With so few columns a kernel is either bound by launching too few workgroups or by
// re-streaming the weights, and k_conv1d_b3's stage / barrier / compute cycle per 32-channel chunk is mostly exposed
// latency.  Here a workgroup owns a 64 x 64 output tile over ONE 192-channel slice of the input (the encoder's hidden
// width; conv_2's 768 input channels are four slices = four workgroups whose raw sums the following LayerNorm launch adds
// up in slice order): the slice (+ K - 1 halo columns) is staged ONCE as three bf16 planes (76 KiB: two workgroups per CU,
// one stages while the other computes), then each of the four waves runs its 32 x 32 tile's whole k-range without another
// barrier (b3_chunk: 36 groups x 6 products at K = 3), weights streamed from L2 (each fragment by two waves).
// Grid at batch 32: q/k/v 576, o 192, conv_1 768, conv_2 4 x 192 workgroups; one utterance of 180 phonemes: 27 / 9 / 36 /
// 36.  The tile shape and the slice order are fixed by the layer: a row's bits do not depend on what it is batched with.
// ------------------------------------------------------------------------------------------------
constexpr int ENC_CS = 192, ENC_NG = ENC_CS / 16, ENC_TB = 64;

// NG 16-channel groups per slice: 12 (192 channels, the encoder's width) or 6 (96: the coupling layers' half of the latent,
// flow.pre — the same kernel serves the two pointwise convs around each WaveNet stack at frame resolution)
template <bool W1, int NG>
__global__ __launch_bounds__(256) void k_enc_b3(ConvArgs a) {
    DYN_SMEM(float, smem);
    uint4* planes = reinterpret_cast<uint4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.z / a.ksplit, sl = blockIdx.z - b * a.ksplit;
    const int t0 = blockIdx.x * ENC_TB;
    const int c0 = sl * 16 * NG;
    int tend = a.in_len ? a.in_len[b] : a.T;
    if (tend > a.T) tend = a.T;
    const int out_len = a.out_len ? a.out_len[b] : a.T;
    const int LD = ENC_TB + (a.K - 1) * a.dil;
    const int PS = NG * 2 * LD;
    if (!(LAB_ABLATE(a) & 2))
        stage_planes<NG, NG == 12 ? 8 : 4>(a.x + (long)b * a.x_bs + (long)c0 * a.x_ld, a.x_ld, LD, t0 - a.pad, tend, a.in_slope, planes, PS, tid, 256);
    __syncthreads();
    // a.rb_loop row blocks of 64 output channels per workgroup, one after the other over the SAME staged slice (large grids, round 6:
    // at batch 256 each of a conv's 3 - 12 row-block workgroups staged and split the slice again — 9.5 VALU per MFMA, MFMA pipe 0.31
    // busy).  No barrier inside the loop; an output element's products and their order do not change: same bits as rb_loop = 1.
    const int R = a.rb_loop > 1 ? a.rb_loop : 1;
    const int ngt = a.Cin / 16;
    const int t = t0 + wn * 32 + bcol;
    for (int rr = 0; rr < R; ++rr) {
        const int rt = (blockIdx.y * R + rr) * 2 + wm;  // 32-row tile of the output
        if (32 * rt >= a.Cout) break;
        f32x16 acc[1][1];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
        const uint4* wp[1] = {reinterpret_cast<const uint4*>(a.wb3) + ((long)rt * a.K * ngt + c0 / 16) * 192 + lane};
        if (!(LAB_ABLATE(a) & 1)) b3_chunk<1, 1, NG, 1, W1>(acc, wp, planes + brow * LD + bcol + wn * 32, PS, LD, a.K, ngt, a.dil);
        if (t >= a.T) continue;
        if (a.ksplit == 1) {
            epi_std_tile(a, b, 32 * rt, t, brow, acc[0][0], out_len);
        } else {
            float* pp = a.part + (((long)sl * a.B + b) * a.Cout + 32 * rt) * a.T + t;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r)
                if (32 * rt + tile_row(r, brow) < a.Cout) pp[(long)tile_row(r, brow) * a.T] = acc[0][0][r];
        }
    }
}

// epi_std_tile for full 32-row tiles (Cout % 32 == 0) with buffer addressing: a row's byte offset rides in an SGPR (wave-uniform:
// tile base + the accumulator row's part), the lane's part (its column and brow) in ONE VGPR, lanes past the row's columns are
// switched off by the buffer range check — no 64-bit address arithmetic, no divergent branch around a tile.  Operation for operation
// the arithmetic of epi_std_tile (a.cond / a.shuf_s not supported: callers check).
__device__ __forceinline__ void epi_std_tile_buf(const ConvArgs& a, int b, int co0, int t, bool t_ok, int brow, const f32x16& acc, int out_len) {
    const BufRsrc yb = buf_rsrc(a.y + (long)b * a.y_bs);
    const unsigned vy = t_ok ? 4u * (unsigned)(t + 4 * brow * a.y_ld) : BUF_OOB;
    float v[16], rv[16], yv[16];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    if (a.bias) {
        const BufRsrc bb = buf_rsrc(a.bias);
        float bv[16];
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) bv[r] = buf_load_f32(bb, 16u * (unsigned)brow, 4u * (unsigned)(co0 + (r & 3) + 8 * (r >> 2)));
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) v[r] += bv[r];
    }
    if (a.res) {
        const BufRsrc rb = buf_rsrc(a.res + (long)b * a.res_bs);
        const unsigned vr = t_ok ? 4u * (unsigned)(t + 4 * brow * a.res_ld) : BUF_OOB;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) rv[r] = buf_load_f32(rb, vr, 4u * (unsigned)((co0 + (r & 3) + 8 * (r >> 2)) * a.res_ld));
    }
    if (a.accumulate) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) yv[r] = buf_load_f32(yb, vy, 4u * (unsigned)((co0 + (r & 3) + 8 * (r >> 2)) * a.y_ld));
    }
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) {
        float q = v[r];
        if (a.relu) q = fmaxf(q, 0.0f);
        if (a.mask_before_res && t >= out_len) q = 0.0f;
        if (a.res) q = a.res_sub ? rv[r] - q : rv[r] + q;
        q *= a.out_scale;
        if (!a.mask_before_res && t >= out_len) q = 0.0f;
        if (a.accumulate) q += yv[r];
        buf_store_f32(yb, vy, 4u * (unsigned)((co0 + (r & 3) + 8 * (r >> 2)) * a.y_ld), q);
    }
}

// k_enc_b3 for LARGE grids (round 6; batch 256: 32,768 phoneme columns, and the flow's pointwise convs at frame resolution).  The
// 64 x 64 form above streams every weight fragment through the L1 for ONE 32-column tile (and twice: its two column-tile waves load
// the same rows): 8 waves x 3 KiB per 192 MFMA cycles = 128 B per clock against the L1's 64 — the matrix pipe cannot pass 0.5
// (measured 0.31 - 0.35 busy).  Here a workgroup owns 32 NCT columns (128) x NW 32-row tiles: the slice of those columns is staged
// once (150 KiB at 192 channels), wave w streams row tile w's fragments — nobody else's — and every fragment feeds NCT column tiles
// (24 MFMAs = 768 cycles per 3 KiB: 32 B per clock and CU with eight waves); a.rb_loop row blocks of NW tiles one after the other.
// b3_chunk walks k-groups and taps in the same order for every accumulator: an output element has the bits of the 64 x 64 form.
template <bool W1, int NG, int NCT, int NW, bool SIX = false>
__global__ __launch_bounds__(64 * NW) void k_enc_b3w(ConvArgs a) {
    static_assert(!SIX || (NW == 8 && NCT == 4), "six row tiles x four column tiles dealt to eight waves");
    DYN_SMEM(float, smem);
    uint4* planes = reinterpret_cast<uint4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int b = blockIdx.z / a.ksplit, sl = blockIdx.z - b * a.ksplit;
    const int t0 = blockIdx.x * (32 * NCT);
    const int c0 = sl * 16 * NG;
    int tend = a.in_len ? a.in_len[b] : a.T;
    if (tend > a.T) tend = a.T;
    const int out_len = a.out_len ? a.out_len[b] : a.T;
    const int LD = 32 * NCT + (a.K - 1) * a.dil;
    const int PS = NG * 2 * LD;
    if (!(LAB_ABLATE(a) & 2))
        stage_planes<NG, NG == 12 ? 8 : 4>(a.x + (long)b * a.x_bs + (long)c0 * a.x_ld, a.x_ld, LD, t0 - a.pad, tend, a.in_slope, planes, PS, tid, 64 * NW);
    __syncthreads();
    const int R = a.rb_loop > 1 ? a.rb_loop : 1;
    const int ngt = a.Cin / 16;
    // one 32 x 32 output tile through the conv's epilogue (lane coordinates opaque: the epilogue's 4 x 16 row addresses are loop-invariant,
    // and hoisted in front of the row-block loop they spill — 544 bytes of scratch and a drain per reload)
    auto store_tile = [&](int rt, int j, const f32x16& acc, int bcol_e, int brow_e) MI355_INLINE_LAMBDA {
        const int t = t0 + 32 * j + bcol_e;
        if (a.ksplit == 1) {
            epi_std_tile_buf(a, b, 32 * rt, t, t < a.T, brow_e, acc, out_len);
        } else {  // a slice's raw sums (the following LayerNorm launch adds the slices up)
            const BufRsrc pb = buf_rsrc(a.part + (((long)sl * a.B + b) * a.Cout + 32 * rt) * a.T);
            const unsigned vp = t < a.T ? 4u * (unsigned)(t + 4 * brow_e * a.T) : BUF_OOB;
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) buf_store_f32(pb, vp, 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * a.T), acc[r]);
        }
    };
    for (int rr = 0; rr < R; ++rr) {
        if constexpr (SIX) {
            // Row blocks of SIX row tiles on eight waves (192-row convs: FFN conv_2, q / k / v, the couplings' pre conv): with one row tile per
            // wave six waves sit on four SIMDs — two SIMDs carry two waves, the workgroup takes 2 x 4 tile units.  Here the 24 (row tile,
            // column tile) units are dealt three to a wave: waves 0 - 5 take column tiles 0 - 2 of row tile w, waves 6 / 7 column tile 3 of
            // row tiles 0 - 2 / 3 - 5 (their fragments feed one column tile each: 12 instead of 6 fragment streams per workgroup, still
            // below the L1's rate with two waves per SIMD).  b3_chunk_lean walks k-groups and taps in one order for every accumulator:
            // an output element has the bits of the other forms.
            const int rt0 = (blockIdx.y * R + rr) * 6;
            if (32 * rt0 >= a.Cout) break;
            int bcol_e = bcol, brow_e = brow;
            if (wid < 6) {
                const int rt = rt0 + wid;
                f32x16 acc[1][3];
                MI355_UNROLL
                for (int j = 0; j < 3; ++j)
                    MI355_UNROLL
                    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
                const uint4* wp[1] = {reinterpret_cast<const uint4*>(a.wb3) + ((long)rt * a.K * ngt + c0 / 16) * 192 + lane};
                if (!(LAB_ABLATE(a) & 1)) b3_chunk_lean<1, 3, NG, W1>(acc, wp, planes + brow * LD + bcol, PS, LD, a.K, ngt, a.dil);
                OPAQUE_V(bcol_e);
                OPAQUE_V(brow_e);
                MI355_UNROLL
                for (int j = 0; j < 3; ++j) store_tile(rt, j, acc[0][j], bcol_e, brow_e);
            } else {
                const int rtb = rt0 + 3 * (wid - 6);
                f32x16 acc[3][1];
                const uint4* wp[3];
                MI355_UNROLL
                for (int i = 0; i < 3; ++i) {
                    MI355_UNROLL
                    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.0f;
                    wp[i] = reinterpret_cast<const uint4*>(a.wb3) + ((long)(rtb + i) * a.K * ngt + c0 / 16) * 192 + lane;
                }
                if (!(LAB_ABLATE(a) & 1)) b3_chunk_lean<3, 1, NG, W1>(acc, wp, planes + brow * LD + bcol + 96, PS, LD, a.K, ngt, a.dil);
                OPAQUE_V(bcol_e);
                OPAQUE_V(brow_e);
                MI355_UNROLL
                for (int i = 0; i < 3; ++i) store_tile(rtb + i, 3, acc[i][0], bcol_e, brow_e);
            }
        } else {
            const int rt = (blockIdx.y * R + rr) * NW + wid;  // 32-row tile of the output
            if (32 * rt >= a.Cout) break;
            f32x16 acc[1][NCT];
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j)
                MI355_UNROLL
                for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
            const uint4* wp[1] = {reinterpret_cast<const uint4*>(a.wb3) + ((long)rt * a.K * ngt + c0 / 16) * 192 + lane};
            if (!(LAB_ABLATE(a) & 1)) b3_chunk_lean<1, NCT, NG, W1>(acc, wp, planes + brow * LD + bcol, PS, LD, a.K, ngt, a.dil);  // (B fragments single-buffered: the double-buffered loop spills at four column tiles)
            int bcol_e = bcol, brow_e = brow;
            OPAQUE_V(bcol_e);
            OPAQUE_V(brow_e);
            MI355_UNROLL
            for (int j = 0; j < NCT; ++j) store_tile(rt, j, acc[0][j], bcol_e, brow_e);
        }
    }
}

// The attention block's tail in one launch:  y = LN_c(res + conv1x1(x) + bias)   (o-proj + residual + LayerNorm; 192 channels).
// A workgroup owns 32 columns of ALL 192 output rows (six waves = six 32-row tiles over the one 192-channel slice, staged once as
// planes like k_enc_b3), so the LayerNorm's channel statistics never leave the CU: the conv result (+ bias + residual) goes through
// LDS into the (column, channel group) layout, 12 groups of 16 channels summed in a fixed order.  In place on the residual allowed
// (a workgroup reads and writes its own columns only).
struct EncOLnArgs {
    const float* x; long x_bs; int x_ld;       // conv input [B, 192, T]
    const float* wb3;                          // layout-1 bf16 planes of the 1x1 conv [6][1][12][3][64] uint4
    const float* bias;                         // [192] or null
    const float* res; long res_bs; int res_ld; // residual [B, 192, T]
    const float* gamma; const float* beta;
    float* y; long y_bs; int y_ld;
    const int* in_len;                         // mask of the conv input or null
    const int* out_len;                        // y = 0 at t >= out_len[b] (after the LN) or null
    int B, T;
    float eps;
};

template <bool W1>
__global__ __launch_bounds__(384) void k_enc_o_ln(EncOLnArgs a) {
    constexpr int C = ENC_CS, NG = ENC_NG, TB = 32, NCG = 12, NPT = 16;
    DYN_SMEM(float, smem);
    uint4* planes = reinterpret_cast<uint4*>(smem);  // 3 planes x [12 groups][2 halves][32 columns] x 16 B = 36 KiB
    float* V = smem;                                 // later: [C][TB] conv result + bias + residual (24 KiB)
    float* red = smem + 3 * NG * 2 * TB * 4;         // [NCG][TB]
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int col = tid & 31, cg = tid >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TB;
    int tend = a.in_len ? a.in_len[b] : a.T;
    if (tend > a.T) tend = a.T;
    constexpr int PS = NG * 2 * TB;
    stage_planes<NG, 2>(a.x + (long)b * a.x_bs, a.x_ld, TB, t0, tend, 1.0f, planes, PS, tid, 384);  // 12 column sets x 2 rows
    // the epilogue's operands, loaded while the planes settle: bias and residual of this lane's 16 rows, gamma / beta of this
    // thread's 16 channels (every load unconditional, clamped column)
    const int t = t0 + bcol, tc = t < a.T ? t : a.T - 1;
    float bv[16], rv[16], gm[NPT], bt[NPT];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * w + tile_row(r, brow);
        bv[r] = a.bias ? a.bias[co] : 0.0f;
        rv[r] = a.res[(long)b * a.res_bs + (long)co * a.res_ld + tc];
    }
    MI355_UNROLL
    for (int i = 0; i < NPT; ++i) {
        gm[i] = a.gamma[cg + NCG * i];
        bt[i] = a.beta[cg + NCG * i];
    }
    __syncthreads();
    f32x16 acc[1][1];
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    const uint4* wp[1] = {reinterpret_cast<const uint4*>(a.wb3) + (long)w * NG * 192 + lane};
    b3_chunk<1, 1, NG, 1, W1>(acc, wp, planes + brow * TB + bcol, PS, TB, 1, NG, 1);
    __syncthreads();  // every wave is done with the planes: the f32 tile takes their place
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) V[(32 * w + tile_row(r, brow)) * TB + bcol] = rv[r] + (acc[0][0][r] + bv[r]);
    __syncthreads();
    float v[NPT];
    float sum = 0.0f;
    MI355_UNROLL
    for (int i = 0; i < NPT; ++i) {
        v[i] = V[(cg + NCG * i) * TB + col];
        sum += v[i];
    }
    auto col_sum = [&](float x) {
        __syncthreads();
        red[cg * TB + col] = x;
        __syncthreads();
        float s = 0.0f;
        MI355_UNROLL
        for (int g = 0; g < NCG; ++g) s += red[g * TB + col];
        return s;
    };
    const float mean = col_sum(sum) / (float)C;
    float sq = 0.0f;
    MI355_UNROLL
    for (int i = 0; i < NPT; ++i) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(col_sum(sq) / (float)C + a.eps);
    const int to = t0 + col;
    if (to >= a.T) return;
    const bool masked = a.out_len && to >= a.out_len[b];
    float* yo = a.y + (long)b * a.y_bs + to;
    MI355_UNROLL
    for (int i = 0; i < NPT; ++i) {
        const float y = (v[i] - mean) * rstd * gm[i] + bt[i];
        yo[(long)(cg + NCG * i) * a.y_ld] = masked ? 0.0f : y;
    }
}

bool enc_o_ln_supported(int Cin, int Cout, int K) { return Cin == ENC_CS && Cout == ENC_CS && K == 1; }

void launch_enc_o_ln(const ConvArgs& c, const float* gamma, const float* beta, const int* ln_out_len, float eps, hipStream_t s) {
    if (c.T <= 0 || c.B <= 0) return;
    if (!enc_o_ln_supported(c.Cin, c.Cout, c.K) || !c.wb3 || !c.res || c.epi != EPI_STD || c.relu || c.cond || c.accumulate || c.res_sub ||
        c.out_scale != 1.0f || c.in_slope != 1.0f || c.out_len || c.shuf_s)
        throw std::runtime_error("enc_o_ln: unsupported conv");
    EncOLnArgs a{c.x, c.x_bs, c.x_ld, c.wb3, c.bias, c.res, c.res_bs, c.res_ld, gamma, beta, c.y, c.y_bs, c.y_ld, c.in_len, ln_out_len, c.B, c.T, eps};
    dim3 grid((c.T + 31) / 32, c.B);
    const size_t shmem = (size_t)3 * ENC_NG * 2 * 32 * 16 + (size_t)12 * 32 * sizeof(float);
    if (c.math == MATH_BF16W) LAUNCH_KERNEL(k_enc_o_ln<true>, grid, dim3(384), shmem, s, a);
    else LAUNCH_KERNEL(k_enc_o_ln<false>, grid, dim3(384), shmem, s, a);
}

bool enc_conv_b3_supported(int Cin, int Cout, int K, int dil) {
    return ((Cin >= ENC_CS && Cin % ENC_CS == 0) || Cin == ENC_CS / 2) && Cout >= 1 && (K == 1 || K == 3) && dil == 1;
}
int enc_conv_b3_slices(int Cin) { return Cin >= ENC_CS ? Cin / ENC_CS : 1; }

void launch_enc_conv_b3(const ConvArgs& a_in, hipStream_t s) {
    if (a_in.T <= 0 || a_in.B <= 0) return;
    ConvArgs a = a_in;
    if (!enc_conv_b3_supported(a.Cin, a.Cout, a.K, a.dil) || a.epi != EPI_STD || !a.wb3 || a.shuf_s || a.Tin >= 0)
        throw std::runtime_error("enc_conv_b3: unsupported conv");
    if (a.ksplit != enc_conv_b3_slices(a.Cin)) throw std::runtime_error("enc_conv_b3: one workgroup per 192-channel slice");
    const int ng = a.Cin >= ENC_CS ? ENC_NG : ENC_NG / 2;
    if (a.ksplit > 1 && !a.part) throw std::runtime_error("enc_conv_b3: split conv without a partial-sum buffer");
    static const int ablate = lab_getenv("MI355VITS_CONV_ABLATE") ? atoi(lab_getenv("MI355VITS_CONV_ABLATE")) : 0;
    a.ablate = ablate;
    // large grids: the 128-column form (k_enc_b3w) when it still gives every CU a workgroup: one staged slice per 128 columns, every
    // row tile's fragments streamed by ONE wave and used for four column tiles; all of the conv's row blocks in one workgroup
    {
        constexpr int NCT = 4;
        const int nrt = (a.Cout + 31) / 32;
        const int nw = nrt % 8 == 0 ? 8 : (nrt % 6 == 0 ? 6 : 8);
        const int nblk = (nrt + nw - 1) / nw;
        const long wgs_w = (long)((a.T + 32 * NCT - 1) / (32 * NCT)) * a.B * a.ksplit;  // (with every row block inside the workgroup)
        // (the buffer-addressed epilogue: whole 32-row tiles; fewer than six row tiles — the couplings' 192 -> 96 post conv — leave most of
        // the workgroup's waves without work: +21 % measured, profiles/r06_enc_wide_ab.txt)
        bool wide = wgs_w >= current_device_cu_count() && a.Cout % 32 == 0 && !a.cond && nrt >= 6;
        if (const char* f = lab_getenv("MI355VITS_ENC_WIDE")) wide = atoi(f) != 0 && a.Cout % 32 == 0 && !a.cond && (nrt >= 6 || atoi(f) > 1);  // lab / tests (2: also below six row tiles)
        if (wide) {
            a.rb_loop = nblk;
            const int LDw = 32 * NCT + (a.K - 1) * a.dil;
            const size_t shw = (size_t)3 * ng * 2 * LDw * 16;
            dim3 gridw((a.T + 32 * NCT - 1) / (32 * NCT), 1, a.B * a.ksplit);
            auto gow = [&](auto kfn, int threads) {
#ifndef MI355_EMU
                set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
#endif
                LAUNCH_KERNEL(kfn, gridw, dim3(threads), shw, s, a);
            };
            const bool w1 = a.math == MATH_BF16W;
            // row blocks of six tiles: dealt to eight waves (SIX) unless the lab switch asks for the round-6a form (one row tile per wave, six waves)
            bool six8 = nw == 6;
            if (const char* f = lab_getenv("MI355VITS_ENC_SIX8")) six8 = six8 && atoi(f) != 0;
            if (ng == ENC_NG) {
                if (nw == 8) { if (w1) gow(k_enc_b3w<true, ENC_NG, NCT, 8>, 512); else gow(k_enc_b3w<false, ENC_NG, NCT, 8>, 512); }
                else if (six8) { if (w1) gow(k_enc_b3w<true, ENC_NG, NCT, 8, true>, 512); else gow(k_enc_b3w<false, ENC_NG, NCT, 8, true>, 512); }
                else { if (w1) gow(k_enc_b3w<true, ENC_NG, NCT, 6>, 384); else gow(k_enc_b3w<false, ENC_NG, NCT, 6>, 384); }
            } else {
                if (nw == 8) { if (w1) gow(k_enc_b3w<true, ENC_NG / 2, NCT, 8>, 512); else gow(k_enc_b3w<false, ENC_NG / 2, NCT, 8>, 512); }
                else if (six8) { if (w1) gow(k_enc_b3w<true, ENC_NG / 2, NCT, 8, true>, 512); else gow(k_enc_b3w<false, ENC_NG / 2, NCT, 8, true>, 512); }
                else { if (w1) gow(k_enc_b3w<true, ENC_NG / 2, NCT, 6>, 384); else gow(k_enc_b3w<false, ENC_NG / 2, NCT, 6>, 384); }
            }
            return;
        }
    }
    const int LD = ENC_TB + (a.K - 1) * a.dil;
    const size_t shmem = (size_t)3 * ng * 2 * LD * 16;
    // otherwise, on grids that are still large: several 64-row blocks per workgroup over one staged slice — the most that divides the conv's row blocks and still
    // leaves the launch four workgroups per CU (batch 256: FFN conv_1 six of twelve, q/k/v and conv_2 three; profiles/r06_enc_row_loop_ab.txt)
    const int nrb = (a.Cout + 63) / 64;
    const long wgs = (long)((a.T + ENC_TB - 1) / ENC_TB) * nrb * a.B * a.ksplit;
    a.rb_loop = 1;
    for (int r : {6, 4, 3, 2})
        if (nrb % r == 0 && wgs / r >= 4L * current_device_cu_count()) { a.rb_loop = r; break; }
    if (const char* f = lab_getenv("MI355VITS_ENC_ROWLOOP")) a.rb_loop = atoi(f) > 1 && nrb % atoi(f) == 0 ? atoi(f) : 1;  // lab / tests
    dim3 grid((a.T + ENC_TB - 1) / ENC_TB, (nrb + a.rb_loop - 1) / a.rb_loop, a.B * a.ksplit);
    auto go = [&](auto kfn) {
#ifndef MI355_EMU
        set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
#endif
        LAUNCH_KERNEL(kfn, grid, dim3(256), shmem, s, a);
    };
    if (ng == ENC_NG) {
        if (a.math == MATH_BF16W) go(k_enc_b3<true, ENC_NG>);
        else go(k_enc_b3<false, ENC_NG>);
    } else {
        if (a.math == MATH_BF16W) go(k_enc_b3<true, ENC_NG / 2>);
        else go(k_enc_b3<false, ENC_NG / 2>);
    }
}

namespace {

struct TileCfg { int MT, NT, WM, WN; };

template <int MT, int NT, int WM, int WN, int EPI>
void launch_cfg(const ConvArgs& a, int n_tiles, hipStream_t s) {
    constexpr int T_B = 32 * NT * WN;
    const int LD = (T_B + (a.K - 1) * a.dil + 3 + 3) & ~3;
    // C_in chunk: fixed by C_in alone (largest even divisor <= 64) so that the summation order — and with it
    // every output bit — does not depend on the tile shape chosen for a batch size; only a receptive field too
    // large for LDS shrinks it further.
    static const size_t lds_cap = [] { const char* e = lab_getenv("MI355VITS_CONV_LDS_KB"); return (size_t)(e ? atoi(e) : 60) * 1024; }();
    auto fits = [&](int c) { return (size_t)c * LD * sizeof(float) <= lds_cap; };
    int ci_c = 0;
    static const int chunk_max = [] { const char* e = lab_getenv("MI355VITS_CONV_CHUNK"); return e ? atoi(e) : 64; }();
    for (int c = chunk_max; c >= 2; c -= 2)
        if (a.Cin % c == 0 && fits(c)) { ci_c = c; break; }
    if (ci_c == 0) throw std::runtime_error("conv1d_mfma: receptive field too large for LDS staging");
    size_t shmem = (size_t)ci_c * LD * sizeof(float);
    dim3 grid((a.T + T_B - 1) / T_B, (n_tiles + MT * WM - 1) / (MT * WM), a.B);
    ConvArgs av = a;
    av.vec = (a.x_ld % 4 == 0) && (a.x_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0);
    av.yvec = (a.y_ld % 4 == 0) && (a.y_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.y) % 16 == 0);
    // row-major epilogue through LDS (EPI_STD): needs 16-byte aligned output / residual rows; for the polyphase
    // scatter also a phase count that divides the 32-row tile and keeps a lane's 4 rows inside one channel
    static const bool no_ovec = lab_getenv("MI355VITS_CONV_NO_OVEC") != nullptr;
    // the polyphase scatter straight from registers already writes 1 KiB contiguous per store instruction (a lane owns
    // 4 consecutive samples); routing it through LDS measured slower (upsample 1.91 -> 2.34 ms/step), so it is opt-in
    static const bool polyphase_via_lds = lab_getenv("MI355VITS_CONV_POLY_LDS") != nullptr;
    av.ovec = EPI == EPI_STD && !no_ovec && av.yvec &&
              (!a.res || ((a.res_ld % 4 == 0) && (a.res_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.res) % 16 == 0))) &&
              (!a.shuf_s || (polyphase_via_lds && 32 % a.shuf_s == 0 && a.shuf_s % 4 == 0 && a.Cout % 4 == 0));
    if (av.ovec) {
        const size_t s_ = a.shuf_s ? a.shuf_s : 1;
        const size_t need = (32 * WM / s_) * (T_B * s_ + 4) * sizeof(float);
        if (need > shmem) shmem = need;
    }
    if (ci_c == 64) {
        auto kfn = k_conv1d_mfma<MT, NT, WM, WN, EPI, 32>;
        LAUNCH_KERNEL(kfn, grid, dim3(256), shmem, s, av, ci_c);
    } else if (((ci_c >> 1) & 7) == 0) {
        auto kfn = k_conv1d_mfma<MT, NT, WM, WN, EPI, 0>;
        LAUNCH_KERNEL(kfn, grid, dim3(256), shmem, s, av, ci_c);
    } else {
        auto kfn = k_conv1d_mfma<MT, NT, WM, WN, EPI, -1>;
        LAUNCH_KERNEL(kfn, grid, dim3(256), shmem, s, av, ci_c);
    }
}

// split-bf16 staged kernel: 32-channel chunks (two 16-channel groups: 6 * 32 * LD bytes of LDS); convs with one or two taps
// (the polyphase upsamplers) take 64-channel chunks when C_in allows — with so few taps a 32-channel chunk is only 72 - 144
// MFMAs per wave between two stage / barrier cycles.  The chunk size is a function of the layer alone (never of the batch),
// so the summation order of an output does not depend on what it is batched with.
// compute units of the current device (persistent grids), looked up once per device
static int device_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        hipDeviceProp_t p;
        n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <int MT, int NT, int WM, int WN, int EPI>
void launch_b3(const ConvArgs& a, int n_tiles, hipStream_t s) {
    constexpr int T_B = 32 * NT * WN;
    const int LD = (T_B + (a.K - 1) * a.dil + 3 + 3) & ~3;
    const bool wide = a.K <= 2 && a.Cin % 64 == 0 && EPI == EPI_STD;
    size_t shmem = (size_t)6 * (wide ? 64 : 32) * LD;
    dim3 grid((a.T + T_B - 1) / T_B, (n_tiles + MT * WM - 1) / (MT * WM), a.B);
    ConvArgs av = a;
    av.vec = (a.x_ld % 4 == 0) && (a.x_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.x) % 16 == 0);
    av.yvec = (a.y_ld % 4 == 0) && (a.y_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.y) % 16 == 0);
    static const bool no_ovec = lab_getenv("MI355VITS_CONV_NO_OVEC") != nullptr;
    av.ovec = EPI == EPI_STD && !no_ovec && av.yvec && !a.shuf_s &&
              (!a.res || ((a.res_ld % 4 == 0) && (a.res_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.res) % 16 == 0)));
    if (av.ovec) {
        const size_t need = (size_t)(32 * WM) * (T_B + 4) * sizeof(float);
        if (need > shmem) shmem = need;
    }
    auto go = [&](auto kfn) {
#ifndef MI355_EMU
        if (shmem > 64 * 1024) {
            set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
        }
#endif
        LAUNCH_KERNEL(kfn, grid, dim3(256), shmem, s, av);
    };
    if constexpr (EPI == EPI_STD) {
        static const bool no_pc = lab_getenv("MI355VITS_NO_B3_PC") != nullptr;
        if (wide && a.shuf_s && (a.shuf_s & 3) == 0 && !no_pc && 2 * shmem <= 160 * 1024) {
            // persistent producer / consumer form: two staging buffers, one workgroup per CU
            const long total = (long)grid.x * grid.y * grid.z;
            const int cus = device_cu_count();
            shmem *= 2;
            grid = dim3((unsigned)(total < cus ? total : cus), 1, 1);
            auto gop = [&](auto kfn) {
#ifndef MI355_EMU
                set_max_dynamic_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);
#endif
                LAUNCH_KERNEL(kfn, grid, dim3(512), shmem, s, av);
            };
            if (a.math == MATH_F16X2) gop(k_conv1d_b3_pc<MT, NT, WM, WN, 4, false, true>);
            else if (a.math == MATH_BF16W) gop(k_conv1d_b3_pc<MT, NT, WM, WN, 4, true>);
            else gop(k_conv1d_b3_pc<MT, NT, WM, WN, 4, false>);
            return;
        }
        if (wide) {
            if (a.math == MATH_F16X2) go(k_conv1d_b3<MT, NT, WM, WN, EPI, 4, false, true>);
            else if (a.math == MATH_BF16W) go(k_conv1d_b3<MT, NT, WM, WN, EPI, 4, true>);
            else go(k_conv1d_b3<MT, NT, WM, WN, EPI, 4, false>);
            return;
        }
        if (a.math == MATH_F16X2) {
            go(k_conv1d_b3<MT, NT, WM, WN, EPI, 2, false, true>);
            return;
        }
    }
    if (a.math == MATH_BF16W) go(k_conv1d_b3<MT, NT, WM, WN, EPI, 2, true>);
    else go(k_conv1d_b3<MT, NT, WM, WN, EPI, 2, false>);
}

template <int MT, int NT, int WM, int WN, int EPI>
void launch_direct(const ConvArgs& a, int n_tiles, hipStream_t s) {
    constexpr int T_B = 32 * NT * WN;
    dim3 grid((a.T + T_B - 1) / T_B, (n_tiles + MT * WM - 1) / (MT * WM), a.B);
    auto kfn = k_conv_direct_mfma<MT, NT, WM, WN, EPI>;
    ConvArgs av = a;
    av.yvec = (a.y_ld % 4 == 0) && (a.y_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.y) % 16 == 0);
    LAUNCH_KERNEL(kfn, grid, dim3(256), 0, s, av);
}

}  // namespace

bool conv1d_b3_supported(int Cin, int Cout, int K, int dil, int T_hint) {
    (void)Cout;
    // 32-channel chunks; the staged window of the widest tile (192 columns) must fit LDS; short sequences (the text
    // encoder) stay on the LDS-free f32 kernels
    return Cin >= 32 && Cin % 32 == 0 && (size_t)6 * 32 * ((192 + (K - 1) * dil + 6) & ~3) <= 150 * 1024 && T_hint > 512;
}

void launch_conv1d_mfma(const ConvArgs& a_in, hipStream_t s) {
    if (a_in.T <= 0 || a_in.B <= 0) return;
    static const int ablate = lab_getenv("MI355VITS_CONV_ABLATE") ? atoi(lab_getenv("MI355VITS_CONV_ABLATE")) : 0;
    ConvArgs a = a_in;
    a.ablate = ablate;
    if (!conv1d_mfma_supported(a.Cin, a.Cout, a.K, a.dil)) throw std::runtime_error("conv1d_mfma: unsupported shape");
    const int n_tiles = n_tiles_for(a.epi, a.Cout, a.H);
    if ((math_on_bf16(a.math) || (a.math == MATH_F16X2 && a.epi == EPI_STD)) && a.wb3 &&
        conv1d_b3_supported(a.Cin, a.Cout, a.K, a.dil, a.fixed_rule ? 1 << 30 : a.T)) {
        // split-bf16 path: one tile shape per epilogue kind (fixed by the layer, never by the batch)
        if (a.epi == EPI_GATE) launch_b3<2, 3, 2, 2, EPI_GATE>(a, n_tiles, s);
        else if (a.epi == EPI_RESSKIP) {
            if (n_tiles >= 4) launch_b3<2, 3, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
            else launch_b3<1, 3, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
        } else {
            // the largest tile whose grid still covers a good part of the chip, else the finest.  The three shapes walk
            // chunks, taps and groups in the same order: an output element gets the same bits from each, so the choice
            // may depend on the grid (batch 1 vs batch 32) without breaking "batched == unbatched"
            auto nwg = [&](int MT, int NT, int WM, int WN) {
                const long tb = 32L * NT * WN;
                return ((a.T + tb - 1) / tb) * ((n_tiles + MT * WM - 1) / (MT * WM)) * (long)a.B;
            };
            if (n_tiles >= 4 && nwg(2, 3, 2, 2) >= 96) launch_b3<2, 3, 2, 2, EPI_STD>(a, n_tiles, s);
            else if (n_tiles >= 2 && nwg(1, 3, 2, 2) >= 96) launch_b3<1, 3, 2, 2, EPI_STD>(a, n_tiles, s);
            else launch_b3<1, 2, 1, 4, EPI_STD>(a, n_tiles, s);
        }
        return;
    }
    // Tile choice.  Every CU works through ceil(blocks / 256) workgroups' worth of MFMA time, so a grid of 576
    // workgroups runs at 576 / 768 = 75 % of one of 1152: take the largest tile whose grid fills the 256 CUs to
    // >= 85 %, else the candidate that fills them best (small problems: the finest tile = most parallelism).
    auto blocks = [&](int MT, int NT, int WM, int WN) {
        const long tb = 32L * NT * WN;
        return ((a.T + tb - 1) / tb) * ((n_tiles + MT * WM - 1) / (MT * WM)) * (long)a.B;
    };
    auto fill = [&](long nb) { return (double)nb / (256.0 * (double)((nb + 255) / 256)); };
    struct Cand { int MT, NT, WM, WN; };
    static const Cand forced = [] {
        Cand f{0, 0, 0, 0};
        const char* e = lab_getenv("MI355VITS_CONV_CFG");
        if (e) sscanf(e, "%d,%d,%d,%d", &f.MT, &f.NT, &f.WM, &f.WN);
        return f;
    }();
    auto choose = [&](const Cand* c, int n) {
        if (forced.MT)
            for (int i = 0; i < n; ++i)
                if (c[i].MT == forced.MT && c[i].NT == forced.NT && c[i].WM == forced.WM && c[i].WN == forced.WN) return i;
        int best = n - 1;
        double bf = -1.0;
        for (int i = 0; i < n; ++i) {  // candidates ordered from the largest tile to the smallest
            const double f = fill(blocks(c[i].MT, c[i].NT, c[i].WM, c[i].WN));
            if (f >= 0.85) return i;
            if (f > bf + 1e-9) { bf = f; best = i; }
        }
        return best;
    };
    // LDS-free streaming kernel: pointwise convs, and short sequences (encoder FFN) where the grid is too small to
    // hide the stage/barrier cycle of the staged kernel
    static const int poly_direct_cin = lab_getenv("MI355VITS_POLY_DIRECT_CIN") ? atoi(lab_getenv("MI355VITS_POLY_DIRECT_CIN")) : 64;
    const bool direct = a.epi != EPI_GATE && ((a.K * (a.Cin >> 1)) % 8) == 0 &&
                        (a.shuf_s ? (a.K <= 2 && a.Cin <= poly_direct_cin && (a.shuf_s & 3) == 0)
                                  : (a.K == 1 || (a.K <= 3 && a.T <= 512 && !a.fixed_rule)));
    // deep + short (encoder FFN conv_2): split the k-steps over the four waves of a workgroup.  The rule looks at the
    // layer shape only, never at the batch size, so a row's bits do not depend on what it is batched with.
    if (direct && a.epi == EPI_STD && a.T <= 512 && a.K * (a.Cin >> 1) >= 512 && ((a.K * (a.Cin >> 1)) % 32) == 0) {
        dim3 grid((a.T + 31) / 32, n_tiles, a.B);
        auto kfn = k_conv_direct_splitk<EPI_STD>;
        LAUNCH_KERNEL(kfn, grid, dim3(256), 4 * 16 * 64 * sizeof(float), s, a);
        return;
    }
    if (a.epi == EPI_GATE) {
        const Cand c[] = {{2, 2, 2, 2}, {2, 1, 2, 2}};
        if (choose(c, 2) == 0) launch_cfg<2, 2, 2, 2, EPI_GATE>(a, n_tiles, s);
        else launch_cfg<2, 1, 2, 2, EPI_GATE>(a, n_tiles, s);
        return;
    }
    if (a.epi == EPI_RESSKIP) {
        const Cand c[] = {{2, 2, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2}};
        const int k = choose(c + (n_tiles >= 4 ? 0 : (n_tiles >= 2 ? 1 : 2)), n_tiles >= 4 ? 3 : (n_tiles >= 2 ? 2 : 1)) +
                      (n_tiles >= 4 ? 0 : (n_tiles >= 2 ? 1 : 2));
        if (direct) {
            if (k == 0) launch_direct<2, 2, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
            else if (k == 1) launch_direct<1, 2, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
            else launch_direct<1, 1, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
        } else {
            if (k == 0) launch_cfg<2, 2, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
            else if (k == 1) launch_cfg<1, 2, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
            else launch_cfg<1, 1, 2, 2, EPI_RESSKIP>(a, n_tiles, s);
        }
        return;
    }
    if (n_tiles == 1) {
        const Cand c[] = {{1, 2, 1, 4}, {1, 1, 1, 4}};
        const int k = choose(c, 2);
        if (direct) {
            if (k == 0) launch_direct<1, 2, 1, 4, EPI_STD>(a, n_tiles, s);
            else launch_direct<1, 1, 1, 4, EPI_STD>(a, n_tiles, s);
        } else {
            if (k == 0) launch_cfg<1, 2, 1, 4, EPI_STD>(a, n_tiles, s);
            else launch_cfg<1, 1, 1, 4, EPI_STD>(a, n_tiles, s);
        }
        return;
    }
    const Cand c[] = {{2, 2, 2, 2}, {1, 2, 2, 2}, {1, 1, 2, 2}};
    const int first = n_tiles >= 4 ? 0 : 1;
    const int k = choose(c + first, 3 - first) + first;
    if (direct) {
        if (k == 0) launch_direct<2, 2, 2, 2, EPI_STD>(a, n_tiles, s);
        else if (k == 1) launch_direct<1, 2, 2, 2, EPI_STD>(a, n_tiles, s);
        else launch_direct<1, 1, 2, 2, EPI_STD>(a, n_tiles, s);
    } else {
        if (k == 0) launch_cfg<2, 2, 2, 2, EPI_STD>(a, n_tiles, s);
        else if (k == 1) launch_cfg<1, 2, 2, 2, EPI_STD>(a, n_tiles, s);
        else launch_cfg<1, 1, 2, 2, EPI_STD>(a, n_tiles, s);
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose1d: y[co,n] = b[co] + sum_ci sum_{k = r, r+s, ... < K} x[ci,(n+p-k)/s] * W[ci,co,k],
// r = (n + p) mod s.  With K = 2s each output sees exactly two taps (polyphase); no scatter, no atomics.
// ------------------------------------------------------------------------------------------------
constexpr int CT_N = 64, CT_CO = 32, CT_CI = 16;

__global__ __launch_bounds__(256) void k_conv_transpose1d(ConvTArgs a) {
    DYN_SMEM(float, smem);
    const int tid = threadIdx.x;
    const int nl = tid & 63, cg = tid >> 6;  // 64 output samples x 4 groups of 8 channels
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * CT_N;
    const int co0 = blockIdx.y * CT_CO;
    const int Tout = a.Tin * a.stride;
    // input positions touched by this tile: i in [i_lo, i_hi]
    const int taps = (a.K + a.stride - 1) / a.stride;
    const int i_hi = (n0 + CT_N - 1 + a.pad) / a.stride;
    const int i_lo = (n0 + a.pad) / a.stride - (taps - 1);
    const int NI = i_hi - i_lo + 1;
    float* xs = smem;                 // [CT_CI][NI]
    float* ws = smem + CT_CI * NI;    // [CT_CI][CT_CO][K]
    const int n = n0 + nl;
    const int r = (n + a.pad) % a.stride;
    const int ibase = (n + a.pad) / a.stride;  // tap m uses input ibase - m with k = r + m*stride
    const int in_len = a.in_len ? a.in_len[b] : a.Tin;
    float acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
    for (int c0 = 0; c0 < a.Cin; c0 += CT_CI) {
        for (int idx = tid; idx < CT_CI * NI; idx += 256) {
            const int ci = idx / NI, ii = idx - ci * NI;
            const int c = c0 + ci, i = i_lo + ii;
            float v = 0.0f;
            if (c < a.Cin && i >= 0 && i < a.Tin && i < in_len) {
                v = a.x[(long)b * a.x_bs + (long)c * a.x_ld + i];
                v = v >= 0.0f ? v : v * a.in_slope;
            }
            xs[idx] = v;
        }
        for (int idx = tid; idx < CT_CI * CT_CO * a.K; idx += 256) {
            const int ci = idx / (CT_CO * a.K);
            const int rem = idx - ci * (CT_CO * a.K);
            const int co = rem / a.K, k = rem - co * a.K;
            float v = 0.0f;
            if (c0 + ci < a.Cin && co0 + co < a.Cout) v = a.w[((long)(c0 + ci) * a.Cout + co0 + co) * a.K + k];
            ws[idx] = v;
        }
        __syncthreads();
        for (int ci = 0; ci < CT_CI; ++ci) {
            for (int m = 0; m < taps; ++m) {
                const int k = r + m * a.stride;
                if (k >= a.K) break;
                const float xv = xs[ci * NI + (ibase - m - i_lo)];
                const float* wp = ws + (ci * CT_CO + cg * 8) * a.K + k;
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(xv, wp[q * a.K], acc[q]);
            }
        }
        __syncthreads();
    }
    if (n < Tout) {
        for (int q = 0; q < 8; ++q) {
            const int co = co0 + cg * 8 + q;
            if (co < a.Cout) a.y[(long)b * a.y_bs + (long)co * a.y_ld + n] = acc[q] + (a.bias ? a.bias[co] : 0.0f);
        }
    }
}

void launch_conv_transpose1d(const ConvTArgs& a, hipStream_t s) {
    if (a.Tin <= 0 || a.B <= 0) return;
    const int Tout = a.Tin * a.stride;
    const int taps = (a.K + a.stride - 1) / a.stride;
    const int NI = CT_N / a.stride + taps + 2;  // upper bound of i_hi - i_lo + 1
    const size_t shmem = sizeof(float) * ((size_t)CT_CI * NI + (size_t)CT_CI * CT_CO * a.K);
    dim3 grid((Tout + CT_N - 1) / CT_N, (a.Cout + CT_CO - 1) / CT_CO, a.B);
    LAUNCH_KERNEL(k_conv_transpose1d, grid, dim3(256), shmem, s, a);
}

// Polyphase view of ConvTranspose1d (A.2): output n = i*s + r - p, r in [0,s) takes taps k = r + m*s (m = 0..taps-1)
// of inputs x[i - m].  As a stride-1 Conv1d over positions i with `taps` taps and left padding taps-1:
//   y'[co*s + r][i] = sum_ci sum_j W'[co*s + r][ci][j] * x[ci][i - (taps-1) + j],  W'[..][j] = W[ci][co][r + (taps-1-j)*s]
// (channel-major / phase-minor rows: the MFMA C layout then gives every lane runs of consecutive output samples)
int convt_taps(int K, int stride) { return (K + stride - 1) / stride; }
void convt_to_polyphase(const float* w, const float* bias, int Cin, int Cout, int K, int stride, float* w_out,
                        float* bias_out) {
    const int taps = convt_taps(K, stride);
    for (int r = 0; r < stride; ++r)
        for (int co = 0; co < Cout; ++co) {
            if (bias_out) bias_out[co * stride + r] = bias ? bias[co] : 0.0f;
            for (int ci = 0; ci < Cin; ++ci)
                for (int j = 0; j < taps; ++j) {
                    const int k = r + (taps - 1 - j) * stride;
                    w_out[(((size_t)co * stride + r) * Cin + ci) * taps + j] = k < K ? w[((size_t)ci * Cout + co) * K + k] : 0.0f;
                }
        }
}

// ------------------------------------------------------------------------------------------------
// conv_post (C_out = 1, no bias) + leaky-relu(0.01) on the input + tanh + per-utterance peak  (K12 + A2)
// HBM-bound: reads the [C_in, L] activation once (3.4 FLOP/B), writes L floats.
// ------------------------------------------------------------------------------------------------
constexpr int CP_T = 1024;  // samples per workgroup (4 per thread)

__global__ __launch_bounds__(256) void k_conv_post_tanh(const float* x, long x_bs, int x_ld, const float* w, int Cin,
                                                        int K, int L, const int* valid_len, float* audio,
                                                        long audio_bs, unsigned* peak_bits) {
    DYN_SMEM(float, smem);
    const int halo = K - 1, pad = (K - 1) / 2;
    const int LD = CP_T + halo;
    constexpr int CIC = 8;
    float* xs = smem;             // [CIC][LD]
    float* ws = smem + CIC * LD;  // [Cin*K]
    float* red = ws + Cin * K;    // [4]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * CP_T;
    for (int i = tid; i < Cin * K; i += 256) ws[i] = w[i];
    const int vl = valid_len ? valid_len[b] : L;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c0 = 0; c0 < Cin; c0 += CIC) {
        __syncthreads();
        for (int ci = wid; ci < CIC; ci += 4) {
            const int c = c0 + ci;
            const float* row = x + (long)b * x_bs + (long)c * x_ld;
            for (int tt = lane; tt < LD; tt += 64) {
                const int t = t0 - pad + tt;
                float v = 0.0f;
                if (c < Cin && t >= 0 && t < L && t < vl) {
                    v = row[t];
                    v = v >= 0.0f ? v : v * 0.01f;
                }
                xs[ci * LD + tt] = v;
            }
        }
        __syncthreads();
        for (int ci = 0; ci < CIC && c0 + ci < Cin; ++ci) {
            for (int k = 0; k < K; ++k) {
                const float wv = ws[(c0 + ci) * K + k];
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(wv, xs[ci * LD + tid + q * 256 + k], acc[q]);
            }
        }
    }
    float pk = 0.0f;
    for (int q = 0; q < 4; ++q) {
        const int t = t0 + tid + q * 256;
        if (t < L) {
            const float y = tanhf(acc[q]);
            audio[(long)b * audio_bs + t] = y;
            if (t < vl) pk = fmaxf(pk, fabsf(y));
        }
    }
    pk = wave_reduce_max(pk);
    if (lane == 0) red[wid] = pk;
    __syncthreads();
    if (tid == 0) {
        pk = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(peak_bits + b, __float_as_uint(pk));  // non-negative floats order like their bit patterns
    }
}

// Row-aligned variant without LDS: a lane owns 8 consecutive samples and reads x[c, t-4 .. t+11] as four 16-byte
// loads per channel (the overlap with its neighbours is served by L1), several channels in flight.  Same (channel,
// tap) summation order as the staged kernel above, so both give identical bits.
constexpr int CPV_T = 2048;  // samples per workgroup (8 per thread)

__global__ __launch_bounds__(256) void k_conv_post_tanh_vec(const float* __restrict__ x, long x_bs, int x_ld,
                                                            const float* __restrict__ w, int Cin, int K, int L,
                                                            const int* valid_len, float* __restrict__ audio, long audio_bs,
                                                            unsigned* peak_bits) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * CPV_T + 8 * tid;
    const int pad = (K - 1) / 2;  // <= 4
    int vl = valid_len ? valid_len[b] : L;
    if (vl > L) vl = L;
    float acc[8];
    MI355_UNROLL
    for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
    const float* xb = x + (long)b * x_bs;
    if (t < L) {
        const bool inner = t - 4 >= 0 && t + 11 < vl;
#pragma unroll 4
        for (int c = 0; c < Cin; ++c) {
            const float* row = xb + (long)c * x_ld;
            float xv[16];
            if (inner) {
                MI355_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(row + t - 4 + 4 * g);
                    xv[4 * g + 0] = v.x; xv[4 * g + 1] = v.y; xv[4 * g + 2] = v.z; xv[4 * g + 3] = v.w;
                }
            } else {
                MI355_UNROLL
                for (int i = 0; i < 16; ++i) {
                    const int tt = t - 4 + i;
                    xv[i] = (tt >= 0 && tt < vl) ? row[tt] : 0.0f;
                }
            }
            MI355_UNROLL
            for (int i = 0; i < 16; ++i) xv[i] = xv[i] >= 0.0f ? xv[i] : xv[i] * 0.01f;
            for (int k = 0; k < K; ++k) {
                const float wv = w[c * K + k];
                MI355_UNROLL
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(wv, xv[q + k + 4 - pad], acc[q]);
            }
        }
    }
    float pk = 0.0f;
    if (t < L) {
        float y[8];
        MI355_UNROLL
        for (int q = 0; q < 8; ++q) {
            y[q] = tanhf(acc[q]);
            if (t + q < vl) pk = fmaxf(pk, fabsf(y[q]));
        }
        float* ap = audio + (long)b * audio_bs + t;
        if (t + 7 < L) {
            *reinterpret_cast<float4*>(ap) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(ap + 4) = make_float4(y[4], y[5], y[6], y[7]);
        } else {
            for (int q = 0; q < 8; ++q)
                if (t + q < L) ap[q] = y[q];
        }
    }
    pk = wave_reduce_max(pk);
    if (lane == 0) red[wid] = pk;
    __syncthreads();
    if (tid == 0) {
        pk = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(peak_bits + b, __float_as_uint(pk));
    }
}

// Round 5: the same conv with (nearly) every byte of x requested ONCE.  k_conv_post_tanh_vec gives a lane eight consecutive samples and
// has it read x[t - 4 .. t + 11] as four 16-byte loads per channel: lanes 32 bytes apart, so every wave-load touches 32 64-byte
// segments at half use and a channel costs 128 segment accesses for 33 segments of data — and its lane-dependent `inner` test sits
// between the loads and their use, so hipcc drains the memory counter (vmcnt(0)) once per channel: 3.6 TB/s.  Here a wave owns NT
// tiles; per channel and tile a lane loads ONE float4 (a wave-load = 1 KiB contiguous) and gets the three samples either side from
// its neighbours through DPP wave shifts (v_mov_b32 wave_shr:1 / wave_shl:1 — one VALU each, no LDS).  Lanes 0 and 63 only SUPPLY
// halo: a tile produces the 248 samples of lanes 1 .. 62 and consecutive tiles overlap by two float4 (3 % of the loads instead of a
// second, sparse halo load per tile and channel).  All loads are unconditional buffer loads (row offset in an SGPR; "before the
// row" is a negative = out-of-range offset and reads as zero), CU channels are in flight per lane, and waves whose span lies inside
// the row take a path without masks.  Same (channel, tap) order of the same fmaf chain: bit-identical to both kernels above.
__device__ __forceinline__ float wave_left(float v) {   // lane l: v of lane l - 1 (lane 0: unspecified)
#ifdef MI355_EMU
    return __shfl_up(v, 1);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
#endif
}
__device__ __forceinline__ float wave_right(float v) {  // lane l: v of lane l + 1 (lane 63: unspecified)
#ifdef MI355_EMU
    return __shfl_down(v, 1);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
#endif
}

constexpr int CPD_NT = 2;                      // tiles per wave
constexpr int CPD_TW = 248;                    // samples a tile produces (lanes 1 .. 62)
constexpr int CPD_T = 4 * CPD_TW * CPD_NT;     // samples per workgroup (four waves)
constexpr int CPD_CU = 4;                      // channels in flight per lane

template <bool MASKED>
__device__ __forceinline__ void conv_post_span(const BufRsrc xb, unsigned xrow, const float* __restrict__ w, int Cin, int ws, int vl,
                                                float (&acc)[CPD_NT][4]) {
    constexpr int K = 7, NT = CPD_NT, CU = CPD_CU;
    const int lane = threadIdx.x & 63;
    unsigned vo[NT];
    bool mo[NT][4];
    MI355_UNROLL
    for (int i = 0; i < NT; ++i) {
        const int t = ws + CPD_TW * i - 4 + 4 * lane;  // this lane's four samples of tile i
        // (t < 0: past the buffer's 2 GiB range = zeros; MASKED: a lane whose four samples all lie past the row's valid length is
        // switched off the same way — its values are masked below anyway, and its 16-byte load would otherwise reach up to ~2 KB
        // past the row, i.e. past the tensor for the last channel of the last row: ADVICE r5)
        vo[i] = (MASKED && t >= vl) ? BUF_OOB : 4u * (unsigned)t;
        MI355_UNROLL
        for (int e = 0; e < 4; ++e) mo[i][e] = t + e < vl;
    }
    for (int c0 = 0; c0 < Cin; c0 += CU) {
        uint4 xo[CU][NT];
        MI355_UNROLL
        for (int u = 0; u < CU; ++u)
            MI355_UNROLL
            for (int i = 0; i < NT; ++i) xo[u][i] = buf_load_u4(xb, vo[i], (unsigned)(c0 + u) * xrow);
        MI355_UNROLL
        for (int u = 0; u < CU; ++u) {
            float wv[K];
            MI355_UNROLL
            for (int k = 0; k < K; ++k) wv[k] = w[(c0 + u) * K + k];
            MI355_UNROLL
            for (int i = 0; i < NT; ++i) {
                float o[4] = {__uint_as_float(xo[u][i].x), __uint_as_float(xo[u][i].y), __uint_as_float(xo[u][i].z), __uint_as_float(xo[u][i].w)};
                MI355_UNROLL
                for (int e = 0; e < 4; ++e) {
                    if (MASKED) o[e] = mo[i][e] ? o[e] : 0.0f;
                    o[e] = fmaxf(o[e], o[e] * 0.01f);  // leaky-relu(0.01): the bits of `o >= 0 ? o : o * 0.01f`
                }
                // xv[j] = lrelu(x[t - 3 + j]), j = 0 .. 9
                float xv[10];
                xv[0] = wave_left(o[1]);
                xv[1] = wave_left(o[2]);
                xv[2] = wave_left(o[3]);
                xv[3] = o[0]; xv[4] = o[1]; xv[5] = o[2]; xv[6] = o[3];
                xv[7] = wave_right(o[0]);
                xv[8] = wave_right(o[1]);
                xv[9] = wave_right(o[2]);
                MI355_UNROLL
                for (int k = 0; k < K; ++k)
                    MI355_UNROLL
                    for (int q = 0; q < 4; ++q) acc[i][q] = fmaf(wv[k], xv[q + k], acc[i][q]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_conv_post_tanh_dpp(const float* __restrict__ x, long x_bs, int x_ld, const float* __restrict__ w, int Cin,
                                                            int L, const int* valid_len, float* __restrict__ audio, long audio_bs,
                                                            unsigned* peak_bits) {
    __shared__ float red[4];
    constexpr int NT = CPD_NT;
    const int tid = threadIdx.x, lane = tid & 63, wid = WAVE_UNIFORM(tid >> 6);
    const int b = blockIdx.y;
    const int ws = blockIdx.x * CPD_T + wid * CPD_TW * NT;  // the first sample this wave produces
    int vl = valid_len ? valid_len[b] : L;
    if (vl > L) vl = L;
    vl = WAVE_UNIFORM(vl);
    float acc[NT][4];
    MI355_UNROLL
    for (int i = 0; i < NT; ++i)
        MI355_UNROLL
        for (int q = 0; q < 4; ++q) acc[i][q] = 0.0f;
    const BufRsrc xb = buf_rsrc(x + (long)b * x_bs);
    const unsigned xrow = 4u * (unsigned)x_ld;
    if (ws - 3 < vl) {  // (a span whose every tap lies past the row's end: zeros)
        if (ws + CPD_TW * NT + 4 <= vl) conv_post_span<false>(xb, xrow, w, Cin, ws, vl, acc);
        else conv_post_span<true>(xb, xrow, w, Cin, ws, vl, acc);
    }
    float pk = 0.0f;
    MI355_UNROLL
    for (int i = 0; i < NT; ++i) {
        const int t = ws + CPD_TW * i - 4 + 4 * lane;
        if (lane >= 1 && lane <= 62 && t < L) {
            float y[4];
            MI355_UNROLL
            for (int q = 0; q < 4; ++q) {
                y[q] = tanhf(acc[i][q]);
                if (t + q < vl) pk = fmaxf(pk, fabsf(y[q]));
            }
            float* ap = audio + (long)b * audio_bs + t;
            if (t + 3 < L) {
                *reinterpret_cast<float4*>(ap) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
                for (int q = 0; q < 4; ++q)
                    if (t + q < L) ap[q] = y[q];
            }
        }
    }
    pk = wave_reduce_max(pk);
    if (lane == 0) red[wid] = pk;
    __syncthreads();
    if (tid == 0) {
        pk = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(peak_bits + b, __float_as_uint(pk));
    }
}

void launch_conv_post_tanh(const float* x, long x_bs, int x_ld, const float* w, int Cin, int K, int B, int L,
                           const int* valid_len, float* audio, long audio_bs, unsigned* peak_bits, hipStream_t s) {
    if (L <= 0 || B <= 0) return;
    static const bool no_vec = lab_getenv("MI355VITS_CONV_POST_STAGED") != nullptr;
    const bool aligned = (x_ld % 4 == 0) && (x_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                         (audio_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(audio) % 16 == 0);
    const bool no_dpp = lab_getenv("MI355VITS_CONV_POST_V1") != nullptr;  // lab / tests: the round-1 kernels
    if (aligned && K == 7 && Cin % CPD_CU == 0 && (long)Cin * x_ld * 4 < 0x7fffffffL && !no_vec && !no_dpp) {
        dim3 grid((L + CPD_T - 1) / CPD_T, B);
        LAUNCH_KERNEL(k_conv_post_tanh_dpp, grid, dim3(256), 0, s, x, x_bs, x_ld, w, Cin, L, valid_len, audio, audio_bs, peak_bits);
        return;
    }
    if (aligned && K <= 9 && (K & 1) && !no_vec) {
        dim3 grid((L + CPV_T - 1) / CPV_T, B);
        LAUNCH_KERNEL(k_conv_post_tanh_vec, grid, dim3(256), 0, s, x, x_bs, x_ld, w, Cin, K, L, valid_len, audio, audio_bs,
                      peak_bits);
        return;
    }
    const size_t shmem = sizeof(float) * ((size_t)8 * (CP_T + K - 1) + (size_t)Cin * K + 4);
    dim3 grid((L + CP_T - 1) / CP_T, B);
    LAUNCH_KERNEL(k_conv_post_tanh, grid, dim3(256), shmem, s, x, x_bs, x_ld, w, Cin, K, L, valid_len, audio, audio_bs,
                  peak_bits);
}

// audio_float_to_int16 (mimic3_tts/utils.py:237-244) per utterance: scale = 32767 / max(0.01, peak),
// clip to +-32767, truncate toward zero.  Rows are zero beyond their valid length.
__global__ __launch_bounds__(256) void k_pcm16(const float* audio, long audio_bs, const unsigned* peak_bits,
                                               const int* valid_len, int L, int16_t* pcm, long pcm_bs, double volume) {
    const int b = blockIdx.y;
    const float peak = fmaxf(0.01f, __uint_as_float(peak_bits[b]));
    const float scale = 32767.0f / peak;
    const int vl = valid_len ? valid_len[b] : L;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < L; t += gridDim.x * 256) {
        float v = 0.0f;
        if (t < vl) {
            v = audio[(long)b * audio_bs + t] * scale;
            v = fminf(fmaxf(v, -32767.0f), 32767.0f);
        }
        int q = (int)v;  // truncation toward zero: numpy's astype("int16") on an in-range float
        if (volume != 1.0) {
            // audioop.mul (CPython Modules/audioop.c, fbound): double product, clip to the int16 range, floor
            double d = (double)q * volume;
            if (d > 32767.0) d = 32767.0;
            else if (d < -32768.0 + 1.0) d = -32768.0;
            q = (int)floor(d);
        }
        pcm[(long)b * pcm_bs + t] = (int16_t)q;
    }
}

void launch_pcm16(const float* audio, long audio_bs, const unsigned* peak_bits, const int* valid_len, int B, int L,
                  int16_t* pcm, long pcm_bs, hipStream_t s, double volume) {
    if (L <= 0 || B <= 0) return;
    int gx = (L + 255) / 256;
    if (gx > 2048) gx = 2048;
    LAUNCH_KERNEL(k_pcm16, dim3(gx, B), dim3(256), 0, s, audio, audio_bs, peak_bits, valid_len, L, pcm, pcm_bs, volume);
}

// ------------------------------------------------------------------------------------------------
// MFMA fragment-layout self test (asymmetric operands; cdna guide §3 "always A=I-check with asymmetric B")
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_mfma_selftest(float* out) {
    const int lane = threadIdx.x & 63;
    // 32x32x2: C[i][j] = sum_k A[i][k] * B[k][j], A[i][k] = i + 100k + 1, B[k][j] = 3j - 7k + 2
    {
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
        const float av = (float)((lane & 31) + 100 * (lane >> 5) + 1);
        const float bv = (float)(3 * (lane & 31) - 7 * (lane >> 5) + 2);
        c = MFMA_32x32x2_F32(av, bv, c);
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[row * 32 + (lane & 31)] = c[r];
        }
    }
    // 16x16x4
    {
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = 0.0f;
        const float av = (float)((lane & 15) + 100 * (lane >> 4) + 1);
        const float bv = (float)(3 * (lane & 15) - 7 * (lane >> 4) + 2);
        c = MFMA_16x16x4_F32(av, bv, c);
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            out[1024 + row * 16 + (lane & 15)] = c[r];
        }
    }
    // 16x16x32 bf16 (k_mrf_p): A[i][k] = i + 2k, B[k][j] = 3j - k + 2 (small integers: exact in bf16); lane = (quarter q, row /
    // column), k-slots 8q .. 8q + 7, slot e in bits [16 (e & 1) ...] of register e >> 1
    {
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = 0.0f;
        const int q = lane >> 4, rc = lane & 15;
        unsigned aw[4], bw[4];
        for (int e2 = 0; e2 < 4; ++e2) {
            const int k0 = 8 * q + 2 * e2, k1 = k0 + 1;
            aw[e2] = (__float_as_uint((float)(rc + 2 * k0)) >> 16) | (__float_as_uint((float)(rc + 2 * k1)) & 0xffff0000u);
            bw[e2] = (__float_as_uint((float)(3 * rc - k0 + 2)) >> 16) | (__float_as_uint((float)(3 * rc - k1 + 2)) & 0xffff0000u);
        }
        uint4 a4, b4;
        a4.x = aw[0]; a4.y = aw[1]; a4.z = aw[2]; a4.w = aw[3];
        b4.x = bw[0]; b4.y = bw[1]; b4.z = bw[2]; b4.w = bw[3];
        c = MFMA_16x16x32_BF16(a4, b4, c);
        for (int r = 0; r < 4; ++r) out[1280 + (4 * q + r) * 16 + rc] = c[r];
    }
}
void launch_mfma_selftest(float* out, hipStream_t s) { LAUNCH_KERNEL(k_mfma_selftest, dim3(1), dim3(64), 0, s, out); }

__global__ __launch_bounds__(256) void k_fill(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// debug taps: zero p[b, c, t] for t >= len[b] * factor (rows of a ragged batch are not computed past their length: the taps the
// tests compare must not carry whatever the workspace held there)
__global__ __launch_bounds__(256) void k_zero_row_tails(float* p, int C, long T, const int* len, int factor) {
    const int b = blockIdx.y;
    const long t0 = (long)len[b] * factor;
    if (t0 >= T) return;
    const long w = T - t0, n = (long)C * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long c = i / w, t = t0 + (i - c * w);
        p[((long)b * C + c) * T + t] = 0.0f;
    }
}
void launch_zero_row_tails(float* p, int B, int C, long T, const int* len, int factor, hipStream_t s) {
    if (B <= 0 || C <= 0 || T <= 0) return;
    LAUNCH_KERNEL(k_zero_row_tails, dim3(64, (unsigned)B), dim3(256), 0, s, p, C, T, len, factor);
}

void launch_fill(float* p, float v, size_t n, hipStream_t s) {
    if (!n) return;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    LAUNCH_KERNEL(k_fill, dim3((unsigned)g), dim3(256), 0, s, p, v, n);
}

}  // namespace m355
