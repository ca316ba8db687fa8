// kernels_attn.cpp — relative-position multi-head attention on the fp32 matrix cores.
//
//   s[i,j] = (q_i/sqrt(d)) . k_j + [|j-i|<=W] (q_i/sqrt(d)) . E_k[j-i+W];  masked -> -1e4;  p = softmax_j s
//   o_i    = sum_j p[i,j] v_j + sum_{|j-i|<=W} p[i,j] E_v[j-i+W]
//
// One wave owns 32 query rows of one (batch, head) and computes the TRANSPOSED score tile S^T[j][i] = K Q^T with
// v_mfma_f32_32x32x2_f32: in the C/D layout a lane then holds one query column i = lane&31 and 16 key rows per key
// tile, so the softmax statistics are in-lane reductions plus one cross-half shuffle, and P^T is directly the B
// operand of the second product O^T[c][i] = V P^T — the probabilities never leave registers (the MFMA k index is
// permuted to "the rows this lane half already holds": k-step (g,m) uses key row 8g+m for lanes 0-31 and 8g+4+m for
// lanes 32-63; V is read with the same permutation).  Both operands of K Q^T are read straight from the [C,T]
// activation layout: 32 consecutive time samples per half-wave (128-byte segments).  The window terms are two tiny
// extra products: (q E_k^T) as a 32-row padded MFMA whose result goes through a 1 KiB LDS table, and E_v^T p_rel with
// the near-diagonal probabilities gathered in the same table.
#include "kernels.h"

namespace m355 {

template <int NKT>  // key tiles of 32 held in registers: T <= 32 * NKT
__global__ __launch_bounds__(64) void k_rel_attention_mfma(const float* __restrict__ qkv, const float* __restrict__ ek,
                                                           const float* __restrict__ ev, const int* __restrict__ len,
                                                           int T, int H, int nh, int W, float* __restrict__ out) {
    DYN_SMEM(float, tab);  // [32][32]: rows 0..2W hold the window table (first rel-k logits, later rel probabilities)
    const int lane = threadIdx.x & 63;
    const int brow = lane >> 5, bcol = lane & 31;
    const int d = H / nh, nrel = 2 * W + 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int i0 = blockIdx.x * 32;
    const int i = i0 + bcol;  // this lane's query
    const int L = len[b];
    const float scale = 1.0f / sqrtf((float)d);
    const float* qb = qkv + ((long)b * 3 * H + h * d) * T;
    const float* kb = qb + (long)H * T;
    const float* vb = qb + (long)2 * H * T;
    const bool iq = i < T;

    // ---- S^T = K Q^T (+ rel-k logits as a 32-row padded product with E_k)
    f32x16 st[NKT];
    f32x16 rl;
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) rl[r] = 0.0f;
    MI355_UNROLL
    for (int t = 0; t < NKT; ++t)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) st[t][r] = 0.0f;
    // U channel pairs per trip: all their loads are issued before the first MFMA needs one (the loop is latency-bound)
    constexpr int U = NKT <= 4 ? 4 : (NKT <= 8 ? 2 : 1);
    for (int cp0 = 0; cp0 < d / 2; cp0 += U) {
        float qv[U], ekv[U], kv[U][NKT];
        MI355_UNROLL
        for (int u = 0; u < U; ++u) {
            const int c = 2 * (cp0 + u) + brow;
            const bool cin = c < d;  // a partial last trip multiplies zeros
            qv[u] = (iq && cin) ? qb[(long)c * T + i] * scale : 0.0f;          // B[k=c][col=i]
            ekv[u] = (bcol < nrel && cin) ? ek[bcol * d + c] : 0.0f;           // A[row=r][k=c]
            MI355_UNROLL
            for (int t = 0; t < NKT; ++t) {
                const int j = t * 32 + bcol;
                kv[u][t] = (j < T && cin) ? kb[(long)c * T + j] : 0.0f;        // A[row=j][k=c]
            }
        }
        MI355_UNROLL
        for (int u = 0; u < U; ++u) {
            rl = MFMA_32x32x2_F32(ekv[u], qv[u], rl);
            MI355_UNROLL
            for (int t = 0; t < NKT; ++t) st[t] = MFMA_32x32x2_F32(kv[u][t], qv[u], st[t]);
        }
    }
    // rel-k logits -> LDS table [r][i]
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * brow;
        tab[row * 32 + bcol] = rl[r];
    }
    __syncthreads();

    // ---- window bias, masks, softmax statistics (per query column, i.e. per lane pair)
    float mx = -3.0e38f;
    MI355_UNROLL
    for (int t = 0; t < NKT; ++t) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
            float sc = st[t][r];
            const int rel = j - i;
            if (rel >= -W && rel <= W) sc += tab[(rel + W) * 32 + bcol];
            if (j >= L || i >= L) sc = -1e4f;
            if (j >= T) sc = -3.0e38f;  // beyond the tensor: not part of the softmax at all
            st[t][r] = sc;
            mx = fmaxf(mx, sc);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    __syncthreads();  // everyone has read the logits table; it is reused for the rel probabilities
    for (int r = brow; r < 32; r += 2) tab[r * 32 + bcol] = 0.0f;
    __syncthreads();
    float sum = 0.0f;
    MI355_UNROLL
    for (int t = 0; t < NKT; ++t) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
            const float e = j < T ? expf(st[t][r] - mx) : 0.0f;
            st[t][r] = e;
            sum += e;
            const int rel = j - i;
            if (j < T && rel >= -W && rel <= W) tab[(rel + W) * 32 + bcol] = e;
        }
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    __syncthreads();

    // ---- O^T[c][i] = sum_j V[c][j] P^T[j][i] + sum_r E_v[r][c] p_rel[r][i]
    for (int c0 = 0; c0 < d; c0 += 32) {
        f32x16 o;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        const int cr = c0 + bcol;  // A row of this lane
        const bool cv = cr < d;
        const float* vr = vb + (long)(cv ? cr : 0) * T;
        MI355_UNROLL
        for (int t = 0; t < NKT; ++t) {
            MI355_UNROLL
            for (int g = 0; g < 4; ++g) {
                MI355_UNROLL
                for (int m = 0; m < 4; ++m) {
                    const int j = t * 32 + 8 * g + 4 * brow + m;  // the key row this lane half holds in register 4g+m
                    const float vv = (cv && j < T) ? vr[j] : 0.0f;
                    o = MFMA_32x32x2_F32(vv, st[t][4 * g + m], o);
                }
            }
        }
        for (int s = 0; s < (nrel + 1) / 2; ++s) {
            const int r = 2 * s + brow;
            const float evv = (cv && r < nrel) ? ev[r * d + cr] : 0.0f;
            const float pv = r < nrel ? tab[r * 32 + bcol] : 0.0f;
            o = MFMA_32x32x2_F32(evv, pv, o);
        }
        if (iq) {
            MI355_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * brow;
                if (c < d) out[((long)b * H + h * d + c) * T + i] = o[r] * inv;
            }
        }
    }
}

// Four waves per (batch, head, 32 queries): key tile t belongs to wave t % 4, so a wave holds T / 128 score tiles
// instead of T / 32 (no spills at T = 512, a quarter of the dependent MFMA chain).  Softmax statistics and the PV
// partial sums are combined across the waves through LDS in a fixed order (deterministic; key tiles beyond a row's
// length contribute exact zeros, so the result does not depend on how far the batch pads it).
// DP > 0 (= d / 2, the head's channel pairs, known at compile time): EVERY global operand of the kernel — q, E_k, this wave's
// K tiles and V tiles — is loaded up front in one batch (48 (3 + 2 NKW) registers at d = 96), so a workgroup waits for L2
// once instead of once per trip of the two product loops (24 dependent round trips, most of the 30 us a launch took).  The
// MFMA sequence per output is the same as with DP = 0: identical bits.
template <int NKW, int DP = 0>  // key tiles per wave: T <= 128 * NKW
__global__ __launch_bounds__(256) void k_rel_attention_mfma4(const float* __restrict__ qkv, const float* __restrict__ ek,
                                                             const float* __restrict__ ev, const int* __restrict__ len,
                                                             int T, int H, int nh, int W, float* __restrict__ out) {
    DYN_SMEM(float, smem);
    float* tab = smem;                 // [32][32] window table: rel-k logits, later rel probabilities
    float* red = smem + 32 * 32;       // [2][4][32] max / sum per wave and query
    float* ored = red + 2 * 4 * 32;    // [4][16][64] partial O^T tiles
    const int tid = threadIdx.x, lane = tid & 63, w = WAVE_UNIFORM(tid >> 6);
    const int brow = lane >> 5, bcol = lane & 31;
    const int d = H / nh, nrel = 2 * W + 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int i0 = blockIdx.x * 32;
    const int i = i0 + bcol;
    const int L = len[b];
    const float scale = 1.0f / sqrtf((float)d);
    const float* qb = qkv + ((long)b * 3 * H + h * d) * T;
    const float* kb = qb + (long)H * T;
    const float* vb = qb + (long)2 * H * T;
    const bool iq = i < T;

    f32x16 st[NKW];
    f32x16 rl;
    MI355_UNROLL
    for (int r = 0; r < 16; ++r) rl[r] = 0.0f;
    MI355_UNROLL
    for (int m = 0; m < NKW; ++m)
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) st[m][r] = 0.0f;
    constexpr int NCT = DP > 0 ? (2 * DP + 31) / 32 : 1;  // 32-channel output tiles of O^T (prefetched V)
    float vpre[NCT][NKW][16];
    float evpre[NCT][16];  // wave 0: E_v[r = 2 s2 + brow][channel of this lane], s2 < 16 (window <= 15)
    if constexpr (DP > 0) {
        if (w == 0) {
            MI355_UNROLL
            for (int ct = 0; ct < NCT; ++ct)
                MI355_UNROLL
                for (int s2 = 0; s2 < 16; ++s2) {
                    const int r = 2 * s2 + brow, cr = ct * 32 + bcol;
                    evpre[ct][s2] = ev[(r < nrel ? r : nrel - 1) * d + (cr < d ? cr : 0)];
                }
        }
        float qv[DP], ekv[DP], kv[DP][NKW];
        MI355_UNROLL
        for (int u = 0; u < DP; ++u) {
            const int c = 2 * u + brow;
            const float qraw = qb[(long)c * T + (iq ? i : T - 1)];
            const float eraw = ek[(bcol < nrel ? bcol : nrel - 1) * d + c];
            qv[u] = iq ? qraw * scale : 0.0f;
            ekv[u] = (w == 0 && bcol < nrel) ? eraw : 0.0f;
            MI355_UNROLL
            for (int m = 0; m < NKW; ++m) {
                const int j = (w + 4 * m) * 32 + bcol;
                const float kraw = kb[(long)c * T + (j < T ? j : T - 1)];
                kv[u][m] = j < T ? kraw : 0.0f;
            }
        }
        MI355_UNROLL
        for (int ct = 0; ct < NCT; ++ct) {
            const int cr = ct * 32 + bcol;
            const float* vr = vb + (long)(cr < d ? cr : 0) * T;
            MI355_UNROLL
            for (int m = 0; m < NKW; ++m)
                MI355_UNROLL
                for (int g = 0; g < 4; ++g)
                    MI355_UNROLL
                    for (int q = 0; q < 4; ++q) {
                        const int j = (w + 4 * m) * 32 + 8 * g + 4 * brow + q;
                        vpre[ct][m][4 * g + q] = vr[j < T ? j : T - 1];
                    }
        }
        SCHED_FENCE();
        MI355_UNROLL
        for (int u = 0; u < DP; ++u) {
            if (w == 0) rl = MFMA_32x32x2_F32(ekv[u], qv[u], rl);
            MI355_UNROLL
            for (int m = 0; m < NKW; ++m) st[m] = MFMA_32x32x2_F32(kv[u][m], qv[u], st[m]);
        }
    } else {
        constexpr int U = 4;
        for (int cp0 = 0; cp0 < d / 2; cp0 += U) {
            float qv[U], ekv[U], kv[U][NKW];
            // loads through clamped indices, the out-of-range test applied to the VALUE: a test around each load makes hipcc
            // wait for every load in turn (a chain of ~100 dependent L2 round trips per workgroup: 50 us per launch)
            MI355_UNROLL
            for (int u = 0; u < U; ++u) {
                const int c = 2 * (cp0 + u) + brow;
                const bool cin = c < d;
                const int cc = cin ? c : d - 1;
                const float qraw = qb[(long)cc * T + (iq ? i : T - 1)];
                const float eraw = ek[(bcol < nrel ? bcol : nrel - 1) * d + cc];
                qv[u] = (iq && cin) ? qraw * scale : 0.0f;
                ekv[u] = (w == 0 && bcol < nrel && cin) ? eraw : 0.0f;
                MI355_UNROLL
                for (int m = 0; m < NKW; ++m) {
                    const int j = (w + 4 * m) * 32 + bcol;
                    const float kraw = kb[(long)cc * T + (j < T ? j : T - 1)];
                    kv[u][m] = (j < T && cin) ? kraw : 0.0f;
                }
            }
            MI355_UNROLL
            for (int u = 0; u < U; ++u) {
                if (w == 0) rl = MFMA_32x32x2_F32(ekv[u], qv[u], rl);
                MI355_UNROLL
                for (int m = 0; m < NKW; ++m) st[m] = MFMA_32x32x2_F32(kv[u][m], qv[u], st[m]);
            }
        }
    }
    if (w == 0) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) tab[((r & 3) + 8 * (r >> 2) + 4 * brow) * 32 + bcol] = rl[r];
    }
    __syncthreads();

    float mx = -3.0e38f;
    MI355_UNROLL
    for (int m = 0; m < NKW; ++m) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int j = (w + 4 * m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
            float sc = st[m][r];
            const int rel = j - i;
            if (rel >= -W && rel <= W) sc += tab[(rel + W) * 32 + bcol];
            if (j >= L || i >= L) sc = -1e4f;
            if (j >= T) sc = -3.0e38f;
            st[m][r] = sc;
            mx = fmaxf(mx, sc);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (brow == 0) red[w * 32 + bcol] = mx;
    __syncthreads();  // maxima visible; everyone has read the logits table
    mx = fmaxf(fmaxf(red[bcol], red[32 + bcol]), fmaxf(red[64 + bcol], red[96 + bcol]));
    for (int r = tid >> 5; r < 32; r += 8) tab[r * 32 + bcol] = 0.0f;
    __syncthreads();
    float sum = 0.0f;
    MI355_UNROLL
    for (int m = 0; m < NKW; ++m) {
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int j = (w + 4 * m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * brow;
            const float e = j < T ? expf(st[m][r] - mx) : 0.0f;
            st[m][r] = e;
            sum += e;
            const int rel = j - i;
            if (j < T && rel >= -W && rel <= W) tab[(rel + W) * 32 + bcol] = e;
        }
    }
    sum += __shfl_xor(sum, 32);
    if (brow == 0) red[128 + w * 32 + bcol] = sum;
    __syncthreads();
    const float inv = 1.0f / (((red[128 + bcol] + red[160 + bcol]) + red[192 + bcol]) + red[224 + bcol]);

    MI355_UNROLL  // (three straight-line tiles when the head size is a compile-time constant)
    for (int c0 = 0; c0 < (DP > 0 ? 2 * DP : d); c0 += 32) {
        f32x16 o;
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        const int cr = c0 + bcol;
        const bool cv = cr < d;
        const float* vr = vb + (long)(cv ? cr : 0) * T;
        MI355_UNROLL
        for (int m = 0; m < NKW; ++m) {
            MI355_UNROLL
            for (int g = 0; g < 4; ++g) {
                float vraw[4];
                MI355_UNROLL
                for (int q = 0; q < 4; ++q) {
                    const int j = (w + 4 * m) * 32 + 8 * g + 4 * brow + q;
                    if constexpr (DP > 0) vraw[q] = vpre[c0 / 32][m][4 * g + q];
                    else vraw[q] = vr[j < T ? j : T - 1];
                }
                MI355_UNROLL
                for (int q = 0; q < 4; ++q) {
                    const int j = (w + 4 * m) * 32 + 8 * g + 4 * brow + q;
                    const float vv = (cv && j < T) ? vraw[q] : 0.0f;
                    o = MFMA_32x32x2_F32(vv, st[m][4 * g + q], o);
                }
            }
        }
        if (w == 0) {
            if constexpr (DP > 0) {
                MI355_UNROLL
                for (int s2 = 0; s2 < 16; ++s2) {
                    if (s2 < (nrel + 1) / 2) {
                        const int r = 2 * s2 + brow;
                        const float evv = (cv && r < nrel) ? evpre[c0 / 32][s2] : 0.0f;
                        const float pv = r < nrel ? tab[r * 32 + bcol] : 0.0f;
                        o = MFMA_32x32x2_F32(evv, pv, o);
                    }
                }
            } else {
                for (int s2 = 0; s2 < (nrel + 1) / 2; ++s2) {
                    const int r = 2 * s2 + brow;
                    const float eraw = ev[(r < nrel ? r : nrel - 1) * d + (cv ? cr : 0)];
                    const float evv = (cv && r < nrel) ? eraw : 0.0f;
                    const float pv = r < nrel ? tab[r * 32 + bcol] : 0.0f;
                    o = MFMA_32x32x2_F32(evv, pv, o);
                }
            }
        }
        if (c0 > 0) __syncthreads();  // the previous tile's partials have been consumed
        MI355_UNROLL
        for (int r = 0; r < 16; ++r) ored[(w * 16 + r) * 64 + lane] = o[r];
        __syncthreads();
        if (iq) {
            MI355_UNROLL
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * w + q;
                const float v = ((ored[(0 * 16 + r) * 64 + lane] + ored[(1 * 16 + r) * 64 + lane]) + ored[(2 * 16 + r) * 64 + lane]) +
                                ored[(3 * 16 + r) * 64 + lane];
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * brow;
                if (c < d) out[((long)b * H + h * d + c) * T + i] = v * inv;
            }
        }
    }
}

bool rel_attention_mfma_supported(int T, int H, int n_heads, int window) {
    const int d = H / n_heads;
    return T >= 1 && T <= 512 && (d % 2) == 0 && window >= 0 && 2 * window + 1 <= 32;
}

void launch_rel_attention_mfma(const float* qkv, const float* emb_rel_k, const float* emb_rel_v, const int* len, int B,
                               int T, int H, int n_heads, int window, float* out, hipStream_t s) {
    if (!rel_attention_mfma_supported(T, H, n_heads, window)) throw std::runtime_error("rel_attention_mfma: unsupported shape");
    dim3 grid((T + 31) / 32, n_heads, B);
    static const bool one_wave = lab_getenv("MI355VITS_ATTN_ONE_WAVE") != nullptr;  // the older single-wave kernel
    if (!one_wave) {
        const size_t sh4 = (32 * 32 + 2 * 4 * 32 + 4 * 16 * 64) * sizeof(float);
        static const bool no_pre = lab_getenv("MI355VITS_ATTN_NO_PREFETCH") != nullptr;
        // the "_low" / default voices' head: all operands prefetched up front (one L2 wait per workgroup: latency, 256 registers, two
        // workgroups per CU) — on grids of >= 4 workgroups per CU the 76-register form runs six workgroups per CU instead and hides its
        // round trips behind them: 1.01 -> 0.55 ms per step at batch 256, 0.152 -> 0.181 at batch 32 (profiles/r06_attention_ab.txt).
        // The MFMA sequence per output is the same: identical bits, so the choice may follow the grid.
        const bool big = (long)grid.x * grid.y * grid.z >= 4L * current_device_cu_count();
        const bool pre = H / n_heads == 96 && !no_pre && (!big || lab_getenv("MI355VITS_ATTN_PREFETCH"));
        if (T <= 128 && pre) {
            auto k = k_rel_attention_mfma4<1, 48>;
            LAUNCH_KERNEL(k, grid, dim3(256), sh4, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
        } else if (T <= 256 && pre) {
            auto k = k_rel_attention_mfma4<2, 48>;
            LAUNCH_KERNEL(k, grid, dim3(256), sh4, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
        } else if (T <= 128) {
            auto k = k_rel_attention_mfma4<1>;
            LAUNCH_KERNEL(k, grid, dim3(256), sh4, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
        } else if (T <= 256) {
            auto k = k_rel_attention_mfma4<2>;
            LAUNCH_KERNEL(k, grid, dim3(256), sh4, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
        } else {
            auto k = k_rel_attention_mfma4<4>;
            LAUNCH_KERNEL(k, grid, dim3(256), sh4, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
        }
        return;
    }
    const size_t shmem = 32 * 32 * sizeof(float);
    if (T <= 128) {
        auto k = k_rel_attention_mfma<4>;
        LAUNCH_KERNEL(k, grid, dim3(64), shmem, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
    } else if (T <= 256) {
        auto k = k_rel_attention_mfma<8>;
        LAUNCH_KERNEL(k, grid, dim3(64), shmem, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
    } else {
        auto k = k_rel_attention_mfma<16>;
        LAUNCH_KERNEL(k, grid, dim3(64), shmem, s, qkv, emb_rel_k, emb_rel_v, len, T, H, n_heads, window, out);
    }
}

}  // namespace m355
