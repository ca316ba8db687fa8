// hipx.h — the one include every source in csrc/ uses for the HIP runtime and device builtins.
//
// Product build: hipcc --offload-arch=gfx950 (real HIP, wave64, MFMA builtins).
// Test build (-DMI355_EMU, g++): tests/emu/hip_emu.h, a CPU model of workgroups / waves /
// MFMA fragment layouts used by the `-m "not gpu"` suite to check kernel logic.  That build is
// a separate library under tests/emu/ and is never opened by the product path.
#pragma once

#ifdef MI355_EMU
#include "hip_emu.h"
typedef hipemu_f32x16 f32x16;
typedef hipemu_f32x4 f32x4;
#define MFMA_32x32x2_F32(a, b, c) hipemu_mfma_32x32x2((a), (b), (c))
#define MFMA_16x16x4_F32(a, b, c) hipemu_mfma_16x16x4((a), (b), (c))
#define MFMA_32x32x16_BF16(a, b, c) hipemu_mfma_32x32x16_bf16((a), (b), (c))
#define MFMA_32x32x16_F16(a, b, c) hipemu_mfma_32x32x16_f16((a), (b), (c))
#define MFMA_16x16x32_BF16(a, b, c) hipemu_mfma_16x16x32_bf16((a), (b), (c))
#define CVT_PK_BF16_F32(lo, hi) hipemu_cvt_pk_bf16_f32((lo), (hi))
#define LAUNCH_KERNEL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((kernel), (grid), (block), (size_t)(shmem), __VA_ARGS__)
#define DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(hipemu::tl_worker->dyn_smem)
#define MI355_UNROLL
#define MI355_NOUNROLL
#define WAVE_UNIFORM(x) (x)
#define FAST_EXPF(x) expf(x)
#define FAST_RCPF(x) (1.0f / (x))
#define SCHED_FENCE() ((void)0)
#define MIN_WAVES_PER_SIMD(n)
#define SCHED_GROUP(mask, n) ((void)0)
#define OPAQUE_V(x) ((void)0)
#define OPAQUE_S(x) ((void)0)
// buffer addressing (see the product side below): base pointer + per-lane byte offset + wave-uniform byte offset
struct BufRsrc { char* p; };
static inline BufRsrc buf_rsrc(const void* base) { return BufRsrc{const_cast<char*>(static_cast<const char*>(base))}; }
static inline float buf_load_f32(const BufRsrc& r, unsigned voff, unsigned soff) {
    return voff < 0x80000000u ? *reinterpret_cast<const float*>(r.p + voff + soff) : 0.0f;
}
constexpr unsigned BUF_OOB = 0x80000000u;  // a lane offset past the buffer's range: loads return 0, stores are dropped
static inline void buf_store_f32(const BufRsrc& r, unsigned voff, unsigned soff, float v) {
    if (voff < BUF_OOB) *reinterpret_cast<float*>(r.p + voff + soff) = v;
}
static inline void buf_store_f4(const BufRsrc& r, unsigned voff, unsigned soff, float a, float b, float c, float d) {
    if (voff < BUF_OOB) { float* p = reinterpret_cast<float*>(r.p + voff + soff); p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
}
static inline void buf_store_f2(const BufRsrc& r, unsigned voff, unsigned soff, float a, float b) {
    if (voff < BUF_OOB) { float* p = reinterpret_cast<float*>(r.p + voff + soff); p[0] = a; p[1] = b; }
}
static inline uint4 buf_load_u4(const BufRsrc& r, unsigned voff, unsigned soff) {  // 16 bytes (weight fragments, column groups)
    if (voff >= 0x80000000u) return uint4{0u, 0u, 0u, 0u};
    return *reinterpret_cast<const uint4*>(r.p + voff + soff);
}
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA_32x32x2_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA_16x16x4_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_32x32x16_bf16: a / b = eight bf16 per lane (four VGPRs, passed as uint4 bit patterns), f32 accumulate.
// Lane l holds row (A) / column (B) l & 31 and the eight k-slots of half l >> 5; products are exact, sums are f32.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }
#define MFMA_32x32x16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), (c), 0, 0, 0)
// v_mfma_f32_16x16x32_bf16: lane l = row (A) / column (B) l & 15, k-slots 8 (l >> 4) .. + 7; C/D col = l & 15, row = 4 (l >> 4) + reg
#define MFMA_16x16x32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), (c), 0, 0, 0)
// v_mfma_f32_32x32x16_f16: the same shape and rate with eight IEEE halves per lane
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8_t as_f16x8(const uint4& v) { return __builtin_bit_cast(f16x8_t, v); }
#define MFMA_32x32x16_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(a), as_f16x8(b), (c), 0, 0, 0)
// v_cvt_pk_bf16_f32: two f32 -> two bf16 (round to nearest even) in one register, `lo` in bits [15:0]
__device__ __forceinline__ unsigned cvt_pk_bf16_f32(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
#define CVT_PK_BF16_F32(lo, hi) cvt_pk_bf16_f32((lo), (hi))
#define LAUNCH_KERNEL(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
// All LDS scratch lives in the dynamic region, base 16-byte aligned (cdna guide, Guideline 17).
#define DYN_SMEM(type, name)                                                   \
    extern __shared__ __attribute__((aligned(16))) unsigned char _dyn_smem_raw[]; \
    type* name = reinterpret_cast<type*>(_dyn_smem_raw)
#define MI355_UNROLL _Pragma("unroll")
#define MI355_NOUNROLL _Pragma("unroll 1")
// value known to be identical in all lanes of a wave: move it to an SGPR so branches on it are scalar
#define WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// hardware transcendental paths (v_exp_f32 / v_rcp_f32, ~1 ulp): used where the result feeds a bounded
// non-linearity (WaveNet gate), not where exact rounding matters (durations, int16 scaling)
#define FAST_EXPF(x) __expf(x)
#define FAST_RCPF(x) __builtin_amdgcn_rcpf(x)  // v_rcp_f32 (1 ulp); __frcp_rn is the IEEE division sequence: ten instructions
// (WN_GATE below is the only user of the two)
// Pin the hand-written software pipeline: hipcc otherwise clusters the ring's prefetch loads into one burst right
// before their first use (prefetch distance collapses from four k-steps to one).  Nothing moves across this point.
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// ask the scheduler for `n` instructions of class `mask` at this point of the region (1 = ALU, 2 = VALU, 4 = SALU, 8 = MFMA,
// 0x10 = VMEM, 0x100 = DS read, 0x200 = DS write): hand-placed interleave of VALU work into an MFMA stream
#define SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
// make a value opaque to the optimiser at this point (vector / scalar register): inside a long persistent loop it keeps
// loop-invariant address arithmetic from being hoisted out — and held in registers across the whole body
#define OPAQUE_V(x) asm volatile("" : "+v"(x))
#define OPAQUE_S(x) asm volatile("" : "+s"(x))
// buffer_load / buffer_store_dword: a wave-uniform base (four SGPRs), a per-lane 32-bit byte offset and a wave-uniform byte
// offset in an SGPR — "row pointer + lane offset" costs no vector ALU at all (a global_load needs a 64-bit address per lane).
// Raw buffer, no stride, 2 GiB range, DATA_FORMAT_32 (gfx9 descriptor word 3 = 0x00020000).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc buf_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(BufRsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// a lane whose offset is >= the range (0x7fffffff bytes) is out of bounds: its load returns 0, its store is dropped — the
// branch-free way to switch single lanes off (the wave-uniform soffset does not take part in the range check)
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ void buf_store_f32(BufRsrc r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// 16- / 8-byte stores of consecutive floats (out-of-range lanes dropped as above)
__device__ __forceinline__ void buf_store_f4(BufRsrc r, unsigned voff, unsigned soff, float a, float b, float c, float d) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    const v4u_t v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_f2(BufRsrc r, unsigned voff, unsigned soff, float a, float b) {
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    const v2u_t v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// buffer_load_dwordx4: a 16-byte fragment per lane, the fragment's offset in an SGPR (streamed weight fragments: no address VALU)
__device__ __forceinline__ uint4 buf_load_u4(BufRsrc r, unsigned voff, unsigned soff) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// register budget: ask the compiler to keep the kernel within 512 / n registers per lane
#define MIN_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif

// lambdas that carry register arrays by reference must be inlined (an outlined one would put them in scratch memory)
#define MI355_INLINE_LAMBDA __attribute__((always_inline))
// timing experiments (skip a kernel's MFMA loops / staging / stores: wrong results on purpose) exist only in the lab
// build (-DMI355_LAB, tools/): the product library carries no such switch
#ifdef MI355_LAB
#define LAB_ABLATE(args) ((args).ablate)
#else
#define LAB_ABLATE(args) 0
#endif
// kernel-choice / A-B switches are read from the environment only in the lab build and in the CPU model the tests run
// on; the product library ignores them (its one knob, MI355VITS_MATH, is read when a handle is created)
#include <cstdlib>
#if defined(MI355_LAB) || defined(MI355_EMU)
static inline const char* lab_getenv(const char* name) { return getenv(name); }
#else
static inline const char* lab_getenv(const char*) { return nullptr; }
#endif

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#define HIP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) +   \
                                     " at " __FILE__ ":" + std::to_string(__LINE__) + " (" #expr ")"); \
        }                                                                                    \
    } while (0)

constexpr int WAVE_SIZE = 64;

__device__ __forceinline__ float wave_reduce_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// ------------------------------------------------------------------------------------------------
// Cooperative global -> LDS staging of a [rows][LD] activation tile by the NW waves of a workgroup.
//   dst[r*LD + c] = f(x[r*x_ld + ts + c])  for ts + c in [0, tend), else 0;   f = leaky-relu(slope) (1 = identity)
// Waves take rows round-robin; a wave issues the loads of up to 8 rows before the first LDS write, so a tile
// costs a handful of memory round trips even with one wave per SIMD.  With `vec` (rows 16-byte aligned, ts and
// LD multiples of 4) interior elements move as 16-byte loads / ds_write_b128.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lrelu_f(float v, float slope) { return v >= 0.0f ? v : v * slope; }
// WaveNet gate tanh(a) * sigmoid(s) with ONE reciprocal: (e^2a - 1) / ((e^2a + 1) (1 + e^-s)); two v_exp_f32, one v_rcp_f32 and
// five plain VALU per element (round 3: two IEEE reciprocals = 2 x 10 instructions — the gate phase was 19 % of a WaveNet layer,
// profiles/r04_wn_experiments.txt).  Arguments clamped so that every factor stays finite: e^30 * e^30 < f32 max.
__device__ __forceinline__ float wn_gate_f(float at, float as) {
    const float e2 = FAST_EXPF(2.0f * fminf(fmaxf(at, -15.0f), 15.0f));
    const float es = FAST_EXPF(-fminf(fmaxf(as, -30.0f), 30.0f));
    return (e2 - 1.0f) * FAST_RCPF((e2 + 1.0f) * (1.0f + es));
}

template <int NW, int QU>  // QU column groups (64 x 16 B each) per batch: RU * QU 16-byte loads in flight per lane
__device__ __forceinline__ void stage_tile(const float* __restrict__ xb, long x_ld, int rows, int LD, int ts, int tend,
                                           float slope, float* __restrict__ dst, int vec) {
    const int lane = threadIdx.x & 63, wid = WAVE_UNIFORM(threadIdx.x >> 6);
    constexpr int RU = 32 / NW;  // rows per wave per batch: 8 with 4 waves, 4 with 8 waves
    if (vec) {
        const int ld4 = LD >> 2;
        for (int cb = 0; cb < ld4; cb += 64 * QU) {
            for (int r0 = wid; r0 < rows; r0 += NW * RU) {
                float4 v[QU][RU];
                MI355_UNROLL
                for (int q = 0; q < QU; ++q) {
                    const int c4 = cb + 64 * q + lane;
                    const int t = ts + 4 * c4;
                    const bool inner = t >= 0 && t + 3 < tend;
                    MI355_UNROLL
                    for (int u = 0; u < RU; ++u) {
                        const int r = r0 + NW * u;
                        v[q][u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (r < rows && c4 < ld4) {
                            const float* row = xb + (long)r * x_ld;
                            if (inner) {
                                v[q][u] = *reinterpret_cast<const float4*>(row + t);
                            } else {
                                if (t >= 0 && t < tend) v[q][u].x = row[t];
                                if (t + 1 >= 0 && t + 1 < tend) v[q][u].y = row[t + 1];
                                if (t + 2 >= 0 && t + 2 < tend) v[q][u].z = row[t + 2];
                                if (t + 3 >= 0 && t + 3 < tend) v[q][u].w = row[t + 3];
                            }
                        }
                    }
                }
                MI355_UNROLL
                for (int q = 0; q < QU; ++q) {
                    const int c4 = cb + 64 * q + lane;
                    MI355_UNROLL
                    for (int u = 0; u < RU; ++u) {
                        const int r = r0 + NW * u;
                        if (r < rows && c4 < ld4) {
                            float4 o = v[q][u];
                            o.x = lrelu_f(o.x, slope); o.y = lrelu_f(o.y, slope); o.z = lrelu_f(o.z, slope); o.w = lrelu_f(o.w, slope);
                            *reinterpret_cast<float4*>(dst + (long)r * LD + 4 * c4) = o;
                        }
                    }
                }
            }
        }
    } else {
        for (int c = lane; c < LD; c += 64) {
            const int t = ts + c;
            const bool ok = t >= 0 && t < tend;
            for (int r0 = wid; r0 < rows; r0 += NW * RU) {
                float v[RU];
                MI355_UNROLL
                for (int u = 0; u < RU; ++u) {
                    const int r = r0 + NW * u;
                    v[u] = (ok && r < rows) ? xb[(long)r * x_ld + t] : 0.0f;
                }
                MI355_UNROLL
                for (int u = 0; u < RU; ++u) {
                    const int r = r0 + NW * u;
                    if (r < rows) dst[(long)r * LD + c] = lrelu_f(v[u], slope);
                }
            }
        }
    }
}

// ---- packed LDS tiles for 16-byte MFMA operand fetches -----------------------------------------------------------
// Element (channel c, column col) of a tile with `ld` columns sits at pk(c, col, ld): the four channels 8g + 2q + brow
// (q = 0..3) of one MFMA half are side by side, so a lane's B fragments of four consecutive k-steps are one
// ds_read_b128 at float4 index (2g + brow) * ld + col.
__device__ __forceinline__ int pk(int c, int col, int ld) { return ((((c >> 3) * 2 + (c & 1)) * ld + col) << 2) + ((c >> 1) & 3); }
__device__ __forceinline__ float f4c(const float4& v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }

// global [rows][x_ld] -> packed LDS tile: a thread takes one column of one (g, brow) record — four 4-byte loads (channels
// 8g + 2q + brow; 256 contiguous bytes per wave and row), leaky-relu, one 16-byte LDS store.  Consecutive lanes store
// consecutive records: conflict-free (a thread owning four columns stores 64 bytes apart from its neighbour, a
// four-way bank conflict per store).  U items per thread and batch: 4 U loads in flight.  rows % 8 == 0.
template <int NTH, int U = 5>
__device__ __forceinline__ void stage_tile_pk(const float* __restrict__ xb, long x_ld, int rows, int LD, int ts, int tend,
                                              float slope, float* __restrict__ dst, int vec, float scale = 1.0f) {
    (void)vec;
    const int n = (rows >> 2) * LD;
    const int last = tend > 0 ? tend - 1 : 0;
    for (int idx0 = threadIdx.x; idx0 < n; idx0 += NTH * U) {
        float v[U][4];
        MI355_UNROLL
        for (int u = 0; u < U; ++u) {
            const int idx = idx0 + NTH * u;
            const int idc = idx < n ? idx : idx0;  // a missing item re-reads the first (discarded)
            const int gb = idc / LD, col = idc - gb * LD;  // gb = g * 2 + brow
            const int c0 = (gb >> 1) * 8 + (gb & 1);
            const int tt = ts + col;
            const int tc = tt < 0 ? 0 : (tt > last ? last : tt);  // clamped: every load unconditional, masked below
            MI355_UNROLL
            for (int q = 0; q < 4; ++q) v[u][q] = xb[(long)(c0 + 2 * q) * x_ld + tc];
        }
        SCHED_FENCE();
        MI355_UNROLL
        for (int u = 0; u < U; ++u) {
            const int idx = idx0 + NTH * u;
            if (idx >= n) continue;
            const int gb = idx / LD, col = idx - gb * LD;
            const int tt = ts + col;
            const bool in = tt >= 0 && tt < tend;
            reinterpret_cast<float4*>(dst)[idx] = in ? make_float4(lrelu_f(v[u][0], slope) * scale, lrelu_f(v[u][1], slope) * scale,
                                                                   lrelu_f(v[u][2], slope) * scale, lrelu_f(v[u][3], slope) * scale)
                                                     : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            (void)gb;
        }
    }
}

// ---- f32 operands on the bf16 matrix cores ---------------------------------------------------------------------------
// x = h + m + l exactly, each a bf16 (8-bit significand): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); both
// subtractions are exact in f32.  With both operands of a product split this way, the six leading partial products
// (hh, hm, mh, mm, hl, lh) carry everything down to 2^-24 relative — the f32 rounding level — and the bf16 MFMA forms
// each product exactly and sums in f32: f32-grade results at 6/16 of the f32 MFMA's time (MI355X: bf16 MFMA = 16x f32).
// In the kernels the ACTIVATIONS are split by truncation instead (h = the top 16 bits of x, and so on): the three terms
// still sum to x exactly, and it needs no conversion instruction — v_and + v_sub per level and one v_perm_b32 to pack a
// pair, 25 % less issue time than v_cvt_pk_bf16_f32 based rounding (tools/bf16x3_probe: V7 vs V2).  Truncated residuals
// are up to 4x larger than rounded ones (m < 2^-7 x, l < 2^-14 x), but the WEIGHTS are split with round-to-nearest on the
// host (m_w <= 2^-9 w, l_w <= 2^-18 w, random signs), so the dropped products m l_w + l m_w + l l_w stay below 2^-22 of
// the product in the worst case, ~2^-26 typically, and unbiased.
// One pair of elements -> one packed register per plane: 4 and + 4 sub + 3 perm.
__device__ __forceinline__ unsigned pack_hi16(unsigned lo, unsigned hi) {
#ifdef MI355_EMU
    return (lo >> 16) | (hi & 0xffff0000u);
#else
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // v_perm_b32: bytes 3:2 of each
#endif
}
__device__ __forceinline__ void split3_pk(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    h = pack_hi16(u0, u1);
#ifdef MI355_EMU
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = pack_hi16(v0, v1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = pack_hi16(__float_as_uint(s0), __float_as_uint(s1));
#else
    // the pair's two subtractions of a level as one packed op (v_pk_add_f32, neg modifiers): 9 VALU per pair, not 11
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f a = {a0, a1};
    const v2f ta = {__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)};
    const v2f r = a - ta;
    const unsigned v0 = __float_as_uint(r.x), v1 = __float_as_uint(r.y);
    m = pack_hi16(v0, v1);
    const v2f tr = {__uint_as_float(v0 & 0xffff0000u), __uint_as_float(v1 & 0xffff0000u)};
    const v2f q = r - tr;
    l = pack_hi16(__float_as_uint(q.x), __float_as_uint(q.y));
#endif
}
// the same split with plain scalar subtractions (11 VALU per pair): packed f32 VALU ops next to MFMAs cost ~13 cycles each
// beyond their issue slot (MI355X_MICROARCH.md, per-instruction constants) — kernels whose VALU runs beside a partner
// wave's MFMAs use this form
__device__ __forceinline__ void split3_sc(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    h = pack_hi16(u0, u1);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = pack_hi16(v0, v1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = pack_hi16(__float_as_uint(s0), __float_as_uint(s1));
}
// ---- MATH_F16X2: two fp16 terms per operand (11 + 11 significant bits), x ~ h + m with |x - h - m| <= 2^-22 |x| while both
// terms are normal halves.  Round-toward-zero conversions (v_cvt_pkrtz_f16_f32: a pair per instruction, saturating, so
// an out-of-range value stays finite); x - float(h) is exact.  5 VALU per pair (9 for the three-term bf16 split).
__device__ __forceinline__ void split2_pk(float a0, float a1, unsigned& h, unsigned& m) {
#ifdef MI355_EMU
    const unsigned h0 = hipemu_f32_to_f16_rtz(a0), h1 = hipemu_f32_to_f16_rtz(a1);
    h = h0 | (h1 << 16);
    const float r0 = a0 - hipemu_f16_to_f32(h0), r1 = a1 - hipemu_f16_to_f32(h1);
    m = hipemu_f32_to_f16_rtz(r0) | (hipemu_f32_to_f16_rtz(r1) << 16);
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    const auto hv = __builtin_amdgcn_cvt_pkrtz(a0, a1);
    h = __builtin_bit_cast(unsigned, hv);
    const v2f a = {a0, a1};
    const v2f hf = {(float)hv[0], (float)hv[1]};
    const v2f r = a - hf;
    m = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r.x, r.y));
#endif
}
__device__ __forceinline__ void split2_x8(const float4& lo4, const float4& hi4, uint4& h, uint4& m) {
    split2_pk(lo4.x, lo4.y, h.x, m.x);
    split2_pk(lo4.z, lo4.w, h.y, m.y);
    split2_pk(hi4.x, hi4.y, h.z, m.z);
    split2_pk(hi4.z, hi4.w, h.w, m.w);
}

// eight k-slots of one lane (two packed-tile float4: channels 16G + brow + 2e | 16G + 8 + brow + 2e) -> three planes
__device__ __forceinline__ void split3_x8(const float4& lo4, const float4& hi4, uint4& h, uint4& m, uint4& l) {
    split3_pk(lo4.x, lo4.y, h.x, m.x, l.x);
    split3_pk(lo4.z, lo4.w, h.y, m.y, l.y);
    split3_pk(hi4.x, hi4.y, h.z, m.z, l.z);
    split3_pk(hi4.z, hi4.w, h.w, m.w, l.w);
}

__device__ __forceinline__ void stage_tile_256(const float* __restrict__ xb, long x_ld, int rows, int LD, int ts, int tend,
                                               float slope, float* __restrict__ dst, int vec) {
    stage_tile<4, 1>(xb, x_ld, rows, LD, ts, tend, slope, dst, vec);
}
