// bf16x3_probe.hip — instruction-rate facts behind the MATH_BF16X3 inner loops on this MI355X (development tool, not part
// of libmi355vits.so):   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_probe.hip -o tools/bf16x3_probe && tools/bf16x3_probe
// One "group" = what a wave does per 16-channel k-group with NT column tiles: 6*NT bf16 MFMAs (32x32x16), and depending
// on the variant the raw-f32 LDS reads + 3-way split (on the fly) or the pre-split LDS reads, plus 3 A-plane loads.
//   V0 mfma only, tile-major (6 dependent MFMAs per accumulator back to back)
//   V1 mfma only, product-major (consecutive MFMAs on different accumulators)
//   V2 split only (LDS raw reads + split3, no MFMA)
//   V3 on-the-fly: LDS raw reads + split + MFMAs, A from registers
//   V4 V3 + A planes from global memory (L2-resident), one group ahead
//   V5 pre-split: 3*NT ds_read_b128 + MFMAs + A planes from global
//   V6 V5 with A planes from LDS as well
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned cvt(float lo, float hi) { f32x2_t v = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ void split3_pk(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt(a0, a1);
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    m = cvt(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt(s0, s1);
}
// truncation split: hi = top 16 bits (no cvt instruction): and + sub per level, v_perm_b32 to pack
__device__ __forceinline__ void split3_pk_t(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
template <bool TR>
__device__ __forceinline__ void split8x(const float4& a, const float4& b, uint4& h, uint4& m, uint4& l) {
    if (TR) {
        split3_pk_t(a.x, a.y, h.x, m.x, l.x); split3_pk_t(a.z, a.w, h.y, m.y, l.y);
        split3_pk_t(b.x, b.y, h.z, m.z, l.z); split3_pk_t(b.z, b.w, h.w, m.w, l.w);
    } else {
        split3_pk(a.x, a.y, h.x, m.x, l.x); split3_pk(a.z, a.w, h.y, m.y, l.y);
        split3_pk(b.x, b.y, h.z, m.z, l.z); split3_pk(b.z, b.w, h.w, m.w, l.w);
    }
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& h, uint4& m, uint4& l) {
    split3_pk(a.x, a.y, h.x, m.x, l.x); split3_pk(a.z, a.w, h.y, m.y, l.y);
    split3_pk(b.x, b.y, h.z, m.z, l.z); split3_pk(b.z, b.w, h.w, m.w, l.w);
}
__device__ __forceinline__ f32x16 mf(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int V, int NT>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ w, float* __restrict__ out, int groups) {
    extern __shared__ float4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) smem4[i] = make_float4(0.001f * (i & 255), 0.5f, -0.25f, 1.0f + i);
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    const uint4* wp = w + lane;
    uint4 ra[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) ra[0][p] = ra[1][p] = make_uint4(0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u);
    if (V == 4 || V == 5) {
#pragma unroll
        for (int p = 0; p < 3; ++p) ra[0][p] = wp[p * 64];
    }
    const float4* xw = smem4 + lane;
    const uint4* xu = reinterpret_cast<const uint4*>(smem4) + lane;
    float4 xb[2][NT][2];
    uint4 pb[2][NT][3];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        xb[0][n][0] = xw[n * 64]; xb[0][n][1] = xw[n * 64 + 2048];
#pragma unroll
        for (int p = 0; p < 3; ++p) pb[0][n][p] = xu[n * 64 + p * 1024];
    }
    unsigned sink = 0;
    for (int g0 = 0; g0 < groups; g0 += 2) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int off = ((g0 + g + 1) * 192) & 1023;
            if (V == 4 || V == 5) {
#pragma unroll
                for (int p = 0; p < 3; ++p) ra[(g + 1) & 1][p] = wp[((g0 + g + 1) & 63) * 192 + p * 64];
            }
            if (V == 6) {
#pragma unroll
                for (int p = 0; p < 3; ++p) ra[(g + 1) & 1][p] = xu[4096 + off + p * 64];
            }
            if (V == 2 || V == 3 || V == 4 || V == 7 || V == 8) {
#pragma unroll
                for (int n = 0; n < NT; ++n) { xb[(g + 1) & 1][n][0] = xw[off + n * 64]; xb[(g + 1) & 1][n][1] = xw[off + n * 64 + 2048]; }
            }
            if (V == 5 || V == 6) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int p = 0; p < 3; ++p) pb[(g + 1) & 1][n][p] = xu[off + n * 64 + p * 1024];
            }
            FENCE();
            const uint4 ah = ra[g & 1][0], am = ra[g & 1][1], al = ra[g & 1][2];
            if (V == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[n] = mf(al, ah, acc[n]); acc[n] = mf(ah, al, acc[n]); acc[n] = mf(am, am, acc[n]);
                    acc[n] = mf(am, ah, acc[n]); acc[n] = mf(ah, am, acc[n]); acc[n] = mf(ah, ah, acc[n]);
                }
            } else if (V == 1) {
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n] = mf(p & 1 ? am : ah, p & 2 ? al : ah, acc[n]);
            } else if (V == 2) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    uint4 bh, bm, bl;
                    split8(xb[g & 1][n][0], xb[g & 1][n][1], bh, bm, bl);
                    sink ^= bh.x ^ bm.y ^ bl.z ^ bh.w ^ bm.x ^ bl.y ^ bh.z ^ bm.w ^ bl.x ^ bh.y ^ bm.z ^ bl.w;
                }
            } else if (V == 7) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    uint4 bh, bm, bl;
                    split8x<true>(xb[g & 1][n][0], xb[g & 1][n][1], bh, bm, bl);
                    sink ^= bh.x ^ bm.y ^ bl.z ^ bh.w ^ bm.x ^ bl.y ^ bh.z ^ bm.w ^ bl.x ^ bh.y ^ bm.z ^ bl.w;
                }
            } else if (V == 8) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    uint4 bh, bm, bl;
                    split8x<true>(xb[g & 1][n][0], xb[g & 1][n][1], bh, bm, bl);
                    acc[n] = mf(al, bh, acc[n]); acc[n] = mf(ah, bl, acc[n]); acc[n] = mf(am, bm, acc[n]);
                    acc[n] = mf(am, bh, acc[n]); acc[n] = mf(ah, bm, acc[n]); acc[n] = mf(ah, bh, acc[n]);
                }
            } else if (V == 3 || V == 4) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    uint4 bh, bm, bl;
                    split8(xb[g & 1][n][0], xb[g & 1][n][1], bh, bm, bl);
                    acc[n] = mf(al, bh, acc[n]); acc[n] = mf(ah, bl, acc[n]); acc[n] = mf(am, bm, acc[n]);
                    acc[n] = mf(am, bh, acc[n]); acc[n] = mf(ah, bm, acc[n]); acc[n] = mf(ah, bh, acc[n]);
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint4 bh = pb[g & 1][n][0], bm = pb[g & 1][n][1], bl = pb[g & 1][n][2];
                    acc[n] = mf(al, bh, acc[n]); acc[n] = mf(ah, bl, acc[n]); acc[n] = mf(am, bm, acc[n]);
                    acc[n] = mf(am, bh, acc[n]); acc[n] = mf(ah, bm, acc[n]); acc[n] = mf(ah, bh, acc[n]);
                }
            }
            FENCE();
        }
    }
    float sum = __uint_as_float(sink);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[n][r];
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + tid] = sum;
}

template <int V, int NT>
void run(const char* name, const uint4* w, float* out, int waves_per_block, int blocks_per_cu = 1) {
    const int groups = 4096;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto fn = k<V, NT>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t sh = blocks_per_cu == 1 ? 150 * 1024 : 72 * 1024;
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves_per_block), sh, 0, w, out, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves_per_block), sh, 0, w, out, groups);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double waves_per_simd = waves_per_block * blocks_per_cu / 4.0;
    const double ns_per_group_per_simd = ms * 1e6 / groups / waves_per_simd;   // time one SIMD spends per wave-group
    const double mfma = (V == 2 || V == 7) ? 0 : 6.0 * NT;
    const double tf = mfma * 2.0 * 32 * 32 * 16 * groups * grid * waves_per_block / (ms * 1e-3) / 1e12;
    printf("%-34s NT=%d waves/CU=%2d  %8.3f ms  %7.1f ns/group/SIMD = %6.0f cyc@2.4GHz (%4.1f cyc/MFMA)  %7.1f TF bf16 = %6.1f TF f32-equivalent\n",
           name, NT, waves_per_block * blocks_per_cu, ms, ns_per_group_per_simd, ns_per_group_per_simd * 2.4,
           mfma ? ns_per_group_per_simd * 2.4 / mfma : 0.0, tf, tf / 6.0);
}

int main() {
    uint4* w; float* out;
    CHECK(hipMalloc(&w, 64 * 192 * 16 + 4096)); CHECK(hipMemset(w, 0x3f, 64 * 192 * 16 + 4096));
    CHECK(hipMalloc(&out, 1 << 22));
    for (int wpb : {4, 8}) {
        run<0, 3>("V0 mfma tile-major", w, out, wpb);
        run<1, 3>("V1 mfma product-major", w, out, wpb);
        run<2, 3>("V2 split only", w, out, wpb);
        run<3, 3>("V3 on-the-fly (A regs)", w, out, wpb);
        run<7, 3>("V7 split only, truncation", w, out, wpb);
        run<8, 3>("V8 on-the-fly, truncation split", w, out, wpb);
        run<4, 3>("V4 on-the-fly + A global", w, out, wpb);
        run<5, 3>("V5 pre-split + A global", w, out, wpb);
        run<6, 3>("V6 pre-split + A from LDS", w, out, wpb);
        run<5, 2>("V5 pre-split + A global", w, out, wpb);
        run<5, 1>("V5 pre-split + A global", w, out, wpb);
        run<6, 1>("V6 pre-split + A from LDS", w, out, wpb);
        run<6, 2>("V6 pre-split + A from LDS", w, out, wpb);
        run<3, 2>("V3 on-the-fly (A regs)", w, out, wpb);
    }
    run<5, 3>("V5 pre-split, 2 WG/CU x 4 waves", w, out, 4, 2);
    run<5, 2>("V5 pre-split, 2 WG/CU x 4 waves", w, out, 4, 2);
    return 0;
}
