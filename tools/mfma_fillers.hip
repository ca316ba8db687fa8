// mfma_fillers.hip — how many single-issue instructions ride for free beside a bf16 MFMA on gfx950, per MFMA shape and per
// waves-per-SIMD?  Development tool (not part of libmi355vits.so):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_fillers.hip -o tools/mfma_fillers && tools/mfma_fillers
// Every wave runs a loop of MFMAs on two independent accumulator chains; between two MFMAs it issues F independent
// v_fma_f32 fillers (and optionally one ds_read_b128 per two MFMAs).  One workgroup per CU.  Prints ns per MFMA and the
// cycles that is at the clock measured by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE, int F, int LDS>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    extern __shared__ uint4 sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += blockDim.x) sm[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.5f + i + lane;
    const float m = 1.0001f, c = 0.25f;
    uint4 r = sm[lane];
    long long t0 = 0;
    if (tid == 0) t0 = __builtin_readcyclecounter();
    if constexpr (SHAPE == 16) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f]) : "v"(m), "v"(c));
                if (LDS) { asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((lane * 16 + u * 1024) & 32767)); }
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f]) : "v"(m), "v"(c));
            }
            if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        out[blockIdx.x * blockDim.x + tid] = acc0[0] + acc1[1] + x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7] + (float)r.x;
    } else {
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) { acc0[i] = 0; acc1[i] = 0; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(m), "v"(c));
                if (LDS) { asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((lane * 16 + u * 1024) & 32767)); }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(m), "v"(c));
            }
            if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        out[blockIdx.x * blockDim.x + tid] = acc0[0] + acc1[1] + x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7] + (float)r.x;
    }
    if (tid == 0 && blockIdx.x == 0) cyc[0] = __builtin_readcyclecounter() - t0;
}

template <int SHAPE, int F, int LDS>
void run(int waves, float* d_out, long long* d_cyc) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<SHAPE, F, LDS><<<grid, waves * 64, 32768>>>(d_out, 100, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<SHAPE, F, LDS><<<grid, waves * 64, 32768>>>(d_out, iters, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc = 0;
    CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * 16 * (waves / 4.0);
    printf("shape %2d  waves/SIMD %d  fillers/MFMA %d  lds %d : %.2f ns per MFMA and SIMD, %.1f shader cycles per MFMA and SIMD (clock %.2f GHz)\n", SHAPE,
           waves / 4, F, LDS, ms * 1e6 / mfma_per_simd, (double)cyc / mfma_per_simd, cyc / (ms * 1e6));
}

int main() {
    float* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 256 * 512 * 4));
    CHECK(hipMalloc(&d_cyc, 8));
    for (int waves : {4, 8}) {
        run<16, 0, 0>(waves, d_out, d_cyc); run<16, 1, 0>(waves, d_out, d_cyc); run<16, 2, 0>(waves, d_out, d_cyc); run<16, 3, 0>(waves, d_out, d_cyc);
        run<16, 4, 0>(waves, d_out, d_cyc); run<16, 6, 0>(waves, d_out, d_cyc); run<16, 2, 1>(waves, d_out, d_cyc);
        run<32, 0, 0>(waves, d_out, d_cyc); run<32, 2, 0>(waves, d_out, d_cyc); run<32, 4, 0>(waves, d_out, d_cyc); run<32, 6, 0>(waves, d_out, d_cyc);
        run<32, 8, 0>(waves, d_out, d_cyc); run<32, 12, 0>(waves, d_out, d_cyc); run<32, 4, 1>(waves, d_out, d_cyc);
    }
    return 0;
}
