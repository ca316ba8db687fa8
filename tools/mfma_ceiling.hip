// mfma_ceiling.hip — what v_mfma_f32_32x32x2_f32 loops can reach on this MI355X under the operand-feeding patterns the
// engine's kernels use.  Development tool (not part of libmi355vits.so):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o tools/mfma_ceiling && tools/mfma_ceiling
// Each wave owns MT x NT accumulator tiles and runs STEPS k-steps; per step it needs MT A fragments and NT B fragments:
//   mode bit 0: A fragments stream from global memory (L2-resident weights), 8-register ring, 4 steps ahead
//   mode bit 1: B fragments are read from LDS, one step ahead
// otherwise the operand is a loop-invariant register.  Prints TFLOP/s for a grid that fills the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MT, int NT, int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, float* __restrict__ out, int steps, int lds_floats, int wstride) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < lds_floats; i += blockDim.x) smem[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const float* wp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wp[m] = w + (long)(wid * MT + m) * wstride + lane;
    float ra[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[m][u] = (MODE & 1) ? wp[m][u * 64] : 0.5f + lane;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 4; u < 8; ++u) ra[m][u] = 0.25f;
    const float* xw = smem + (lane >> 5) * 44 + (lane & 31);
    float bb[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { bb[0][n] = (MODE & 2) ? xw[n * 32] : 1.0f + lane; bb[1][n] = 0.75f; }
    const int ldmask = lds_floats / 2 - 1;
    for (int s0 = 0; s0 < steps; s0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = ra[m][u];
            if (MODE & 1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) ra[m][(u + 4) & 7] = wp[m][(u + 4) * 64];
            }
            if (MODE & 2) {
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[(u + 1) & 1][n] = xw[(((s0 + u + 1) * 88) & ldmask) + n * 32];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[u & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) wp[m] += 8 * 64;
    }
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + tid] = sum;
}

// Same loop with 16-byte operand fetches: one global_load_dwordx4 / ds_read_b128 brings the fragments of four
// consecutive k-steps (weights packed [k-step group][lane][4], LDS tile packed [channel-pair group][column][4]).
template <int MT, int NT, int MODE>
__global__ __launch_bounds__(512) void kv(const float* __restrict__ w, float* __restrict__ out, int steps, int lds_floats, int wstride) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < lds_floats; i += blockDim.x) smem[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const float4* wp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wp[m] = reinterpret_cast<const float4*>(w + (long)(wid * MT + m) * wstride) + lane;
    float4 ra[MT][4];  // ring of 4 groups (16 steps), 2 groups ahead
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) ra[m][g] = (MODE & 1) ? wp[m][(g & 1) * 64] : make_float4(0.5f, 0.25f, 0.125f, lane);
    const float4* xw = reinterpret_cast<const float4*>(smem) + (lane >> 5) * 44 + (lane & 31);
    float4 bb[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { bb[0][n] = (MODE & 2) ? xw[n * 32] : make_float4(1.0f, 2.0f, 3.0f, lane); bb[1][n] = bb[0][n]; }
    const int ldmask = lds_floats / 8 - 1;
    for (int s0 = 0; s0 < steps; s0 += 16) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = ra[m][g];
            if (MODE & 1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) ra[m][(g + 2) & 3] = wp[m][(g + 2) * 64];
            }
            if (MODE & 2) {
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[(g + 1) & 1][n] = xw[(((s0 / 4 + g + 1) * 88) & ldmask) + n * 32];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float av = q == 0 ? a[m].x : q == 1 ? a[m].y : q == 2 ? a[m].z : a[m].w;
                        const float4 b4 = bb[g & 1][n];
                        const float bv = q == 0 ? b4.x : q == 1 ? b4.y : q == 2 ? b4.z : b4.w;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) wp[m] += 4 * 64;
    }
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + tid] = sum;
}

// A fragments one dword per step through the 8-register ring (as in k), B fragments as one ds_read_b128 per four k-steps
// (LDS tile packed [channel-pair group][column][4]), fetched one group ahead.
template <int MT, int NT>
__global__ __launch_bounds__(512) void kb4(const float* __restrict__ w, float* __restrict__ out, int steps, int lds_floats, int wstride) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < lds_floats; i += blockDim.x) smem[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const float* wp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wp[m] = w + (long)(wid * MT + m) * wstride + lane;
    float ra[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[m][u] = u < 4 ? wp[m][u * 64] : 0.25f;
    const float4* xw = reinterpret_cast<const float4*>(smem) + (lane >> 5) * 44 + (lane & 31);
    float4 bb[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { bb[0][n] = xw[n * 32]; bb[1][n] = bb[0][n]; }
    const int ldmask = lds_floats / 8 - 1;
    for (int s0 = 0; s0 < steps; s0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = ra[m][u];
#pragma unroll
            for (int m = 0; m < MT; ++m) ra[m][(u + 4) & 7] = wp[m][(u + 4) * 64];
            if ((u & 3) == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[((u >> 2) + 1) & 1][n] = xw[(((s0 / 4 + (u >> 2) + 1) * 88) & ldmask) + n * 32];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float4 b4 = bb[(u >> 2) & 1][n];
                    const float bv = (u & 3) == 0 ? b4.x : (u & 3) == 1 ? b4.y : (u & 3) == 2 ? b4.z : b4.w;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv, acc[m][n], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) wp[m] += 8 * 64;
    }
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + tid] = sum;
}

template <int MT, int NT, int MODE>
static void run(const char* name, int waves, int blocks_per_cu, int lds_bytes, const float* w, float* out, int wstride,
                bool vec = false, int steps = 1152, int reps = 10, bool b4 = false) {
    const int cus = 256;
    const int grid = cus * blocks_per_cu;
    auto kk = b4 ? kb4<MT, NT> : (vec ? kv<MT, NT, MODE> : k<MT, NT, MODE>);
    CHECK(hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kk, dim3(grid), dim3(64 * waves), lds_bytes, 0, w, out, steps, lds_bytes / 4, wstride);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kk, dim3(grid), dim3(64 * waves), lds_bytes, 0, w, out, steps, lds_bytes / 4, wstride);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = (double)grid * waves * steps * MT * NT * 4096.0;
    printf("%s%-34s MTxNT=%dx%d waves/blk=%d blk/CU=%d (%.1f waves/SIMD) lds=%3dKB  %7.3f ms  %6.1f TFLOP/s\n", b4 ? "B4 " : (vec ? "x4 " : "   "), name, MT, NT, waves,
           blocks_per_cu, waves * blocks_per_cu / 4.0, lds_bytes / 1024, ms, flops / ms * 1e-9);
}

int main(int argc, char** argv) {
    const bool sustained = argc > 1;
    const int wstride = 1152 * 64 + 4096;
    const size_t wn = (size_t)48 * wstride + 65536;  // up to 8 waves x 6 streams
    std::vector<float> hw(wn, 0.001f);
    float *w, *out;
    CHECK(hipMalloc(&w, wn * 4));
    CHECK(hipMalloc(&out, 1 << 26));
    CHECK(hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice));
    printf("nominal peak: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz = 157.3 TFLOP/s\n");
    if (argc > 1 && argv[1][0] == 'b') {  // B-side 16-byte reads only
        run<1, 3, 3>("mrf C=32 shape", 8, 1, 65536, w, out, wstride);
        run<1, 3, 3>("mrf C=32 shape", 8, 1, 65536, w, out, wstride, false, 1152, 10, true);
        run<2, 3, 3>("mrf C=64 shape", 8, 1, 65536, w, out, wstride);
        run<2, 3, 3>("mrf C=64 shape", 8, 1, 65536, w, out, wstride, false, 1152, 10, true);
        run<3, 1, 3>("wn_layer 4-wave shape", 4, 3, 49152, w, out, wstride);
        run<3, 1, 3>("wn_layer 4-wave shape", 4, 3, 49152, w, out, wstride, false, 1152, 10, true);
        run<2, 2, 3>("generic conv shape", 4, 3, 53248, w, out, wstride);
        run<2, 2, 3>("generic conv shape", 4, 3, 53248, w, out, wstride, false, 1152, 10, true);
        run<2, 4, 3>("2x4 tiles", 4, 2, 65536, w, out, wstride);
        run<2, 4, 3>("2x4 tiles", 4, 2, 65536, w, out, wstride, false, 1152, 10, true);
        run<4, 2, 3>("4x2 tiles", 4, 2, 65536, w, out, wstride);
        run<4, 2, 3>("4x2 tiles", 4, 2, 65536, w, out, wstride, false, 1152, 10, true);
        run<2, 4, 3>("2x4 tiles, 8 waves", 8, 1, 65536, w, out, wstride);
        run<2, 4, 3>("2x4 tiles, 8 waves", 8, 1, 65536, w, out, wstride, false, 1152, 10, true);
        return 0;
    }
    if (sustained) {  // ~0.3 s of back-to-back launches per line: what the clocks settle at
        printf("sustained (800 launches each)\n");
        run<2, 2, 0>("registers only", 4, 2, 1024, w, out, wstride, false, 1152, 800);
        run<2, 1, 3>("A L2 + B LDS (wn_layer shape)", 6, 3, 33792, w, out, wstride, false, 1152, 800);
        run<2, 3, 3>("A L2 + B LDS (mrf C=64 shape)", 8, 1, 65536, w, out, wstride, false, 1152, 800);
        run<2, 3, 3>("A L2 + B LDS (mrf C=64 shape)", 8, 1, 65536, w, out, wstride, true, 1152, 800);
        return 0;
    }
    run<2, 1, 0>("registers only", 4, 1, 1024, w, out, wstride);
    run<2, 1, 0>("registers only", 4, 2, 1024, w, out, wstride);
    run<2, 1, 0>("registers only", 8, 2, 1024, w, out, wstride);
    run<2, 2, 0>("registers only", 4, 2, 1024, w, out, wstride);
    run<2, 1, 1>("A from L2", 4, 2, 1024, w, out, wstride);
    run<2, 1, 1>("A from L2", 8, 2, 1024, w, out, wstride);
    run<2, 1, 1>("A from L2", 4, 2, 1024, w, out, wstride, true);
    run<2, 1, 1>("A from L2", 8, 2, 1024, w, out, wstride, true);
    run<2, 1, 2>("B from LDS", 4, 2, 32768, w, out, wstride);
    run<2, 1, 2>("B from LDS", 8, 2, 32768, w, out, wstride);
    run<2, 1, 2>("B from LDS", 8, 2, 32768, w, out, wstride, true);
    run<2, 1, 3>("A L2 + B LDS (wn_layer shape)", 6, 3, 33792, w, out, wstride);
    run<2, 1, 3>("A L2 + B LDS (wn_layer shape)", 6, 3, 33792, w, out, wstride, true);
    run<2, 1, 3>("A L2 + B LDS", 4, 2, 32768, w, out, wstride);
    run<2, 1, 3>("A L2 + B LDS", 4, 2, 32768, w, out, wstride, true);
    run<2, 1, 3>("A L2 + B LDS", 8, 2, 32768, w, out, wstride);
    run<2, 1, 3>("A L2 + B LDS", 8, 2, 32768, w, out, wstride, true);
    run<2, 1, 3>("A L2 + B LDS", 4, 4, 32768, w, out, wstride);
    run<2, 2, 3>("A L2 + B LDS", 4, 2, 32768, w, out, wstride);
    run<2, 2, 3>("A L2 + B LDS", 4, 2, 32768, w, out, wstride, true);
    run<2, 2, 3>("A L2 + B LDS", 8, 2, 32768, w, out, wstride);
    run<2, 2, 3>("A L2 + B LDS", 8, 2, 32768, w, out, wstride, true);
    run<2, 3, 3>("A L2 + B LDS (mrf C=64 shape)", 8, 1, 65536, w, out, wstride);
    run<2, 3, 3>("A L2 + B LDS (mrf C=64 shape)", 8, 1, 65536, w, out, wstride, true);
    run<1, 3, 3>("A L2 + B LDS (mrf C=32 shape)", 8, 1, 65536, w, out, wstride);
    run<1, 3, 3>("A L2 + B LDS (mrf C=32 shape)", 8, 1, 65536, w, out, wstride, true);
    run<1, 1, 3>("A L2 + B LDS", 8, 2, 32768, w, out, wstride);
    run<1, 1, 3>("A L2 + B LDS", 8, 2, 32768, w, out, wstride, true);
    run<4, 1, 3>("A L2 + B LDS", 4, 2, 32768, w, out, wstride);
    run<4, 1, 3>("A L2 + B LDS", 4, 2, 32768, w, out, wstride, true);
    run<3, 1, 3>("A L2 + B LDS (wn_layer 4-wave shape)", 4, 3, 49152, w, out, wstride);
    run<3, 1, 3>("A L2 + B LDS (wn_layer 4-wave shape)", 4, 3, 49152, w, out, wstride, true);
    run<3, 2, 3>("A L2 + B LDS", 4, 1, 98304, w, out, wstride);
    run<3, 2, 3>("A L2 + B LDS", 4, 2, 65536, w, out, wstride);
    run<3, 2, 3>("A L2 + B LDS", 4, 2, 65536, w, out, wstride, true);
    run<2, 2, 3>("A L2 + B LDS", 4, 3, 49152, w, out, wstride);
    run<2, 2, 3>("A L2 + B LDS (generic conv shape)", 4, 3, 53248, w, out, wstride);
    return 0;
}
