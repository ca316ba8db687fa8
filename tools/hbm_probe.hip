// hbm_probe.hip — what this box's HBM gives a plain streaming kernel: write-only (fill), read-only (sum), copy.
// The roofline's 8 TB/s is the pin rate; these are the ceilings a store- or load-dominated epilogue can reach.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill(float4* p, long n4, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void k_fill_nt(float4* p, long n4, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f q = {v, v, v, v};
        __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(p + i));
    }
}
__global__ void k_sum(const float4* p, long n4, float* out) {
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1.2345f) out[0] = s;
}
__global__ void k_copy(const float4* a, float4* b, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
// the upsampler's store shape: a lane owns 4 consecutive floats, 2 floats off a 16-byte boundary (two 8-byte stores)
__global__ void k_fill_f2_misaligned(float* p, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; 4 * i + 6 <= n; i += (long)gridDim.x * blockDim.x) {
        float* q = p + 4 * i + 2;
        *reinterpret_cast<float2*>(q) = make_float2(v, v);
        *reinterpret_cast<float2*>(q + 2) = make_float2(v, v);
    }
}

int main() {
    const long bytes = 1L << 30, n4 = bytes / 16;
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double gb, auto fn) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f GB/s  (%.3f ms per GiB pass)\n", name, gb * 10 / (ms * 1e-3) / 1e9, ms / 10);
    };
    for (int blocks : {256 * 4, 256 * 8, 256 * 32}) {
        printf("grid %d x 256 threads\n", blocks);
        time("fill (float4)", bytes / 1.0, [&] { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, a, n4, 1.0f); });
        time("fill nontemporal", bytes / 1.0, [&] { hipLaunchKernelGGL(k_fill_nt, dim3(blocks), dim3(256), 0, 0, a, n4, 1.0f); });
        time("fill 2 x float2 misaligned", bytes / 1.0, [&] { hipLaunchKernelGGL(k_fill_f2_misaligned, dim3(blocks), dim3(256), 0, 0, (float*)a, bytes / 4, 1.0f); });
        time("sum (float4 loads)", bytes / 1.0, [&] { hipLaunchKernelGGL(k_sum, dim3(blocks), dim3(256), 0, 0, a, n4, out); });
        time("copy (read + write)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n4); });
    }
    time("hipMemsetAsync", bytes / 1.0, [&] { hipMemsetAsync(a, 0, bytes, 0); });
    return 0;
}
