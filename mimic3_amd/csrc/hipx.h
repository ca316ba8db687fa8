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
#define LAUNCH_KERNEL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((kernel), (grid), (block), (size_t)(shmem), __VA_ARGS__)
#define DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(hipemu::tl_worker->dyn_smem)
#define MI355_UNROLL
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA_32x32x2_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA_16x16x4_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define LAUNCH_KERNEL(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
// All LDS scratch lives in the dynamic region, base 16-byte aligned (cdna guide, Guideline 17).
#define DYN_SMEM(type, name)                                                   \
    extern __shared__ __attribute__((aligned(16))) unsigned char _dyn_smem_raw[]; \
    type* name = reinterpret_cast<type*>(_dyn_smem_raw)
#define MI355_UNROLL _Pragma("unroll")
#endif

#include <cstdint>
#include <cstdio>
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
