/* mi355vits_lab.h — test / benchmark / probe hooks around the kernels of libmi355vits (NOT the product ABI: that is
 * include/mi355vits.h, which mirrors the reference's onnxruntime boundary, mimic3_tts/voice.py:230,403-405).
 *
 * Exported by
 *   mimic3_amd/csrc/libmi355vits_hooks.so  — the product library's own object files + csrc/lab_api.cpp: the kernel-level tests
 *                                            (tests/test_gpu_parity.py ...) and bench.py's box probe run the PRODUCT's kernels through it;
 *   mimic3_amd/csrc/libmi355vits_lab.so    — the -DMI355_LAB build (kernel-choice switches for A/B runs, tools/);
 *   tests/emu/libmi355vits_emu.so          — the CPU model of the kernels.
 * libmi355vits.so itself exports none of these symbols. */
#ifndef MI355VITS_LAB_H
#define MI355VITS_LAB_H

#include "mi355vits.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel unit-test hook: one Conv1d through a chosen implementation on host buffers.
 * impl: 0 = generic VALU kernel, 1 = fp32-MFMA kernel, 2 = split-bf16 staged kernel (MI355VITS_MATH_BF16X3; needs
 * Cin % 32 == 0 and T > 512), 3 = the text encoder's slice kernel (MI355VITS_MATH_BF16X3; Cin % 192 == 0, K in {1, 3}, dilation
 * 1; Cin > 192: the raw sums of the 192-channel slices added up, no bias / residual), 4 = the 128-channel resblock conv with
 * every input channel resident in LDS (MI355VITS_MATH_BF16X3; Cin = Cout = 128, (K, dilation) in {(3,1), (3,2), (5,2), (5,6),
 * (7,3), (7,12)}, needs res and in_len).  mi355vits_test_conv_transpose1d: impl 0 = generic, 1 = f32-MFMA polyphase, 2 = the
 * staged split-bf16 polyphase kernels, 3 = the resident-input polyphase kernels (256 -> 128 and 128 -> 64 with stride 8 / K 16,
 * 64 -> 32 with stride 4 / K 8).  See tests/test_gpu_parity.py, tests/test_emu_engine.py. */
typedef struct mi355vits_conv_test {
    int32_t impl, B, Cin, Cout, T, K, dilation;
    const float* x;       /* [B,Cin,T] */
    const float* w;       /* [Cout,Cin,K] */
    const float* bias;    /* [Cout] or NULL */
    const float* res;     /* [B,Cout,T] or NULL */
    const int32_t* in_len;  /* [B] or NULL */
    const int32_t* out_len; /* [B] or NULL */
    float in_slope;       /* leaky-relu slope on the input, 1 = identity */
    int32_t relu;         /* relu on the output */
    float out_scale;
    int32_t res_sub;      /* y = res - conv instead of res + conv */
    float* y;             /* [B,Cout,T], also the accumulate source when accumulate != 0 */
    int32_t accumulate;
} mi355vits_conv_test;
int mi355vits_test_conv1d(int device, const mi355vits_conv_test* t);
int mi355vits_test_conv_transpose1d(int device, int impl, int B, int Cin, int Cout, int Tin, int K, int stride,
                                    const float* x, const float* w, const float* bias, float in_slope, float* y);
/* Kernel micro-benchmark hook (tools/convbench.py): times `reps` launches of one MFMA Conv1d on random device data.
 * epi: 0 = standard epilogue (bias + residual), 1 = WaveNet gate (Cout = 2*H), 2 = res/skip.  (The tile-shape overrides
 * MI355VITS_CONV_CFG / MI355VITS_CONV_CHUNK exist in the lab build of the library only, csrc/hipx.h lab_getenv.) */
int mi355vits_bench_conv1d(int device, int B, int Cin, int Cout, int T, int K, int dilation, int epi, int reps,
                           float* ms_per_launch);
/* MFMA fragment-layout self test: returns 0 when the 32x32x2 and 16x16x4 f32 MFMA lane maps
 * assumed by the kernels hold on this device; max abs error in *err. */
int mi355vits_test_mfma_layout(int device, float* err);
/* Box probe (bench.py: the numbers ride in the JSON line so that a slow lease can be told from a slow kernel; ~30 ms, 1 GiB of
 * scratch): out[0] = GB/s all CUs together reach streaming ONE 2.6 MB table out of the L2 with 16-byte buffer loads (the
 * weight-fragment pattern of the WaveNet / resident-input kernels), out[1] = ns per dependent vector load over 2 MB (L2 hits),
 * out[2] = GB/s (read + written) of a 256 MiB HBM copy, out[3] = compute units, out[4] = the table stream of out[0] again while
 * every workgroup also copies its slice of 256 MiB through the same L2 (8 bytes of table per byte of copy: what the cache sees of a
 * weight-streaming kernel), out[5] = GB/s streaming a 24 MB table (fits the memory-side cache, not an XCD's L2), out[6] / out[7] =
 * ns per dependent load over 32 MB (memory-side cache) / 1 GiB (HBM, mostly TLB misses). */
int mi355vits_probe_device(int device, double out[8]);
/* The same L2 stream over 2.6 MB windows of THIS handle's weight arena (the bytes the kernels actually stream): out[0..2] = min /
 * median / max GB/s over the windows with eight 16-byte loads in flight per lane, out[3..5] = with one (the latency a kernel sees
 * that fetches its fragments a step ahead), out[6] = windows measured, out[7] = low 36 bits of the arena's device address. */
int mi355vits_probe_weights(mi355vits_handle h, double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* MI355VITS_LAB_H */
