// mrfs.h — device building blocks of the row-sweep MRF kernel (kernels_mrfs.cpp: one pass per resblock, 64 channels): LDS rings of bf16 planes addressed modulo their length, and one
// wave's 16-column tile of one conv on v_mfma_f32_16x16x32_bf16 in k_mrf_p's order of operations (bit-identical results).
#pragma once
#include "kernels.h"

namespace m355 {

// ring offset (any unit) -> [0, ring): valid for c < 2 * ring (unsigned: c - ring wraps around when c < ring)
__device__ __forceinline__ unsigned mrfs_wrap(unsigned c, unsigned ring) {
    const unsigned t = c - ring;
    return c < t ? c : t;
}

// Ring wrap without per-lane arithmetic in the steps.  A tile's lane n reads ring slot (sb + n + s d) mod ring at tap s; the
// lanes that have passed the ring's end at tap s are n >= ring - sb - s d — the same 16-lane pattern in all four quarters and
// a function of wave-uniform values only, so the mask is built on the scalar unit and ONE v_cndmask per step picks between
// the lane's two addresses (base, base - ring); the tap's s d columns ride in the instruction's immediate offset together
// with the k-group and plane offsets.
__device__ __forceinline__ unsigned long long mrfs_lane_mask(int th) {  // lanes n >= th of every 16-lane quarter
#ifdef MI355_EMU
    th = th < 0 ? 0 : (th > 16 ? 16 : th);
    const unsigned m16 = (0xffffu << th) & 0xffffu;
    const unsigned m32 = m16 * 0x10001u;
#else
    // on the scalar unit, whatever the compiler would pick for the clamp (it selects v_med3_i32 and drags the rest onto the VALU)
    unsigned m32;
    asm("s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, 16\n\ts_lshl_b32 %0, 0xffff, %0\n\ts_and_b32 %0, %0, 0xffff\n\ts_mul_i32 %0, %0, 0x10001"
        : "=&s"(m32)
        : "s"(th)
        : "scc");
#endif
    return ((unsigned long long)m32 << 32) | m32;
}
__device__ __forceinline__ unsigned mrfs_sel(unsigned a, unsigned b, unsigned long long mask, int lane) {  // mask bit set ? b : a
#ifdef MI355_EMU
    return ((mask >> lane) & 1ull) ? b : a;
#else
    (void)lane;
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
#endif
}

// the three plane fragments of step st of a tile: lds0 = the LDS window's base; a0 / a1 = this lane's byte offset of (plane 0,
// k-group 0, its quarter, the tile's column at tap 0) without / with the ring subtracted; wr = ring - sb (columns)
template <int G, int K>
__device__ __forceinline__ void mrfs_rd(uint4 (&f)[3], int st, const char* __restrict__ lds0, unsigned a0, unsigned a1, int wr, int lane, unsigned PS16,
                                        unsigned ring16, int d) {
    const int g = st / K, s = st % K;
    const unsigned a = mrfs_sel(a0, a1, mrfs_lane_mask(wr - s * d), lane);
    const char* p = lds0 + a + ((unsigned)g * 4u * ring16 + 16u * (unsigned)(s * d));
    MI355_UNROLL
    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const uint4*>(p + (unsigned)pl * PS16);
}

// the same with the step's lane mask given (mrfs_tile builds a tile's K masks in one batch ahead of its steps: five dependent
// scalar instructions per mask would otherwise sit in front of every step's fragment reads in the wave's in-order stream)
template <int G, int K>
__device__ __forceinline__ void mrfs_rd_m(uint4 (&f)[3], int st, const char* __restrict__ lds0, unsigned a0, unsigned a1, unsigned long long mask, int lane,
                                          unsigned PS16, unsigned ring16, int d) {
    const int g = st / K, s = st % K;
    const unsigned a = mrfs_sel(a0, a1, mask, lane);
    const char* p = lds0 + a + ((unsigned)g * 4u * ring16 + 16u * (unsigned)(s * d));
    MI355_UNROLL
    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const uint4*>(p + (unsigned)pl * PS16);
}

// One 16-column tile of one conv for this wave's 16 output rows: acc (+)= sum over the C / 32 k-groups and K taps, in
// k_mrf_p's order: per k-group two accumulator chains (small terms -> as, large terms -> ab), joined (ab + as) after the
// group's last tap.  W: this wave's fragments [k-group][tap][plane].  rq: this lane's byte offset of (plane 0, k-group 0, its
// quarter, slot 0) + 16 n; sb: ring slot of the tile's column 0 at tap 0 (wave-uniform, < ring); d: the dilation.
// bfirst: on entry the tile's step-0 fragments (read by the previous tile, or by the caller for a block's first), on exit the
// NEXT tile's (ring slot sbn), read behind this tile's last step — a tile never starts with an exposed LDS round trip.
template <int G, int K, int AH>
__device__ __forceinline__ void mrfs_tile(f32x4& acc, const uint4 (&W)[G][K][3], const char* __restrict__ lds0, unsigned rq, unsigned PS16, unsigned ring,
                                          unsigned sb, int d, int lane, uint4 (&bfirst)[3], unsigned sbn) {
    constexpr int NSTEP = G * K, RING = AH + 1;
    static_assert(AH >= 1 && NSTEP > AH, "the ring holds the running step and AH steps ahead");
    const unsigned ring16 = 16u * ring;
    const unsigned a0 = rq + 16u * sb, a1 = a0 - ring16;
    const int wr = WAVE_UNIFORM((int)ring - (int)sb);
    unsigned long long masks[K];
    MI355_UNROLL
    for (int s = 0; s < K; ++s) masks[s] = mrfs_lane_mask(wr - s * d);
    uint4 bf[RING][3];
    MI355_UNROLL
    for (int pl = 0; pl < 3; ++pl) bf[0][pl] = bfirst[pl];
    MI355_UNROLL
    for (int st = 1; st < AH; ++st) mrfs_rd_m<G, K>(bf[st], st, lds0, a0, a1, masks[st % K], lane, PS16, ring16, d);
    f32x4 ab = acc, as;
    MI355_UNROLL
    for (int r = 0; r < 4; ++r) as[r] = 0.0f;
    MI355_UNROLL
    for (int st = 0; st < NSTEP; ++st) {
        const int g = st / K, s = st % K;
        if (st + AH < NSTEP) mrfs_rd_m<G, K>(bf[(st + AH) % RING], st + AH, lds0, a0, a1, masks[(st + AH) % K], lane, PS16, ring16, d);
        if (st == NSTEP - 1) {  // the next tile's first step
            const unsigned n0 = rq + 16u * sbn;
            mrfs_rd<G, K>(bfirst, 0, lds0, n0, n0 - ring16, WAVE_UNIFORM((int)ring - (int)sbn), lane, PS16, ring16, d);
        }
        SCHED_FENCE();
        const int c = st % RING;
        as = MFMA_16x16x32_BF16(W[g][s][2], bf[c][0], as);  // small terms first
        ab = MFMA_16x16x32_BF16(W[g][s][1], bf[c][0], ab);
        as = MFMA_16x16x32_BF16(W[g][s][0], bf[c][2], as);
        ab = MFMA_16x16x32_BF16(W[g][s][0], bf[c][1], ab);
        as = MFMA_16x16x32_BF16(W[g][s][1], bf[c][1], as);
        ab = MFMA_16x16x32_BF16(W[g][s][0], bf[c][0], ab);
        SCHED_FENCE();
        if (s == K - 1) {  // the k-group is done: join the chains (k_mrf_p: acc = ab + as after every k-group)
            MI355_UNROLL
            for (int r = 0; r < 4; ++r) {
                ab[r] = ab[r] + as[r];
                as[r] = 0.0f;
            }
        }
    }
    acc = ab;
}

}  // namespace m355
