// Device helpers shared by the in-launch hand-off kernels (csrc/draft_persist.hip, csrc/topp_multi.hip): agent-scope accesses
// for data one workgroup hands to another inside a launch, and the pieces of the exact top-p select (2^-40 fixed-point masses,
// DPP prefix sums, the 512-thread scan of a 1 024-bin mass histogram).  Reference arithmetic: utils/sampling.py:5-27,43-60 via
// csrc/sampling.hip (topp_probs_kernel) — same integers, hence the same bits.
#pragma once
#include "common.h"

typedef unsigned long long u64;
#define DP_FIX_SHIFT 40
#define DP_WAVES 8                           // both kernels run 512-thread workgroups

namespace {
// ---- agent-scope (sc1) accesses: everything one workgroup hands to another inside the launch ----
__device__ __forceinline__ u64 ld8(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st8(void* p, u64 v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld4u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld4f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct U64x2 { u64 lo, hi; };
// ---- top-p helpers (csrc/sampling.hip: same exact-integer select) ----
__device__ __forceinline__ u64 dp_fix(float e) {                         // floor(e * 2^40), 0 <= e <= 1
    const unsigned b = __float_as_uint(e);
    const int ex = (int)(b >> 23);
    const u64 man = (u64)((b & 0x7FFFFFu) | 0x800000u);
    const int sh = ex - (127 + 23 - DP_FIX_SHIFT);
    if (ex == 0 || sh <= -24) return 0ull;
    return sh >= 0 ? (man << sh) : (man >> (-sh));
}
__device__ __forceinline__ u64 dp_wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        v += ((u64)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ u64 dp_shfl_u64(u64 v, int src) {
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src, 64);
    const unsigned hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}
#define DP_HB(b) ((b) + ((b) >> 4))                                      // one pad slot per 16 bins
struct DpTopp {                                                          // LDS
    u64 hist[1024 + 64];
    unsigned cnt[1024 + 64];
    u64 zpart[DP_WAVES];
    float wmax[DP_WAVES];
    int red_i[DP_WAVES];
    u64 S, tau, nkeep, zk, Z, hsel;
    unsigned ties, nlist;
    int digit;                                                           // >= 0 boundary bin, -1 keep everything, -2 below the candidate cut
    long long istar;                                                     // index of the last kept tie (tie ranking)
    unsigned nfin;                                                       // entries gathered from the boundary bin (in-wave finish)
};
// hist[digit] += m for the active lanes of a wave (wave-uniform call); a wave whose active lanes all name one bin adds ONE
// pre-summed value (degenerate rows: huge tie groups)
__device__ __forceinline__ void dp_hist_add(DpTopp* sh, bool active, int digit, u64 m, bool count, int lane) {
    const u64 act = __ballot(active);
    if (!act) return;
    const int first = __ffsll((long long)act) - 1;
    const int dref = __shfl(digit, first, 64);
    if (__popcll(act) > 8 && __all(!active || digit == dref)) {
        const u64 tot = dp_wave_sum_u64(active ? m : 0ull);
        if (lane == first) {
            atomicAdd(&sh->hist[DP_HB(dref)], tot);
            if (count) atomicAdd(&sh->cnt[DP_HB(dref)], (unsigned)__popcll(act));
        }
    } else if (active) {
        atomicAdd(&sh->hist[DP_HB(digit)], m);
        if (count) atomicAdd(&sh->cnt[DP_HB(digit)], 1u);
    }
}
// Inclusive prefix sum of a 64-bit value over the 64 lanes of a wave on the DPP path (row shifts inside the rows of 16, then
// the two row broadcasts): 6 steps of two v_mov_dpp + a 64-bit add.  (__shfl_* is ds_bpermute — an LDS round trip per step
// and per 32-bit half: the one-wave scan of the first build spent more than a microsecond per round in them.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dp_dpp_add(u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + (((u64)hi << 32) | lo);
}
__device__ __forceinline__ u64 dp_wave_prefix_u64(u64 v) {
    v = dp_dpp_add<0x111, 0xf>(v);                                       // row_shr:1
    v = dp_dpp_add<0x112, 0xf>(v);                                       // row_shr:2
    v = dp_dpp_add<0x114, 0xf>(v);                                       // row_shr:4
    v = dp_dpp_add<0x118, 0xf>(v);                                       // row_shr:8
    v = dp_dpp_add<0x142, 0xa>(v);                                       // row_bcast:15 -> rows 1, 3
    v = dp_dpp_add<0x143, 0xc>(v);                                       // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ unsigned dp_wave_prefix_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
// All 512 threads scan 1024 bins from the top: thread t holds the masses of bins 1023 - 2t (hA) and 1022 - 2t (hB).  Names the
// bin d with S(d) <= tau < S(d) + mass(d), S(d) = base + mass of the bins above d: the one thread that owns it writes
// tp->digit / tp->S / tp->hsel (tp->digit must be -1 on entry).  Two barriers; the caller reads the result behind them.
__device__ __forceinline__ void dp_scan512(DpTopp* tp, u64 hA, u64 hB, u64 base, u64 tau, int tid, int lane, int wave) {
    const u64 tot = hA + hB;
    const u64 inc = dp_wave_prefix_u64(tot);
    if (lane == 63) tp->zpart[wave] = inc;
    __syncthreads();
    u64 S = base + (inc - tot);
#pragma unroll
    for (int k = 0; k < DP_WAVES; ++k)
        if (k < wave) S += tp->zpart[k];
    if (hA != 0ull && S <= tau && tau - S < hA) {
        tp->digit = 1023 - 2 * tid;
        tp->S = S;
        tp->hsel = hA;
    } else {
        S += hA;
        if (hB != 0ull && S <= tau && tau - S < hB) {
            tp->digit = 1022 - 2 * tid;
            tp->S = S;
            tp->hsel = hB;
        }
    }
    __syncthreads();
}

}  // namespace
