// Shared device helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/triforce_hip.h"

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TF_WAVE 64

#define TF_LAUNCH_CHECK()                      \
    do {                                       \
        hipError_t e__ = hipGetLastError();    \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// fp16 product / sum with torch's half semantics: compute in fp32, round once to fp16.
__device__ __forceinline__ h16 hmul_rn(h16 a, h16 b) { return (h16)((float)a * (float)b); }
__device__ __forceinline__ h16 hadd_rn(h16 a, h16 b) { return (h16)((float)a + (float)b); }

__device__ __forceinline__ half8 load_half8(const h16* p) { return *reinterpret_cast<const half8*>(p); }
// Streamed-once data (KV pages, weights in a decode forward): non-temporal 16-B load (global_load_dwordx4 ... nt)
// — shorter issue->landed latency for data no other CU will re-read (MI355X_MICROARCH.md, row nt-weights).
__device__ __forceinline__ half8 load_half8_stream(const h16* p) {
#ifdef TF_NO_NT
    return *reinterpret_cast<const half8*>(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const half8*>(p));
#endif
}
__device__ __forceinline__ void store_half8(h16* p, half8 v) { *reinterpret_cast<half8*>(p) = v; }
