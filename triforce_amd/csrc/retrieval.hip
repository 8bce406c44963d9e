// Retrieval-cache build and KV row movement for gfx950.
//
// Replaces RetrievalCache.init_graph_cache (models/cache.py:146-178; distributed twin :517-556):
//   chunk_k = K[:prefill].view(C, chunk, H, D).mean(dim=-3)            -> tf_retrieval_score
//   chunk_attn = q @ chunk_k^T   (fp16, un-scaled)                      -> tf_retrieval_score
//   topk(chunk_attn[:, :, 1:], k = sets-1) + 1, chunk 0 prepended       -> tf_retrieval_topk
//   gather of whole chunks for K and V                                  -> tf_retrieval_gather
// and RetrievalCache.update_graph_cache / StreamingLLMEvictionCache.evict_* (cache.py:180-182,
// 252-265) -> tf_kv_copy_rows / tf_kv_shift_rows.
//
// All of it is HBM-bound byte work: K is streamed exactly once (16-B loads, 256-B row segments
// per 16-lane group), nothing is staged in a temporary, the top-k runs in LDS.
#include "common.h"

#ifndef TF_RSCORE_CAP_CEIL
#define TF_RSCORE_CAP_CEIL 0
#endif

// ---- scoring ------------------------------------------------------------------------------
// One 16-lane group per chunk (a wave scores 4 chunks per step); lane (c4, li) owns 8 d's.
// Rounding points follow the reference: mean -> fp16, dot -> fp16.
template <int D>
__global__ __launch_bounds__(256) void retrieval_score_kernel(const h16* __restrict__ k, int64_t stride_t,
                                                              int64_t stride_h, const h16* __restrict__ q,
                                                              h16* __restrict__ scores, int C, int chunk) {
    static_assert(D == 128 || D == 64, "head_dim");
    constexpr int LPR = D / 8;                   // lanes per row (16 for D=128, 8 for D=64)
    constexpr int GPB = 256 / LPR;               // chunk groups per block
    const int h = blockIdx.y;
    const int grp = threadIdx.x / LPR, li = threadIdx.x % LPR;
    const h16* kb = k + (int64_t)h * stride_h + 8 * li;
    float qf[8];
    {
        const half8 qv = load_half8(q + (int64_t)h * D + 8 * li);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e];
    }
    const float inv = 1.0f / (float)chunk;
    for (int c = blockIdx.x * GPB + grp; c < C; c += gridDim.x * GPB) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const h16* kp = kb + (int64_t)c * chunk * stride_t;
        for (int r = 0; r < chunk; ++r) {          // sequential fp32 accumulation over the chunk rows
            const half8 kv = load_half8_stream(kp + (int64_t)r * stride_t);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)kv[e];
        }
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const h16 mean = (h16)(acc[e] * inv);    // chunk mean rounded to fp16 (cache.py:154)
            dot = fmaf(qf[e], (float)mean, dot);
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        if (li == 0) scores[(int64_t)h * C + c] = (h16)dot;
    }
}

// ---- per-head top-k: full bitonic sort of (score, chunk) keys in LDS ------------------------
__device__ __forceinline__ uint32_t sortable_fp16(uint32_t b) {
    return (b & 0x8000u) ? ((~b) & 0xFFFFu) : (b | 0x8000u);
}

__global__ __launch_bounds__(1024) void retrieval_topk_kernel(const uint16_t* __restrict__ scores,
                                                               int32_t* __restrict__ idx, int C, int sets,
                                                               int npow2) {
    extern __shared__ uint32_t keys[];
    const int h = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint16_t* sc = scores + (int64_t)h * C;
    // candidates are chunks 1..C-1; key = sortable(score) << 16 | (65535 - chunk): descending key
    // order == descending score, ascending chunk among equal scores.  Padding sorts last.
    for (int i = tid; i < npow2; i += nt) {
        const int c = i + 1;
        keys[i] = (c < C) ? ((sortable_fp16(sc[c]) << 16) | (uint32_t)(65535 - (c & 0xFFFF))) : 0u;
    }
    __syncthreads();
    for (int kk = 2; kk <= npow2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool desc = ((i & kk) == 0);
                    const uint32_t a = keys[i], b = keys[ixj];
                    if ((a < b) == desc) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    int32_t* out = idx + (int64_t)h * sets;
    if (tid == 0) out[0] = 0;
    for (int i = tid; i < sets - 1; i += nt) out[1 + i] = 65535 - (int32_t)(keys[i] & 0xFFFFu);
}

// ---- per-head top-k, select-then-sort form ---------------------------------------------------
// The keys are distinct 32-bit integers, so the m-th largest one is found EXACTLY by a 4-pass radix select
// (8 bits per pass, one 256-bin LDS histogram each, integer atomics: deterministic), the m keys >= it are compacted
// and only those are bitonic-sorted: for cfg2 (m = 511 of 15 615) 4 histogram passes + 45 passes over 512 keys
// instead of 105 passes over 16 384 keys (213 us -> ~25 us per layer).  Same output, bit for bit, as the full sort.
__global__ __launch_bounds__(1024) void retrieval_topk_select_kernel(const uint16_t* __restrict__ scores,
                                                                      int32_t* __restrict__ idx, int C, int sets,
                                                                      int n, int mpow2) {
    extern __shared__ uint32_t smem_u[];
    uint32_t* keys = smem_u;                 // [n]   candidate keys (chunks 1..C-1)
    uint32_t* sel = smem_u + n;              // [mpow2] the winners, then sorted in place
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_remaining, s_count;
    const int h = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint16_t* sc = scores + (int64_t)h * C;
    const int m = sets - 1;
    for (int i = tid; i < n; i += nt) {
        const int c = i + 1;
        keys[i] = (sortable_fp16(sc[c]) << 16) | (uint32_t)(65535 - (c & 0xFFFF));
    }
    for (int i = tid; i < mpow2; i += nt) sel[i] = 0u;                   // padding sorts last
    if (tid == 0) { s_prefix = 0u; s_remaining = m; s_count = 0; }
    __syncthreads();
    if (m > 0) {
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            for (int i = tid; i < n; i += nt) {
                const uint32_t k = keys[i];
                if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid == 0) {                                              // digit of the remaining-th largest key
                int rem = s_remaining, b = 255, cum = 0;
                for (; b > 0; --b) {
                    if (cum + hist[b] >= rem) break;
                    cum += hist[b];
                }
                s_remaining = rem - cum;
                s_prefix = prefix | ((uint32_t)b << shift);
            }
            __syncthreads();
        }
        const uint32_t T = s_prefix;                                     // the m-th largest key (keys are distinct)
        for (int i = tid; i < n; i += nt) {
            const uint32_t k = keys[i];
            if (k >= T) {
                const int p = atomicAdd(&s_count, 1);
                if (p < mpow2) sel[p] = k;
            }
        }
        __syncthreads();
        for (int kk = 2; kk <= mpow2; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < mpow2; i += nt) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const bool desc = ((i & kk) == 0);
                        const uint32_t a = sel[i], b = sel[ixj];
                        if ((a < b) == desc) {
                            sel[i] = b;
                            sel[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    int32_t* out = idx + (int64_t)h * sets;
    if (tid == 0) out[0] = 0;
    for (int i = tid; i < m; i += nt) out[1 + i] = 65535 - (int32_t)(sel[i] & 0xFFFFu);
}

// ---- gather ---------------------------------------------------------------------------------
// One 128-thread workgroup per (slot, head, K | V): a 2-KiB chunk, one 16-byte vector per thread.  In situ 24.9 us for the
// 134 MB (67 read + 67 written) of a cfg2 layer = 5.4 TB/s.  A coarser form — one workgroup per 8 slots, all chunk ids
// read first, 8 loads in flight per lane before the 8 stores — measured 26.0 us (round 3, bench.py roofline_stages on two
// boxes): the copy is bound by the mixed read / write stream, not by loads in flight; the simple form stays.
__global__ __launch_bounds__(128) void retrieval_gather_kernel(
    const h16* __restrict__ k_src, const h16* __restrict__ v_src, int64_t sst, int64_t ssh,
    const int32_t* __restrict__ idx, h16* __restrict__ k_dst, h16* __restrict__ v_dst, int64_t dst_t, int64_t dsh,
    int sets, int chunk, int D) {
    const int slot = blockIdx.x, h = blockIdx.y;
    const bool is_v = blockIdx.z != 0;
    const h16* src = (is_v ? v_src : k_src) + (int64_t)h * ssh;
    h16* dst = (is_v ? v_dst : k_dst) + (int64_t)h * dsh;
    const int c = idx[(int64_t)h * sets + slot];
    const int vec_per_row = D / 8;
    for (int e = threadIdx.x; e < chunk * vec_per_row; e += blockDim.x) {
        const int r = e / vec_per_row, dv = e - r * vec_per_row;
        const half8 x = load_half8(src + ((int64_t)c * chunk + r) * sst + 8 * dv);
        store_half8(dst + ((int64_t)slot * chunk + r) * dst_t + 8 * dv, x);
    }
}

// ---- row copies -----------------------------------------------------------------------------
// (blockIdx.z = 1: the second (src, dst) pair of the same geometry — K and V of one cache in one launch)
__global__ __launch_bounds__(256) void kv_copy_rows_kernel(const h16* __restrict__ src, int64_t ssl, int64_t sst,
                                                           int64_t ssh, h16* __restrict__ dst, int64_t dsl,
                                                           int64_t dst_t, int64_t dsh, int src_t0, int dst_t0, int n,
                                                           int H, int D, const h16* __restrict__ src2,
                                                           h16* __restrict__ dst2) {
    const int l = blockIdx.y / H, h = blockIdx.y % H;
    const h16* s = (blockIdx.z ? src2 : src) + (int64_t)l * ssl + (int64_t)h * ssh;
    h16* d = (blockIdx.z ? dst2 : dst) + (int64_t)l * dsl + (int64_t)h * dsh;
    const int vpr = D / 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * vpr; e += gridDim.x * blockDim.x) {
        const int r = e / vpr, dv = e - r * vpr;
        store_half8(d + (int64_t)(dst_t0 + r) * dst_t + 8 * dv, load_half8(s + (int64_t)(src_t0 + r) * sst + 8 * dv));
    }
}

// In-place downward shift (dst_t0 <= src_t0), overlap allowed: one workgroup per (layer, head)
// walks the rows in ascending blocks; inside a block every thread reads before anyone writes.
__global__ __launch_bounds__(256) void kv_shift_rows_kernel(h16* __restrict__ cache, int64_t sl, int64_t st,
                                                            int64_t sh, int src_t0, int dst_t0, int n, int H, int D,
                                                            h16* __restrict__ cache2) {
    const int l = blockIdx.x / H, h = blockIdx.x % H;
    h16* base = (blockIdx.y ? cache2 : cache) + (int64_t)l * sl + (int64_t)h * sh;
    const int vpr = D / 8;
    const int rows_per_blk = 256 / vpr;
    const int r_in = threadIdx.x / vpr, dv = threadIdx.x % vpr;
    for (int r0 = 0; r0 < n; r0 += rows_per_blk) {
        const int r = r0 + r_in;
        half8 x;
        const bool ok = (r < n) && (r_in < rows_per_blk);
        if (ok) x = load_half8(base + (int64_t)(src_t0 + r) * st + 8 * dv);
        __syncthreads();
        if (ok) store_half8(base + (int64_t)(dst_t0 + r) * st + 8 * dv, x);
        __syncthreads();
    }
}

// ---- C ABI ------------------------------------------------------------------------------------
extern "C" int tf_retrieval_score(const void* k, int64_t stride_t, int64_t stride_h, const void* q, void* scores,
                                  int C, int chunk, int H, int D, void* stream) {
    if (!k || !q || !scores || C < 1 || chunk < 1 || H < 1) return TF_EINVAL;
    if ((stride_t % 8) || (stride_h % 8)) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int gpb = 256 / (D / 8);
    int gx = (C + gpb - 1) / gpb;
#if TF_RSCORE_CAP_CEIL
    const int cap = (2048 + H - 1) / H;              // round 1-3 rule: 2 080 workgroups at H = 40, 2 050 at H = 5
#else
    // 8 resident workgroups per CU and NOT ONE MORE: with ceil(2048 / H) the 32 (H = 40) or 2 (H = 5) workgroups past
    // 2 048 start when the others retire and stream their share alone, latency-bound (0.59 of peak at H = 40 against
    // 0.81 at H = 32, profiles/r03_bench_13b_cfg4_world1.json); grid-stride beyond
    const int cap = H <= 2048 ? 2048 / H : 1;
#endif
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    if (D == 128)
        hipLaunchKernelGGL((retrieval_score_kernel<128>), dim3(gx, H), dim3(256), 0, st, (const h16*)k, stride_t,
                           stride_h, (const h16*)q, (h16*)scores, C, chunk);
    else if (D == 64)
        hipLaunchKernelGGL((retrieval_score_kernel<64>), dim3(gx, H), dim3(256), 0, st, (const h16*)k, stride_t,
                           stride_h, (const h16*)q, (h16*)scores, C, chunk);
    else
        return TF_EINVAL;
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_retrieval_topk(const void* scores, int32_t* idx, int C, int sets, int H, void* stream) {
    if (!scores || !idx || C < 2 || sets < 1 || sets > C || H < 1) return TF_EINVAL;
    if (C > 32768) return TF_ERANGE;
    // select-then-sort when candidates + winners fit in LDS together (always for the paper's shapes)
    int mpow2 = 2;
    while (mpow2 < sets - 1) mpow2 <<= 1;
    const size_t lds_sel = (size_t)((C - 1) + mpow2) * sizeof(uint32_t);
    if (lds_sel <= 150 * 1024 && mpow2 <= 8192) {
        if (lds_sel > 48 * 1024) {
            static bool raised_sel = false;
            if (!raised_sel) {
                hipError_t e = hipFuncSetAttribute((const void*)retrieval_topk_select_kernel,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                if (e != hipSuccess) return (int)e;
                raised_sel = true;
            }
        }
        hipLaunchKernelGGL(retrieval_topk_select_kernel, dim3(H), dim3(1024), lds_sel, (hipStream_t)stream,
                           (const uint16_t*)scores, idx, C, sets, C - 1, mpow2);
        TF_LAUNCH_CHECK();
        return TF_OK;
    }
    int npow2 = 2;
    while (npow2 < C - 1) npow2 <<= 1;
    const size_t lds = (size_t)npow2 * sizeof(uint32_t);
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute((const void*)retrieval_topk_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return (int)e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(retrieval_topk_kernel, dim3(H), dim3(1024), lds, (hipStream_t)stream,
                       (const uint16_t*)scores, idx, C, sets, npow2);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_retrieval_gather(const void* k_src, const void* v_src, int64_t src_stride_t, int64_t src_stride_h,
                                   const int32_t* idx, void* k_dst, void* v_dst, int64_t dst_stride_t,
                                   int64_t dst_stride_h, int sets, int chunk, int H, int D, void* stream) {
    if (!k_src || !v_src || !idx || !k_dst || !v_dst || sets < 1 || chunk < 1 || H < 1 || (D % 8)) return TF_EINVAL;
    if ((src_stride_t % 8) || (src_stride_h % 8) || (dst_stride_t % 8) || (dst_stride_h % 8)) return TF_EINVAL;
    hipLaunchKernelGGL(retrieval_gather_kernel, dim3(sets, H, 2), dim3(128), 0, (hipStream_t)stream,
                       (const h16*)k_src, (const h16*)v_src, src_stride_t, src_stride_h, idx, (h16*)k_dst, (h16*)v_dst,
                       dst_stride_t, dst_stride_h, sets, chunk, D);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// Compaction of accepted tree nodes (DistributedSimpleCache.gather_kv_incremental, models/cache.py:333-343):
// rows offset + idx[j] -> offset + j for j in [0, n), K and V, every layer and head.  idx must be strictly
// increasing (a root-to-leaf path of the Sequoia tree: node ids grow with depth), so idx[j] >= j and walking the
// rows in ascending batches with read-all / barrier / write-all is overlap-safe without the reference's clone().
__global__ __launch_bounds__(256) void kv_gather_rows_kernel(h16* __restrict__ kc, h16* __restrict__ vc, int64_t sl,
                                                             int64_t st, int64_t sh, int offset,
                                                             const int32_t* __restrict__ idx, int n, int H, int D) {
    const int l = blockIdx.x / H, h = blockIdx.x % H;
    h16* base = (blockIdx.y ? vc : kc) + (int64_t)l * sl + (int64_t)h * sh;
    const int vpr = D / 8;
    const int rows_per_blk = 256 / vpr;
    const int r_in = threadIdx.x / vpr, dv = threadIdx.x % vpr;
    for (int r0 = 0; r0 < n; r0 += rows_per_blk) {
        const int r = r0 + r_in;
        half8 x;
        const bool ok = (r < n) && (r_in < rows_per_blk);
        if (ok) x = load_half8(base + (int64_t)(offset + idx[r]) * st + 8 * dv);
        __syncthreads();
        if (ok) store_half8(base + (int64_t)(offset + r) * st + 8 * dv, x);
        __syncthreads();
    }
}

extern "C" int tf_kv_gather_rows(void* k_cache, void* v_cache, int64_t stride_l, int64_t stride_t, int64_t stride_h,
                                 int offset, const int32_t* idx, int n, int L, int H, int D, void* stream) {
    if (!k_cache || !v_cache || !idx || n < 1 || offset < 0 || L < 1 || H < 1 || D < 8 || (D % 8) || D > 2048)
        return TF_EINVAL;
    hipLaunchKernelGGL(kv_gather_rows_kernel, dim3(L * H, 2), dim3(256), 0, (hipStream_t)stream, (h16*)k_cache,
                       (h16*)v_cache, stride_l, stride_t, stride_h, offset, idx, n, H, D);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

static int kv_copy_rows_launch(const void* src, const void* src2, int64_t src_stride_l, int64_t src_stride_t,
                               int64_t src_stride_h, void* dst, void* dst2, int64_t dst_stride_l, int64_t dst_stride_t,
                               int64_t dst_stride_h, int src_t0, int dst_t0, int n, int L, int H, int D, void* stream) {
    if (n == 0) return TF_OK;
    if (!src || !dst || n < 0 || L < 1 || H < 1 || (D % 8) || src_t0 < 0 || dst_t0 < 0) return TF_EINVAL;
    if ((src_stride_t % 8) || (src_stride_h % 8) || (src_stride_l % 8) || (dst_stride_t % 8) || (dst_stride_h % 8) ||
        (dst_stride_l % 8))
        return TF_EINVAL;
    int gx = (n * (D / 8) + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(kv_copy_rows_kernel, dim3(gx, L * H, src2 ? 2 : 1), dim3(256), 0, (hipStream_t)stream,
                       (const h16*)src, src_stride_l, src_stride_t, src_stride_h, (h16*)dst, dst_stride_l, dst_stride_t,
                       dst_stride_h, src_t0, dst_t0, n, H, D, (const h16*)src2, (h16*)dst2);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_kv_copy_rows(const void* src, int64_t src_stride_l, int64_t src_stride_t, int64_t src_stride_h,
                               void* dst, int64_t dst_stride_l, int64_t dst_stride_t, int64_t dst_stride_h,
                               int src_t0, int dst_t0, int n, int L, int H, int D, void* stream) {
    return kv_copy_rows_launch(src, nullptr, src_stride_l, src_stride_t, src_stride_h, dst, nullptr, dst_stride_l,
                               dst_stride_t, dst_stride_h, src_t0, dst_t0, n, L, H, D, stream);
}

// K and V rows of one cache pair in ONE launch (same strides and row ranges for both): the decode step's tail copies
extern "C" int tf_kv_copy_rows_pair(const void* src_k, const void* src_v, int64_t src_stride_l, int64_t src_stride_t,
                                    int64_t src_stride_h, void* dst_k, void* dst_v, int64_t dst_stride_l,
                                    int64_t dst_stride_t, int64_t dst_stride_h, int src_t0, int dst_t0, int n, int L, int H,
                                    int D, void* stream) {
    if (n != 0 && (!src_v || !dst_v)) return TF_EINVAL;
    return kv_copy_rows_launch(src_k, src_v, src_stride_l, src_stride_t, src_stride_h, dst_k, dst_v, dst_stride_l,
                               dst_stride_t, dst_stride_h, src_t0, dst_t0, n, L, H, D, stream);
}

static int kv_shift_rows_launch(void* cache, void* cache2, int64_t stride_l, int64_t stride_t, int64_t stride_h,
                                int src_t0, int dst_t0, int n, int L, int H, int D, void* stream) {
    if (n == 0 || src_t0 == dst_t0) return TF_OK;
    if (!cache || n < 0 || L < 1 || H < 1 || (D % 8) || D > 2048 || dst_t0 > src_t0 || dst_t0 < 0) return TF_EINVAL;
    if ((stride_t % 8) || (stride_h % 8) || (stride_l % 8)) return TF_EINVAL;
    hipLaunchKernelGGL(kv_shift_rows_kernel, dim3(L * H, cache2 ? 2 : 1), dim3(256), 0, (hipStream_t)stream, (h16*)cache,
                       stride_l, stride_t, stride_h, src_t0, dst_t0, n, H, D, (h16*)cache2);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_kv_shift_rows(void* cache, int64_t stride_l, int64_t stride_t, int64_t stride_h, int src_t0,
                                int dst_t0, int n, int L, int H, int D, void* stream) {
    return kv_shift_rows_launch(cache, nullptr, stride_l, stride_t, stride_h, src_t0, dst_t0, n, L, H, D, stream);
}

extern "C" int tf_kv_shift_rows_pair(void* k_cache, void* v_cache, int64_t stride_l, int64_t stride_t, int64_t stride_h,
                                     int src_t0, int dst_t0, int n, int L, int H, int D, void* stream) {
    if (n != 0 && src_t0 != dst_t0 && !v_cache) return TF_EINVAL;
    return kv_shift_rows_launch(k_cache, v_cache, stride_l, stride_t, stride_h, src_t0, dst_t0, n, L, H, D, stream);
}
