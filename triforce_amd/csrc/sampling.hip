// Sampling and speculative accept/rollback for gfx950 — device-side replacement for the host
// loops of utils/decoding.py:97-134 (outer accept chain, residual resample, bonus token) and
// :190-220 (Middle_Spec accept test + follow-up sample), and for torch.multinomial
// (utils/sampling.py:63-66) with an explicit uniform.
//
// The reference syncs the host once per examined token (`if r < ...`, `.item()`); here the whole
// chain is one single-workgroup kernel: the accept flags of all drafted tokens are computed by one
// wavefront, `__ballot` + find-first-set gives the accepted prefix, and the correction token is
// drawn by a block-wide prefix sum over the (residual) distribution.  One D2H of 4 int64 follows.
#include "common.h"

#define SAMP_THREADS 1024

struct SampleShared {
    float wave_tot[SAMP_THREADS / 64];
    int first_idx;
    int last_nz;
    float total;
};

// value(i) = RESID ? max(p[i]-q[i],0) : p[i];  returns the first index whose inclusive cumulative
// sum exceeds u * total; all threads of the block get the result.
template <bool RESID>
__device__ int block_sample(const float* __restrict__ p, const float* __restrict__ q, int V, float u,
                            SampleShared* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = (V + SAMP_THREADS - 1) / SAMP_THREADS;
    const int i0 = tid * seg, i1 = min(V, i0 + seg);
    float local = 0.f;
    int my_last_nz = -1;
    for (int i = i0; i < i1; ++i) {
        float v = p[i];
        if (RESID) { v -= q[i]; v = v > 0.f ? v : 0.f; }
        local += v;
        if (v > 0.f) my_last_nz = i;
    }
    // inclusive scan of `local` inside the wave
    float inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (tid == 0) { sh->first_idx = V; sh->last_nz = -1; }
    if (lane == 63) sh->wave_tot[wave] = inc;
    __syncthreads();
    float wave_prefix = 0.f, total = 0.f;
#pragma unroll
    for (int w = 0; w < SAMP_THREADS / 64; ++w) {
        const float t = sh->wave_tot[w];
        if (w < wave) wave_prefix += t;
        total += t;
    }
    const float target = u * total;
    // Every thread walks its own segment from its exclusive prefix: the answer is the smallest index
    // with non-zero mass whose inclusive cumulative sum exceeds the target (robust to the small
    // disagreement between the tree-scanned prefixes and the sequential in-segment sums).
    float c = wave_prefix + (inc - local);
    for (int i = i0; i < i1; ++i) {
        float v = p[i];
        if (RESID) { v -= q[i]; v = v > 0.f ? v : 0.f; }
        c += v;
        if (v > 0.f && c > target) { atomicMin(&sh->first_idx, i); break; }
    }
    if (my_last_nz >= 0) atomicMax(&sh->last_nz, my_last_nz);
    __syncthreads();
    int r = sh->first_idx;
    if (r >= V) r = sh->last_nz >= 0 ? sh->last_nz : 0;   // u*total rounded past the end
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(SAMP_THREADS) void sample_kernel(const float* __restrict__ probs,
                                                              const float* __restrict__ u,
                                                              int64_t* __restrict__ token_out, int V) {
    __shared__ SampleShared sh;
    const int t = block_sample<false>(probs, nullptr, V, *u, &sh);
    if (threadIdx.x == 0) *token_out = (int64_t)t;
}

__global__ __launch_bounds__(SAMP_THREADS) void accept_chain_kernel(
    const float* __restrict__ p, const float* __restrict__ q, const int64_t* __restrict__ tokens,
    const float* __restrict__ uniforms, int g2, int V, int inclusive, int64_t eos, int64_t* __restrict__ out) {
    __shared__ SampleShared sh;
    __shared__ int s_count, s_reason;
    const int tid = threadIdx.x;
    if (tid < 64) {
        bool f = false, is_eos = false;
        if (tid < g2) {
            const int64_t t = tokens[tid];
            const float ratio = p[(int64_t)tid * V + t] / q[(int64_t)tid * V + t];
            const float m = (ratio != ratio) ? ratio : fminf(1.0f, ratio);     // torch.min keeps NaN
            const float r = uniforms[tid];
            f = inclusive ? (r <= m) : (r < m);
            is_eos = (t == eos);
        }
        const unsigned long long all = (g2 >= 64) ? ~0ull : ((1ull << g2) - 1ull);
        const unsigned long long acc = __ballot(f) & all;
        const unsigned long long rej = (~acc) & all;
        int count = rej ? (__ffsll((long long)rej) - 1) : g2;                    // accepted prefix length
        int reason = (count == g2) ? 1 : 0;
        const unsigned long long pre = (count >= 64) ? ~0ull : ((1ull << count) - 1ull);
        const unsigned long long em = __ballot(is_eos) & pre;
        if (em) {                                  // stop right after an accepted eos (decoding.py:108-110) ...
            const int e = __ffsll((long long)em);
            if (e < g2) { count = e; reason = 2; }  // ... unless it is the last token: then the bonus path runs (:127)
        }
        if (tid == 0) { s_count = count; s_reason = reason; }
    }
    __syncthreads();
    const int count = s_count, reason = s_reason;
    const int examined = (reason == 0) ? count + 1 : count;
    int64_t next;
    if (reason == 2) {
        next = eos;
    } else if (reason == 0) {
        next = block_sample<true>(p + (int64_t)count * V, q + (int64_t)count * V, V, uniforms[examined], &sh);
    } else {
        next = block_sample<false>(p + (int64_t)g2 * V, nullptr, V, uniforms[examined], &sh);
    }
    if (tid == 0) {
        out[0] = count;
        out[1] = next;
        out[2] = reason;
        out[3] = examined + (reason != 2 ? 1 : 0);
    }
}

__global__ __launch_bounds__(SAMP_THREADS) void middle_accept_kernel(
    const float* __restrict__ p, const float* __restrict__ q_d, int64_t* __restrict__ tokens,
    const float* __restrict__ uniforms, int n, int gamma, int V, int64_t* __restrict__ out) {
    __shared__ SampleShared sh;
    const int64_t d = tokens[n + 1];
    const float ratio = p[(int64_t)n * V + d] / q_d[d];
    const float m = (ratio != ratio) ? ratio : fminf(1.0f, ratio);
    const int acc = (uniforms[0] < m) ? 1 : 0;
    const int b = block_sample<false>(p + (int64_t)(n + acc) * V, nullptr, V, uniforms[1], &sh);
    if (threadIdx.x == 0) {
        out[0] = acc;
        out[1] = b;
        out[2] = d;
        if (n + 1 + acc <= gamma) tokens[n + 1 + acc] = b;
    }
}

extern "C" int tf_sample_inverse_cdf(const float* probs, const float* u, int64_t* token_out, int V, void* stream) {
    if (!probs || !u || !token_out || V < 1) return TF_EINVAL;
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, probs, u, token_out, V);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_accept_chain(const float* p, const float* q, const int64_t* tokens, const float* uniforms, int g2,
                               int V, int inclusive, int64_t eos_token_id, int64_t* out, void* stream) {
    if (!p || !q || !tokens || !uniforms || !out || g2 < 1 || g2 > 63 || V < 1) return TF_EINVAL;
    hipLaunchKernelGGL(accept_chain_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q, tokens,
                       uniforms, g2, V, inclusive, eos_token_id, out);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_middle_accept(const float* p, const float* q_d, int64_t* tokens, const float* uniforms, int n,
                                int gamma, int V, int64_t* out, void* stream) {
    if (!p || !q_d || !tokens || !uniforms || !out || n < 0 || n >= gamma || V < 1) return TF_EINVAL;
    hipLaunchKernelGGL(middle_accept_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q_d, tokens,
                       uniforms, n, gamma, V, out);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused temperature + top-p + softmax  (utils/sampling.py:5-27,43-60 `norm_logits`, top_k = -1).
//
// The reference sorts the whole vocabulary (torch.sort + softmax + cumsum + scatter + softmax: ~10 kernels,
// ~350 us for 8 x 32000 on MI355X).  The kept set is a threshold set, so no sort is needed:
//   e_i = exp(l_i/T - max), Z = sum e.  With G(u) = mass of the entries strictly greater than u, an entry of
//   value w is kept iff G(w) + (mass of equal entries with a lower index) <= top_p * Z  — exactly the
//   "drop rank r when the inclusive cumulative mass of rank r-1 exceeds top_p" rule with a stable descending
//   sort.  The smallest u with G(u) <= top_p*Z is found by a 31-step search over the fp32 bit pattern (G is
//   monotone), each step one block-wide deterministic reduction; ties at u are resolved by one index-ordered
//   block scan.  One workgroup per row, the row lives in registers (V <= 32768).
// ------------------------------------------------------------------------------------------------
#define TOPP_THREADS 1024
#define TOPP_EPT 32

__device__ __forceinline__ float block_sum_1024(float v, float* sm, int tid) {
    v = wave_sum(v);
    __syncthreads();                       // sm reuse across consecutive calls
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < TOPP_THREADS / 64; ++w) t += sm[w];
    return t;
}
__device__ __forceinline__ float block_max_1024(float v, float* sm, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    float t = sm[0];
#pragma unroll
    for (int w = 1; w < TOPP_THREADS / 64; ++w) t = fmaxf(t, sm[w]);
    return t;
}

__global__ __launch_bounds__(TOPP_THREADS) void topp_probs_kernel(const float* __restrict__ logits,
                                                                  float* __restrict__ probs, int V,
                                                                  float inv_temperature_is_div, float temperature,
                                                                  float top_p) {
    __shared__ float sm[TOPP_THREADS / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (int64_t)row * V;
    float* pr = probs + (int64_t)row * V;
    // thread t owns the CONTIGUOUS indices [t*EPT, (t+1)*EPT): index order == (thread, slot) order
    const int i0 = tid * TOPP_EPT;
    float e[TOPP_EPT];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < TOPP_EPT; ++s) {
        const int i = i0 + s;
        const float x = (i < V) ? lr[i] / temperature : -INFINITY;       // logits / temperature (sampling.py:56)
        e[s] = x;
        mx = fmaxf(mx, x);
    }
    mx = block_max_1024(mx, sm, tid);
    float zl = 0.f;
#pragma unroll
    for (int s = 0; s < TOPP_EPT; ++s) {
        e[s] = (i0 + s < V) ? expf(e[s] - mx) : 0.f;
        zl += e[s];
    }
    const float Z = block_sum_1024(zl, sm, tid);
    const float tau = top_p * Z;
    // v = largest bit pattern with G(v) > tau  (G(0) = Z > tau for top_p < 1; e >= 0 so patterns order like values)
    unsigned v = 0u;
    for (int bit = 30; bit >= 0; --bit) {
        const unsigned cand = v | (1u << bit);
        float g = 0.f;
#pragma unroll
        for (int s = 0; s < TOPP_EPT; ++s) g += (__float_as_uint(e[s]) > cand) ? e[s] : 0.f;
        g = block_sum_1024(g, sm, tid);
        if (g > tau) v = cand;
    }
    const unsigned u = v + 1u;                       // smallest pattern with G(u) <= tau
    float gl = 0.f, tl = 0.f;
#pragma unroll
    for (int s = 0; s < TOPP_EPT; ++s) {
        const unsigned b = __float_as_uint(e[s]);
        gl += (b > u) ? e[s] : 0.f;
        tl += (b == u) ? e[s] : 0.f;
    }
    const float G = block_sum_1024(gl, sm, tid);
    // exclusive prefix (index order) of the tie mass held by earlier threads
    float inc = tl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    __syncthreads();
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    float before = inc - tl;
    for (int w = 0; w < wave; ++w) before += sm[w];
    // keep: > u always; == u while G + (ties before) <= tau.  Always keep rank 0 (filter[...,0] = 0).
    float kept_sum = 0.f;
    unsigned keepmask = 0u;
    float run = before;
#pragma unroll
    for (int s = 0; s < TOPP_EPT; ++s) {
        const unsigned b = __float_as_uint(e[s]);
        bool keep = b > u;
        if (b == u) {
            keep = (G + run <= tau) || (G == 0.f && run == 0.f);
            run += e[s];
        }
        if (keep && e[s] > 0.f) { keepmask |= (1u << s); kept_sum += e[s]; }
    }
    const float Zk = block_sum_1024(kept_sum, sm, tid);
#pragma unroll
    for (int s = 0; s < TOPP_EPT; ++s) {
        const int i = i0 + s;
        if (i < V) pr[i] = ((keepmask >> s) & 1u) ? e[s] / Zk : 0.f;
    }
}

extern "C" int tf_topp_probs(const float* logits, float* probs, int rows, int V, float temperature, float top_p,
                             void* stream) {
    if (!logits || !probs || rows < 1 || V < 1 || !(temperature > 0.f) || !(top_p > 0.f)) return TF_EINVAL;
    if (V > TOPP_THREADS * TOPP_EPT) return TF_ERANGE;
    hipLaunchKernelGGL(topp_probs_kernel, dim3(rows), dim3(TOPP_THREADS), 0, (hipStream_t)stream, logits, probs, V,
                       0.f, temperature, top_p);
    TF_LAUNCH_CHECK();
    return TF_OK;
}
