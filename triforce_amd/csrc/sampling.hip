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
    // thread t owns the contiguous slice [t*seg, (t+1)*seg), seg a multiple of 4 (<= 32 for V <= 32768): the slice is
    // fetched ONCE with 16-byte loads and both passes run from registers (the first version re-read it element by
    // element: 28 us per call, pure load latency)
    constexpr int MAXSEG = 32;
    const int seg = (((V + SAMP_THREADS - 1) / SAMP_THREADS) + 3) & ~3;
    const int i0 = tid * seg, i1 = min(V, i0 + seg);
    float val[MAXSEG];
    const bool vec = seg <= MAXSEG && (V % 4) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0) &&
                     (!RESID || (reinterpret_cast<uintptr_t>(q) & 15) == 0);
    float local = 0.f;
    int my_last_nz = -1;
    if (vec) {
#pragma unroll
        for (int s4 = 0; s4 < MAXSEG; s4 += 4) {
            const int i = i0 + s4;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (s4 < seg && i < V) {                                  // V % 4 == 0: a float4 never straddles the end
                a = *reinterpret_cast<const f32x4*>(p + i);
                if (RESID) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(q + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] -= b[e]; a[e] = a[e] > 0.f ? a[e] : 0.f; }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) val[s4 + e] = a[e];
        }
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) {
            if (i0 + s < i1) {
                local += val[s];
                if (val[s] > 0.f) my_last_nz = i0 + s;
            }
        }
    } else {
        for (int i = i0; i < i1; ++i) {
            float v = p[i];
            if (RESID) { v -= q[i]; v = v > 0.f ? v : 0.f; }
            local += v;
            if (v > 0.f) my_last_nz = i;
        }
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
    if (vec) {
        bool found = false;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) {
            if (!found && i0 + s < i1) {
                c += val[s];
                if (val[s] > 0.f && c > target) { atomicMin(&sh->first_idx, i0 + s); found = true; }
            }
        }
    } else {
        for (int i = i0; i < i1; ++i) {
            float v = p[i];
            if (RESID) { v -= q[i]; v = v > 0.f ? v : 0.f; }
            c += v;
            if (v > 0.f && c > target) { atomicMin(&sh->first_idx, i); break; }
        }
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

// ------------------------------------------------------------------------------------------------
// Sequoia tree verification (utils/SpecTree_TP.py:147-199 `accept_step` + the loop of `verify`): walk the static
// tree from the root; at a node, test its children in order — child token x is accepted iff p[x] > r * q[x]
// (r a fresh uniform, q = softmax(draft_logits / T) with the already rejected sibling tokens forced to -inf);
// a rejection replaces p by the normalised residual relu(p - q).  The walk stops at the first node with no
// accepted child (or a leaf, or an accepted token 0 / 2), then the next token is drawn from the residual.
// The reference syncs the host at every child test (`if p[token] > r * q[token]`); here the whole walk is ONE
// single-workgroup kernel: the node's target row and draft row live in registers (thread t owns the contiguous
// vocabulary slice [32t, 32t+32)), every test is two block reductions, and the host reads one record.
//   out[0] = len(accept_list) (root included)   out[1] = next token   out[2] = terminal (0/1)
//   out[3] = uniforms consumed                  out[4 + j] = accept_list[j]
// ------------------------------------------------------------------------------------------------
#define TREE_EPT 32
#define TREE_MAX_PATH 60

__device__ __forceinline__ float block_sum_s(float v, float* sm, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SAMP_THREADS / 64; ++w) t += sm[w];
    return t;
}
__device__ __forceinline__ float block_max_s(float v, float* sm, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    float t = sm[0];
#pragma unroll
    for (int w = 1; w < SAMP_THREADS / 64; ++w) t = fmaxf(t, sm[w]);
    return t;
}

__global__ __launch_bounds__(SAMP_THREADS) void tree_accept_kernel(
    const float* __restrict__ p_rows, const float* __restrict__ draft_logits, const int64_t* __restrict__ tokens,
    const int32_t* __restrict__ succ_off, const int32_t* __restrict__ succ, const float* __restrict__ uniforms,
    int V, float temperature, int64_t* __restrict__ out) {
    __shared__ float sm[SAMP_THREADS / 64];
    __shared__ int s_flag, s_nan, s_first, s_lastnz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = tid * TREE_EPT;
    float p[TREE_EPT], dl[TREE_EPT], qv[TREE_EPT];
    int node = 0, nacc = 1, consumed = 0, terminal = 0;
    if (tid == 0) out[4] = 0;
    for (int depth = 0; depth < TREE_MAX_PATH; ++depth) {
#pragma unroll
        for (int s = 0; s < TREE_EPT; ++s) {
            const int i = i0 + s;
            p[s] = (i < V) ? p_rows[(int64_t)node * V + i] : 0.f;
            dl[s] = (i < V) ? draft_logits[(int64_t)node * V + i] : -INFINITY;
        }
        const int c0 = succ_off[node], c1 = succ_off[node + 1];
        int accepted = -1;
        int64_t acc_tok = -1;
        for (int c = c0; c < c1; ++c) {
            const int pos = succ[c];
            const int64_t tok = tokens[pos];
            float mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < TREE_EPT; ++s) {
                qv[s] = dl[s] / temperature;                       // softmax(draft_logits / T), SpecTree_TP.py:159
                mx = fmaxf(mx, qv[s]);
            }
            mx = block_max_s(mx, sm, tid);
            float zl = 0.f;
#pragma unroll
            for (int s = 0; s < TREE_EPT; ++s) {
                qv[s] = (i0 + s < V) ? expf(qv[s] - mx) : 0.f;
                zl += qv[s];
            }
            const float Z = block_sum_s(zl, sm, tid);
            const float r = uniforms[consumed];
            ++consumed;
            if (tid == 0) s_flag = 0;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < TREE_EPT; ++s) {
                qv[s] = qv[s] / Z;
                if ((int64_t)(i0 + s) == tok && p[s] > r * qv[s]) s_flag = 1;       // :162
            }
            __syncthreads();
            if (s_flag) {
                accepted = pos;
                acc_tok = tok;
                break;
            }
            float rl = 0.f;                                           // residual (offloading_seqouia.py:24-27)
#pragma unroll
            for (int s = 0; s < TREE_EPT; ++s) {
                const float df = p[s] - qv[s];
                p[s] = (df != df) ? df : fmaxf(df, 0.f);              // torch relu keeps NaN (fmaxf would drop it)
                rl += p[s];
                if ((int64_t)(i0 + s) == tok) dl[s] = -3.4028234663852886e38f;    // finfo(float32).min, :166
            }
            const float S = block_sum_s(rl, sm, tid);
#pragma unroll
            for (int s = 0; s < TREE_EPT; ++s) p[s] = p[s] / S;
        }
        if (accepted < 0) break;                                      // all children rejected, or a leaf
        if (tid == 0 && nacc < TREE_MAX_PATH) out[4 + nacc] = accepted;
        ++nacc;
        if (acc_tok == 0 || acc_tok == 2) {                           // :188-190
            terminal = 1;
            break;
        }
        node = accepted;
    }
    int64_t next = 0;
    if (!terminal) {
        if (tid == 0) { s_nan = 0; s_first = V; s_lastnz = -1; }
        __syncthreads();
        bool has_nan = false;
        float local = 0.f;
        int my_last = -1;
#pragma unroll
        for (int s = 0; s < TREE_EPT; ++s) {
            if (i0 + s < V) {
                has_nan |= (p[s] != p[s]);
                local += p[s];
                if (p[s] > 0.f) my_last = i0 + s;
            }
        }
        if (has_nan) s_nan = 1;
        __syncthreads();
        if (s_nan) {                                                  // :199-200
            terminal = 1;
        } else {                                                      // residual.multinomial -> inverse CDF
            float inc = local;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float nb = __shfl_up(inc, o, 64);
                if (lane >= o) inc += nb;
            }
            __syncthreads();
            if (lane == 63) sm[wave] = inc;
            __syncthreads();
            float prefix = 0.f, total = 0.f;
#pragma unroll
            for (int w = 0; w < SAMP_THREADS / 64; ++w) {
                const float t = sm[w];
                if (w < wave) prefix += t;
                total += t;
            }
            const float target = uniforms[consumed] * total;
            float cacc = prefix + (inc - local);
#pragma unroll
            for (int s = 0; s < TREE_EPT; ++s) {
                if (i0 + s < V) {
                    cacc += p[s];
                    if (p[s] > 0.f && cacc > target) { atomicMin(&s_first, i0 + s); break; }
                }
            }
            if (my_last >= 0) atomicMax(&s_lastnz, my_last);
            __syncthreads();
            int rtok = s_first;
            if (rtok >= V) rtok = s_lastnz >= 0 ? s_lastnz : 0;
            next = rtok;
            ++consumed;
        }
    }
    if (tid == 0) {
        out[0] = nacc;
        out[1] = next;
        out[2] = terminal;
        out[3] = consumed;
    }
}

extern "C" int tf_tree_accept(const float* p_rows, const float* draft_logits, const int64_t* tokens,
                              const int32_t* succ_off, const int32_t* succ, const float* uniforms, int V,
                              float temperature, int64_t* out, void* stream) {
    if (!p_rows || !draft_logits || !tokens || !succ_off || !succ || !uniforms || !out) return TF_EINVAL;
    if (V < 1 || !(temperature > 0.f)) return TF_EINVAL;
    if (V > SAMP_THREADS * TREE_EPT) return TF_ERANGE;
    hipLaunchKernelGGL(tree_accept_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p_rows, draft_logits,
                       tokens, succ_off, succ, uniforms, V, temperature, out);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// ------------------------------------------------------------------------------------------------
// Sampling WITHOUT replacement for the Sequoia tree growth (test/offloading_seqouia.py:29-39):
//   position = (rand.log() / softmax(logits / T)).topk(k).indices
// — the exponential race: the k largest log(u_i)/q_i (all negative; closest to zero wins) are a draw of k
// distinct tokens proportional to q.  One workgroup per row: softmax statistics by two block reductions, the keys
// stay in registers, then k rounds of a block-wide arg-max (ties -> lowest token id) each retiring its winner.
// The torch formulation (softmax, log, div, multi-block topk over 32 000 columns) is ~6 launches per tree level
// and its multi-block top-k does not survive hipGraph replay (second replay of a captured growth hangs); this
// kernel is what makes the whole-growth graph possible.  rand is fp16 like the reference's table; its log is
// rounded to fp16 exactly as `rand.log()` does.  k <= 16, V <= 32768.
// ------------------------------------------------------------------------------------------------
#define SWOR_EPT 32
__global__ __launch_bounds__(SAMP_THREADS) void sample_wor_kernel(const float* __restrict__ logits,
                                                                   const h16* __restrict__ rnd,
                                                                   int64_t* __restrict__ out, int V, int k,
                                                                   float temperature) {
    __shared__ float sm[SAMP_THREADS / 64];
    __shared__ float sm_v[SAMP_THREADS / 64];
    __shared__ int sm_i[SAMP_THREADS / 64];
    __shared__ int s_win;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (int64_t)row * V;
    const h16* rr = rnd + (int64_t)row * V;
    float key[SWOR_EPT];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < SWOR_EPT; ++s) {
        const int i = s * SAMP_THREADS + tid;                       // coalesced; ownership order is irrelevant here
        key[s] = (i < V) ? lr[i] / temperature : -INFINITY;
        mx = fmaxf(mx, key[s]);
    }
    mx = block_max_s(mx, sm, tid);
    float zl = 0.f;
#pragma unroll
    for (int s = 0; s < SWOR_EPT; ++s) {
        key[s] = (s * SAMP_THREADS + tid < V) ? expf(key[s] - mx) : 0.f;
        zl += key[s];
    }
    const float Z = block_sum_s(zl, sm, tid);
#pragma unroll
    for (int s = 0; s < SWOR_EPT; ++s) {
        const int i = s * SAMP_THREADS + tid;
        if (i < V) {
            const float lu = (float)(h16)logf((float)rr[i]);        // fp16 log like torch's half log
            key[s] = lu / (key[s] / Z);
            if (key[s] != key[s]) key[s] = -INFINITY;               // -inf / 0: never a winner
        } else {
            key[s] = -INFINITY;
        }
    }
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < SWOR_EPT; ++s) {
            const int i = s * SAMP_THREADS + tid;
            if (i < V && (key[s] > bv || (key[s] == bv && i < bi))) { bv = key[s]; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { sm_v[wave] = bv; sm_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float wv = sm_v[0];
            int wi = sm_i[0];
            for (int w = 1; w < SAMP_THREADS / 64; ++w)
                if (sm_v[w] > wv || (sm_v[w] == wv && sm_i[w] < wi)) { wv = sm_v[w]; wi = sm_i[w]; }
            if (wi >= V) wi = 0;                                    // fewer than k finite keys: degenerate row
            s_win = wi;
            out[(int64_t)row * k + r] = wi;
        }
        __syncthreads();
        const int win = s_win;
        if ((win % SAMP_THREADS) == tid) {
            const int ws = win / SAMP_THREADS;
#pragma unroll
            for (int s = 0; s < SWOR_EPT; ++s)
                if (s == ws) key[s] = -INFINITY;                    // retire the winner
        }
    }
}

extern "C" int tf_sample_without_replacement(const float* logits, const void* rand_f16, int64_t* out, int rows, int V,
                                             int k, float temperature, void* stream) {
    if (!logits || !rand_f16 || !out || rows < 1 || V < 1 || k < 1 || k > 16 || !(temperature > 0.f)) return TF_EINVAL;
    if (V > SAMP_THREADS * SWOR_EPT) return TF_ERANGE;
    hipLaunchKernelGGL(sample_wor_kernel, dim3(rows), dim3(SAMP_THREADS), 0, (hipStream_t)stream, logits,
                       (const h16*)rand_f16, out, V, k, temperature);
    TF_LAUNCH_CHECK();
    return TF_OK;
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
#define TOPP_BLOCK_BITS 12       // bits decided with block-wide reductions before the single-wave finish
#define TOPP_CAND_CAP 2048       // undecided entries the finishing wave can hold (32 per lane)

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
    __shared__ int sm_i[TOPP_THREADS / 64];
    __shared__ float sm_cand[TOPP_CAND_CAP];
    __shared__ unsigned sm_v;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (int64_t)row * V;
    float* pr = probs + (int64_t)row * V;
    // thread t owns the CONTIGUOUS indices [t*EPT, (t+1)*EPT): index order == (thread, slot) order
    const int i0 = tid * TOPP_EPT;
    // The row goes through LDS so that global traffic is coalesced while every thread still owns a CONTIGUOUS slice
    // (thread t, slot s) <-> index 32t + s at LDS word 33t + s (one word of padding per thread: conflict-free both
    // ways).  Reading the slice straight from global memory costs 32 loads per lane with a 128-byte lane stride, i.e.
    // 64 cache lines per instruction on ONE CU: 2/3 of the 66 us this kernel used to take.
    extern __shared__ float row_lds[];                  // TOPP_THREADS * (TOPP_EPT + 1) floats
    for (int i = tid; i < TOPP_THREADS * TOPP_EPT; i += TOPP_THREADS)
        row_lds[(i / TOPP_EPT) * (TOPP_EPT + 1) + (i % TOPP_EPT)] = (i < V) ? lr[i] : 0.f;
    __syncthreads();
    float e[TOPP_EPT];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < TOPP_EPT; ++s) {
        const int i = i0 + s;
        const float x = (i < V) ? row_lds[tid * (TOPP_EPT + 1) + s] / temperature : -INFINITY;   // sampling.py:56
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
    // v = largest bit pattern with G(v) > tau  (G(0) = Z > tau for top_p < 1; e >= 0 so patterns order like values).
    // e <= 1.0f = 0x3F800000, so bit 30 is never part of the answer.  The first TOPP_BLOCK_BITS bits are decided with
    // block-wide reductions (two barriers each, ~2 us per bit); by then the open interval (v, v + 2^(bit+1)) holds a few
    // hundred of the 32000 entries, which are compacted into LDS and finished by ONE wave with shuffle reductions only
    // (31 block-wide steps cost 67 us per call — a third of a 68M draft step).
    unsigned v = 0u;
    int bit = 29;
    for (; bit > 29 - TOPP_BLOCK_BITS; --bit) {
        const unsigned cand = v | (1u << bit);
        float g = 0.f;
#pragma unroll
        for (int s = 0; s < TOPP_EPT; ++s) g += (__float_as_uint(e[s]) > cand) ? e[s] : 0.f;
        g = block_sum_1024(g, sm, tid);
        if (g > tau) v = cand;
    }
    {
        const unsigned span = 1u << (bit + 1);               // remaining candidates: v | x, x < span
        float gh = 0.f;
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < TOPP_EPT; ++s) {
            const unsigned b = __float_as_uint(e[s]);
            if (b - v >= span && b > v) gh += e[s];          // above every remaining candidate
            else if (b > v) ++cnt;                           // undecided: v < b < v + span
        }
        const float Ghigh = block_sum_1024(gh, sm, tid);
        // exclusive prefix of cnt over the block (wave scan + wave totals)
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o, 64);
            if (lane >= o) inc += n;
        }
        __syncthreads();
        if (lane == 63) sm_i[wave] = inc;
        __syncthreads();
        int before = inc - cnt, total = 0;
#pragma unroll
        for (int w = 0; w < TOPP_THREADS / 64; ++w) {
            const int t = sm_i[w];
            if (w < wave) before += t;
            total += t;
        }
        if (total <= TOPP_CAND_CAP) {                        // block-uniform
#pragma unroll
            for (int s = 0; s < TOPP_EPT; ++s) {
                const unsigned b = __float_as_uint(e[s]);
                if (b > v && b - v < span) sm_cand[before++] = e[s];
            }
            __syncthreads();
            if (wave == 0) {
                float x[TOPP_CAND_CAP / 64];
#pragma unroll
                for (int j = 0; j < TOPP_CAND_CAP / 64; ++j) {
                    const int i = lane + 64 * j;
                    x[j] = (i < total) ? sm_cand[i] : 0.f;   // pattern 0 is never > cand
                }
                for (int b2 = bit; b2 >= 0; --b2) {
                    const unsigned cand = v | (1u << b2);
                    float g = 0.f;
#pragma unroll
                    for (int j = 0; j < TOPP_CAND_CAP / 64; ++j) g += (__float_as_uint(x[j]) > cand) ? x[j] : 0.f;
                    g = Ghigh + wave_sum(g);
                    if (g > tau) v = cand;
                }
                if (lane == 0) sm_v = v;
            }
            __syncthreads();
            v = sm_v;
        } else {                                             // degenerate rows (huge tie groups): stay block-wide
            for (; bit >= 0; --bit) {
                const unsigned cand = v | (1u << bit);
                float g = 0.f;
#pragma unroll
                for (int s = 0; s < TOPP_EPT; ++s) g += (__float_as_uint(e[s]) > cand) ? e[s] : 0.f;
                g = block_sum_1024(g, sm, tid);
                if (g > tau) v = cand;
            }
        }
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
    for (int s = 0; s < TOPP_EPT; ++s) row_lds[tid * (TOPP_EPT + 1) + s] = ((keepmask >> s) & 1u) ? e[s] / Zk : 0.f;
    __syncthreads();
    for (int i = tid; i < V; i += TOPP_THREADS) pr[i] = row_lds[(i / TOPP_EPT) * (TOPP_EPT + 1) + (i % TOPP_EPT)];
}

extern "C" int tf_topp_probs(const float* logits, float* probs, int rows, int V, float temperature, float top_p,
                             void* stream) {
    if (!logits || !probs || rows < 1 || V < 1 || !(temperature > 0.f) || !(top_p > 0.f)) return TF_EINVAL;
    if (V > TOPP_THREADS * TOPP_EPT) return TF_ERANGE;
    const size_t lds = (size_t)TOPP_THREADS * (TOPP_EPT + 1) * sizeof(float);          // 132 KiB: above the 64 KiB default
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)topp_probs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(topp_probs_kernel, dim3(rows), dim3(TOPP_THREADS), lds, (hipStream_t)stream, logits, probs, V,
                       0.f, temperature, top_p);
    TF_LAUNCH_CHECK();
    return TF_OK;
}
