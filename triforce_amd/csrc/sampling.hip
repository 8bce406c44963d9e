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

// Uniforms behind a DEVICE CURSOR (round 5): `uniforms` is then the base of the stream's buffer and u_k = uniforms[*cursor + k],
// so that a kernel captured inside a hipGraph (its arguments are frozen) consumes fresh numbers at every replay; the kernel
// that closes a decision advances the cursor itself by what the decision consumed (utils.sampling.UniformSource mirrors the
// position on the host).  cursor == NULL: `uniforms` points at u_0 (the eager form).
__device__ __forceinline__ const float* cur_uniforms(const float* uniforms, const int64_t* cursor, int64_t* at) {
    const int64_t c = cursor ? __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    *at = c;
    return uniforms + c;
}
// What a record kernel leaves for the NEXT chain of launches (token ids in the shared token buffer, the cursor) is stored
// write-through (agent scope) and drained BEFORE the record store: the host enqueues the next chain as soon as it sees the
// record — on the SAME stream (include/triforce_hip.h: a chain on another stream would be ordered by the record alone and
// would have to acquire what it reads; no caller does that).
__device__ __forceinline__ void st_agent_i64(int64_t* p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(SAMP_THREADS) void sample_kernel(const float* __restrict__ probs,
                                                              const float* __restrict__ u, const int64_t* cursor, int off,
                                                              int64_t* __restrict__ token_out, int V) {
    __shared__ SampleShared sh;
    int64_t at;
    const float* uu = cur_uniforms(u, cursor, &at) + off;
    const int t = block_sample<false>(probs, nullptr, V, *uu, &sh);
    if (threadIdx.x == 0) *token_out = (int64_t)t;
}

// What the outer accept can leave on the DEVICE for the launches that follow its record (round 5, tf_accept_chain_step), so that
// neither of them needs an eager set-up launch behind a record read (the two largest idle sources left in
// profiles/r05_gap_analysis_decode_steps.txt): the PASS TOKENS of the catch-up draft forward, written into the shared token
// buffer (reference decoding.py:137: [next, accepted..., resampled | bonus, PAD...]), and the positions / append slot / key
// count of the NEXT target verify for both captured lengths (the cache length after the roll-back, S + count + 1, :124).
struct ChainStep {
    int64_t* tok;                    // token buffer base (tok[0] = next, tok[1 + i] = drafted token i); NULL = not used
    int64_t pad;
    const int32_t* s_src;            // cache length BEFORE this verify (the replayed graph's append slot)
    int64_t* pos[2];
    int32_t* slot[2];
    int32_t* sk[2];
    int n_pos[2], qlen[2];
    int nsets;
};

__global__ __launch_bounds__(SAMP_THREADS) void accept_chain_kernel(
    const float* __restrict__ p, const float* __restrict__ q, const int64_t* __restrict__ tokens,
    const float* __restrict__ uniforms_base, int64_t* cursor, int g2, int V, int inclusive, int64_t eos,
    int64_t* __restrict__ out, ChainStep cs) {
    __shared__ SampleShared sh;
    __shared__ int s_count, s_reason;
    const int tid = threadIdx.x;
    int64_t at;
    const float* uniforms = cur_uniforms(uniforms_base, cursor, &at);
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
    if (cs.tok && tid < 64) {
        // pass tokens: tok[0 .. count] stay; tok[count + 1] = the resampled / bonus token (PAD after an accepted eos); PAD up to
        // g2 + 1 — and the next verify's scalars; all write-through, drained (below) before the record
        const int i = tid;
        if (i >= count + 1 && i <= g2 + 1) st_agent_i64(&cs.tok[i], (i == count + 1 && reason != 2) ? next : cs.pad);
        const int32_t s_new = *cs.s_src + count + 1;
        for (int k = 0; k < cs.nsets; ++k) {
            if (i < cs.n_pos[k]) st_agent_i64(&cs.pos[k][i], (int64_t)s_new + i);
            if (i == 0) {
                __hip_atomic_store(cs.slot[k], s_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cs.sk[k], s_new + cs.qlen[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        drain_stores();
    }
    __syncthreads();
    if (tid == 0) {
        const int consumed = examined + (reason != 2 ? 1 : 0);
        if (cursor) {
            st_agent_i64(cursor, at + consumed);
            drain_stores();
        }
        out[0] = count;
        out[1] = next;
        out[2] = reason;
        out[3] = consumed;
    }
}

__global__ __launch_bounds__(SAMP_THREADS) void middle_accept_kernel(
    const float* __restrict__ p, const float* __restrict__ q_d, int64_t* __restrict__ tokens,
    const float* __restrict__ uniforms_base, int64_t* cursor, int uoff, int n, int gamma, int V, int64_t* __restrict__ out,
    int write_limit) {
    __shared__ SampleShared sh;
    int64_t at;
    const float* uniforms = cur_uniforms(uniforms_base, cursor, &at) + uoff;
    const int64_t d = tokens[n + 1];
    const float ratio = p[(int64_t)n * V + d] / q_d[d];
    const float m = (ratio != ratio) ? ratio : fminf(1.0f, ratio);
    const int acc = (uniforms[0] < m) ? 1 : 0;
    const int b = block_sample<false>(p + (int64_t)(n + acc) * V, nullptr, V, uniforms[1], &sh);
    if (threadIdx.x == 0) {
        // first what the next chain reads on the device (write-through, drained), then the record the host waits for
        // (write_limit = gamma: the verify block's gamma + 1 entries; gamma + 1 when `tokens` is the engine's longer token buffer,
        //  so that after the last iteration it holds ALL of [next, t_1 .. t_g2] for the target verify to read in place)
        if (n + 1 + acc <= write_limit) st_agent_i64(&tokens[n + 1 + acc], (int64_t)b);
        if (cursor) st_agent_i64(cursor, at + uoff + 2);
        drain_stores();
        out[0] = acc;
        out[1] = b;
        out[2] = d;
        if (cursor) out[3] = at;             // where this decision's numbers began (the host checks its mirror)
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
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, probs, u, (const int64_t*)nullptr, 0,
                       token_out, V);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_accept_chain(const float* p, const float* q, const int64_t* tokens, const float* uniforms, int g2,
                               int V, int inclusive, int64_t eos_token_id, int64_t* out, void* stream) {
    if (!p || !q || !tokens || !uniforms || !out || g2 < 1 || g2 > 63 || V < 1) return TF_EINVAL;
    hipLaunchKernelGGL(accept_chain_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q, tokens,
                       uniforms, (int64_t*)nullptr, g2, V, inclusive, eos_token_id, out, ChainStep{});
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_middle_accept(const float* p, const float* q_d, int64_t* tokens, const float* uniforms, int n,
                                int gamma, int V, int64_t* out, void* stream) {
    if (!p || !q_d || !tokens || !uniforms || !out || n < 0 || n >= gamma || V < 1) return TF_EINVAL;
    hipLaunchKernelGGL(middle_accept_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q_d, tokens,
                       uniforms, (int64_t*)nullptr, 0, n, gamma, V, out, gamma);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// Tensor-parallel loop: the record every rank goes on with is RANK 0's (broadcast; the reference's sample_dist / r broadcast,
// utils/decoding.py:230-239,452-470).  tokens[n + 1], tokens[n + 2] as that record implies — accepted: the drafted token d stays
// at n + 1 and the follow-up b goes to n + 2 (when it fits), rejected: b at n + 1 — in one launch (torch: six tiny kernels).
__global__ void mid_record_tokens_kernel(const int64_t* __restrict__ rec, int64_t* __restrict__ tokens, int n, int limit) {
    const int64_t acc = rec[0], b = rec[1], d = rec[2];
    if (n + 1 <= limit) tokens[n + 1] = acc > 0 ? d : b;
    if (acc > 0 && n + 2 <= limit) tokens[n + 2] = b;
}
extern "C" int tf_mid_record_tokens(const int64_t* rec, int64_t* tokens, int tokens_len, int n, void* stream) {
    if (!rec || !tokens || n < 0 || tokens_len < n + 2) return TF_EINVAL;
    hipLaunchKernelGGL(mid_record_tokens_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rec, tokens, n, tokens_len - 1);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// The same three kernels with their uniforms behind a device cursor (see cur_uniforms): u_k = ubuf[*cursor + k].
//   tf_sample_inverse_cdf_cur : token = sample(probs) with u = ubuf[*cursor + off]; the cursor is left alone
//   tf_middle_accept_cur      : accept test with ubuf[*cursor + 1], follow-up sample with ubuf[*cursor + 2] (the draw of the
//                               drafted token used + 0); cursor += 3; out[3] = the cursor value the decision started from
//   tf_accept_chain_cur       : accept chain over ubuf[*cursor ...]; cursor += out[3] (the numbers it consumed)
// In all of them the device-visible results (token ids, cursor) are stored write-through and drained before the record.
extern "C" int tf_sample_inverse_cdf_cur(const float* probs, const float* ubuf, const int64_t* cursor, int off,
                                         int64_t* token_out, int V, void* stream) {
    if (!probs || !ubuf || !cursor || !token_out || V < 1 || off < 0) return TF_EINVAL;
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, probs, ubuf, cursor, off, token_out, V);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_middle_accept_cur(const float* p, const float* q_d, int64_t* tokens, int tokens_len, const float* ubuf,
                                    int64_t* cursor, int n, int gamma, int V, int64_t* out, void* stream) {
    if (!p || !q_d || !tokens || !ubuf || !cursor || !out || n < 0 || n >= gamma || V < 1 || tokens_len < gamma + 1) return TF_EINVAL;
    // a token buffer with room for it also receives the follow-up token of the LAST position (index gamma + 1)
    const int limit = tokens_len >= gamma + 2 ? gamma + 1 : gamma;
    hipLaunchKernelGGL(middle_accept_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q_d, tokens, ubuf, cursor,
                       1, n, gamma, V, out, limit);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_accept_chain_cur(const float* p, const float* q, const int64_t* tokens, const float* ubuf, int64_t* cursor,
                                   int g2, int V, int inclusive, int64_t eos_token_id, int64_t* out, void* stream) {
    if (!p || !q || !tokens || !ubuf || !cursor || !out || g2 < 1 || g2 > 63 || V < 1) return TF_EINVAL;
    hipLaunchKernelGGL(accept_chain_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q, tokens, ubuf, cursor,
                       g2, V, inclusive, eos_token_id, out, ChainStep{});
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// tf_accept_chain_cur that also prepares, on the device, what follows its record (see ChainStep): tok_buf[0 .. g2] must hold
// [next, t_1 .. t_g2] (the chain tests tok_buf[1 ..]); afterwards tok_buf[0 .. g2 + 1] are the pass tokens of the catch-up draft
// forward, and for each of the nsets (<= 2) captured verify lengths pos / slot / sk describe a verify at the rolled-back cache
// length *s_src + count + 1.  tok_buf needs g2 + 2 entries.
extern "C" int tf_accept_chain_step(const float* p, const float* q, int64_t* tok_buf, int tok_len, const float* ubuf,
                                    int64_t* cursor, int g2, int V, int inclusive, int64_t eos_token_id, int64_t pad,
                                    const int32_t* s_src, int nsets, int64_t* pos_a, int n_pos_a, int32_t* slot_a, int32_t* sk_a,
                                    int qlen_a, int64_t* pos_b, int n_pos_b, int32_t* slot_b, int32_t* sk_b, int qlen_b,
                                    int64_t* out, void* stream) {
    if (!p || !q || !tok_buf || !ubuf || !cursor || !out || !s_src || g2 < 1 || g2 > 62 || V < 1 || tok_len < g2 + 2) return TF_EINVAL;
    if (nsets < 0 || nsets > 2 || (nsets >= 1 && (!pos_a || !slot_a || !sk_a || n_pos_a < 0 || n_pos_a > 64)) ||
        (nsets == 2 && (!pos_b || !slot_b || !sk_b || n_pos_b < 0 || n_pos_b > 64)))
        return TF_EINVAL;
    ChainStep cs = {};
    cs.tok = tok_buf, cs.pad = pad, cs.s_src = s_src, cs.nsets = nsets;
    cs.pos[0] = pos_a, cs.slot[0] = slot_a, cs.sk[0] = sk_a, cs.n_pos[0] = n_pos_a, cs.qlen[0] = qlen_a;
    cs.pos[1] = pos_b, cs.slot[1] = slot_b, cs.sk[1] = sk_b, cs.n_pos[1] = n_pos_b, cs.qlen[1] = qlen_b;
    hipLaunchKernelGGL(accept_chain_kernel, dim3(1), dim3(SAMP_THREADS), 0, (hipStream_t)stream, p, q, tok_buf + 1, ubuf, cursor,
                       g2, V, inclusive, eos_token_id, out, cs);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused temperature + top-p + softmax  (utils/sampling.py:5-27,43-60 `norm_logits`, top_k = -1).
//
// The reference sorts the whole vocabulary (torch.sort + softmax + cumsum + scatter + softmax: ~10 kernels,
// ~350 us for 8 x 32000 on MI355X).  The kept set is a threshold set, so no sort is needed:
//   e_i = exp(l_i/T - max).  With M(i) = mass of the entries strictly greater than e_i plus the mass of equal entries
//   with a lower index, entry i is kept iff M(i) <= top_p * Z — exactly the "drop rank r when the inclusive cumulative
//   mass of rank r-1 exceeds top_p" rule with a stable descending sort.
// The boundary value is found by a 3-round radix select over the fp32 bit pattern of e (10 bits per round; patterns
// order like values for e >= 0, and e <= 1.0f = 0x3F800000 leaves 30 significant bits): every round builds a 1024-bin
// histogram of MASS in LDS, one wavefront scans it from the top and names the bin in which the cumulative mass
// crosses top_p * Z, and the next round looks only at the entries of that bin.  Masses are accumulated as 2^-40
// fixed-point integers with LDS integer atomics, so the sums are exact and independent of the order in which the
// atomics land (a float histogram would make the kept set depend on scheduling), and the normaliser of the kept set
// falls out of the select (mass above the boundary + kept ties) without another reduction.
// One workgroup per row, the row lives in registers (V <= 32768), global traffic is 16 bytes per lane both ways.
// (The first version decided one bit of the boundary per block-wide reduction — 12 reductions of 16 waves, then a
// single-wave finish — behind a 132 KiB LDS staging of the row: 50 us per call, every draft step and every verify.)
// ------------------------------------------------------------------------------------------------
#define TOPP_THREADS 1024
#define TOPP_SLABS 8                       // 16-byte slabs per thread: 8 * 4 * 1024 = 32768 entries
#define TOPP_EPT (4 * TOPP_SLABS)
#define TOPP_BINS 1024
#define TOPP_HB(b) ((b) + ((b) >> 4))      // one pad slot per 16 bins: the scan's 16-bins-per-lane reads spread over banks
#define TOPP_FIX_SHIFT 40                  // masses in units of 2^-40 (row total < 2^15 * 2^40)

struct ToppShared {
    unsigned long long hist[TOPP_BINS + TOPP_BINS / 16];
    unsigned cnt[TOPP_BINS + TOPP_BINS / 16];
    float red[TOPP_THREADS / 64];
    int red_i[TOPP_THREADS / 64];
    unsigned long long S, tau, nkeep, zk;
    unsigned long long zpart[TOPP_THREADS / 64];   // per-wave sums of all masses: Z does not come from the histogram
    unsigned ties;
    int digit;                                     // >= 0 boundary bin, -1 keep everything, -2 boundary below the candidate cut
};

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

// floor(e * 2^40) for 0 <= e <= 1 straight from the bit pattern (monotone in e; 0 below 2^-40)
__device__ __forceinline__ unsigned long long topp_fix(float e) {
    const unsigned b = __float_as_uint(e);
    const int ex = (int)(b >> 23);
    const unsigned long long man = (unsigned long long)((b & 0x7FFFFFu) | 0x800000u);
    const int sh = ex - (127 + 23 - TOPP_FIX_SHIFT);
    if (ex == 0 || sh <= -24) return 0ull;
    return sh >= 0 ? (man << sh) : (man >> (-sh));
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// hist[digit] += m for the active lanes of a wave (wave-uniform call).  When more than 8 lanes are active and all of
// them name the same bin (degenerate rows: huge tie groups) the wave adds ONE pre-summed value instead of serialising up
// to 64 same-address atomics.
__device__ __forceinline__ void topp_hist_add(ToppShared* sh, bool active, int digit, unsigned long long m, bool count) {
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const int first = __ffsll((long long)act) - 1;
    const int dref = __shfl(digit, first, 64);
    if (__popcll(act) > 8 && __all(!active || digit == dref)) {
        const unsigned long long tot = wave_sum_u64(active ? m : 0ull);
        if ((int)(threadIdx.x & 63) == first) {
            atomicAdd(&sh->hist[TOPP_HB(dref)], tot);
            if (count) atomicAdd(&sh->cnt[TOPP_HB(dref)], (unsigned)__popcll(act));
        }
    } else if (active) {
        atomicAdd(&sh->hist[TOPP_HB(digit)], m);
        if (count) atomicAdd(&sh->cnt[TOPP_HB(digit)], 1u);
    }
}

// One wavefront scans the histogram from the top: lane l owns bins [16l, 16l+16).  Names the bin d with
//   S(d) <= tau < S(d) + mass(d),   S(d) = base + mass of the bins above d
// (sh->digit, sh->S = S(d)); no such bin (tau >= total: top_p >= 1) -> digit = -1.  The first round also fixes
// tau = floor(top_p * Z); the last round resolves the ties at the boundary pattern: the j-th tie (index order) is
// kept iff S + j * m <= tau.  Clears the bins it read for the next round.
__device__ __forceinline__ void topp_scan(ToppShared* sh, int lane, bool first_round, bool last_round, float top_p) {
    unsigned long long h[16];
    unsigned long long tot = 0ull;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        h[k] = sh->hist[TOPP_HB(16 * lane + k)];
        sh->hist[TOPP_HB(16 * lane + k)] = 0ull;
        tot += h[k];
    }
    unsigned long long inc = tot;                                   // inclusive suffix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned lo = (unsigned)__shfl_down((int)(unsigned)inc, o, 64);
        const unsigned hi = (unsigned)__shfl_down((int)(unsigned)(inc >> 32), o, 64);
        if (lane + o < 64) inc += ((unsigned long long)hi << 32) | lo;
    }
    unsigned long long base, tau, Z = 0ull;
    if (first_round) {
#pragma unroll
        for (int w = 0; w < TOPP_THREADS / 64; ++w) Z += sh->zpart[w];          // exact integers: any order
        const double t = (double)top_p * (double)Z;
        tau = (t >= 18446744073709549568.0) ? ~0ull : __double2ull_rd(t);
        base = 0ull;
        if (lane == 0) {
            sh->tau = tau;
            sh->zk = Z;
        }
    } else {
        base = sh->S;
        tau = sh->tau;
    }
    unsigned long long S = base + (inc - tot);
    int found = -1;
    unsigned long long Sf = 0ull, hf = 0ull;
#pragma unroll
    for (int k = 15; k >= 0; --k) {
        if (found < 0 && h[k] != 0ull && S <= tau && tau - S < h[k]) {
            found = k;
            Sf = S;
            hf = h[k];
        }
        S += h[k];
    }
    const unsigned long long hit = __ballot(found >= 0);
    if (!hit) {
        // nothing crossed: tau >= Z (top_p >= 1: keep everything) — or, first round only, the crossing lies among the
        // entries the candidate cut left out of the histogram (cannot happen for the cut topp_probs_kernel derives)
        if (lane == 0) sh->digit = (first_round && tau < Z) ? -2 : -1;
    } else if (found >= 0) {
        sh->digit = 16 * lane + found;
        sh->S = Sf;
        if (last_round) {
            const unsigned T = sh->cnt[TOPP_HB(16 * lane + found)];
            const unsigned long long m = hf / (unsigned long long)T;      // all ties share one pattern, hence one mass
            unsigned long long nk = (tau - Sf) / m + 1ull;
            if (nk > (unsigned long long)T) nk = T;
            sh->ties = T;
            sh->nkeep = nk;
            sh->zk = Sf + nk * m;
        }
    }
    if (last_round) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sh->cnt[TOPP_HB(16 * lane + k)] = 0u;
    }
}

__global__ __launch_bounds__(TOPP_THREADS) void topp_probs_kernel(const float* __restrict__ logits,
                                                                  float* __restrict__ probs, int V,
                                                                  float temperature, float top_p) {
    // dynamic LDS only (a static block in front would push the 16-byte row accesses off their alignment):
    // [ row: nslab * 4096 floats | ToppShared ]
    extern __shared__ __attribute__((aligned(16))) unsigned char topp_smem[];
    const int nslab = (V + 4095) >> 12;
    float* rowe = reinterpret_cast<float*>(topp_smem);
    ToppShared* sh = reinterpret_cast<ToppShared*>(topp_smem + (size_t)nslab * 4096 * sizeof(float));
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (int64_t)row * V;
    float* pr = probs + (int64_t)row * V;
    // thread t owns entries 4096 s + 4 t + j (s < nslab slabs, j < 4): one 16-byte access per slab, global and LDS
    // alike; index order is (s, t, j) lexicographic — only the tie ranking at the boundary ever needs it
    const bool vec = (V % 4) == 0 && ((reinterpret_cast<uintptr_t>(lr) | reinterpret_cast<uintptr_t>(pr)) & 15) == 0;
    f32x4* mine = reinterpret_cast<f32x4*>(rowe) + tid;                  // slab s at mine[1024 * s]

    // ---- pass 0: all loads in flight, then x = l / T (sampling.py:56) one slab at a time, row -> LDS, row max ----
    f32x4 a[TOPP_SLABS];
#pragma unroll
    for (int s = 0; s < TOPP_SLABS; ++s) {
        const int i0 = 4096 * s + 4 * tid;
        a[s] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // entries past V: exp(-inf) = 0 mass
        if (s < nslab) {
            if (vec) {
                if (i0 < V) a[s] = *reinterpret_cast<const f32x4*>(lr + i0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < V) a[s][j] = lr[i0 + j];
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < TOPP_SLABS; ++s) {
        if (s < nslab) {
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = a[s][j] / temperature;
                mx = fmaxf(mx, x[j]);
            }
            mine[1024 * s] = x;
        }
        __builtin_amdgcn_sched_barrier(0);                                // interleaving 32 IEEE divisions spills
    }
    sh->hist[TOPP_HB(tid)] = 0ull;
    sh->cnt[TOPP_HB(tid)] = 0u;
    mx = block_max_1024(mx, sh->red, tid);

    // ---- round 1: e = exp(x - max) back to LDS; mass histogram over bits 29..20 of the CANDIDATES ----
    // Z is summed in registers from every entry; the histogram (LDS atomics) only takes entries >= cut, with
    // cut = (1 - top_p) / (2 V) rounded down to a bin edge: the entries below it weigh less than (1 - top_p) / 2 of the
    // largest entry alone, so the crossing of top_p * Z always lies among the candidates — in a peaked row (what a
    // language model emits) that is a few hundred of 32 000 entries, and rounds 2 and 3 then visit only those (a bit
    // per entry in `cand`).  Same Z, tau, boundary and kept set as with the full histogram.
    unsigned cutpat = 0u;
    if (top_p < 1.0f) cutpat = __float_as_uint((1.0f - top_p) / (2.0f * (float)V)) & ~((1u << 20) - 1u);
    unsigned cand = 0u;
    unsigned long long zacc = 0ull;
#pragma unroll 1
    for (int s = 0; s < nslab; ++s) {
        f32x4 x = mine[1024 * s];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = expf(x[j] - mx);
        mine[1024 * s] = x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long m = topp_fix(x[j]);
            zacc += m;
            const bool c = m != 0ull && __float_as_uint(x[j]) >= cutpat;
            cand |= (c ? 1u : 0u) << (4 * s + j);
            topp_hist_add(sh, c, (int)(__float_as_uint(x[j]) >> 20), m, false);
        }
    }
    {
        const unsigned long long zw = wave_sum_u64(zacc);
        if (lane == 0) sh->zpart[wave] = zw;
    }
    __syncthreads();
    if (wave == 0) topp_scan(sh, lane, true, false, top_p);
    __syncthreads();
    if (sh->digit == -2) {                                       // block-uniform; unreachable for the cut above (kept as
        __syncthreads();                                         // the exact fallback): histogram of ALL entries
#pragma unroll 1
        for (int s = 0; s < nslab; ++s) {
            const f32x4 x = mine[1024 * s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long m = topp_fix(x[j]);
                cand |= (m != 0ull ? 1u : 0u) << (4 * s + j);
                topp_hist_add(sh, m != 0ull, (int)(__float_as_uint(x[j]) >> 20), m, false);
            }
        }
        __syncthreads();
        if (wave == 0) topp_scan(sh, lane, true, false, top_p);
        __syncthreads();
    }
    const int d1 = sh->digit;
    unsigned ustar = 0u, ties = 0u;
    unsigned long long nkeep = 0ull;
    // a wave whose lanes hold many candidates walks all its entries (unrolled, 16-byte LDS reads); a wave with few walks
    // only the set bits of `cand`
    const bool dense = wave_sum_u64((unsigned long long)__popc(cand)) > 4ull * 64ull;
    if (d1 >= 0) {                                               // block-uniform; -1: top_p >= 1 keeps everything
        // ---- round 2: bits 19..10 of the entries inside the boundary bin ----
        if (dense) {
#pragma unroll 1
            for (int s = 0; s < nslab; ++s) {
                const f32x4 x = mine[1024 * s];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned b = __float_as_uint(x[j]);
                    const bool in = (int)(b >> 20) == d1;
                    topp_hist_add(sh, in, (int)((b >> 10) & 1023u), in ? topp_fix(x[j]) : 0ull, false);
                }
            }
        } else {
            unsigned left = cand;
            while (__ballot(left != 0u)) {
                const bool has = left != 0u;
                const int bit = has ? __ffs((int)left) - 1 : 0;
                left &= left - 1u;
                const float xv = rowe[4096 * (bit >> 2) + 4 * tid + (bit & 3)];
                const unsigned b = __float_as_uint(xv);
                const bool in = has && (int)(b >> 20) == d1;
                topp_hist_add(sh, in, (int)((b >> 10) & 1023u), in ? topp_fix(xv) : 0ull, false);
            }
        }
        __syncthreads();
        if (wave == 0) topp_scan(sh, lane, false, false, top_p);
        __syncthreads();
        const unsigned pre = ((unsigned)d1 << 10) | (unsigned)sh->digit;
        // ---- round 3: bits 9..0, with tie counts ----
        if (dense) {
#pragma unroll 1
            for (int s = 0; s < nslab; ++s) {
                const f32x4 x = mine[1024 * s];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned b = __float_as_uint(x[j]);
                    const bool in = (b >> 10) == pre;
                    topp_hist_add(sh, in, (int)(b & 1023u), in ? topp_fix(x[j]) : 0ull, true);
                }
            }
        } else {
            unsigned left = cand;
            while (__ballot(left != 0u)) {
                const bool has = left != 0u;
                const int bit = has ? __ffs((int)left) - 1 : 0;
                left &= left - 1u;
                const float xv = rowe[4096 * (bit >> 2) + 4 * tid + (bit & 3)];
                const unsigned b = __float_as_uint(xv);
                const bool in = has && (b >> 10) == pre;
                topp_hist_add(sh, in, (int)(b & 1023u), in ? topp_fix(xv) : 0ull, true);
            }
        }
        __syncthreads();
        if (wave == 0) topp_scan(sh, lane, false, true, top_p);
        __syncthreads();
        ustar = (pre << 10) | (unsigned)sh->digit;
        ties = sh->ties;
        nkeep = sh->nkeep;
    }
    const float Zk = (float)((double)sh->zk * (1.0 / 1099511627776.0));
    // kept: everything above the boundary pattern, and the first nkeep entries equal to it in INDEX order.  Usually
    // nkeep == ties (one entry carries the boundary value); otherwise the ties are ranked slab by slab with a
    // block-wide exclusive scan (block-uniform branch).
    const bool rank_ties = d1 >= 0 && nkeep < (unsigned long long)ties;
    const unsigned ulow = d1 < 0 ? 0u : (rank_ties ? ustar + 1u : ustar);      // patterns >= ulow stay unconditionally
    int basecnt = 0;
#pragma unroll 1
    for (int s = 0; s < nslab; ++s) {
        const f32x4 x = mine[1024 * s];
        unsigned tiekeep = 0u;
        if (rank_ties) {
            int c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) c += (__float_as_uint(x[j]) == ustar) ? 1 : 0;
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_up(inc, o, 64);
                if (lane >= o) inc += n;
            }
            __syncthreads();
            if (lane == 63) sh->red_i[wave] = inc;
            __syncthreads();
            int rank = basecnt + inc - c, total = 0;
#pragma unroll
            for (int w = 0; w < TOPP_THREADS / 64; ++w) {
                const int t = sh->red_i[w];
                if (w < wave) rank += t;
                total += t;
            }
            basecnt += total;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (__float_as_uint(x[j]) == ustar) {
                    if ((unsigned long long)rank < nkeep) tiekeep |= 1u << j;
                    ++rank;
                }
        }
        const int i0 = 4096 * s + 4 * tid;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool keep = __float_as_uint(x[j]) >= ulow || ((tiekeep >> j) & 1u);
            o[j] = keep ? x[j] / Zk : 0.f;
        }
        if (vec) {
            if (i0 < V) *reinterpret_cast<f32x4*>(pr + i0) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < V) pr[i0 + j] = o[j];
        }
    }
}

static size_t topp_lds_bytes(int V) { return (size_t)((V + 4095) >> 12) * 4096 * sizeof(float) + sizeof(ToppShared); }

extern "C" int tf_topp_probs(const float* logits, float* probs, int rows, int V, float temperature, float top_p,
                             void* stream) {
    if (!logits || !probs || rows < 1 || V < 1 || !(temperature > 0.f) || !(top_p > 0.f)) return TF_EINVAL;
    if (V > TOPP_THREADS * TOPP_EPT) return TF_ERANGE;
    static bool attr_set = false;                     // up to 141 KiB of dynamic LDS: above the 64 KiB default limit
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)topp_probs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)topp_lds_bytes(TOPP_THREADS * TOPP_EPT));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(topp_probs_kernel, dim3(rows), dim3(TOPP_THREADS), topp_lds_bytes(V), (hipStream_t)stream, logits,
                       probs, V, temperature, top_p);
    TF_LAUNCH_CHECK();
    return TF_OK;
}
