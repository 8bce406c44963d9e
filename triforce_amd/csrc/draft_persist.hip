// tf_draft_forward_68m_persist: one decode call of the Llama-68M draft model as ONE LAUNCH (round 6).
// (models/modeling_llama_68m.py:129-190 forward + utils/graph_infer.py:52-57 draft_run + utils/sampling.py:5-27,43-60.)
//
// tf_draft_forward_68m (csrc/draft.hip) issues the forward as 13 launches; 11 of them sit at the ~4.8 us launch floor and the
// last two (lm_head, one-row top-p) run at 1.6 TB/s and on ONE compute unit: 113.8 us for 87 MB = 0.095 of the HBM roofline
// (BENCH_r05).  Here the whole forward is one grid of 256 co-resident workgroups (one per CU, 8 waves); the stages of the
// forward are ROLES of fixed workgroup ranges and meet through per-edge arrival counters:
//
//     role            workgroups   work item                          waits for            arrives at
//     q|k|v (+RoPE)   0..143       one 16-row panel, 8 K-splits       down[l-1] (l > 0)    qkv[l]
//     attention       192..203     one head                           qkv[l]               attn[l]
//     o_proj          204..251     one panel, 8 K-splits              attn[l]              o[l]
//     gate|up         0..191       one panel pair, 4 + 4 K-splits     o[l]                 gu[l]
//     down_proj       204..251     one panel, 8 K-splits              gu[l]                down[l]
//     lm_head         0..255       one panel per WAVE, 4 K-ranges     down[L-1]            lm
//     top-p (a)       0..255       own 128 entries: e, mass, bins     lm                   t2
//     top-p (b)       0..255       whole row, redundantly             t2                   -
//
// An edge is a monotone counter (sharded 8 ways above 64 producers): a producer stores its outputs WRITE-THROUGH (agent-scope
// relaxed atomics = `global_store ... sc1`), drains them (s_waitcnt vmcnt(0)) and adds 1; a consumer polls until the count
// reaches (epoch + 1) x producers and then reads the data with agent-scope loads (`global_load ... sc1`: they bypass this CU's
// L1, the only cache another CU's stores do not reach — MI355X_MICROARCH.md, inter-workgroup visibility).  No fences, no
// cooperative launch.  `epoch` counts the launches on this control block (read at entry, bumped by workgroup 0 once every
// workgroup has passed the last edge), so nothing is re-zeroed between launches or graph replays.  Every role loads its weight
// tiles into registers BEFORE it waits, so the weight stream of a stage hides behind the hand-off of the one before.
//
// Arithmetic: every stage computes exactly what its launch in the 13-launch chain computes — the same packed weight tiles
// through the same v_mfma_f32_16x16x32_f16 sequence per K-range, the partial sums added in the same order, the same fp16
// rounding points, the same exact-integer top-p select — so logits, probabilities and the K / V rows written are BIT-IDENTICAL
// to tf_draft_forward_68m (tests/test_gpu_draft_persist.py).  Shapes: the 68M draft (hidden 768, 12 heads of 64, inter 3072,
// vocab <= 32768), 1..16 rows, <= 384 keys; anything else returns TF_EINVAL and the caller keeps the chain.
//
// Failure: every wait is bounded by wall-clock time (tf_draft_persist_tune key 0, default 2 s; frozen into captured graphs).
// A time-out (a workgroup that never became resident, a lost arrival) sets the sticky error word (+ optional pinned host
// mirror), every workgroup that sees it poisons its share of the outputs with NaN and leaves; later launches on the same
// control block return at once with NaN outputs until tf_draft_persist_reset.
#include "select_common.h"

#define DP_THREADS 512
#define DP_GRID 256
#define DP_MAX_LAYERS 2
#define DP_EDGES (5 * DP_MAX_LAYERS + 2)
#define DP_MAX_KEYS 384
#define DP_KPAD 8
#define DP_WALL_HZ 100000000ull
#define DP_STAMPS 48                         // wall-clock stamps per workgroup (instrumented launches)
#ifndef DP_LM_EARLY
#define DP_LM_EARLY 0                       // lm_head k-chunks (KiB per wave) requested behind the workgroup's first role; the
#endif                                       // rest (24 - this) in front of the last wait.  24 spills: the down_proj role holds 96 registers of its own

namespace {
constexpr int HID = 768, NH = 12, HD = 64, INTER = 3072, KC = HID / 32, KCI = INTER / 32;
constexpr int QKV_PANELS = 3 * HID / 16, O_PANELS = HID / 16, GU_PANELS = INTER / 16, SS_PARTS = HID / 16;
// role ranges
constexpr int QKV_LO = 0, QKV_HI = QKV_PANELS;          // 144
constexpr int GU_LO = 0, GU_HI = GU_PANELS;             // 192
constexpr int ATT_LO = 192, ATT_HI = ATT_LO + NH;       // 204
constexpr int OD_LO = 204, OD_HI = OD_LO + O_PANELS;    // 252
static_assert(OD_HI <= DP_GRID && GU_HI <= ATT_LO, "role ranges");
enum { E_QKV = 0, E_ATT = 1, E_O = 2, E_GU = 3, E_DOWN = 4 };

struct DpCtl {                               // device memory, zero-filled once (tf_draft_persist_reset)
    unsigned epoch, error, pad0[2];
    u64 mirror;                              // 0 or a pinned host word that also receives the error code
    unsigned pad1[10];
    unsigned cnt[DP_EDGES][9][16];           // arrival counters: edge, 8 shards + the shards' top counter, one 64-byte line each
    unsigned flag[DP_EDGES][8][16];          // READY flags: 8 copies per edge (workgroup w polls copy w & 7), = epoch + 1
};

struct DpParams {
    const h16* embed;
    const h16* ln1[DP_MAX_LAYERS];
    const half8* wqkv[DP_MAX_LAYERS];
    const half8* wo[DP_MAX_LAYERS];
    const h16* ln2[DP_MAX_LAYERS];
    const half8* wgate[DP_MAX_LAYERS];
    const half8* wup[DP_MAX_LAYERS];
    const half8* wdown[DP_MAX_LAYERS];
    const h16* norm;
    const half8* lm_head;
    const h16* cosb;
    const h16* sinb;
    h16* kc[DP_MAX_LAYERS];
    h16* vc[DP_MAX_LAYERS];
    int64_t stride_t, stride_h;
    const int64_t* ids;
    float* logits;
    float* probs;
    h16* x;                                  // [16][HID]   residual stream
    h16* q;                                  // [16][NH][HD]
    h16* a;                                  // [16][HID]   attention output
    h16* act;                                // [16][INTER] SwiGLU output
    float* ss;                               // [48][32]    per-panel sums of squares (norm hand-off)
    float* wgmax;                            // [256]       top-p: per-workgroup row maximum
    u64* cand;                               // [256][128]  top-p: every workgroup's candidates (pattern << 32) | index
    u64* zslice;                             // [256]       top-p: every workgroup's exact mass sum
    unsigned* ccount;                        // [256]       top-p: every workgroup's candidate count
    DpCtl* ctl;
    u64* stamps;                             // 0 or [DP_GRID][DP_STAMPS]
    u64 timeout_ticks;
    int n, slot0, kv_len, layers, vocab, skip_edge;
    float eps, scale, temperature, top_p;
};

__device__ __forceinline__ half8 ld_act8(const h16* p) {             // 16 bytes of an activation another workgroup wrote
    U64x2 v;
    v.lo = ld8(p);
    v.hi = ld8(p + 4);
    return __builtin_bit_cast(half8, v);
}
__device__ __forceinline__ void st_half4(h16* p, half4 v) { st8(p, __builtin_bit_cast(u64, v)); }
__device__ __forceinline__ half8 ld_w(const half8* p) { return __builtin_nontemporal_load(p); }   // weights: read once

// h = w_ln * fp16(x * inv)  (csrc/gemv.hip sg_normalise; modeling_llama.py:141-143)
__device__ __forceinline__ half8 dp_normalise(half8 xv, half8 wv, float inv) {
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = hmul_rn(wv[e], (h16)((float)xv[e] * inv));
    return o;
}

struct DpCtx {                               // per-thread view of the launch
    const DpParams* P;
    DpCtl* ctl;
    unsigned epoch;
    int w, tid, lane, wave, li, g;
    int nstamp;
    int* sh_fail;                            // LDS word: a wait of this workgroup failed
};

__device__ __forceinline__ void dp_stamp(DpCtx& c) {
    if (c.P->stamps && c.tid == 0 && c.nstamp < DP_STAMPS) c.P->stamps[(int64_t)c.w * DP_STAMPS + c.nstamp] = wall_clock64();
    ++c.nstamp;
}

__device__ __forceinline__ int dp_shard_count(int lo, int hi, int s) {   // producers w in [lo, hi) with w % 8 == s
    return (hi - s + 7) / 8 - (lo - s + 7) / 8;
}

// Arrive at an edge: every storing wave has drained its stores; ONE lane adds 1 to the edge's counter (one of 8 shards above 64
// producers).  The LAST arriver — it sees the count complete — publishes epoch + 1 in the edge's 8 READY flags (two levels when
// sharded: the last of a shard adds 1 to the top counter, the last of those publishes).  Waiters poll a flag copy, never the
// counters: with 48-256 workgroups polling the line the arrivals add to, polls and atomics queued behind each other and an
// edge cost 5-7 us (profiles/r06_draft_persist_timeline_a.jsonl).  Called by the epilogue wave, or by everyone (`all`).
__device__ __forceinline__ void dp_arrive(DpCtx& c, int edge, int lo, int hi, bool all) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (all) __syncthreads();
    if (c.tid == 0) {
        const bool sharded = hi - lo > 64;
        const bool skip = c.P->skip_edge == edge + 1 && c.w == lo;        // fault injection (tests): one arrival is lost
        if (!skip) {
            const int s = sharded ? (c.w & 7) : 0;
            const unsigned expect = (unsigned)(sharded ? dp_shard_count(lo, hi, s) : (hi - lo));
            unsigned old = __hip_atomic_fetch_add(&c.ctl->cnt[edge][s][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool last = old + 1u == (c.epoch + 1u) * expect;
            if (last && sharded) {
                old = __hip_atomic_fetch_add(&c.ctl->cnt[edge][8][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = old + 1u == (c.epoch + 1u) * 8u;
            }
            if (last) {
#pragma unroll
                for (int k = 0; k < 8; ++k) st4u(&c.ctl->flag[edge][k][0], c.epoch + 1u);
            }
        }
    }
}

// Wait until every producer of `edge` has arrived in THIS launch: lane 0 of wave 0 polls this workgroup's copy of the edge's
// READY flag, the other waves park at the barrier.  false (workgroup-uniform): time-out or error word set — the caller leaves.
__device__ __forceinline__ bool dp_wait(DpCtx& c, int edge, int lo, int hi) {
    (void)lo;
    (void)hi;
    if (c.wave == 0) {
        const unsigned tgt = c.epoch + 1u;
        const unsigned* slot = &c.ctl->flag[edge][c.w & 7][0];
        const u64 t0 = wall_clock64();
        bool fail = false;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = ld4u(slot);                                // (wave-uniform address: one request)
            if ((int)(v - tgt) >= 0) break;
            __builtin_amdgcn_s_sleep(3);
            if ((spins & 31u) == 31u) {
                if (ld4u(&c.ctl->error) != 0u) { fail = true; break; }
                if (wall_clock64() - t0 > c.P->timeout_ticks) {
                    if (c.lane == 0) {                                       // the FIRST time-out names the edge (word and mirror agree)
                        unsigned expected = 0u;
                        if (__hip_atomic_compare_exchange_strong(&c.ctl->error, &expected, (unsigned)(edge + 1), __ATOMIC_RELAXED,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            unsigned* mir = reinterpret_cast<unsigned*>(c.ctl->mirror);
                            if (mir) __hip_atomic_store(mir, (unsigned)(edge + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                    fail = true;
                    break;
                }
            }
        }
        if (fail && c.lane == 0) *c.sh_fail = 1;
    }
    __syncthreads();
    return *c.sh_fail == 0;
}

// Fold the 48 per-panel sums of squares a residual epilogue left (ss[panel][32 rows]) into sum(x^2) of row li, in the order
// skinny_gemm_kernel folds them with GRP = 4 x its waves thread groups (csrc/gemv.hip, NORM prologue): thread (pg, m) adds
// the partials of panels pg, pg + GRP, pg + 2 GRP of row m, then row li adds the GRP group sums in order.
template <int GRP>
__device__ __forceinline__ float dp_fold_ss(const float* ss, float* red, int tid, int li) {
    if (tid < GRP * 16) {
        const int pg = tid >> 4, m16 = tid & 15;
        const float v0 = ld4f(ss + pg * 32 + m16);
        const float v1 = (pg + GRP < SS_PARTS) ? ld4f(ss + (pg + GRP) * 32 + m16) : 0.f;
        const float v2 = (pg + 2 * GRP < SS_PARTS) ? ld4f(ss + (pg + 2 * GRP) * 32 + m16) : 0.f;
        float part = 0.f;
        part += v0;
        part += v1;
        part += v2;
        red[pg * 16 + m16] = part;
    }
    __syncthreads();
    float tot = 0.f;
    for (int j = 0; j < GRP; ++j) tot += red[j * 16 + li];
    __syncthreads();
    return tot;
}

__device__ __forceinline__ const h16* dp_embed_row(const DpParams& P, int m) {
    int64_t id = P.ids[m];
    if (id < 0) id = 0;
    if (id >= P.vocab) id = P.vocab - 1;
    return P.embed + id * HID;
}

// ------------------------------------------------------------------------------------------------------------------------
// q|k|v + RoPE(q) + KV append of layer l: panel = workgroup, wave = K-split (3 k-chunks)      [tf_skinny_qkv_rope, 8 waves]
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dp_role_qkv(DpCtx& c, int l, float* sm, float* red, float* sm_ss) {
    const DpParams& P = *c.P;
    const int panel = c.w - QKV_LO, lane = c.lane, wave = c.wave, li = c.li, g = c.g;
    const half8* wa = P.wqkv[l] + (int64_t)panel * KC * 64 + lane;
    const int c0 = wave * 3;
    half8 a[3], b[3], lw[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) a[u] = ld_w(wa + (int64_t)(c0 + u) * 64);
#pragma unroll
    for (int u = 0; u < 3; ++u) lw[u] = load_half8(P.ln1[l] + 32 * (c0 + u) + 8 * g);
    // RoPE tables of this lane's output columns (epilogue wave): positions are slot0 + row
    const int sec = panel / (NH * 4), hd = (panel / 4) % NH, pp = panel % 4;
    half4 rope_cs = {0, 0, 0, 0}, rope_sn = {0, 0, 0, 0};
    const bool rowv = li < P.n;
    if (wave == 0 && rowv && sec == 0) {
        const int d = 8 * pp + 4 * (g & 1) + ((g >= 2) ? (HD >> 1) : 0);
        const int64_t pos = P.slot0 + li;
        rope_cs = *reinterpret_cast<const half4*>(P.cosb + pos * HD + d);
        rope_sn = *reinterpret_cast<const half4*>(P.sinb + pos * HD + d);
    }
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float inv;
    if (l == 0) {
        // x rows straight from the embedding table; sum(x^2) computed here: this wave's share (its own 3 k-chunks: the
        // 8-wave form gives wave w chunks 3w..3w+2 of the sum as well), then across the waves
        const h16* xr = dp_embed_row(P, rowv ? li : 0) + 8 * g;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const half8 v = load_half8(xr + 32 * (c0 + u));
            b[u] = rowv ? v : zero8;
        }
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)b[u][e];
                ss = fmaf(f, f, ss);
            }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (g == 0) sm_ss[wave * 16 + li] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < DP_WAVES; ++w2) tot += sm_ss[w2 * 16 + li];
        inv = 1.0f / sqrtf(tot / (float)HID + P.eps);
    } else {
        if (!dp_wait(c, 5 * (l - 1) + E_DOWN, OD_LO, OD_HI)) return false;
        const h16* xr = P.x + (int64_t)(rowv ? li : 0) * HID + 8 * g;
#pragma unroll
        for (int u = 0; u < 3; ++u) b[u] = zero8;
        if (rowv) {                                                      // (rows past n: no request at all)
#pragma unroll
            for (int u = 0; u < 3; ++u) b[u] = ld_act8(xr + 32 * (c0 + u));
        }
        const float tot = dp_fold_ss<32>(P.ss, red, c.tid, li);
        inv = 1.0f / sqrtf(tot / (float)HID + P.eps);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], dp_normalise(b[u], lw[u], inv), acc, 0, 0, 0);
    *reinterpret_cast<f32x4*>(sm + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
        float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w2 = 0; w2 < DP_WAVES; ++w2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sm + (w2 * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[r] += v[r];
        }
        h16 val[4], oth[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            val[r] = (h16)S[r];
            oth[r] = (h16)__shfl_xor((float)val[r], 32, 64);             // rotary partner: lane g <-> g ^ 2 (exact)
        }
        if (rowv) {
            const int slot = P.slot0 + li;
            half4 out;
            if (sec == 2) {                                              // v: natural row order
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = val[r];
                st_half4(P.vc[l] + (int64_t)slot * P.stride_t + (int64_t)hd * P.stride_h + 16 * pp + 4 * g, out);
            } else {
                const bool hi = g >= 2;
                const int d = 8 * pp + 4 * (g & 1) + (hi ? (HD >> 1) : 0);
                if (sec == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const h16 rh = hi ? oth[r] : (h16)(-(float)oth[r]);
                        out[r] = hadd_rn(hmul_rn(val[r], rope_cs[r]), hmul_rn(rh, rope_sn[r]));
                    }
                    st_half4(P.q + ((int64_t)li * NH + hd) * HD + d, out);
                } else {                                                 // k: cached UN-rotated (68m.py:151-178)
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = val[r];
                    st_half4(P.kc[l] + (int64_t)slot * P.stride_t + (int64_t)hd * P.stride_h + d, out);
                }
            }
        }
        dp_arrive(c, 5 * l + E_QKV, QKV_LO, QKV_HI, false);
    }
    __syncthreads();                                                     // sm / sm_ss are reused by the next role
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// plain GEMM + residual + sums of squares (o_proj: K = 768, 3 chunks per wave; down_proj: K = 3072, 12 chunks per wave)
//                                                                                 [tf_skinny_gemm_ex, 8-wave "wide" form]
// ------------------------------------------------------------------------------------------------------------------------
template <int CPW>
__device__ __forceinline__ bool dp_role_plain(DpCtx& c, const half8* wp, const h16* xin, int ldx, int l, bool resid_embed,
                                              int edge_in, int in_lo, int in_hi, int edge_out, float* sm) {
    const DpParams& P = *c.P;
    const int panel = c.w - OD_LO, lane = c.lane, wave = c.wave, li = c.li, g = c.g;
    constexpr int NCH = CPW * DP_WAVES;
    const half8* wa = wp + (int64_t)panel * NCH * 64 + lane;
    const int c0 = wave * CPW;
    half8 a[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) a[u] = ld_w(wa + (int64_t)(c0 + u) * 64);
    if (!dp_wait(c, edge_in, in_lo, in_hi)) return false;
    const bool rowv = li < P.n;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const h16* xr = xin + (int64_t)(rowv ? li : 0) * ldx + 8 * g;
    half8 b[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) b[u] = zero8;
    if (rowv) {                                                          // (rows past n: no request at all)
#pragma unroll
        for (int u = 0; u < CPW; ++u) b[u] = ld_act8(xr + 32 * (c0 + u));
    }
    const int y_off = 16 * panel + 4 * g;
    half4 res = {0, 0, 0, 0};
    if (wave == 0 && rowv) {
        if (resid_embed) res = *reinterpret_cast<const half4*>(dp_embed_row(P, li) + y_off);
        else res = __builtin_bit_cast(half4, ld8(P.x + (int64_t)li * HID + y_off));
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CPW; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u], acc, 0, 0, 0);
    *reinterpret_cast<f32x4*>(sm + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
        float S[4] = {0.f, 0.f, 0.f, 0.f}, s2[4];
#pragma unroll
        for (int w2 = 0; w2 < DP_WAVES; ++w2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sm + (w2 * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[r] += v[r];
        }
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = (h16)S[r];
            o[r] = hadd_rn(res[r], o[r]);                                // residual + hidden, fp16 add
            s2[r] = rowv ? (float)o[r] : 0.f;
        }
        if (rowv) st_half4(P.x + (int64_t)li * HID + y_off, o);
        float qq = s2[0] * s2[0] + s2[1] * s2[1] + s2[2] * s2[2] + s2[3] * s2[3];
        qq += __shfl_xor(qq, 16, 64);
        qq += __shfl_xor(qq, 32, 64);
        if (g == 0) st4f(P.ss + panel * 32 + li, qq);
        dp_arrive(c, edge_out, OD_LO, OD_HI, false);
    }
    __syncthreads();
    (void)l;
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// RMSNorm + gate|up + SwiGLU: panel pair = workgroup; waves 0-3 = the gate stream's 4 K-splits, waves 4-7 = the up stream's
//                                                                                     [tf_skinny_gemm_swiglu_ex, 4 waves]
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dp_role_gateup(DpCtx& c, int l, float* sm, float* red) {
    const DpParams& P = *c.P;
    const int panel = c.w - GU_LO, lane = c.lane, wave = c.wave, li = c.li, g = c.g;
    const int ks = wave & 3, c0 = ks * 6;
    const half8* wa = (wave < 4 ? P.wgate[l] : P.wup[l]) + (int64_t)panel * KC * 64 + lane;
    half8 a[6], lw[6], b[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) a[u] = ld_w(wa + (int64_t)(c0 + u) * 64);
#pragma unroll
    for (int u = 0; u < 6; ++u) lw[u] = load_half8(P.ln2[l] + 32 * (c0 + u) + 8 * g);
    if (!dp_wait(c, 5 * l + E_O, OD_LO, OD_HI)) return false;
    const bool rowv = li < P.n;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const h16* xr = P.x + (int64_t)(rowv ? li : 0) * HID + 8 * g;
#pragma unroll
    for (int u = 0; u < 6; ++u) b[u] = zero8;
    if (rowv) {
#pragma unroll
        for (int u = 0; u < 6; ++u) b[u] = ld_act8(xr + 32 * (c0 + u));
    }
    const float tot = dp_fold_ss<16>(P.ss, red, c.tid, li);
    const float inv = 1.0f / sqrtf(tot / (float)HID + P.eps);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 6; ++u)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], dp_normalise(b[u], lw[u], inv), acc, 0, 0, 0);
    *reinterpret_cast<f32x4*>(sm + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
        float S[4] = {0.f, 0.f, 0.f, 0.f}, S2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sm + (w2 * 64 + lane) * 4);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(sm + ((w2 + 4) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                S[r] += v[r];
                S2[r] += v2[r];
            }
        }
        if (rowv) {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const h16 gt = (h16)S[r], up = (h16)S2[r];
                const float gf = (float)gt;
                const h16 act = (h16)(gf / (1.0f + expf(-gf)));
                o[r] = hmul_rn(act, up);
            }
            st_half4(P.act + (int64_t)li * INTER + 16 * panel + 4 * g, o);
        }
        dp_arrive(c, 5 * l + E_GU, GU_LO, GU_HI, false);
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// attention of one head, RoPE applied to the cached keys on read                      [attn_rope_on_read_mfma_kernel<64>]
// The rows below slot0 (written by earlier launches) are rotated / transposed into LDS BEFORE the wait.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dp_role_attn(DpCtx& c, int l, unsigned char* smem) {
    const DpParams& P = *c.P;
    const int h = c.w - ATT_LO, tid = c.tid, lane = c.lane, wave = c.wave, li = c.li, g = c.g;
    const int kv_len = P.kv_len, sq = P.n;
    constexpr int KS = HD + DP_KPAD;
    const int kvp = (kv_len + 31) & ~31, PS = kvp + DP_KPAD;
    h16* sK = reinterpret_cast<h16*>(smem);               // [kvp][KS]
    h16* sVt = sK + (size_t)kvp * KS;                     // [HD][PS]
    h16* sP = sVt + (size_t)HD * PS;                      // [16][PS]
    h16* sPl = sP + (size_t)16 * PS;                      // [16][PS]  low-order parts of P
    float* sS = reinterpret_cast<float*>(sPl + (size_t)16 * PS);   // [16][kvp]
    float* sL = sS + (size_t)16 * kvp;                    // [16]
    const h16* kb = P.kc[l] + (int64_t)h * P.stride_h;
    const h16* vb = P.vc[l] + (int64_t)h * P.stride_h;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int old = min(P.slot0, kv_len);                 // rows [0, old): untouched by this launch

    auto rotate_rows = [&](int j0, int j1, bool fresh) {
        for (int e = j0 * 4 + tid; e < j1 * 4; e += DP_THREADS) {
            const int j = e >> 2, i = e & 3;              // key j, 8-wide vector i of the low half
            half8 lo = zero8, hi = zero8;
            if (j < kv_len) {
                const h16* kp = kb + (int64_t)j * P.stride_t;
                const half8 x1 = fresh ? ld_act8(kp + 8 * i) : load_half8(kp + 8 * i);
                const half8 x2 = fresh ? ld_act8(kp + 8 * i + 32) : load_half8(kp + 8 * i + 32);
                const half8 c1 = load_half8(P.cosb + (int64_t)j * HD + 8 * i), c2 = load_half8(P.cosb + (int64_t)j * HD + 8 * i + 32);
                const half8 s1 = load_half8(P.sinb + (int64_t)j * HD + 8 * i), s2 = load_half8(P.sinb + (int64_t)j * HD + 8 * i + 32);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    lo[t] = hadd_rn(hmul_rn(x1[t], c1[t]), hmul_rn((h16)(-(float)x2[t]), s1[t]));
                    hi[t] = hadd_rn(hmul_rn(x2[t], c2[t]), hmul_rn(x1[t], s2[t]));
                }
            }
            store_half8(sK + (size_t)j * KS + 8 * i, lo);
            store_half8(sK + (size_t)j * KS + 8 * i + 32, hi);
        }
        for (int e = j0 * 8 + tid; e < j1 * 8; e += DP_THREADS) {
            const int j = e >> 3, i = e & 7;
            half8 x = zero8;
            if (j < kv_len) x = fresh ? ld_act8(vb + (int64_t)j * P.stride_t + 8 * i) : load_half8(vb + (int64_t)j * P.stride_t + 8 * i);
#pragma unroll
            for (int t = 0; t < 8; ++t) sVt[(size_t)(8 * i + t) * PS + j] = x[t];
        }
    };
    rotate_rows(0, old, false);
    // The rows this launch writes (and the padding up to kvp): at most 16 + 31 rows — ONE K element and ONE V element per
    // thread.  Their RoPE table rows are fetched before the wait; behind it every hand-off load (K, V, q) is issued before the
    // first is used: one memory round trip instead of three.
    const int ek = old * 4 + tid, ev = old * 8 + tid;
    const int jk = ek >> 2, ik = ek & 3, jv = ev >> 3, iv = ev & 7;
    const bool kin = ek < kvp * 4 && jk < kv_len, vin = ev < kvp * 8 && jv < kv_len;
    half8 c1 = zero8, c2 = zero8, s1 = zero8, s2 = zero8;
    if (kin) {
        c1 = load_half8(P.cosb + (int64_t)jk * HD + 8 * ik), c2 = load_half8(P.cosb + (int64_t)jk * HD + 8 * ik + 32);
        s1 = load_half8(P.sinb + (int64_t)jk * HD + 8 * ik), s2 = load_half8(P.sinb + (int64_t)jk * HD + 8 * ik + 32);
    }
    if (!dp_wait(c, 5 * l + E_QKV, QKV_LO, QKV_HI)) return false;
    dp_stamp(c);
    half8 x1 = zero8, x2 = zero8, xv = zero8;
    half8 qf[HD / 32];
    {
        const h16* kp = kb + (int64_t)(kin ? jk : 0) * P.stride_t + 8 * ik;
        const h16* vp = vb + (int64_t)(vin ? jv : 0) * P.stride_t + 8 * iv;
        const h16* qp = P.q + ((int64_t)(li < sq ? li : 0) * NH + h) * HD + 8 * g;
        const half8 t1 = ld_act8(kp), t2 = ld_act8(kp + 32), t3 = ld_act8(vp), t4 = ld_act8(qp), t5 = ld_act8(qp + 32);
        x1 = kin ? t1 : zero8;
        x2 = kin ? t2 : zero8;
        xv = vin ? t3 : zero8;
        qf[0] = (li < sq) ? t4 : zero8;
        qf[1] = (li < sq) ? t5 : zero8;
    }
    if (ek < kvp * 4) {
        half8 lo, hi;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            lo[t] = hadd_rn(hmul_rn(x1[t], c1[t]), hmul_rn((h16)(-(float)x2[t]), s1[t]));
            hi[t] = hadd_rn(hmul_rn(x2[t], c2[t]), hmul_rn(x1[t], s2[t]));
        }
        if (!kin) lo = zero8, hi = zero8;
        store_half8(sK + (size_t)jk * KS + 8 * ik, lo);
        store_half8(sK + (size_t)jk * KS + 8 * ik + 32, hi);
    }
    if (ev < kvp * 8) {
#pragma unroll
        for (int t = 0; t < 8; ++t) sVt[(size_t)(8 * iv + t) * PS + jv] = xv[t];
    }
    __syncthreads();
    dp_stamp(c);
    // S[q][key] = scale * Q K^T
    for (int t = wave; t < kvp / 16; t += DP_WAVES) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < HD / 32; ++cc) {
            const half8 bk = load_half8(sK + (size_t)(t * 16 + li) * KS + 32 * cc + 8 * g);
            s = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[cc], bk, s, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[(size_t)(4 * g + r) * kvp + t * 16 + li] = s[r] * P.scale;
    }
    __syncthreads();
    dp_stamp(c);
    // softmax: wave owns rows w, w + 8; rows past sq are skipped (their P rows stay whatever LDS held: an MFMA output row
    // depends on its own A row only, and rows past sq are never stored)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave + DP_WAVES * rr;
        if (row >= sq) continue;
        const int kmax = kv_len - sq + row;                              // bottom-right causal
        float mloc = -1.0e30f;
        for (int j = lane; j <= kmax; j += 64) mloc = fmaxf(mloc, sS[(size_t)row * kvp + j]);
        const float mx = wave_max(mloc);
        float lsum = 0.f;
        for (int j = lane; j < kvp; j += 64) {
            float p = 0.f;
            if (j <= kmax) {
                p = __expf(sS[(size_t)row * kvp + j] - mx);
                lsum += p;
            }
            const h16 ph = (h16)p;
            sP[(size_t)row * PS + j] = ph;
            sPl[(size_t)row * PS + j] = (h16)(p - (float)ph);
        }
        lsum = wave_sum(lsum);
        if (lane == 0) sL[row] = lsum;
    }
    __syncthreads();
    dp_stamp(c);
    // O[:, 16w..16w+15] = P V, waves 0..3; staged through LDS (the sS block is free now) for 8-byte write-through stores
    h16* sO = reinterpret_cast<h16*>(sS);                 // [16][HD]
    if (wave < 4) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < kvp / 32; ++cc) {
            const half8 ap = load_half8(sP + (size_t)li * PS + 32 * cc + 8 * g);
            const half8 bv = load_half8(sVt + (size_t)(16 * wave + li) * PS + 32 * cc + 8 * g);
            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap, bv, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(load_half8(sPl + (size_t)li * PS + 32 * cc + 8 * g), bv, o, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sO[(4 * g + r) * HD + 16 * wave + li] = (h16)(o[r] / sL[4 * g + r]);
    }
    __syncthreads();
    dp_stamp(c);
    if (tid < 256) {
        const int row = tid >> 4, piece = tid & 15;
        if (row < sq) st_half4(P.a + (int64_t)row * HID + h * HD + 4 * piece, *reinterpret_cast<const half4*>(sO + row * HD + 4 * piece));
    }
    dp_arrive(c, 5 * l + E_ATT, ATT_LO, ATT_HI, true);
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// The kernel
// ------------------------------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(DP_THREADS) void draft_persist_kernel(DpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dp_smem[];
    // LDS: [ 0, 64 )  flags | then role-specific carving (roles of one workgroup never overlap in time)
    int* sh_fail = reinterpret_cast<int*>(dp_smem);
    unsigned char* role_smem = dp_smem + 64;
    float* sm = reinterpret_cast<float*>(role_smem);                     // [8][64][4] split-K merge
    float* red = sm + DP_WAVES * 64 * 4;                                 // [32][16]
    float* sm_ss = red + 32 * 16;                                        // [8][16]

    DpCtx c;
    c.P = &P;
    c.ctl = P.ctl;
    c.w = blockIdx.x;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = c.tid >> 6;
    c.li = c.lane & 15;
    c.g = c.lane >> 4;
    c.nstamp = 0;
    c.sh_fail = sh_fail;
    const int w = c.w, tid = c.tid, lane = c.lane, wave = c.wave, li = c.li, g = c.g;
    const int n = P.n, V = P.vocab;
    if (tid == 0) *sh_fail = 0;
    c.epoch = ld4u(&P.ctl->epoch);
    const unsigned err0 = ld4u(&P.ctl->error);
    dp_stamp(c);
    __syncthreads();
    const int npan = V / 16;
    const int lm_panel = w * DP_WAVES + wave;                            // lm_head: one panel per wave
    const bool lm_ok = lm_panel < npan;

    auto poison = [&]() {                                                // NaN over this workgroup's share of the outputs
        const float nanv = __builtin_nanf("");
        for (int i = w * DP_THREADS + tid; i < n * V; i += DP_GRID * DP_THREADS) P.logits[i] = nanv;
        if (P.probs)
            for (int i = w * DP_THREADS + tid; i < V; i += DP_GRID * DP_THREADS) P.probs[i] = nanv;
    };
    if (err0 != 0u) {                                                    // sticky: a failed control block runs nothing
        poison();
        return;
    }
    const bool r_qkv = w >= QKV_LO && w < QKV_HI, r_gu = w >= GU_LO && w < GU_HI;
    const bool r_att = w >= ATT_LO && w < ATT_HI, r_od = w >= OD_LO && w < OD_HI;
    // lm_head weights: this wave's panel, 24 KiB in registers, requested as early as the workgroup's own first role allows —
    // right away where the first role is late or absent, else behind the first role (a wave's loads return in order: issued
    // in front of it they would stand between the first role and its weights).  The 49 MB stream — more than half the bytes of
    // the forward — then runs under the layer phases instead of behind the last one.
    half8 a_lm[KC];
    const half8* wa_lm = P.lm_head + (int64_t)(lm_ok ? lm_panel : 0) * KC * 64 + lane;
    auto lm_prefetch = [&]() {
#pragma unroll
        for (int u = 0; u < DP_LM_EARLY; ++u) a_lm[u] = ld_w(wa_lm + (int64_t)u * 64);
    };
    if (!r_qkv && !r_att && !r_od) lm_prefetch();
    bool ok = true;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (ok && r_qkv) {
            ok = dp_role_qkv(c, l, sm, red, sm_ss);
            if (l == 0) lm_prefetch();
            dp_stamp(c);
        }
        if (ok && r_att) {
            ok = dp_role_attn(c, l, role_smem);
            if (l == 0) lm_prefetch();
            dp_stamp(c);
        }
        if (ok && r_od) {
            ok = dp_role_plain<3>(c, P.wo[l], P.a, HID, l, l == 0, 5 * l + E_ATT, ATT_LO, ATT_HI, 5 * l + E_O, sm);
            if (l == 0) lm_prefetch();
            dp_stamp(c);
        }
        if (ok && r_gu) {
            ok = dp_role_gateup(c, l, sm, red);
            dp_stamp(c);
        }
        if (ok && r_od) {
            ok = dp_role_plain<12>(c, P.wdown[l], P.act, INTER, l, false, 5 * l + E_GU, GU_LO, GU_HI, 5 * l + E_DOWN, sm);
            dp_stamp(c);
        }
    }
    if (!ok) {
        poison();
        return;
    }

    // ---------------- lm_head: final RMSNorm + one 16-column panel per wave ----------------   [tf_skinny_gemm_ex, P = 2 form]
    constexpr int E_LM = 5 * L, E_T2 = 5 * L + 1;
    float xrow[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};        // this lane's 4 entries of the sampled row: l / T
    const bool own = lm_ok && li == n - 1;                               // lanes that hold entries of the last row
    {
#pragma unroll
        for (int u = DP_LM_EARLY; u < KC; ++u) a_lm[u] = ld_w(wa_lm + (int64_t)u * 64);   // the rest: behind the last wait
        if (!dp_wait(c, 5 * (L - 1) + E_DOWN, OD_LO, OD_HI)) {
            poison();
            return;
        }
        dp_stamp(c);
        // the normalised rows once per workgroup, in LDS: h[m][k] = w_ln * fp16(x * inv)
        constexpr int HS = HID + 8;                                      // row stride (halfs): 16-byte reads spread over the banks
        h16* sh_h = reinterpret_cast<h16*>(role_smem + 16384);           // [16][HS]   (behind sm / red / sm_ss)
        const float tot = dp_fold_ss<16>(P.ss, red, tid, li);
        const float inv = 1.0f / sqrtf(tot / (float)HID + P.eps);
        // thread -> (row m = li of the fold, k-octets): inv is per row li, so lane (li, ...) normalises row li
        for (int o8 = (tid >> 4); o8 < HID / 8; o8 += DP_THREADS / 16) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (li < n) v = ld_act8(P.x + (int64_t)li * HID + 8 * o8);
            store_half8(sh_h + li * HS + 8 * o8, dp_normalise(v, load_half8(P.norm + 8 * o8), inv));
        }
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            const half8 b = load_half8(sh_h + li * HS + 32 * u + 8 * g);
            acc[u / 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lm[u], b, acc[u / 6], 0, 0, 0);
        }
        float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[r] += acc[s][r];
        if (lm_ok && li < n) {
            float* dst = P.logits + (int64_t)li * V + lm_panel * 16 + 4 * g;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (float)(h16)S[r];         // logits.float(): fp16 GEMM, then cast
            *reinterpret_cast<f32x4*>(dst) = o;
            if (own) {
#pragma unroll
                for (int r = 0; r < 4; ++r) xrow[r] = o[r] / P.temperature;
            }
        }
    }
    DpTopp* tp = reinterpret_cast<DpTopp*>(role_smem);                   // (aliases the GEMM scratch: a barrier separates them)
    __syncthreads();
    if (P.probs) {
        float mx = fmaxf(fmaxf(xrow[0], xrow[1]), fmaxf(xrow[2], xrow[3]));
        mx = wave_max(mx);
        if (lane == 0) tp->wmax[wave] = mx;
        for (int i = tid; i < 1024 + 64; i += DP_THREADS) {
            tp->hist[i] = 0ull;
            tp->cnt[i] = 0u;
        }
        __syncthreads();
        if (tid == 0) {
            float m2 = tp->wmax[0];
#pragma unroll
            for (int k = 1; k < DP_WAVES; ++k) m2 = fmaxf(m2, tp->wmax[k]);
            st4f(P.wgmax + w, m2);
        }
    }
    dp_stamp(c);
    dp_arrive(c, E_LM, 0, DP_GRID, true);
    if (!P.probs) {
        dp_arrive(c, E_T2, 0, DP_GRID, false);                           // every edge counts every launch
        if (w == 0) {
            if (!dp_wait(c, E_LM, 0, DP_GRID)) return;
            if (tid == 0) st4u(&P.ctl->epoch, c.epoch + 1u);
        }
        dp_stamp(c);
        return;
    }

    // ---------------- top-p, part (a): e = exp(x - max), masses, this workgroup's CANDIDATES ----------------
    // Candidates = entries at or above the cut (1 - top_p) / (2 V): they provably contain the top-p crossing
    // (tests/test_host_edges_cpu.py::test_topp_candidate_cut_always_contains_the_crossing).  Every workgroup leaves its <= 128 as
    // (pattern, index) pairs + a count + its exact mass sum; part (b) then reads a few KB of candidates instead of the 128 KB row
    // (round 6's first build read the row — 4.9 us through the fabric, write-through lines sit in no L2 — and added masses to a
    // global histogram with atomics that 256 workgroups of a flat row all aimed at the same three bins).
    if (!dp_wait(c, E_LM, 0, DP_GRID)) {
        poison();
        return;
    }
    dp_stamp(c);
    const float top_p = P.top_p;
    unsigned cutpat = 0u;
    if (top_p < 1.0f) cutpat = __float_as_uint((1.0f - top_p) / (2.0f * (float)V)) & ~((1u << 20) - 1u);
    float erow[4] = {0.f, 0.f, 0.f, 0.f};
    {
        float mx = -INFINITY;
        if (wave == 0) {
            // 256 per-workgroup maxima: 4 per lane
            const U64x2 v = {ld8(P.wgmax + 4 * lane), ld8(P.wgmax + 4 * lane + 2)};
            const f32x4 f = __builtin_bit_cast(f32x4, v);
            mx = wave_max(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])));
            if (lane == 0) {
                tp->wmax[0] = mx;
                tp->nlist = 0u;
            }
        }
        __syncthreads();
        mx = tp->wmax[0];
        u64 zacc = 0ull;
        u64* mycand = P.cand + (int64_t)w * 128;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = own ? expf(xrow[r] - mx) : 0.f;
            erow[r] = e;
            const u64 m = own ? dp_fix(e) : 0ull;
            zacc += m;
            if (m != 0ull && __float_as_uint(e) >= cutpat) {             // (order within the segment is irrelevant: sums are exact)
                const unsigned at = atomicAdd(&tp->nlist, 1u);
                st8(mycand + at, ((u64)__float_as_uint(e) << 32) | (unsigned)(lm_panel * 16 + 4 * g + r));
            }
        }
        const u64 zw = dp_wave_sum_u64(zacc);
        if (lane == 0) tp->zpart[wave] = zw;
        __syncthreads();
        if (tid == 0) {
            u64 z = 0ull;
#pragma unroll
            for (int k = 0; k < DP_WAVES; ++k) z += tp->zpart[k];
            st8(P.zslice + w, z);
            st4u(P.ccount + w, tp->nlist);
        }
    }
    dp_stamp(c);
    dp_arrive(c, E_T2, 0, DP_GRID, true);
    if (!dp_wait(c, E_T2, 0, DP_GRID)) {
        poison();
        return;
    }
    dp_stamp(c);
    if (w == 0 && tid == 0) st4u(&P.ctl->epoch, c.epoch + 1u);           // every workgroup has passed its last wait's arrival

    // ---------------- top-p, part (b): every workgroup finds the boundary of the WHOLE row ----------------
    // Z, tau and the candidate offsets from the 256 per-workgroup sums / counts; the candidates gathered into an LDS list (each wave
    // its 32 segments, one entry per lane and segment in flight together: a model's row leaves a few dozen per segment); the three
    // radix rounds on the list with 512-thread DPP scans; a cut tie group ranked by a two-round radix select on the index.  More
    // candidates than the list holds (near-flat rows): round 1 streams them and only the boundary bin is filed; a boundary bin
    // that still does not fit is streamed by every round.
    constexpr int LIST_CAP = 16384;
    u64* list = reinterpret_cast<u64*>(role_smem + 16384);               // [LIST_CAP]
    unsigned* offs = reinterpret_cast<unsigned*>(role_smem + 16384 + LIST_CAP * 8);    // [DP_GRID + 1] candidate offsets
    u64* fin = reinterpret_cast<u64*>(role_smem + 16384 + LIST_CAP * 8 + 1088);        // [64] the boundary bin's entries
    // (the first 16 entries — one 128-byte line — of this wave's 32 segments are requested WITH the counts, a quarter wave per segment:
    //  lanes past a segment's count read stale workspace and drop it.  One round trip and 32 KB per workgroup for the usual row; a
    //  segment with more than 16 candidates costs a second trip for the rest.)
    constexpr int SPEC = 16;
    u64 spec[8];
    {
        const unsigned cmine = tid < DP_GRID ? ld4u(P.ccount + tid) : 0u;
        const u64 zmine = tid < DP_GRID ? ld8(P.zslice + tid) : 0ull;
#pragma unroll
        for (int j = 0; j < 8; ++j) spec[j] = ld8(P.cand + (int64_t)(32 * wave + 4 * j + (lane >> 4)) * 128 + (lane & 15));
        const unsigned incl = dp_wave_prefix_u32(cmine);
        const u64 zincl = dp_wave_prefix_u64(zmine);
        if (lane == 63) {
            tp->red_i[wave] = (int)incl;
            tp->zpart[wave] = zincl;
        }
        __syncthreads();
        unsigned before = 0u;
        u64 Z = 0ull;
#pragma unroll
        for (int k = 0; k < DP_WAVES; ++k) {
            if (k < wave) before += (unsigned)tp->red_i[k];
            Z += tp->zpart[k];
        }
        if (tid < DP_GRID) offs[tid + 1] = before + incl;
        if (tid == 0) {
            offs[0] = 0u;
            const double t = (double)top_p * (double)Z;
            tp->tau = (t >= 18446744073709549568.0) ? ~0ull : __double2ull_rd(t);
            tp->Z = Z;
            tp->zk = Z;
            tp->S = 0ull;
            tp->digit = -1;
            tp->ties = 0u;
            tp->nkeep = 0ull;
            tp->nlist = 0u;
            tp->nfin = 0u;
        }
        __syncthreads();
    }
    const unsigned C = offs[DP_GRID];
    const u64 tau = tp->tau, Zall = tp->Z;
    unsigned maxc = 0u;                                                  // largest segment of THIS wave's 32
    for (int q = 0; q < 32; ++q) maxc = max(maxc, offs[32 * wave + q + 1] - offs[32 * wave + q]);
    // walk the row's candidates where they lie: wave w owns the segments of workgroups 32 w .. 32 w + 31
    auto stream = [&](auto fn) {                                         // (every lane calls fn: it may use wave collectives)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sg = 32 * wave + 4 * j + (lane >> 4);
            const unsigned cn = offs[sg + 1] - offs[sg], en = (unsigned)(lane & 15);
            fn(spec[j], offs[sg] + en, en < cn);
        }
        if (maxc <= (unsigned)SPEC) return;
        for (int q0 = 0; q0 < 32; q0 += 16) {                            // entries 16 .. 127: 16 segments' loads in flight together
            u64 v[16][2];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int sg = 32 * wave + q0 + q;
                const unsigned cn = offs[sg + 1] - offs[sg];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned en = SPEC + lane + 64 * k;
                    v[q][k] = (en < cn && en < 128u) ? ld8(P.cand + (int64_t)sg * 128 + en) : 0ull;
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int sg = 32 * wave + q0 + q;
                const unsigned cn = offs[sg + 1] - offs[sg];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned en = SPEC + lane + 64 * k;
                    fn(v[q][k], offs[sg] + en, en < cn && en < 128u);
                }
            }
        }
    };
    unsigned nlist = 0u;
    bool use_list = C <= (unsigned)LIST_CAP;
    if (use_list) {
        stream([&](u64 e, unsigned at, bool ok) {
            if (ok) list[at] = e;
        });
        nlist = C;
        __syncthreads();
    }
    dp_stamp(c);
    auto for_each = [&](auto fn) {
        if (use_list) {
            for (unsigned i = tid; i < nlist; i += DP_THREADS) fn(list[i]);
        } else {
            stream([&](u64 e, unsigned, bool ok) {
                if (ok) fn(e);
            });
        }
    };
    // one round: bins dig(pattern) of the candidates with sel(pattern) (+ how many fell in each when `count`); then the 512-thread
    // scan from base tp->S; tp->ties = the entries of the bin the scan names (count rounds only)
    auto round = [&](auto sel, auto dig, bool count) {
        for_each([&](u64 e) {
            const unsigned bb = (unsigned)(e >> 32);
            if (sel(bb)) {
                const int d = dig(bb);
                atomicAdd(&tp->hist[DP_HB(d)], dp_fix(__uint_as_float(bb)));
                if (count) atomicAdd(&tp->cnt[DP_HB(d)], 1u);
            }
        });
        __syncthreads();
        const int bA = DP_HB(1023 - 2 * tid), bB = DP_HB(1022 - 2 * tid);
        const u64 hA = tp->hist[bA], hB = tp->hist[bB];
        tp->hist[bA] = 0ull;
        tp->hist[bB] = 0ull;
        const u64 base = tp->S;
        if (tid == 0) tp->digit = -1;
        dp_scan512(tp, hA, hB, base, tau, tid, lane, wave);
        if (count) {
            const int d = tp->digit;
            if (d >= 0 && (d == 1023 - 2 * tid || d == 1022 - 2 * tid)) tp->ties = tp->cnt[DP_HB(d)];
            __syncthreads();
            tp->cnt[bA] = 0u;
            tp->cnt[bB] = 0u;
        }
    };
    auto tie_round = [&](auto sel, auto dig, u64 base, u64 want) {
        for_each([&](u64 e) {
            if (sel(e)) atomicAdd(&tp->hist[DP_HB(1023 - dig(e))], 1ull);
        });
        __syncthreads();
        const int bA = DP_HB(1023 - 2 * tid), bB = DP_HB(1022 - 2 * tid);
        const u64 hA = tp->hist[bA], hB = tp->hist[bB];
        tp->hist[bA] = 0ull;
        tp->hist[bB] = 0ull;
        if (tid == 0) tp->digit = -1;
        dp_scan512(tp, hA, hB, base, want, tid, lane, wave);
    };
    int d1 = -1;
    unsigned ustar = 0u, ties = 0u;
    u64 nkeep = 0ull, zk = Zall;
    long long istar = -1;
    if (tau < Zall) {                                                    // (tau >= Z: top_p >= 1 keeps everything)
        round([](unsigned) { return true; }, [](unsigned bb) { return (int)(bb >> 20); }, false);
        d1 = tp->digit;
        if (d1 < 0) {                                                    // cannot happen: the cut holds the crossing among the candidates
            if (tid == 0) {
                unsigned expected = 0u;
                __hip_atomic_compare_exchange_strong(&P.ctl->error, &expected, 99u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            poison();
            return;
        }
        dp_stamp(c);
        const int dd1 = d1;
        // the boundary bin's entries (an eighth of an octave of e), gathered in any order — up to 64 of them; one counter bump
        // per wave and pass (a flat row has thousands of matches)
        for_each([&](u64 e) {
            const bool hit = (int)((unsigned)(e >> 32) >> 20) == dd1;
            const unsigned long long hm = __ballot(hit);                 // (the lanes still in the loop)
            if (hit) {
                const int leader = (int)__builtin_ctzll(hm);
                unsigned at0 = 0u;
                if (lane == leader) at0 = atomicAdd(&tp->nfin, (unsigned)__popcll(hm));
                const unsigned at = (unsigned)__shfl((int)at0, leader, 64) + (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
                if (at < 64u) fin[at] = e;
            }
        });
        __syncthreads();
        const unsigned nbin = tp->nfin;
        if (nbin <= 64u) {
            // The usual row: the boundary bin holds a handful of entries and every WAVE finishes by itself, lane i holding entry i:
            // the mass above it, its tie group and its rank by index inside the group come from one pass over the <= 64 entries
            // — the boundary the two remaining radix rounds and the tie ranking would name (integers throughout), without their
            // eight barriers.
            const u64 mine = (unsigned)lane < nbin ? fin[lane] : 0ull;
            const unsigned pi = (unsigned)(mine >> 32), ii = (unsigned)mine;
            const u64 mi = dp_fix(__uint_as_float(pi));
            u64 above = 0ull;
            unsigned eq = 0u, lower = 0u;
            for (unsigned j = 0; j < nbin; ++j) {
                const u64 ej = fin[j];                                   // (one address for the wave: a broadcast read)
                const unsigned pj = (unsigned)(ej >> 32), ij = (unsigned)ej;
                above += pj > pi ? dp_fix(__uint_as_float(pj)) : 0ull;
                eq += pj == pi ? 1u : 0u;
                lower += (pj == pi && ij < ii) ? 1u : 0u;
            }
            const u64 Sx = tp->S + above;                                // mass of everything above this lane's pattern
            const bool cross = (unsigned)lane < nbin && Sx <= tau && tau - Sx < (u64)eq * mi;
            const unsigned long long cm = __ballot(cross);               // the lanes of ONE tie group
            const int first = (int)__builtin_ctzll(cm);
            ustar = (unsigned)__shfl((int)pi, first, 64);
            ties = (unsigned)__shfl((int)eq, first, 64);
            const u64 Sg = dp_shfl_u64(Sx, first), mg = dp_shfl_u64(mi, first);
            nkeep = (tau - Sg) / mg + 1ull;
            if (nkeep > (u64)ties) nkeep = ties;
            zk = Sg + nkeep * mg;
            if (nkeep < (u64)ties) {                                     // the tie group is cut: its nkeep lowest indices stay
                const unsigned long long lm = __ballot(cross && (u64)lower + 1ull == nkeep);
                istar = (long long)(unsigned)__shfl((int)ii, (int)__builtin_ctzll(lm), 64);
            }
        } else {
            if (!use_list) {
                // file the boundary bin's entries (LDS counter: only the matches pay for it); overflow -> the rounds keep streaming
                stream([&](u64 e, unsigned, bool ok) {
                    if (ok && (int)((unsigned)(e >> 32) >> 20) == dd1) {
                        const unsigned at = atomicAdd(&tp->nlist, 1u);
                        if (at < (unsigned)LIST_CAP) list[at] = e;
                    }
                });
                __syncthreads();
                nlist = tp->nlist;
                use_list = nlist <= (unsigned)LIST_CAP;
            }
            round([dd1](unsigned bb) { return (int)(bb >> 20) == dd1; }, [](unsigned bb) { return (int)((bb >> 10) & 1023u); }, false);
            const unsigned pre = ((unsigned)d1 << 10) | (unsigned)tp->digit;
            dp_stamp(c);
            round([pre](unsigned bb) { return (bb >> 10) == pre; }, [](unsigned bb) { return (int)(bb & 1023u); }, true);
            ustar = (pre << 10) | (unsigned)tp->digit;
            ties = tp->ties;
            {
                const u64 m = tp->hsel / (u64)ties;                      // all ties share one pattern, hence one mass
                nkeep = (tau - tp->S) / m + 1ull;
                if (nkeep > (u64)ties) nkeep = ties;
                zk = tp->S + nkeep * m;
            }
            __syncthreads();                                             // (everyone has read tp->S before a tie round moves it)
            if (nkeep < (u64)ties) {                                     // the tie group is cut: its nkeep lowest indices stay
                const u64 want = nkeep - 1ull;
                const unsigned us = ustar;
                tie_round([us](u64 e) { return (unsigned)(e >> 32) == us; }, [](u64 e) { return (int)(((unsigned)e >> 5) & 1023u); }, 0ull, want);
                const unsigned hi_idx = (unsigned)(1023 - tp->digit);
                const u64 before = tp->S;
                __syncthreads();
                tie_round([us, hi_idx](u64 e) { return (unsigned)(e >> 32) == us && (((unsigned)e >> 5) & 1023u) == hi_idx; },
                          [](u64 e) { return (int)((unsigned)e & 31u); }, before, want);
                istar = (long long)((hi_idx << 5) | (unsigned)(1023 - tp->digit));
            }
        }
    }
    dp_stamp(c);
    const float Zk = (float)((double)zk * (1.0 / 1099511627776.0));
    const bool rank_ties = d1 >= 0 && nkeep < (u64)ties;
    if (own) {
        const unsigned ulow = d1 < 0 ? 0u : (rank_ties ? ustar + 1u : ustar);   // patterns >= ulow stay unconditionally
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned b = __float_as_uint(erow[r]);
            const long long idx = lm_panel * 16 + 4 * g + r;
            const bool keep = b >= ulow || (rank_ties && b == ustar && idx <= istar);
            o[r] = keep ? erow[r] / Zk : 0.f;
        }
        *reinterpret_cast<f32x4*>(P.probs + lm_panel * 16 + 4 * g) = o;
    }
    dp_stamp(c);
}

int g_dp_timeout_ms = 2000;
int g_dp_skip_edge = 0;
u64* g_dp_stamps = nullptr;

size_t dp_lds_bytes(int kv_len) {
    const size_t kvp = (size_t)((kv_len + 31) & ~31), PS = kvp + DP_KPAD;
    const size_t attn = kvp * (HD + DP_KPAD) * 2 + HD * PS * 2 + 2 * 16 * PS * 2 + 16 * kvp * 4 + 64;
    const size_t gemm = 16384 + 16 * (HID + 8) * 2;                       // merge scratch + normalised rows of lm_head
    const size_t topp = 16384 + (size_t)16384 * 8 + 1088 + 64 * 8;        // select scratch + the candidate list + the segment offsets + the boundary bin
    size_t m = attn > gemm ? attn : gemm;
    if (topp > m) m = topp;
    return 64 + ((m + 15) & ~(size_t)15);
}
inline int64_t a256(int64_t v) { return (v + 255) & ~(int64_t)255; }
}  // namespace

extern "C" int64_t tf_draft_persist_ctl_bytes(void) { return (int64_t)sizeof(DpCtl); }

extern "C" int64_t tf_draft_persist_ws_bytes(const TfDraftModel* m) {
    if (!m) return 0;
    return a256(16 * HID * 2) * 3 + a256(16 * INTER * 2) + a256(SS_PARTS * 32 * 4) + a256(DP_GRID * 4) + a256((int64_t)DP_GRID * 128 * 8) + a256(DP_GRID * 8) + a256(DP_GRID * 4);
}

// 0 when the shape is one the persistent launch takes (the 68M draft, <= 16 rows, <= 384 keys, a device with >= 256 CUs)
extern "C" int tf_draft_persist_supported(const TfDraftModel* m, int n, int kv_len) {
    if (!m) return TF_EINVAL;
    if (m->hidden != HID || m->heads != NH || m->head_dim != HD || m->inter != INTER) return TF_EINVAL;
    if (m->layers < 1 || m->layers > DP_MAX_LAYERS || (m->vocab % 16) || m->vocab > DP_GRID * DP_WAVES * 16) return TF_EINVAL;
    if (m->vocab / 16 <= 512) return TF_EINVAL;        // (the chain's lm_head splits K 8 ways up to 512 panels, 4 ways above: this form is the 4-way one)
    if (n < 1 || n > 16 || kv_len < n || kv_len > DP_MAX_KEYS) return TF_EINVAL;
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return TF_EINVAL;
        cus = prop.multiProcessorCount;
    }
    return cus >= DP_GRID ? TF_OK : TF_ERANGE;
}

extern "C" int tf_draft_forward_68m_persist(const TfDraftModel* m, const TfDraftCache* cch, const int64_t* ids, int n, int slot0,
                                            int kv_len, float* logits_out, float* probs_out, float temperature, float top_p,
                                            void* ws, int64_t ws_bytes, void* ctl, void* stream) {
    if (!m || !cch || !ids || !logits_out || !ws || !ctl) return TF_EINVAL;
    const int rc = tf_draft_persist_supported(m, n, kv_len);
    if (rc) return rc;
    if (slot0 < 0 || kv_len < slot0 + n) return TF_EINVAL;
    if (((kv_len + 31) & ~31) - slot0 > 64) return TF_EINVAL;       // rows read behind the hand-off: one K / V element per thread
    if ((cch->stride_t % 8) || (cch->stride_h % 8)) return TF_EINVAL;
    if (probs_out && (!(temperature > 0.f) || !(top_p > 0.f))) return TF_EINVAL;
    if (ws_bytes < tf_draft_persist_ws_bytes(m)) return TF_ENOSPC;
    if ((reinterpret_cast<uintptr_t>(ws) % 256) || (reinterpret_cast<uintptr_t>(ctl) % 64)) return TF_EINVAL;
    DpParams P = {};
    P.embed = (const h16*)m->embed;
    for (int l = 0; l < m->layers; ++l) {
        P.ln1[l] = (const h16*)m->ln1[l];
        P.wqkv[l] = (const half8*)m->wqkv[l];
        P.wo[l] = (const half8*)m->wo[l];
        P.ln2[l] = (const h16*)m->ln2[l];
        P.wgate[l] = (const half8*)m->wgate[l];
        P.wup[l] = (const half8*)m->wup[l];
        P.wdown[l] = (const half8*)m->wdown[l];
        P.kc[l] = (h16*)cch->k[l];
        P.vc[l] = (h16*)cch->v[l];
        if (!P.ln1[l] || !P.wqkv[l] || !P.wo[l] || !P.ln2[l] || !P.wgate[l] || !P.wup[l] || !P.wdown[l] || !P.kc[l] || !P.vc[l])
            return TF_EINVAL;
    }
    P.norm = (const h16*)m->norm;
    P.lm_head = (const half8*)m->lm_head;
    P.cosb = (const h16*)m->cos;
    P.sinb = (const h16*)m->sin;
    if (!P.embed || !P.norm || !P.lm_head || !P.cosb || !P.sinb) return TF_EINVAL;
    P.stride_t = cch->stride_t;
    P.stride_h = cch->stride_h;
    P.ids = ids;
    P.logits = logits_out;
    P.probs = probs_out;
    char* p = static_cast<char*>(ws);
    P.x = reinterpret_cast<h16*>(p);      p += a256(16 * HID * 2);
    P.q = reinterpret_cast<h16*>(p);      p += a256(16 * HID * 2);
    P.a = reinterpret_cast<h16*>(p);      p += a256(16 * HID * 2);
    P.act = reinterpret_cast<h16*>(p);    p += a256(16 * INTER * 2);
    P.ss = reinterpret_cast<float*>(p);   p += a256(SS_PARTS * 32 * 4);
    P.wgmax = reinterpret_cast<float*>(p); p += a256(DP_GRID * 4);
    P.cand = reinterpret_cast<u64*>(p);   p += a256((int64_t)DP_GRID * 128 * 8);
    P.zslice = reinterpret_cast<u64*>(p); p += a256(DP_GRID * 8);
    P.ccount = reinterpret_cast<unsigned*>(p);
    P.ctl = reinterpret_cast<DpCtl*>(ctl);
    P.stamps = g_dp_stamps;
    P.timeout_ticks = (u64)(g_dp_timeout_ms > 0 ? g_dp_timeout_ms : 1) * (DP_WALL_HZ / 1000ull);
    P.n = n;
    P.slot0 = slot0;
    P.kv_len = kv_len;
    P.layers = m->layers;
    P.vocab = m->vocab;
    P.skip_edge = g_dp_skip_edge;
    P.eps = m->eps;
    P.scale = m->scale;
    P.temperature = temperature;
    P.top_p = top_p;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)draft_persist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)dp_lds_bytes(DP_MAX_KEYS));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)draft_persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)dp_lds_bytes(DP_MAX_KEYS));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (m->layers == 1)
        hipLaunchKernelGGL(draft_persist_kernel<1>, dim3(DP_GRID), dim3(DP_THREADS), dp_lds_bytes(kv_len), (hipStream_t)stream, P);
    else
        hipLaunchKernelGGL(draft_persist_kernel<2>, dim3(DP_GRID), dim3(DP_THREADS), dp_lds_bytes(kv_len), (hipStream_t)stream, P);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// key 0: wall-clock limit of one wait in ms (default 2000; frozen into captured graphs); key 1: fault injection — edge index
// + 1 whose first producer loses its arrival (0 = off; tests only); returns the previous value, -1 for an unknown key
extern "C" int tf_draft_persist_tune(int key, int value) {
    int* slot = key == 0 ? &g_dp_timeout_ms : key == 1 ? &g_dp_skip_edge : nullptr;
    if (!slot) return -1;
    const int old = *slot;
    if (key == 0 && value < 1) return old;
    if (key == 1 && (value < 0 || value > DP_EDGES)) return old;
    *slot = value;
    return old;
}

// Instrumented launches: `buf` = device memory of DP_GRID x DP_STAMPS u64 wall-clock stamps (100 MHz) per workgroup, or NULL
extern "C" int tf_draft_persist_stamps(void* buf) {
    g_dp_stamps = reinterpret_cast<u64*>(buf);
    return DP_STAMPS;
}

// error word of a control block (blocking host read): 0, or edge index + 1 of the wait that timed out
extern "C" int tf_draft_persist_error(const void* ctl) {
    if (!ctl) return TF_EINVAL;
    unsigned head[4];
    hipError_t e = hipMemcpy(head, ctl, sizeof(head), hipMemcpyDeviceToHost);
    return e == hipSuccess ? (int)head[1] : (int)e;
}

// Back to a freshly zeroed control block (blocking); the pinned host mirror pointer is kept or replaced (set_mirror)
extern "C" int tf_draft_persist_reset(void* ctl, void* host_mirror, int set_mirror) {
    if (!ctl) return TF_EINVAL;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    DpCtl* c = reinterpret_cast<DpCtl*>(ctl);
    u64 mirror = 0;
    e = hipMemcpy(&mirror, &c->mirror, sizeof(mirror), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (set_mirror) mirror = (u64)(uintptr_t)host_mirror;
    e = hipMemset(ctl, 0, sizeof(DpCtl));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(&c->mirror, &mirror, sizeof(mirror), hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
    if (mirror) *reinterpret_cast<volatile unsigned*>(mirror) = 0u;
    return (int)hipDeviceSynchronize();
}
