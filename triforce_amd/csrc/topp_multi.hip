// tf_topp_probs_multi: temperature + top-p + softmax (utils/sampling.py:5-27,43-60 `norm_logits`, top_k = -1) with every ROW
// spread over S workgroups of ONE launch (round 6).
//
// tf_topp_probs (csrc/sampling.hip) runs one 1 024-thread workgroup per row: 32 000 correctly rounded divisions (twice) and
// expf on ONE compute unit — ~13 us of arithmetic before any select, 26-42 us per call for <= 8 rows, while 248 CUs idle.
// Here grid = rows x S (S = 16 slices of <= 2 048 entries; 8 slices of <= 4 096 from 17 rows up), 512 threads, all resident,
// and the slices of a row meet through an in-launch edge — the machinery of csrc/draft_persist.hip (arrival counter, the
// last arriver publishes the launch epoch in a READY flag, write-through stores / agent-scope loads, no fences):
//
//   (a) x = l / T; the row maximum: every workgroup reads the WHOLE row (128 KB out of L2, 16 float4 loads in flight per thread —
//       cheaper than an edge; tune key 3 = 0 brings back the slice maxima exchanged through edge A[row]), or takes the per-panel
//       maxima the lm_head GEMM left (panel_max)
//   (b) e = exp(x - max), 2^-40 fixed-point masses, the slice's mass sum, and the slice's CANDIDATES — entries at or above the
//       cut (1 - top_p) / (2 V), which provably contain the top-p crossing — compacted as (pattern, index) pairs
//                                                                  -> edge B[row]
//   (c) every workgroup of the row: the 16 slice counts / sums AND the first 32 candidates of every slice in one round trip (lanes
//       past a count drop what they read), Z and tau, the rest of the candidates gathered into LDS (<= 16 384; beyond that round 1
//       streams them and only the boundary bin is kept), round 1 of the radix select (top 10 pattern bits, exact integers, 512-thread
//       DPP scan) names the boundary bin — an eighth of an octave of e.  <= 64 entries there (the usual row): every wave finishes by
//       itself from one pass over them (mass above, tie group, rank by index); more: two more radix rounds and a radix select on the
//       index for a cut tie group.  Own entries: p = keep ? e / Z_kept : 0.
//
// The select is the one of topp_probs_kernel on the same integers (mass = floor(e 2^40), tau = floor(top_p Z), boundary
// pattern, kept ties = lowest indices, Z_kept): the probabilities are BIT-IDENTICAL to tf_topp_probs (tests/test_gpu_ops.py).
// Control block (zero-filled once, owned by the caller; launches on it must not overlap): sticky error word, per-row arrival
// counters and READY generations (nothing is re-zeroed between launches: see tm_arrive).  Every wait is bounded (tf_topp_multi_tune key 0, ms); a time-out sets the error word and the
// rows are NaN-filled.
#include "select_common.h"

#define TM_THREADS 512
#define TM_MAX_ROWS 32
#define TM_MAX_SLICES 16
#define TM_LIST_CAP 16384                   // candidates the rounds keep in LDS (128 KiB)
#define TM_WALL_HZ 100000000ull

namespace {
struct TmCtl {                               // device memory, zero-filled once
    unsigned spare, error, pad0[14];
    unsigned cnt[2][TM_MAX_ROWS][16];        // arrival counters: edge A / B, row, one 64-byte line each
    unsigned flag[2][TM_MAX_ROWS][16];       // READY flags (= epoch + 1)
};

struct TmParams {
    const float* logits;
    const float* panel_max;                  // NULL, or [V / 16][32] per-panel row maxima of the logits (lm_head epilogue)
    float* probs;
    float* wgmax;                            // [rows][S]
    u64* zslice;                             // [rows][S]
    unsigned* ccount;                        // [rows][S]
    u64* cand;                               // [rows][S][slice capacity]  (pattern << 32) | index
    TmCtl* ctl;
    u64 timeout_ticks;
    int rows, V, S, slice_f4, skip_edge, list_cap;     // slice_f4: float4s per slice; list_cap: entries of the LDS list
    int rowmax;                              // 1: every workgroup takes the row maximum from the whole row (no edge A)
    float temperature, top_p;
};

struct TmShared {
    DpTopp tp;
    int fail;
    unsigned off[TM_MAX_SLICES + 1];         // candidate offsets of the row's slices
    float red[DP_WAVES];
    u64 red64[DP_WAVES];
    unsigned red32[DP_WAVES];
    u64 fin[64];                             // the boundary bin's entries (in-wave finish)
};

// An edge of ONE row: its S workgroups add 1 to the row's counter; the last arriver puts the counter back to 0 and advances the
// row's READY generation.  Every workgroup of the row read the generation at entry — before the flag can move, which takes all S
// arrivals, each behind its own read — so "flag != generation read at entry" is this launch's edge and nothing has to be
// re-zeroed or counted across launches (rows x S may differ from launch to launch).
__device__ __forceinline__ void tm_arrive(const TmParams& P, unsigned gen, int edge, int row, int slice, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const bool skip = P.skip_edge == edge + 1 && slice == 0 && row == 0;          // fault injection (tests)
        if (!skip) {
            const unsigned old = __hip_atomic_fetch_add(&P.ctl->cnt[edge][row][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == (unsigned)P.S) {
                st4u(&P.ctl->cnt[edge][row][0], 0u);
                st4u(&P.ctl->flag[edge][row][0], gen + 1u);
            }
        }
    }
}

__device__ __forceinline__ bool tm_wait(const TmParams& P, TmShared* sh, unsigned gen, int edge, int row, int tid) {
    if (tid < 64) {
        const unsigned* slot = &P.ctl->flag[edge][row][0];
        const u64 t0 = wall_clock64();
        bool fail = false;
        for (unsigned spins = 0;; ++spins) {
            if (ld4u(slot) != gen) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 31u) == 31u) {
                if (ld4u(&P.ctl->error) != 0u) { fail = true; break; }
                if (wall_clock64() - t0 > P.timeout_ticks) {
                    if (tid == 0) {
                        unsigned expected = 0u;
                        __hip_atomic_compare_exchange_strong(&P.ctl->error, &expected, (unsigned)(edge + 1), __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    fail = true;
                    break;
                }
            }
        }
        if (fail && tid == 0) sh->fail = 1;
    }
    __syncthreads();
    return sh->fail == 0;
}

__device__ __forceinline__ float tm_block_max(float v, TmShared* sh, int lane, int wave) {
    v = wave_max(v);
    if (lane == 0) sh->red[wave] = v;
    __syncthreads();
    float m = sh->red[0];
#pragma unroll
    for (int k = 1; k < DP_WAVES; ++k) m = fmaxf(m, sh->red[k]);
    __syncthreads();
    return m;
}

template <int NPT>                           // float4s per thread of a slice
__global__ __launch_bounds__(TM_THREADS) void topp_multi_kernel(TmParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tm_smem[];
    TmShared* sh = reinterpret_cast<TmShared*>(tm_smem);
    u64* list = reinterpret_cast<u64*>(tm_smem + ((sizeof(TmShared) + 15) & ~(size_t)15));    // [TM_LIST_CAP]
    DpTopp* tp = &sh->tp;
    const int S = P.S, row = blockIdx.x / S, slice = blockIdx.x % S, V = P.V;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = P.logits + (int64_t)row * V;
    float* pr = P.probs + (int64_t)row * V;
    const int nf4 = V / 4, f0 = slice * P.slice_f4;
    if (tid == 0) sh->fail = 0;
    for (int i = tid; i < 1024 + 64; i += TM_THREADS) {
        tp->hist[i] = 0ull;
        tp->cnt[i] = 0u;
    }
    const unsigned genA = ld4u(&P.ctl->flag[0][row][0]), genB = ld4u(&P.ctl->flag[1][row][0]);   // this row's READY generations
    const unsigned err0 = ld4u(&P.ctl->error);
    __syncthreads();
    // this thread's entries: float4 f0 + tid + 512 k of the row
    int fidx[NPT];
    bool fok[NPT];
    f32x4 x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int fl = tid + TM_THREADS * k;
        fidx[k] = f0 + fl;
        fok[k] = fl < P.slice_f4 && fidx[k] < nf4;
        x[k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (fok[k]) x[k] = *reinterpret_cast<const f32x4*>(lr + 4 * fidx[k]);
    }
    auto poison = [&]() {
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            if (fok[k]) *reinterpret_cast<f32x4*>(pr + 4 * fidx[k]) = f32x4{nanv, nanv, nanv, nanv};
    };
    if (err0 != 0u) {
        poison();
        return;
    }
    float lmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < NPT; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[k][j] = x[k][j] / P.temperature;                            // (entries past the slice stay -inf: zero mass)
            lmax = fmaxf(lmax, x[k][j]);
        }
    // ---- (a) the row maximum ----
    float mx;
    if (P.panel_max != nullptr) {
        // max over the row's V / 16 panel maxima of the LOGITS, then one division: x -> x / T is monotone, so max(l) / T is
        // max(l / T) bit for bit
        float pm = -INFINITY;
        for (int p = tid; p < V / 16; p += TM_THREADS) pm = fmaxf(pm, P.panel_max[p * 32 + row]);
        mx = tm_block_max(pm, sh, lane, wave) / P.temperature;
    } else if (P.rowmax) {
        // every workgroup reads the WHOLE row (128 KB from L2, 16 float4 loads in flight per thread): cheaper than an edge (the
        // arrival + the flag poll cost ~3 us) and max(l) / T is max(l / T) bit for bit
        float pm = -INFINITY;
        f32x4 r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int f = tid + TM_THREADS * k;
            r[k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (f < nf4) r[k] = *reinterpret_cast<const f32x4*>(lr + 4 * f);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) pm = fmaxf(pm, fmaxf(fmaxf(r[k][0], r[k][1]), fmaxf(r[k][2], r[k][3])));
        mx = tm_block_max(pm, sh, lane, wave) / P.temperature;
    } else {
        const float smax = tm_block_max(lmax, sh, lane, wave);
        if (tid == 0) st4f(P.wgmax + row * S + slice, smax);
        tm_arrive(P, genA, 0, row, slice, tid);
        if (!tm_wait(P, sh, genA, 0, row, tid)) {
            poison();
            return;
        }
        float m = -INFINITY;
        if (lane < S) m = ld4f(P.wgmax + row * S + lane);
        mx = wave_max(m);
    }
    // ---- (b) e, masses, the slice's candidates ----
    unsigned cutpat = 0u;
    if (P.top_p < 1.0f) cutpat = __float_as_uint((1.0f - P.top_p) / (2.0f * (float)V)) & ~((1u << 20) - 1u);
    u64 zacc = 0ull;
    unsigned mycnt = 0u;
#pragma unroll
    for (int k = 0; k < NPT; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float e = fok[k] ? expf(x[k][j] - mx) : 0.f;
            x[k][j] = e;
            const u64 m = dp_fix(e);
            zacc += m;
            mycnt += (m != 0ull && __float_as_uint(e) >= cutpat) ? 1u : 0u;
        }
    {
        const unsigned incl = dp_wave_prefix_u32(mycnt);
        const u64 zincl = dp_wave_prefix_u64(zacc);
        if (lane == 63) {
            sh->red32[wave] = incl;
            sh->red64[wave] = zincl;
        }
        __syncthreads();
        unsigned at = incl - mycnt, total = 0u;
        u64 z = 0ull;
#pragma unroll
        for (int k = 0; k < DP_WAVES; ++k) {
            if (k < wave) at += sh->red32[k];
            total += sh->red32[k];
            z += sh->red64[k];
        }
        u64* mine = P.cand + ((int64_t)row * S + slice) * (int64_t)(P.slice_f4 * 4);
#pragma unroll
        for (int k = 0; k < NPT; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned b = __float_as_uint(x[k][j]);
                if (dp_fix(x[k][j]) != 0ull && b >= cutpat) st8(mine + at++, ((u64)b << 32) | (unsigned)(4 * fidx[k] + j));
            }
        if (tid == 0) {
            st8(P.zslice + row * S + slice, z);
            st4u(P.ccount + row * S + slice, total);
        }
    }
    tm_arrive(P, genB, 1, row, slice, tid);
    if (!tm_wait(P, sh, genB, 1, row, tid)) {
        poison();
        return;
    }
    // ---- (c) the select ----
    // The slice counts / sums AND the first 32 candidates of every slice (half a wave each) are requested together: lanes past a
    // slice's count read stale workspace and drop it.  One round trip for a sharp row; a slice with more costs a second for the rest.
    constexpr int SPEC = 32;
    const int cap = P.slice_f4 * 4;                                      // candidate capacity of a slice
    const u64* cbase = P.cand + (int64_t)row * S * (int64_t)cap;
    const int sp_slice = 2 * wave + (lane >> 5), sp_en = lane & 31;
    u64 spec = 0ull;
    {
        unsigned c = 0u;
        u64 z = 0ull;
        if (tid < S) {
            c = ld4u(P.ccount + row * S + tid);
            z = ld8(P.zslice + row * S + tid);
        }
        if (sp_slice < S) spec = ld8(cbase + (int64_t)sp_slice * cap + sp_en);
        if (wave == 0) {
            const unsigned incl = dp_wave_prefix_u32(c);
            const u64 zin = dp_wave_prefix_u64(z);
            if (lane < S) sh->off[lane + 1] = incl;
            if (lane == 0) sh->off[0] = 0u;
            if (lane == 63) {
                const double t = (double)P.top_p * (double)zin;
                tp->Z = zin;
                tp->zk = zin;
                tp->tau = (t >= 18446744073709549568.0) ? ~0ull : __double2ull_rd(t);
                tp->S = 0ull;
                tp->digit = -1;
                tp->ties = 0u;
                tp->nkeep = 0ull;
                tp->nlist = 0u;
                tp->nfin = 0u;
            }
        }
    }
    __syncthreads();
    const unsigned C = sh->off[S];
    const u64 tau = tp->tau, Z = tp->Z;
    // Walk the row's candidates where they lie (per slice, coalesced; the loads of a batch in flight before the first is used).
    constexpr int NK = 4 * NPT;                                          // a slice holds at most 512 x NK entries
    unsigned maxc = 0u;
#pragma unroll
    for (int k = 0; k < TM_MAX_SLICES; ++k)
        if (k < S) maxc = max(maxc, sh->off[k + 1] - sh->off[k]);
    auto stream = [&](auto fn) {
        if (sp_slice < S && (unsigned)sp_en < sh->off[sp_slice + 1] - sh->off[sp_slice]) fn(spec, sh->off[sp_slice] + sp_en);
        if (maxc <= (unsigned)SPEC) return;
        if (maxc <= (unsigned)(SPEC + TM_THREADS)) {
            // no slice holds more than 32 + 512 candidates: ONE more entry per thread and slice, all S loads in flight together
            u64 v[TM_MAX_SLICES];
#pragma unroll
            for (int q = 0; q < TM_MAX_SLICES; ++q) {
                const unsigned cn = q < S ? (sh->off[q + 1] - sh->off[q]) : 0u;
                v[q] = (unsigned)(SPEC + tid) < cn ? ld8(cbase + (int64_t)q * cap + SPEC + tid) : 0ull;
            }
#pragma unroll
            for (int q = 0; q < TM_MAX_SLICES; ++q) {
                const unsigned cn = q < S ? (sh->off[q + 1] - sh->off[q]) : 0u;
                if ((unsigned)(SPEC + tid) < cn) fn(v[q], sh->off[q] + SPEC + tid);
            }
            return;
        }
        if (maxc <= (unsigned)(SPEC + 2 * TM_THREADS)) {                  // up to 32 + 1 024: two entries per thread and slice, still one trip
            u64 v[TM_MAX_SLICES][2];
#pragma unroll
            for (int q = 0; q < TM_MAX_SLICES; ++q) {
                const unsigned cn = q < S ? (sh->off[q + 1] - sh->off[q]) : 0u;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned j = SPEC + tid + TM_THREADS * k;
                    v[q][k] = j < cn ? ld8(cbase + (int64_t)q * cap + j) : 0ull;
                }
            }
#pragma unroll
            for (int q = 0; q < TM_MAX_SLICES; ++q) {
                const unsigned cn = q < S ? (sh->off[q + 1] - sh->off[q]) : 0u;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned j = SPEC + tid + TM_THREADS * k;
                    if (j < cn) fn(v[q][k], sh->off[q] + j);
                }
            }
            return;
        }
        for (int s0 = 0; s0 < S; s0 += 4) {
            u64 v[4][NK];
            unsigned cn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cn[q] = (s0 + q < S) ? (sh->off[s0 + q + 1] - sh->off[s0 + q]) : 0u;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const unsigned j = SPEC + tid + TM_THREADS * k;
                    v[q][k] = j < cn[q] ? ld8(cbase + (int64_t)(s0 + q) * cap + j) : 0ull;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const unsigned j = SPEC + tid + TM_THREADS * k;
                    if (j < cn[q]) fn(v[q][k], sh->off[s0 + q] + j);
                }
        }
    };
    // The working set of the rounds: ALL candidates when they fit the LDS list (a model's row: hundreds to a few thousand), else
    // round 1 streams them and only the boundary bin's entries are filed (near-flat rows); a boundary bin that still does not
    // fit (tens of thousands of equal entries) is streamed by every round.
    unsigned nlist = 0u;
    bool use_list = C <= (unsigned)P.list_cap;
    if (use_list) {
        stream([&](u64 e, unsigned at) { list[at] = e; });
        nlist = C;
        __syncthreads();
    }
    auto for_each = [&](auto fn) {
        if (use_list) {
            for (unsigned i = tid; i < nlist; i += TM_THREADS) fn(list[i]);
        } else {
            stream([&](u64 e, unsigned) { fn(e); });
        }
    };
    // one round: bins dig(pattern) of the candidates with sel(pattern) (+ how many fell in each when `count`); then the 512-thread
    // scan from base tp->S; tp->ties = the entries of the bin the scan names (count rounds only)
    auto round = [&](auto sel, auto dig, bool count) {
        for_each([&](u64 e) {
            const unsigned b = (unsigned)(e >> 32);
            if (sel(b)) {
                const int d = dig(b);
                atomicAdd(&tp->hist[DP_HB(d)], dp_fix(__uint_as_float(b)));
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
            tp->cnt[DP_HB(1023 - 2 * tid)] = 0u;
            tp->cnt[DP_HB(1022 - 2 * tid)] = 0u;
        }
    };
    // index of the nkeep-th tie by a two-round radix select on the 15-bit index (bins reversed: "from the top" = ascending index)
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
    u64 nkeep = 0ull, zk = Z;
    long long istar = -1;
    if (tau < Z) {                                                       // (tau >= Z: top_p >= 1 keeps everything)
        round([](unsigned) { return true; }, [](unsigned b) { return (int)(b >> 20); }, false);
        d1 = tp->digit;
        if (d1 < 0) {                                                    // the cut guarantees the crossing among the candidates
            if (tid == 0) {
                unsigned expected = 0u;
                __hip_atomic_compare_exchange_strong(&P.ctl->error, &expected, 99u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            poison();
            return;
        }
        const int dd1 = d1;
        // the boundary bin's entries (an eighth of an octave of e), gathered in any order — up to 64 of them; one counter bump
        // per wave and pass (a flat row has thousands of matches)
        u64* fin = sh->fin;
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
                stream([&](u64 e, unsigned) {
                    if ((int)((unsigned)(e >> 32) >> 20) == dd1) {
                        const unsigned at = atomicAdd(&tp->nlist, 1u);
                        if (at < (unsigned)P.list_cap) list[at] = e;
                    }
                });
                __syncthreads();
                nlist = tp->nlist;
                use_list = nlist <= (unsigned)P.list_cap;
            }
            round([dd1](unsigned b) { return (int)(b >> 20) == dd1; }, [](unsigned b) { return (int)((b >> 10) & 1023u); }, false);
            const unsigned pre = ((unsigned)d1 << 10) | (unsigned)tp->digit;
            round([pre](unsigned b) { return (b >> 10) == pre; }, [](unsigned b) { return (int)(b & 1023u); }, true);
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
    const float Zk = (float)((double)zk * (1.0 / 1099511627776.0));
    const bool rank_ties = d1 >= 0 && nkeep < (u64)ties;
    const unsigned ulow = d1 < 0 ? 0u : (rank_ties ? ustar + 1u : ustar);
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        if (!fok[k]) continue;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned b = __float_as_uint(x[k][j]);
            const long long idx = 4ll * fidx[k] + j;
            const bool keep = b >= ulow || (rank_ties && b == ustar && idx <= istar);
            o[j] = keep ? x[k][j] / Zk : 0.f;
        }
        *reinterpret_cast<f32x4*>(pr + 4 * fidx[k]) = o;
    }
}

int g_tm_timeout_ms = 2000;
int g_tm_skip_edge = 0;
inline int64_t tm_a256(int64_t v) { return (v + 255) & ~(int64_t)255; }
int g_tm_rowmax = 1;                       // 1: the row maximum from the whole row, no edge A (tf_topp_multi_tune key 3)
int g_tm_list_cap = 16384;                 // candidates the rounds keep in LDS (tf_topp_multi_tune key 2; <= TM_LIST_CAP)
inline size_t tm_lds_bytes(int cap) { return ((sizeof(TmShared) + 15) & ~(size_t)15) + (size_t)cap * 8; }
}  // namespace

extern "C" int64_t tf_topp_multi_ctl_bytes(void) { return (int64_t)sizeof(TmCtl); }

// slices per row and float4s per slice for `rows` rows of V entries; 0 when the launch does not take the shape
static int tm_plan(int rows, int V, int* slice_f4) {
    if (rows < 1 || rows > TM_MAX_ROWS || V < 64 || (V % 4) || V > 32768) return 0;
    const int S = rows <= 16 ? 16 : 8;
    const int nf4 = V / 4;
    const int per = (nf4 + S - 1) / S;
    if (per > TM_THREADS * 2 || (S == 16 && per > TM_THREADS)) return 0;
    *slice_f4 = per;
    return S;
}

extern "C" int64_t tf_topp_multi_ws_bytes(int rows, int V) {
    int per = 0;
    const int S = tm_plan(rows, V, &per);
    if (!S) return 0;
    return tm_a256((int64_t)rows * S * 4) + tm_a256((int64_t)rows * S * 8) + tm_a256((int64_t)rows * S * 4) +
           tm_a256((int64_t)rows * S * per * 4 * 8);
}

extern "C" int tf_topp_probs_multi(const float* logits, const float* panel_max, float* probs, int rows, int V, float temperature,
                                   float top_p, void* ctl, void* ws, int64_t ws_bytes, void* stream) {
    if (!logits || !probs || !ctl || !ws || !(temperature > 0.f) || !(top_p > 0.f)) return TF_EINVAL;
    int per = 0;
    const int S = tm_plan(rows, V, &per);
    if (!S) return TF_ERANGE;
    if (panel_max && (V % 16)) return TF_EINVAL;
    if (ws_bytes < tf_topp_multi_ws_bytes(rows, V)) return TF_ENOSPC;
    if ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(probs)) & 15) return TF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ws) % 256) || (reinterpret_cast<uintptr_t>(ctl) % 64)) return TF_EINVAL;
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return TF_EINVAL;
        cus = prop.multiProcessorCount;
    }
    if (rows * S > cus) return TF_ERANGE;                               // every workgroup must be resident
    TmParams P = {};
    P.logits = logits, P.panel_max = panel_max, P.probs = probs;
    char* p = static_cast<char*>(ws);
    P.wgmax = reinterpret_cast<float*>(p);       p += tm_a256((int64_t)rows * S * 4);
    P.zslice = reinterpret_cast<u64*>(p);        p += tm_a256((int64_t)rows * S * 8);
    P.ccount = reinterpret_cast<unsigned*>(p);   p += tm_a256((int64_t)rows * S * 4);
    P.cand = reinterpret_cast<u64*>(p);
    P.ctl = reinterpret_cast<TmCtl*>(ctl);
    P.timeout_ticks = (u64)(g_tm_timeout_ms > 0 ? g_tm_timeout_ms : 1) * (TM_WALL_HZ / 1000ull);
    P.rows = rows, P.V = V, P.S = S, P.slice_f4 = per, P.skip_edge = g_tm_skip_edge, P.list_cap = g_tm_list_cap;
    P.rowmax = (g_tm_rowmax != 0 && V <= 32768) ? 1 : 0;
    P.temperature = temperature, P.top_p = top_p;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)topp_multi_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tm_lds_bytes(TM_LIST_CAP));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)topp_multi_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tm_lds_bytes(TM_LIST_CAP));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (per <= TM_THREADS)
        hipLaunchKernelGGL(topp_multi_kernel<1>, dim3(rows * S), dim3(TM_THREADS), tm_lds_bytes(g_tm_list_cap), st, P);
    else
        hipLaunchKernelGGL(topp_multi_kernel<2>, dim3(rows * S), dim3(TM_THREADS), tm_lds_bytes(g_tm_list_cap), st, P);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// key 0: wall-clock limit of one wait in ms (default 2000); key 1: fault injection (edge + 1 whose first arrival of row 0 is lost);
// key 2: entries of the LDS candidate list (512 .. 16384 = the default: the launch is one workgroup per CU for <= 16 rows);
// key 3: 1 (default) every workgroup takes the row maximum from the whole row, 0 slice maxima through edge A
extern "C" int tf_topp_multi_tune(int key, int value) {
    int* slot = key == 0 ? &g_tm_timeout_ms : key == 1 ? &g_tm_skip_edge : key == 2 ? &g_tm_list_cap : key == 3 ? &g_tm_rowmax : nullptr;
    if (!slot) return -1;
    const int old = *slot;
    if (key == 0 && value < 1) return old;
    if (key == 1 && (value < 0 || value > 2)) return old;
    if (key == 2 && (value < 512 || value > TM_LIST_CAP)) return old;
    if (key == 3 && (value < 0 || value > 1)) return old;
    *slot = value;
    return old;
}

extern "C" int tf_topp_multi_error(const void* ctl) {
    if (!ctl) return TF_EINVAL;
    unsigned head[4];
    hipError_t e = hipMemcpy(head, ctl, sizeof(head), hipMemcpyDeviceToHost);
    return e == hipSuccess ? (int)head[1] : (int)e;
}

extern "C" int tf_topp_multi_reset(void* ctl) {
    if (!ctl) return TF_EINVAL;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    e = hipMemset(ctl, 0, sizeof(TmCtl));
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceSynchronize();
}
