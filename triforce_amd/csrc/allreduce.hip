// One-shot all-reduce over peer-mapped buffers (xGMI) for the decode-sized messages of the tensor-parallel path.
//
// Replaces dist.all_reduce at models/tensor_op.py:179,326,359 (after o_proj and after down_proj of every layer).  A
// decode forward issues 2 * L of them on 7..18 rows x hidden fp16 = 57..184 KB each: pure latency.  A ring collective
// pays 2 (W - 1) hops per call; here every rank stages its partial in a buffer its peers have mapped (hipIpc) and
// each rank reads ALL partials itself and adds them — one exchange of flags, one pass of remote reads:
//
//   producer GEMM  -> writes this rank's partial into its staging buffer (kernel boundary = visible system-wide)
//   phase READY    -> a rank tells every peer "my partial of epoch e is staged", waits for all peers' READY
//   reduce         -> out[i] = fp16( sum over ranks r = 0..W-1 (in THIS order, fp32) of staging_r[i] )
//   phase DONE     -> the last workgroup tells every peer "I have read your partial", waits for all peers' DONE:
//                     when the kernel ends the staging buffer may be overwritten by the next producer
//
// Alternating form (tf_allreduce_oneshot_alt, half_elems > 0): the staging buffer has two halves and exchange e uses
// half e & 1, which makes DONE unnecessary.  Rank B overwrites half e & 1 next when it produces the partial of exchange
// e + 2, i.e. after its kernel e + 1 ended; that kernel got past READY only once every peer A had signalled e + 1, which A
// does at the start of ITS kernel e + 1 — stream-ordered after A's kernel e, whose reads of B's half e & 1 are therefore
// over.  No rank can run more than one exchange ahead of another, so two halves suffice.  The half is taken from the
// device epoch (replays of a captured launch stay correct); the host, which had to point the producer at that half,
// passes the half it assumed and a disagreement is a sticky error (3) with NaN output, never a silent race.
//
// Every rank adds the same values in the same order, so all ranks hold bit-identical results (the reference's NCCL
// ring gives each rank the same bits too, with a different — sequential fp16 — rounding order; at world size 2 the two
// are identical: one correctly rounded fp16 addition).  The epoch lives in device memory and is advanced by the kernel
// itself, so a captured launch replays correctly.  Staging and flag buffers must be FINE-GRAINED device memory
// (tf_ar_alloc): peers' stores and loads bypass the caches; plain device memory is only coherent across GPUs at kernel
// boundaries.  Every spin is bounded (about 25 s; a peer may legitimately be
// seconds late right after start-up, when the ranks leave graph capture at different times); a
// timeout sets the sticky error word, after which every later call returns immediately instead of waiting again — and
// on every error path `out` is filled with NaN, so no token can be computed from a reduction that did not happen; the
// host polls tf_ar_error once per decode step (utils/decoding.py) and raises.
#include "common.h"
#include <string.h>

#define AR_MAX_WORLD 8
#define AR_THREADS 256
#define AR_SPIN_LIMIT (1u << 27)            // ~0.2 us per poll: 3 s measured at 1 << 24 -> about 25 s

// One rank's control block (fine-grained memory, mapped by every peer)
struct ArFlags {
    unsigned ready[AR_MAX_WORLD];     // ready[p] = last epoch for which peer p staged its partial   (written by p)
    unsigned done[AR_MAX_WORLD];      // done[p]  = last epoch whose reads of MY staging peer p finished (written by p)
    unsigned epoch;                   // completed all-reduces on this rank (owner only)
    unsigned ticket;                  // workgroups of the running launch that finished reducing (owner only)
    unsigned error;                   // sticky: 1 = READY wait timed out, 2 = DONE wait timed out, 3 = the producer
                                      //         staged into the other half than the epoch selects (alternating form)
    unsigned pad0;
    unsigned long long mirror;        // 0, or the address of a pinned HOST word that receives every error code as well
                                      // (tf_ar_set_error_mirror): the host then polls the error without a copy
    unsigned pad[10];
};
static_assert(sizeof(ArFlags) == 128, "control block layout");

// error <- code, also into the host mirror when one is registered (owner only)
__device__ __forceinline__ void ar_set_error(ArFlags* mine, unsigned code);

struct ArComm {
    const h16* data[AR_MAX_WORLD];    // staging buffer of every rank (own entry = local pointer)
    ArFlags* flags[AR_MAX_WORLD];     // control block of every rank
    int rank, world;
};

__device__ __forceinline__ unsigned ar_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ar_store(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void ar_set_error(ArFlags* mine, unsigned code) {
    ar_store(&mine->error, code);
    unsigned* m = reinterpret_cast<unsigned*>(mine->mirror);
    if (m) ar_store(m, code);
}

// lane p < world waits until *slot(p) reaches `epoch` (epochs only grow; wrap-safe compare); false on timeout
__device__ __forceinline__ bool ar_wait(const unsigned* slot, unsigned epoch) {
    for (unsigned spins = 0; spins < AR_SPIN_LIMIT; ++spins) {
        if ((int)(ar_load(slot) - epoch) >= 0) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// error path: out <- NaN (fp16 0x7e00), so a timed-out reduction can never pass for a result
__device__ __forceinline__ void ar_poison(h16* out, int64_t n_vec8) {
    half8 nan8;
#pragma unroll
    for (int e = 0; e < 8; ++e) nan8[e] = __builtin_bit_cast(h16, (unsigned short)0x7e00);
    for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < n_vec8; i += (int64_t)gridDim.x * AR_THREADS)
        *reinterpret_cast<half8*>(out + 8 * i) = nan8;
}

// out[piece i] = [resid +] fp16( sum over ranks, rank order, fp32 ) of the staged pieces i; returns what was stored
__device__ __forceinline__ half8 ar_reduce_piece(const ArComm& c, int64_t base, int64_t i, const h16* resid, h16* out) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    half8 v[AR_MAX_WORLD];
#pragma unroll
    for (int r = 0; r < AR_MAX_WORLD; ++r)
        if (r < c.world) v[r] = *reinterpret_cast<const half8*>(c.data[r] + base + 8 * i);
#pragma unroll
    for (int r = 0; r < AR_MAX_WORLD; ++r)
        if (r < c.world) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[r][e];
        }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)acc[e];
    if (resid) {                                      // hidden = residual + all_reduce(partial): fp16 add of the
        const half8 rv = *reinterpret_cast<const half8*>(resid + 8 * i);   // ROUNDED sum (tensor_op.py:179-181)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = hadd_rn(rv[e], o[e]);
    }
    *reinterpret_cast<half8*>(out + 8 * i) = o;
    return o;
}

__global__ __launch_bounds__(AR_THREADS) void allreduce_oneshot_kernel(ArComm c, const h16* resid, h16* out, int64_t n_vec8,
                                                                        float* ss_out, int hidden, int64_t half_elems,
                                                                        int expect_half, int pack_rows) {
    __shared__ unsigned s_epoch;
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    ArFlags* mine = c.flags[c.rank];
    if (tid == 0) {
        s_epoch = ar_load(&mine->epoch) + 1u;       // bumped only after every workgroup of this launch has read it
        s_ok = ar_load(&mine->error) == 0u;         // sticky: after one timeout every later call returns at once
    }
    __syncthreads();
    if (!s_ok) {                                     // (block-uniform) the host polls tf_ar_error once per decode step;
        ar_poison(out, n_vec8);                      // until it does, nothing computed from `out` may look plausible
        return;
    }
    const unsigned epoch = s_epoch;
    const int64_t base = half_elems * (int64_t)(epoch & 1u);       // alternating form: exchange e lives in half e & 1
    if (half_elems > 0 && (int)(epoch & 1u) != expect_half) {      // (block-uniform) the producer wrote the other half
        if (tid == 0) ar_set_error(mine, 3u);
        ar_poison(out, n_vec8);
        return;
    }
    // ---- READY: my partial was staged by the previous kernel in this stream ----
    if (blockIdx.x == 0 && tid < c.world) ar_store(&c.flags[tid]->ready[c.rank], epoch);
    if (tid < c.world && !ar_wait(&mine->ready[tid], epoch)) {
        ar_set_error(mine, 1u);
        s_ok = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");    // system scope: nothing read below may predate the flags
    __syncthreads();
    // ---- reduce: 16-byte vectors, fixed rank order, fp32 accumulation, one rounding ----
    if (s_ok && ss_out && pack_rows > 0) {
        // k-octet-major block (ops.Act, R = pack_rows rows): piece i = (k-octet i / R, row i % R).  A thread reduces the
        // TWO pieces of one (16-column panel, row) — i0 = 2 p R + m and i0 + R — and owns that entry of the hand-off.
        const int R = pack_rows;
        const int64_t pairs = n_vec8 >> 1;
        for (int64_t j = (int64_t)blockIdx.x * AR_THREADS + tid; j < pairs; j += (int64_t)gridDim.x * AR_THREADS) {
            const int64_t pnl = j / R;
            const int m = (int)(j - pnl * R);
            const int64_t i0 = 2 * pnl * R + m;
            float sq = 0.f;
#pragma unroll
            for (int hpiece = 0; hpiece < 2; ++hpiece) {
                const int64_t i = i0 + (int64_t)hpiece * R;
                const half8 o = ar_reduce_piece(c, base, i, resid, out);
#pragma unroll
                for (int e = 0; e < 8; ++e) sq = fmaf((float)o[e], (float)o[e], sq);
            }
            ss_out[pnl * 32 + m] = sq;
        }
    } else if (s_ok) {
        for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + tid; i < n_vec8; i += (int64_t)gridDim.x * AR_THREADS) {
            const half8 o = ar_reduce_piece(c, base, i, resid, out);
            if (ss_out) {
                // sum of squares of the 16-column panel this vector is half of -> ss_out[panel][row]: the hand-off the
                // RMSNorm prologue of the consuming GEMM folds (tf_skinny_gemm_ex ss_in) instead of re-reading the rows.
                // Vectors 2j and 2j+1 sit in adjacent lanes of one loop trip (n_vec8 and AR_THREADS are even).
                float sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sq = fmaf((float)o[e], (float)o[e], sq);
                const float other = __shfl_xor(sq, 1, 64);
                if ((i & 1) == 0) {
                    const int64_t col0 = (8 * i) % hidden, row = (8 * i) / hidden;
                    ss_out[(col0 >> 4) * 32 + row] = sq + other;
                }
            }
        }
    } else {
        ar_poison(out, n_vec8);                      // READY timed out: `out` would otherwise keep whatever it held
    }
    // ---- DONE: the last workgroup of this launch releases the peers' staging buffers and waits for mine ----
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        const unsigned t = __hip_atomic_fetch_add(&mine->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    if (half_elems == 0) {                           // single staging buffer: nobody may leave before every peer has read
        if (tid < c.world) ar_store(&c.flags[tid]->done[c.rank], epoch);
        if (tid < c.world && !ar_wait(&mine->done[tid], epoch)) ar_set_error(mine, 2u);
        __syncthreads();
    }
    if (tid == 0) {
        __hip_atomic_store(&mine->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ar_store(&mine->epoch, epoch);               // next launch (stream-ordered after this one) sees epoch + 1
    }
}

// ---- C ABI -------------------------------------------------------------------------------------------------------
extern "C" int tf_ar_flags_bytes(void) { return (int)sizeof(ArFlags); }
extern "C" int tf_ar_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// Fine-grained (cross-device coherent) device memory for staging / control blocks, zero-filled.
extern "C" int tf_ar_alloc(int64_t bytes, void** ptr) {
    if (!ptr || bytes < 1) return TF_EINVAL;
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(p, 0, (size_t)bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return (int)e;
    }
    *ptr = p;
    return TF_OK;
}

extern "C" int tf_ar_free(void* ptr) {
    if (!ptr) return TF_EINVAL;
    hipError_t e = hipFree(ptr);
    return e == hipSuccess ? TF_OK : (int)e;
}

extern "C" int tf_ar_get_ipc_handle(void* ptr, void* handle_out) {
    if (!ptr || !handle_out) return TF_EINVAL;
    hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), ptr);
    return e == hipSuccess ? TF_OK : (int)e;
}

extern "C" int tf_ar_open_ipc_handle(const void* handle, void** ptr_out) {
    if (!handle || !ptr_out) return TF_EINVAL;
    hipIpcMemHandle_t h;
    memcpy((void*)&h, handle, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? TF_OK : (int)e;
}

extern "C" int tf_ar_close_ipc_handle(void* ptr) {
    if (!ptr) return TF_EINVAL;
    hipError_t e = hipIpcCloseMemHandle(ptr);
    return e == hipSuccess ? TF_OK : (int)e;
}

// out[0, n) = sum over ranks of staging_r[0, n) (fp16, fp32 accumulation in rank order).  peer_data / peer_flags: `world`
// device-visible pointers (own entry included) to every rank's staging buffer / ArFlags block; this rank's partial must
// already be in peer_data[rank] (written by an earlier kernel of `stream`); `out` is ordinary device memory and must not
// alias the staging buffer.  n % 8 == 0.  Capturable: the epoch is kept in the control block.
// tf_allreduce_oneshot_add: out = resid + (sum over ranks), the residual added in fp16 to the rounded sum — the
// `hidden_states = residual + all_reduce(o)` of tensor_op.py:179-181,359-360 in the same launch.  resid may equal out
// (in-place residual stream); resid == NULL is the plain all-reduce.
static int ar_launch(void* const* peer_data, void* const* peer_flags, int rank, int world, const void* resid, void* out,
                     int64_t n, float* ss_out, int hidden, void* stream, int64_t half_elems = 0, int expect_half = 0,
                     int pack_rows = 0) {
    if (!peer_data || !peer_flags || !out || world < 1 || world > AR_MAX_WORLD || rank < 0 || rank >= world) return TF_EINVAL;
    if (n < 8 || (n % 8)) return TF_EINVAL;
    if (ss_out && (hidden < 16 || (hidden % 16) || (n % hidden) || n / hidden > 32)) return TF_EINVAL;
    if (pack_rows < 0 || (pack_rows > 0 && (!ss_out || (int64_t)pack_rows * hidden != n))) return TF_EINVAL;
    ArComm c;
    for (int r = 0; r < AR_MAX_WORLD; ++r) {
        c.data[r] = (r < world) ? (const h16*)peer_data[r] : nullptr;
        c.flags[r] = (r < world) ? (ArFlags*)peer_flags[r] : nullptr;
        if (r < world && (!c.data[r] || !c.flags[r])) return TF_EINVAL;
    }
    if ((const h16*)out == c.data[rank] || (const h16*)resid == c.data[rank]) return TF_EINVAL;
    if (half_elems < 0 || (half_elems % 8) || (half_elems > 0 && (n > half_elems || (expect_half & ~1)))) return TF_EINVAL;
    if (half_elems > 0 && ((const h16*)out == c.data[rank] + half_elems || (const h16*)resid == c.data[rank] + half_elems))
        return TF_EINVAL;
    c.rank = rank;
    c.world = world;
    const int64_t n_vec8 = n / 8;
    int blocks = (int)((n_vec8 + AR_THREADS - 1) / AR_THREADS);
    if (blocks > 64) blocks = 64;                     // <= 64 workgroups: co-resident with anything, latency-bound anyway
    hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(blocks), dim3(AR_THREADS), 0, (hipStream_t)stream, c,
                       (const h16*)resid, (h16*)out, n_vec8, ss_out, hidden, half_elems, expect_half, pack_rows);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_allreduce_oneshot_add(void* const* peer_data, void* const* peer_flags, int rank, int world,
                                        const void* resid, void* out, int64_t n, void* stream) {
    return ar_launch(peer_data, peer_flags, rank, world, resid, out, n, nullptr, 0, stream);
}

// ... and the sums of squares of the result rows, per 16-column panel: out is [n / hidden][hidden], ss_out[panel * 32 + row]
// (fp32, hidden / 16 panels x 32 rows — the layout tf_skinny_gemm_ex writes as ss_out and folds as ss_in).  Lets the
// tensor-parallel decode layer keep the single-GPU engine's fused form: the RMSNorm that follows an all-reduce runs in the
// prologue of the next GEMM without another pass over the residual stream.  n / hidden <= 32 rows.
extern "C" int tf_allreduce_oneshot_add_ss(void* const* peer_data, void* const* peer_flags, int rank, int world,
                                           const void* resid, void* out, int64_t n, int hidden, float* ss_out,
                                           void* stream) {
    if (!ss_out) return TF_EINVAL;
    return ar_launch(peer_data, peer_flags, rank, world, resid, out, n, ss_out, hidden, stream);
}

// Alternating form (see the header comment): every rank's staging buffer holds 2 * half_elems values and exchange e uses
// the half e & 1 — no DONE phase.  `expect_half` is the half this rank's producer wrote (the caller counts its exchanges:
// the first one after allocation uses half 1); resid and ss_out may be NULL.  All ranks must use the same form for
// the lifetime of a control block.
extern "C" int tf_allreduce_oneshot_alt(void* const* peer_data, void* const* peer_flags, int rank, int world,
                                        const void* resid, void* out, int64_t n, int hidden, float* ss_out,
                                        int64_t half_elems, int expect_half, void* stream) {
    if (half_elems < 8) return TF_EINVAL;
    return ar_launch(peer_data, peer_flags, rank, world, resid, out, n, ss_out, hidden, stream, half_elems, expect_half);
}

// Either form (half_elems == 0: READY / reduce / DONE; > 0: alternating halves) for a residual stream kept K-OCTET-MAJOR
// (include/triforce_hip.h, tf_skinny_gemm_act; pack_rows = R rows, n = R * hidden): the sum itself is layout-blind, the
// sum-of-squares hand-off needs to know which pieces make up (panel, row).  ss_out is required (without it the plain
// entry points serve any layout).
extern "C" int tf_allreduce_oneshot_act(void* const* peer_data, void* const* peer_flags, int rank, int world,
                                        const void* resid, void* out, int64_t n, int hidden, int pack_rows, float* ss_out,
                                        int64_t half_elems, int expect_half, void* stream) {
    if (pack_rows < 1 || !ss_out) return TF_EINVAL;
    return ar_launch(peer_data, peer_flags, rank, world, resid, out, n, ss_out, hidden, stream, half_elems, expect_half,
                     pack_rows);
}

extern "C" int tf_allreduce_oneshot(void* const* peer_data, void* const* peer_flags, int rank, int world, void* out,
                                    int64_t n, void* stream) {
    return tf_allreduce_oneshot_add(peer_data, peer_flags, rank, world, nullptr, out, n, stream);
}

// Sets the sticky error word of a control block from the host (0 clears it): fault injection for the tests of the
// error path (NaN-filled outputs, the per-step host poll, the bench line's "allreduce_error" field).
extern "C" int tf_ar_inject_error(void* flags_local, int code) {
    if (!flags_local || code < 0) return TF_EINVAL;
    const unsigned v = (unsigned)code;
    hipError_t e = hipMemcpy(&reinterpret_cast<ArFlags*>(flags_local)->error, &v, sizeof(v), hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
    unsigned long long m = 0;
    e = hipMemcpy(&m, &reinterpret_cast<ArFlags*>(flags_local)->mirror, sizeof(m), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (m) *reinterpret_cast<volatile unsigned*>(m) = v;       // the mirror is host memory
    return TF_OK;
}

// Registers (or with NULL removes) a pinned, device-mapped HOST word that every error code is also written to, so the
// decode loop's per-step health poll is a plain host read instead of a blocking copy of the control block.  The word must
// stay allocated for the lifetime of the control block; it is set to the block's current error value here.
extern "C" int tf_ar_set_error_mirror(void* flags_local, void* host_word) {
    if (!flags_local) return TF_EINVAL;
    ArFlags* f = reinterpret_cast<ArFlags*>(flags_local);
    unsigned cur = 0;
    hipError_t e = hipMemcpy(&cur, &f->error, sizeof(cur), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (host_word) *reinterpret_cast<volatile unsigned*>(host_word) = cur;
    const unsigned long long m = (unsigned long long)(uintptr_t)host_word;
    e = hipMemcpy(&f->mirror, &m, sizeof(m), hipMemcpyHostToDevice);
    return e == hipSuccess ? TF_OK : (int)e;
}

// Error word of a control block (0 = never timed out); host-side read for the self-check and the per-step poll.
extern "C" int tf_ar_error(const void* flags_local) {
    if (!flags_local) return TF_EINVAL;
    ArFlags f;
    hipError_t e = hipMemcpy(&f, flags_local, sizeof(f), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    return (int)f.error;
}

// Completed exchanges of this rank's control block (the device-side epoch): a blocking host read, for the caller that
// counts its exchanges (the alternating form's `expect_half`) to resynchronise after a launch or capture it issued failed.
// Call it with the stream idle.  Negative: -(hip error).
extern "C" int64_t tf_ar_epoch(const void* flags_local) {
    if (!flags_local) return TF_EINVAL;
    ArFlags f;
    hipError_t e = hipMemcpy(&f, flags_local, sizeof(f), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return -(int64_t)e;
    return (int64_t)f.epoch;
}

// ---- message-passing litmus for the exchange (tools/xgmi_litmus.py; DESIGN section 12.5 assumptions 1-3) ----------------
// tf_ar_litmus_stage: the "producer GEMM" of the litmus — PLAIN stores of a pattern that changes every iteration into
// this rank's staging buffer, by an ordinary kernel that precedes the exchange in stream order (exactly how the o_proj /
// down_proj epilogues stage their partials).  it_dev is a device counter the kernel of rank-local thread 0 of block 0
// one-thread launch advances FIRST (so a captured launch stages fresh data on every replay); small integer values:
//     v_r[i] = (7 i + 13 it + 101 r) mod 509        (sum over <= 8 ranks < 4 072: exact in the fp32 accumulation,
//                                                      and the fp16 result is exact up to 2 048 — the check below
//                                                      compares against the correctly rounded fp16 of the exact sum)
// tf_ar_litmus_check: after the exchange, compares out[i] with fp16(sum over ranks of v_r[i]) for the SAME iteration and
// adds the number of mismatching elements to *bad (device counter, never reset by the kernel); a stale staging line, a
// READY flag that overtook its partial, or a torn read all show up as a count, not as a hang.
__global__ __launch_bounds__(256) void ar_litmus_stage_kernel(h16* __restrict__ staging, int64_t n, int rank,
                                                              const unsigned* __restrict__ it_dev) {
    const unsigned it = *it_dev;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        staging[i] = (h16)(float)((7u * (unsigned)i + 13u * it + 101u * (unsigned)rank) % 509u);
}
// the counter is advanced by its own one-thread launch in front of the stage kernel: every block of the stage / check
// kernels of one iteration then reads the same value
__global__ void ar_litmus_advance_kernel(unsigned* it_dev) { *it_dev += 1u; }

__global__ __launch_bounds__(256) void ar_litmus_check_kernel(const h16* __restrict__ out, int64_t n, int world,
                                                              const unsigned* __restrict__ it_dev,
                                                              unsigned long long* __restrict__ bad) {
    const unsigned it = *it_dev;                       // already advanced past the iteration being checked
    unsigned mism = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float acc = 0.f;
        for (int r = 0; r < world; ++r) acc += (float)((7u * (unsigned)i + 13u * it + 101u * (unsigned)r) % 509u);
        if ((float)out[i] != (float)(h16)acc) ++mism;
    }
    mism += __shfl_xor(mism, 32, 64);
    mism += __shfl_xor(mism, 16, 64);
    mism += __shfl_xor(mism, 8, 64);
    mism += __shfl_xor(mism, 4, 64);
    mism += __shfl_xor(mism, 2, 64);
    mism += __shfl_xor(mism, 1, 64);
    if ((threadIdx.x & 63) == 0 && mism) atomicAdd(bad, (unsigned long long)mism);
}

// stage: it_dev is advanced (by its own one-thread launch) BEFORE the pattern is written, so stage / exchange / check of
// one iteration all see the same value.
extern "C" int tf_ar_litmus_stage(void* staging, int64_t n, int rank, uint32_t* it_dev, void* stream) {
    if (!staging || !it_dev || n < 1 || rank < 0 || rank >= AR_MAX_WORLD) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ar_litmus_advance_kernel, dim3(1), dim3(1), 0, st, it_dev);
    TF_LAUNCH_CHECK();
    int blocks = (int)((n + 255) / 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(ar_litmus_stage_kernel, dim3(blocks), dim3(256), 0, st, (h16*)staging, n, rank, it_dev);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_ar_litmus_check(const void* out, int64_t n, int world, const uint32_t* it_dev, uint64_t* bad,
                                  void* stream) {
    if (!out || !it_dev || !bad || n < 1 || world < 1 || world > AR_MAX_WORLD) return TF_EINVAL;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(ar_litmus_check_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const h16*)out, n, world,
                       it_dev, (unsigned long long*)bad);
    TF_LAUNCH_CHECK();
    return TF_OK;
}
