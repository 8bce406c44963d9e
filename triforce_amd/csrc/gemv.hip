// Skinny (decode) GEMM for gfx950: y[M,N] = x[M,K] . W[N,K]^T with M <= 32 rows — the weight-streaming side of
// every decode forward (reference: nn.Linear / F.linear at models/modeling_llama.py:156-159,212-214,243,408,
// models/tensor_op.py:140-142,175,353-357).  A 7B forward streams 13.2 GB of weights against <= 18 activation
// rows, so the kernel is a pure HBM stream; the matrix core is used only because it performs the K-reduction
// without any cross-lane traffic.
//
//   * Weights are PRE-PACKED once at load time into MFMA-operand order: for every panel of 16 output rows and
//     every 32-wide k-chunk, the 16x32 tile is stored as 64 consecutive 16-byte pieces, piece (g*16 + i) holding
//     W[n0+i][k0+8g .. k0+8g+7].  A wavefront's A-operand load is then ONE fully contiguous 1 KiB read and a
//     panel is a single sequential stream of K*32 bytes — no 64-byte row fragments, no LDS staging.
//   * grid = N/16 panels, the waves of a workgroup split the panel's K range; partial 16x16 accumulators meet in
//     LDS.  x (<= 32 x K fp16, L2-resident) is read straight into the B operand.
//
// Everything that used to sit BETWEEN the GEMMs of a decoder layer as its own launch (4-5 us each at <= 18 rows:
// pure launch latency) is folded into the GEMM that consumes or produces it, with the reference's rounding
// points kept:
//   * NORM prologue — RMSNorm of the input rows (modeling_llama.py:138-143): every workgroup reduces the rows'
//     sum of squares itself (x is <= 64 KiB and L2-resident; the first weight chunks are already in flight) and
//     normalises its B operand on the fly: h = w_ln * fp16(x * rsqrt(mean(x^2) + eps)).
//   * residual epilogue — y = fp16(res + fp16(acc))  (hidden_states = residual + o_proj(...), :278, :284).
//   * SwiGLU epilogue (gate|up pair, :156-159) — act = fp16(silu(fp16 gate)) * fp16 up.
//   * RoPE + KV-append epilogue (fused q|k|v, :212-238): the packed row order pairs rotary partners d and d + D/2
//     inside one 16-row panel, so the epilogue rotates q / k in registers (one cross-lane exchange) and writes q
//     to [M][H][D] and the k / v rows straight into the cache slots.
#include "common.h"

#ifdef TF_NO_NT
#define SG_LOAD(p) (*(p))
#else
#define SG_LOAD(p) __builtin_nontemporal_load(p)   // weights are read once per forward: non-temporal
#endif

// Waves per workgroup = K-splits of one 16-row panel.  A wave keeps SG_U KiB of weights in flight, so a grid of
// few panels (o_proj / down_proj: N = 4096 -> 256 workgroups, one per CU) needs more waves per panel to cover the
// HBM latency-bandwidth product (cold-cache graph replay, tools/tune.py): o_proj 9.0 -> 7.6 us, down_proj 20.8 ->
// 18.3 us with 8 waves per panel; 16 waves measured the same as 8.
#ifndef SG_WAVES
#define SG_WAVES 4
#endif
#ifndef SG_WAVES_WIDE
#define SG_WAVES_WIDE 8
#endif
#ifndef SG_WIDE_MAX_PANELS
#define SG_WIDE_MAX_PANELS 512   // <= 2 workgroups per CU -> use the wide (more K-splits) variant
#endif
#ifndef SG_U
#define SG_U 4              // k-chunks (KiB of weights) in flight per wave
#endif
#ifndef SG_TAIL_BATCH
#define SG_TAIL_BATCH 1         // 1: the last 1..U-1 k-chunks of a wave as one batch of loads (round 4); 0: one chunk at a time
#endif
#ifndef SG_LN_PRE
#define SG_LN_PRE 0             // 1: the first batch's norm weights are loaded with the prologue too (+16 registers: the one-tile
                                // gate|up form drops from 4 to 3 waves per SIMD)
#endif
#ifndef SG_EPI_LATE
#define SG_EPI_LATE 1           // forms without a norm prologue fetch their epilogue operands behind the first weight batch
#endif
#ifndef SG_PROLOGUE_ORDER
#define SG_PROLOGUE_ORDER 2     // 2: norm partials, x rows, norm weights, weights in ONE basic block (unconditional loads,
                                // pinned in front of the fold); 1: the same order behind branches — which the compiler
                                // threads into "partials, WAIT, fold, then the rest"; 0: weights first, x after the fold (round 3)
#endif
#ifndef TF_SG_RES_EARLY
#define TF_SG_RES_EARLY 1   // residual epilogue operands fetched before the weight loop (0: at the tail, round 2's form)
#endif

enum { SG_PLAIN = 0, SG_GATEUP = 1, SG_F32 = 2, SG_QKV = 3 };

// Activation layout (round 4).  Every activation operand is addressed through two element strides,
//     element (m, k) at base[m * sm + (k / 8) * sk + (k % 8)],
// so one kernel serves the reference's row-major [M][ld] blocks (sm = ld, sk = 8) and the k-octet-major form
// (sm = 8, sk = 8 * R, R >= M): the 16-byte pieces of one k-octet of all rows are contiguous, a 32-wide k-chunk of an
// M-row block is ONE contiguous run of 4 * R pieces.  The B operand of a 16-row tile is then 4 runs of 256 B instead of
// 16 row fragments of 64 B (twice the cache lines per load instruction), which is what bounded the 17...32-row GEMMs of the
// gamma = 16 verifies (profiles/r03_gemm_rows_ab.jsonl).
struct SgAct {
    int64_t sm, sk;
};

// P = panels per wave: a wave that multiplies P weight panels against ONE B operand reads x once per P KiB of weights
// (x traffic through L2 -> L1 is panels * M * K * 2 bytes: as large as the weight stream itself at 17 rows with P = 1).
// Measured (profiles/r04_gemm_layout_ab.jsonl, k-octet-major x): P = 2 pays when the halved grid still puts two
// workgroups on most CUs — 13B q|k|v (480 groups) 32.0 -> 27.6 us at 17 rows, 26.9 -> 25.2 at 8; 13B gate|up (432)
// 49.8 -> 47.5 / 48.4 -> 44.7 — and LOSES below that: 7B q|k|v (384 groups) 20.5 -> 23.4, 7B gate|up (344) 32.2 -> 35.4.
// With 4 waves per workgroup it is bit-identical to the P = 1 form (same K ranges per wave, same merge order).
#ifndef SG_P_WIDE_MIN_GROUPS
#define SG_P_WIDE_MIN_GROUPS 420
#endif
// K split ACROSS workgroups (round 4).  A GEMM with few output panels — the q|k|v and gate|up shards of a tensor-parallel
// rank: 96 / 86 panels at 7B TP 8 — ran one workgroup per panel, i.e. on 86-96 of the 256 CUs: 13.2 / 9.1 us for 22.5 /
// 12.6 MB (1.7 / 1.4 TB/s, profiles/r04_tp8_7b_kernel_timeline_before.json).  With gridDim.y = KS workgroups per panel
// group each streams 1 / KS of the panel's K range; the per-panel partial sums (1 KiB per row tile) meet through a
// workspace: written with agent-scope write-through stores, drained, one ticket per panel — the LAST workgroup to arrive
// reads all KS partials back (agent-scope loads), adds them IN SPLIT ORDER (deterministic whoever arrives last) and runs
// the epilogue.  Same hand-off as the one-launch attention merge (csrc/attn.hip).  The workspace (tickets first) belongs
// to the caller: tf_sg_workspace registers one per device; without it KS = 1.
#define SG_KSPLIT_MAX 4
#define SG_TICKETS 4096
struct SgKsplit {
    float* ws;            // [panel][KS][NA][MT][64][4] fp32
    unsigned* tickets;    // [SG_TICKETS], zero between launches
    unsigned long long* stamps;   // -DSG_STAMPS=1 builds only (tools/gemm_stamps.py): 16 cycle stamps per workgroup
};
// Phase stamps of wave 0 of every workgroup (instrumented build only): 0 entry, 1 prologue loads issued, 2 norm scale known,
// 3 first k-batch done, 4 K loop done, 5 waves merged, 6 exit; 8 / 9 = 100 MHz wall clock at entry / exit.
#ifndef SG_STAMPS
#define SG_STAMPS 0
#endif
#if SG_STAMPS
#define SG_STAMP(i)                                                                                                      \
    do {                                                                                                                 \
        if (tid == 0 && kx.stamps) {                                                                                     \
            kx.stamps[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (i)] = __builtin_readcyclecounter();        \
            if ((i) == 0 || (i) == 6) kx.stamps[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 8 + (i) / 6] = wall_clock64(); \
        }                                                                                                                \
    } while (0)
#else
#define SG_STAMP(i) do { } while (0)
#endif
static void* g_sg_ws[16] = {};
static int64_t g_sg_ws_bytes[16] = {};
// The workspace is cut into SG_WS_SLOTS equal slots (tickets + partials each), one per LAUNCH STREAM, first come first served:
// split GEMMs issued on different streams of one device (virtual ranks of the tests, a side stream) can then run concurrently
// without meeting in one ticket array (advisor, round 4).  A fifth stream gets no slot: its GEMMs run unsplit.  (A captured
// launch keeps the slot of its capture stream; graphs of one engine replay one at a time.)
#define SG_WS_SLOTS 4
static hipStream_t g_sg_slot_stream[16][SG_WS_SLOTS] = {};
static int g_sg_slots_used[16] = {};
#include <mutex>
static std::mutex g_sg_slot_mutex;
static int sg_slot_of(int dev, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_sg_slot_mutex);
    for (int i = 0; i < g_sg_slots_used[dev]; ++i)
        if (g_sg_slot_stream[dev][i] == st) return i;
    if (g_sg_slots_used[dev] >= SG_WS_SLOTS) return -1;
    g_sg_slot_stream[dev][g_sg_slots_used[dev]] = st;
    return g_sg_slots_used[dev]++;
}
// Measured (profiles/r04_tp_shard_structural_ab.jsonl, r04_tp8_7b_kernel_timeline_after_splitk.json): NO gain in situ — the
// 7B TP-8 gate|up GEMM stays at 13.2 us with 258 workgroups instead of 86, q|k|v goes 9.1 -> 10.9 us: at 12-22 MB these
// launches are made of fixed costs (dispatch, the norm prologue's dependent loads, merge, epilogue, drain), not of the
// stream the extra CUs would shorten, and the hand-off adds a round trip.  Default off (0); tf_sg_tune key 3 turns it on.
// Measured (profiles/r04_gemm_deep_prefetch_ab.jsonl) and OFF: twice the bytes per trip halves the trips but not the time —
// 7B TP-8 gate|up K loop 7.3 -> 6.4 us with the prologue 0.8 us longer, workgroup lifetime 11.0 us either way; retrieval verify
// 1 671 -> 1 685 us, 13B TP 8 3 109 -> 3 200.  The K loop of a few-panel GEMM is bound by what ONE CU can pull (~37 GB/s:
// 256 KiB in 7 us, whatever is in flight), not by the number of round trips.
static int g_sg_deep_panels = 0;           // grids of up to this many panels (and K >= 2048) keep twice the weights in flight per wave (key 6)
static int g_sg_few_panels = 200;          // gate|up GEMMs of up to this many panels run 8 waves per panel at two row tiles (tf_sg_tune key 5)
static int g_sg_ksplit_force = 0;          // tf_sg_tune key 4 (A/B): > 1 that many K-splits across workgroups for EVERY P = 1 GEMM,
                                           // 1 never split, 0 the rule in sg_pick_ksplit
static int g_sg_ksplit_max_groups = 0;     // split K across workgroups below this many panel groups (tf_sg_tune key 3; 0 = never)
static int g_sg_n8_u = 0;                  // narrow-panel form (skinny_gemm_n8_kernel), tf_sg_tune key 7 (A/B): 0 = the rule in
                                           // launch_sg_n8, 5 / 8 = that many super-chunks per batch for every launch

__device__ __forceinline__ void sg_st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sg_ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// GEMM + exchange in ONE launch (round 4; tf_skinny_gemm_xchg).  The tensor-parallel layer's o_proj / down_proj are
// followed by an all-reduce of their (rows x hidden) partial outputs (models/tensor_op.py:179-181,359-360); as its own
// launch that exchange is 6-7 us on the device before any xGMI hop — a chain of uncached round trips (epoch, READY, the
// staged partials, ticket, DONE) behind a launch boundary (profiles/r04_tp8_7b_kernel_timeline_before.json) — twice per
// layer of a 60 us layer.  Here every workgroup exchanges ITS OWN 16-column panel with the same workgroup of the other
// ranks and nobody waits for a whole grid:
//   partial panel (fp16) -> own staging half (system-scope 8-byte stores), drained        [what the GEMM epilogue wrote anyway]
//   flag[rank][panel] = epoch on every peer; wait for the peers' flags of this panel      [one hop, per panel]
//   read the peers' panels, add in rank order (fp32), round, + residual, store, sums of squares   [one remote round trip]
// Exchange e OF A PANEL uses staging half e & 1 (every panel counts its own exchanges on the device): rank B rewrites a
// panel's half at that panel's exchange e + 2, after its exchange e + 1, which waited for every peer's flag e + 1 — set
// after that peer's exchange e of the panel had finished reading.  No DONE phase, no host-side half bookkeeping (producer and exchange are one kernel).  Same
// arithmetic as GEMM -> tf_allreduce_oneshot_add: fp16 partials, fp32 sum in rank order, one rounding, fp16 residual add.
// Every spin is bounded; a time-out poisons the panel with NaN and sets the sticky error word (+ host mirror).
#define XC_MAXP 512                          // panels (N / 16) an exchange GEMM may have
#define XC_MAX_WORLD 8
#define XC_SPIN_LIMIT (1u << 27)            // hard cap on polls; the wall-clock limit below ends a wait long before it
#define XC_WALL_HZ 100000000ull              // wall_clock64(): 100 MHz
// A peer that never flags (dead rank; virtual ranks of one device that do not fit the chip together and starve each other)
// is reported after g_xc_timeout_ms of WALL time — every poll is an uncached round trip of ~1-2 us, so a poll count alone
// stood for minutes (advisor, round 4).  tf_xchg_tune key 1.
static int g_xc_timeout_ms = 5000;
// 1: the exchange runs with the release / acquire FENCES the first build had (system-scope release before the flag stores,
// system-scope acquire = buffer_inv sc0 sc1 behind the flag wait, ~1.7 us each) instead of relying on s_waitcnt vmcnt(0) +
// system-scope (sc0 sc1) accesses in issue order.  The engine's start-up litmus (utils/oneshot_ar.GemmExchange.litmus)
// runs both forms on the REAL group and selects this one on any mismatch of the fence-free form.  tf_xchg_tune key 0.
static int g_xc_fence = 0;
struct XcCtl {                               // head of a rank's control buffer (fine-grained memory); flags follow at +1024 B,
    unsigned epoch, ticket, error, pad;      // then this rank's own per-panel epochs (XC_PEPOCH_OFF); epoch / ticket unused since
    unsigned long long mirror;               // the epochs went per panel.  mirror: 0 or a pinned host word that also gets error codes
};
#define XC_PEPOCH_OFF (1024 + XC_MAX_WORLD * XC_MAXP * 4)
struct SgXchg {
    h16* stage[XC_MAX_WORLD];                // every rank's staging buffer (2 halves), own entry = local pointer
    unsigned* pf[XC_MAX_WORLD];              // every rank's flag array [world][XC_MAXP]: pf[r][q * XC_MAXP + p] written by rank q
    XcCtl* ctl;                              // own control block
    unsigned* pepoch;                        // own per-panel exchange counts [XC_MAXP] (no peer reads them)
    int64_t half_elems;
    int rank, world;
    int fence;                               // see g_xc_fence
    unsigned long long timeout_ticks;        // wall-clock budget of one flag wait
};
__device__ __forceinline__ unsigned xc_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void xc_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void xc_st8(h16* p, half4 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ half4 xc_ld8(const h16* p) {
    return __builtin_bit_cast(half4, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_SYSTEM));
}

static int g_sg_p2_rows = 1;       // P = 2 from this many rows (tf_sg_tune key 0; 33 = never)
static int g_sg_p2_waves = 4;      // waves per workgroup of the P = 2 form (key 1)
static int g_sg_p2_groups = SG_P_WIDE_MIN_GROUPS;   // ... while panels / 2 >= this (key 2)

struct SgRope {                       // arguments of the RoPE + KV-append epilogue (SG_QKV)
    const h16* cosb;                  // [max_pos][D] fp16
    const h16* sinb;
    const int64_t* positions;         // [M]
    h16* q_out;                       // [M][H][D]
    h16* k_cache;                     // layer base, element strides below
    h16* v_cache;
    int64_t stride_t, stride_h;
    const int32_t* slot0_dev;
    int slot0, H, D, rotate_k;
};

// h = w_ln * fp16(x * inv): the cast precedes the weight multiply (modeling_llama.py:141-143); the fp16 product of
// two fp16 values rounded once is the native half multiply.
__device__ __forceinline__ half8 sg_normalise(half8 xv, half8 wv, float inv) {
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = hmul_rn(wv[e], (h16)((float)xv[e] * inv));
    return o;
}

template <int MT, int MODE, bool NORM, int WAVES, int P, bool KSPLIT, bool XCHG = false, int UX = 1>
// Argument order: what the PROLOGUE needs comes first, as plain pointers / scalars — weights, x and its two strides, the
// norm partials and weights, K, M: 14 dwords, the most the hardware preloads into SGPRs with the dispatch (built with
// -mllvm -amdgpu-kernarg-preload-count=14, triforce_amd/build.py): the first loads are issued without waiting for the
// kernel-argument segment to arrive through the scalar cache; everything the epilogue needs follows (by-value structs end
// the preloadable run).
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(const half8* __restrict__ wp,
                                                                 const half8* __restrict__ wp_up,
                                                                 const h16* __restrict__ x,
                                                                 const float* __restrict__ ss_in,
                                                                 const h16* __restrict__ ln_w, int K, int M,
                                                                 int xa_sm, int xa_sk, float eps, int N,
                                                                 const h16* resid, SgAct ra, void* yv, SgAct ya,
                                                                 SgRope rp, float* __restrict__ ss_out, SgKsplit kx,
                                                                 SgXchg xc) {
    const SgAct xa = {(int64_t)xa_sm, (int64_t)xa_sk};          // (32-bit in the argument list: two more preloaded dwords)
    static_assert(!XCHG || (MODE == SG_PLAIN && P == 1 && !KSPLIT), "the exchange form is the plain one-panel GEMM");
    constexpr bool GATEUP = MODE == SG_GATEUP;
    constexpr int NA = GATEUP ? 2 : 1;                       // weight streams (accumulator sets) per panel
    // k-chunks in flight per wave and weight stream: 8 KiB of weights per wave (4 for P = NA = 1) — times UX.  UX = 2 was
    // built for the FEW-PANEL grids (a tensor-parallel rank's shards), where a wave's K share looks like a chain of load ->
    // wait -> MFMA round trips of ~0.9 us — 7.5 of the 11.1 us a 7B TP-8 gate|up workgroup lives (tools/gemm_stamps.py,
    // profiles/r04_gemm_phase_stamps.json).  Measured: half the trips take as long (see g_sg_deep_panels) — off by default.
    constexpr int U = ((P * NA >= 2) ? (8 / (P * NA)) : SG_U) * UX;
    static_assert(P <= WAVES && U >= 1, "one epilogue wave per panel");
    const int panel0 = blockIdx.x * P;                       // this workgroup's P consecutive panels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    SG_STAMP(0);
    const int nchunks = K >> 5;
    // this workgroup's share of K (gridDim.y = KS workgroups per panel group), then its waves' shares of that
    const int KS = KSPLIT ? (int)gridDim.y : 1, ks = KSPLIT ? (int)blockIdx.y : 0;   // (own instantiation: the fold below
                                                                                      //  costs the one-workgroup form registers)
    const int g0c = (int)((int64_t)nchunks * ks / KS), g1c = (int)((int64_t)nchunks * (ks + 1) / KS);
    const int cpw = (g1c - g0c + WAVES - 1) / WAVES;
    const int c0 = g0c + wave * cpw, c1 = min(g1c, c0 + cpw);

    __shared__ float sm[WAVES][P * NA][MT][64][4];
    __shared__ float sm_ss[NORM ? WAVES : 1][MT][16];

    f32x4 acc[P][NA][MT];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[j][a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t pstride = (int64_t)nchunks * 64;           // half8 pieces per panel
    const half8* wa = wp + (int64_t)panel0 * pstride + lane;
    const half8* wu = GATEUP ? (wp_up + (int64_t)panel0 * pstride + lane) : nullptr;
    const int64_t xcs = 4 * xa.sk;                           // elements per 32-wide k-chunk step of the B operand
    const h16* xr[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + li;
        xok[t] = m < M;
        xr[t] = x + (int64_t)(xok[t] ? m : 0) * xa.sm + (int64_t)g * xa.sk;
    }
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // Loads of the prologue, ALL issued before anything waits, in the order they are consumed (the vector-memory counter
    // retires in order): the sum-of-squares partials of the norm (L2 / Infinity-Cache hits: the producer ran on other
    // XCDs), the x rows of this wave's first k-chunks, then its first weight chunks (HBM).  Round 3 issued the weights
    // first and the x rows only after the fold: the fold then waited for the HBM loads, and the first MFMA for one more
    // round trip of x — ~1.5 us of every norm-prologue GEMM in situ (q|k|v 21.7 us in the layer against 17.8 alone).
    constexpr int SSB = MT == 1 ? 3 : 2;                     // batches of 8 partials per thread loaded up front
    constexpr bool BPRE = MT == 1 && SG_PROLOGUE_ORDER != 0; // (two row tiles: the x prefetch would cost a wave per SIMD)
    constexpr int G = WAVES * 4;
    const int nparts = K >> 4;
    const int pg = tid >> 4, m16 = tid & 15;
    float ssv[MT][SSB][8], part0[MT];
    constexpr bool LNPRE = BPRE && SG_PROLOGUE_ORDER == 2 && SG_LN_PRE != 0;   // norm weights of the first batch fetched up front
    half8 a_pre[U][P][NA], b_pre[BPRE ? U : 1][MT], ln_pre[LNPRE ? U : 1];
    const bool pre = NORM && (c0 + U <= c1);
#if SG_PROLOGUE_ORDER == 2
    if constexpr (NORM) {
        // ONE basic block: every load below is unconditional (clamped index + select; a null ss_in reads the head of the
        // weights instead — the values are not used), nothing between them can wait, and the block is pinned in front of
        // the fold.  With the loads behind `if (ss_in)` / `if (pre)` / `p < nparts ? :` the compiler (a) branched around
        // every partial load and sank the first addition into the first branch — `global_load; s_waitcnt vmcnt(0); v_add` at
        // the very top of the kernel — and (b) threaded the two `if (ss_in)` regions together: partials, s_waitcnt
        // vmcnt(0), the fold's additions, and only THEN the x rows and the first weights: two dependent memory round trips
        // (Infinity Cache, then HBM) where the source asked for one.
        const float* ssp = ss_in ? ss_in : reinterpret_cast<const float*>(wp);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int bb = 0; bb < SSB; ++bb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = pg + bb * 8 * G + j * G;
                    const float v = ssp[(int64_t)min(p, nparts - 1) * 32 + t * 16 + m16];
                    ssv[t][bb][j] = (p < nparts) ? v : 0.f;
                }
        if constexpr (BPRE) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cu = min(c0 + u, nchunks - 1);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const half8 v = load_half8(xr[t] + xcs * cu);
                    b_pre[u][t] = xok[t] ? v : zero8;
                }
                if constexpr (LNPRE) ln_pre[u] = load_half8(ln_w + 32 * cu + 8 * g);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = min(c0 + u, nchunks - 1);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                a_pre[u][j][0] = SG_LOAD(wa + j * pstride + (int64_t)cu * 64);
                if (GATEUP) a_pre[u][j][NA - 1] = SG_LOAD(wu + j * pstride + (int64_t)cu * 64);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // the partials' first additions stay in THIS block (their loads can then not be sunk below the weights' loads into
        // the fold's branch): they wait for the partials only — the x rows and the weights stay in flight behind them.
        // Adding the zeros of the entries past nparts is the identity (the sums are non-negative): same bits as the
        // guarded form.
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float part = 0.f;
#pragma unroll
            for (int bb = 0; bb < SSB; ++bb)
#pragma unroll
                for (int j = 0; j < 8; ++j) part += ssv[t][bb][j];
            asm volatile("" : "+v"(part));                              // materialised HERE: nothing above may sink below
            part0[t] = part;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#if SG_PROLOGUE_ORDER == 0
    if (pre) {                                               // round 3's order (A/B): weights first
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                a_pre[u][j][0] = SG_LOAD(wa + j * pstride + (int64_t)(c0 + u) * 64);
                if (GATEUP) a_pre[u][j][NA - 1] = SG_LOAD(wu + j * pstride + (int64_t)(c0 + u) * 64);
            }
    }
#endif
    if (NORM && ss_in) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int bb = 0; bb < SSB; ++bb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = pg + bb * 8 * G + j * G;
                    ssv[t][bb][j] = (p < nparts) ? ss_in[(int64_t)p * 32 + t * 16 + m16] : 0.f;
                }
    }
#if SG_PROLOGUE_ORDER != 0
    if (pre) {
        if constexpr (BPRE) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t) b_pre[u][t] = xok[t] ? load_half8(xr[t] + xcs * (c0 + u)) : zero8;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                a_pre[u][j][0] = SG_LOAD(wa + j * pstride + (int64_t)(c0 + u) * 64);
                if (GATEUP) a_pre[u][j][NA - 1] = SG_LOAD(wu + j * pstride + (int64_t)(c0 + u) * 64);
            }
    }
#endif
#endif

    SG_STAMP(1);
    float inv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) inv[t] = 1.f;
    if (NORM) {
        float tot[MT];
        if (ss_in) {
            // The producer of x (a residual-epilogue GEMM over K/16 panels) left the per-panel sums of squares of
            // every row: ss_in[panel][32 rows].  Fold them in a fixed order — one round trip instead of a second
            // pass over x.  Thread (pg, m): partials of panels pg, pg + G, ... of row m; G = threads / 16.
            float* red = &sm[0][0][0][0][0];                            // reuse the merge buffer: [MT][G][16] floats
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#if SG_PROLOGUE_ORDER == 2
                float part = part0[t];                                  // summed next to the loads (see there)
#else
                float part = 0.f;
#pragma unroll
                for (int bb = 0; bb < SSB; ++bb)                        // (same order of additions as the loop below)
                    if (pg + bb * 8 * G < nparts) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) part += ssv[t][bb][j];
                    }
#endif
                for (int p0 = pg + SSB * 8 * G; p0 < nparts; p0 += 8 * G) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int p = p0 + j * G;
                        v[j] = (p < nparts) ? ss_in[(int64_t)p * 32 + t * 16 + m16] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += v[j];
                }
                red[(t * G + pg) * 16 + m16] = part;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                tot[t] = 0.f;
                for (int j = 0; j < G; ++j) tot[t] += red[(t * G + j) * 16 + li];
            }
            __syncthreads();                                             // red aliases the split-K merge buffer
        } else {
            float ss[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) ss[t] = 0.f;
            // the sum of squares runs over ALL of K (this wave's share of the whole row, not of the workgroup's K slice)
            const int cpa = (nchunks + WAVES - 1) / WAVES;
            const int n0 = wave * cpa, n1 = min(nchunks, n0 + cpa);
            int cc = n0;
            for (; cc + 8 <= n1; cc += 8) {                             // 8 independent loads per row tile in flight
                half8 v[8][MT];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < MT; ++t) v[j][t] = xok[t] ? load_half8(xr[t] + xcs * (cc + j)) : zero8;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[j][t][e];
                            ss[t] = fmaf(f, f, ss[t]);
                        }
            }
            for (; cc < n1; ++cc) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const half8 v = xok[t] ? load_half8(xr[t] + xcs * cc) : zero8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        ss[t] = fmaf(f, f, ss[t]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                ss[t] += __shfl_xor(ss[t], 16, 64);
                ss[t] += __shfl_xor(ss[t], 32, 64);
                if (g == 0) sm_ss[wave][t][li] = ss[t];
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                tot[t] = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) tot[t] += sm_ss[w][t][li];
            }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) inv[t] = 1.0f / sqrtf(tot[t] / (float)K + eps);
    }

    SG_STAMP(2);
    // Epilogue waves: wave j < P finishes panel panel0 + j.
    const bool epi = wave < P;
    const int panel = panel0 + (epi ? wave : 0);

    // Operands of the epilogue, fetched early so that no dependent round trip is left at the tail of the kernel: the
    // exchange's epoch and sticky error word (uncached), the RoPE tables (positions -> table: two dependent loads), this
    // lane's residual values.  They need the TAIL of the argument list (not preloaded into SGPRs): the forms without a norm
    // prologue run this after their first batch of weight loads has been issued (prefetch_epi below) — issued before, the
    // wait for the argument segment stood in front of the first weight load of every o_proj / down_proj.
    unsigned xepoch = 0, xerr = 0;
    half4 rope_cs[MT], rope_sn[MT], res_pre[MT];
    int64_t r_off = 0, y_off = 0;
    bool res_early = false;
    auto prefetch_epi = [&]() {
        if constexpr (XCHG) {
            if (epi) {
                // every panel counts its own exchanges (written at the end of this panel's previous exchange, an earlier
                // launch of the stream): no grid-wide epoch, so no ticket and no last-workgroup tail at the end of the kernel
                xepoch = xc_ld(&xc.pepoch[panel]) + 1u;
                xerr = xc_ld(&xc.ctl->error);
            }
        }
        if (MODE == SG_QKV && epi) {
            const int H = rp.H, D = rp.D, pph = D >> 4;
            const int sec = panel / (H * pph), pp = panel % pph;
            const int d = 8 * pp + 4 * (g & 1) + ((g >= 2) ? (D >> 1) : 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = t * 16 + li;
                rope_cs[t] = half4{0, 0, 0, 0};
                rope_sn[t] = half4{0, 0, 0, 0};
                if (m < M && sec != 2 && (sec == 0 || rp.rotate_k)) {
                    const int64_t pos = rp.positions[m];
                    rope_cs[t] = *reinterpret_cast<const half4*>(rp.cosb + pos * D + d);
                    rope_sn[t] = *reinterpret_cast<const half4*>(rp.sinb + pos * D + d);
                }
            }
        }
        // (resid may alias y: every element is read and written by the same lane only.)
        r_off = (int64_t)(2 * panel + (g >> 1)) * ra.sk + 4 * (g & 1);   // this lane's 4 columns, piece form
        y_off = (int64_t)(2 * panel + (g >> 1)) * ya.sk + 4 * (g & 1);
        res_early = TF_SG_RES_EARLY && MODE == SG_PLAIN && resid != nullptr && epi && (ra.sm % 4) == 0 &&
                    (ra.sk % 4) == 0 && (reinterpret_cast<uintptr_t>(resid) % 8) == 0;
        if (res_early) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = t * 16 + li;
                res_pre[t] = half4{0, 0, 0, 0};
                if (m < M) res_pre[t] = *reinterpret_cast<const half4*>(resid + (int64_t)m * ra.sm + r_off);
            }
        }
    };
    constexpr bool EPI_LATE = !NORM && SG_EPI_LATE != 0;         // prefetch_epi inside the first K batch
    if (!EPI_LATE || !(c0 + U <= c1)) prefetch_epi();

    int c = c0;
    for (; c + U <= c1; c += U) {
        half8 a[U][P][NA], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (NORM && c == c0 && pre) {
                    a[u][j][0] = a_pre[u][j][0];
                    if (GATEUP) a[u][j][NA - 1] = a_pre[u][j][NA - 1];
                } else {
                    a[u][j][0] = SG_LOAD(wa + j * pstride + (int64_t)(c + u) * 64);
                    if (GATEUP) a[u][j][NA - 1] = SG_LOAD(wu + j * pstride + (int64_t)(c + u) * 64);
                }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (BPRE && NORM && c == c0 && pre) b[u][t] = b_pre[BPRE ? u : 0][t];
                else b[u][t] = xok[t] ? load_half8(xr[t] + xcs * (c + u)) : zero8;
                if (NORM) {
                    half8 lw;
                    if (LNPRE && c == c0 && pre) lw = ln_pre[LNPRE ? u : 0];
                    else lw = load_half8(ln_w + 32 * (c + u) + 8 * g);
                    b[u][t] = sg_normalise(b[u][t], lw, inv[t]);
                }
            }
        }
        if constexpr (EPI_LATE) {
            if (c == c0) prefetch_epi();                             // (behind this batch's loads, in front of its MFMAs)
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < P; ++j)
#pragma unroll
                    for (int aa = 0; aa < NA; ++aa)
                        acc[j][aa][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][j][aa], b[u][t], acc[j][aa][t], 0, 0, 0);
#if SG_STAMPS
        if (c == c0) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // let the batch's MFMAs issue before the stamp
            SG_STAMP(3);
        }
#endif
    }
    SG_STAMP(4);
    if constexpr (NORM || !SG_TAIL_BATCH) {
        // (norm-prologue forms: K = hidden, whose k-chunks divide evenly among the waves in every configuration that
        //  matters; the batch below would only cost them registers — a wave per SIMD on the gate|up form)
        for (; c < c1; ++c) {
            half8 a[P][NA];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                a[j][0] = SG_LOAD(wa + j * pstride + (int64_t)c * 64);
                if (GATEUP) a[j][NA - 1] = SG_LOAD(wu + j * pstride + (int64_t)c * 64);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                half8 b = xok[t] ? load_half8(xr[t] + xcs * c) : zero8;
                if (NORM) b = sg_normalise(b, load_half8(ln_w + 32 * c + 8 * g), inv[t]);
#pragma unroll
                for (int j = 0; j < P; ++j)
#pragma unroll
                    for (int aa = 0; aa < NA; ++aa)
                        acc[j][aa][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j][aa], b, acc[j][aa][t], 0, 0, 0);
            }
        }
    } else if (c < c1) {
        // the last 1 .. U-1 k-chunks of this wave as ONE batch: every load issued before the first wait (one memory
        // round trip; round 3 walked them one chunk — one round trip — at a time: 3 of them in a 7B down_proj wave, 43 =
        // 10 x 4 + 3 chunks).  Loads past the end re-read the last chunk instead of being predicated: a conditional load
        // makes the waitcnt pass wait for everything (compiler trap); its MFMAs are skipped.  Same
        // chunk order, same accumulators: bit-identical to the chunk-by-chunk walk.
        half8 a[U][P][NA], b[U][MT];
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
            const int cc = min(c + u, c1 - 1);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                a[u][j][0] = SG_LOAD(wa + j * pstride + (int64_t)cc * 64);
                if (GATEUP) a[u][j][NA - 1] = SG_LOAD(wu + j * pstride + (int64_t)cc * 64);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                b[u][t] = xok[t] ? load_half8(xr[t] + xcs * cc) : zero8;
                if (NORM) b[u][t] = sg_normalise(b[u][t], load_half8(ln_w + 32 * cc + 8 * g), inv[t]);
            }
        }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (c + u < c1) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int j = 0; j < P; ++j)
#pragma unroll
                        for (int aa = 0; aa < NA; ++aa)
                            acc[j][aa][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][j][aa], b[u][t], acc[j][aa][t], 0, 0, 0);
            }
    }

    // split-K merge across the waves; C layout: lane holds D[n = 4g + r][m = li]
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int aa = 0; aa < NA; ++aa)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                *reinterpret_cast<f32x4*>(&sm[wave][j * NA + aa][t][lane][0]) = acc[j][aa][t];
    __syncthreads();
    SG_STAMP(5);
    if (!epi) return;
    float S[MT][4], S2[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S[t][r] = 0.f;
            S2[t][r] = 0.f;
        }
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&sm[w][wave * NA][t][lane][0]);
            f32x4 v2 = {0.f, 0.f, 0.f, 0.f};
            if (GATEUP) v2 = *reinterpret_cast<const f32x4*>(&sm[w][wave * NA + NA - 1][t][lane][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                S[t][r] += v[r];
                if (GATEUP) S2[t][r] += v2[r];
            }
        }
    }
    if constexpr (KSPLIT) {
        // publish this workgroup's partial of the panel, take a ticket; only the last arriver goes on (wave-uniform)
        float* base = kx.ws + ((int64_t)panel * KS * NA * MT) * 256 + lane * 4;
        float* mine = base + (int64_t)ks * NA * MT * 256;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sg_st_agent(mine + t * 256 + r, S[t][r]);
                if (GATEUP) sg_st_agent(mine + (MT + t) * 256 + r, S2[t][r]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own write-through stores landed before the ticket
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(&kx.tickets[panel], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk != (unsigned)(KS - 1)) return;
        float P1[SG_KSPLIT_MAX][MT][4], P2[SG_KSPLIT_MAX][MT][4];
#pragma unroll
        for (int k = 0; k < SG_KSPLIT_MAX; ++k)                 // every load issued up front: one memory round trip
            if (k < KS) {
                const float* src = base + (int64_t)k * NA * MT * 256;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        P1[k][t][r] = sg_ld_agent(src + t * 256 + r);
                        if (GATEUP) P2[k][t][r] = sg_ld_agent(src + (MT + t) * 256 + r);
                    }
            }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int k = 0; k < SG_KSPLIT_MAX; ++k)        // split order, whoever arrived last
                    if (k < KS) {
                        a += P1[k][t][r];
                        if (GATEUP) b += P2[k][t][r];
                    }
                S[t][r] = a;
                S2[t][r] = b;
            }
        if (lane == 0) __hip_atomic_store(&kx.tickets[panel], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (XCHG) {
        // ---- this panel's all-reduce, inside the launch (see SgXchg) ----
        xepoch = __builtin_amdgcn_readfirstlane(xepoch);
        bool ok = __builtin_amdgcn_readfirstlane(xerr) == 0u;       // sticky: after one time-out nothing waits again
        const int64_t hb = xc.half_elems * (int64_t)(xepoch & 1u);
        half4 part[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[t][r] = (h16)S[t][r];
            if (ok && t * 16 + li < M) xc_st8(xc.stage[xc.rank] + hb + (int64_t)(t * 16 + li) * ya.sm + y_off, part[t]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the partial has left this CU before any flag says so
        if (xc.fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (fenced form: system-scope release as well)
        if (ok) {
            const bool peer = lane < xc.world && lane != xc.rank;
            if (peer) xc_st(xc.pf[lane] + xc.rank * XC_MAXP + panel, xepoch);
            bool seen = true;
            if (peer) {
                seen = false;
                const unsigned* slot = xc.pf[xc.rank] + lane * XC_MAXP + panel;
                const unsigned long long t0 = wall_clock64();
                for (unsigned spins = 0; spins < XC_SPIN_LIMIT; ++spins) {
                    if ((int)(xc_ld(slot) - xepoch) >= 0) { seen = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if ((spins & 63u) == 63u && wall_clock64() - t0 > xc.timeout_ticks) break;
                }
            }
            ok = __builtin_amdgcn_ballot_w64(!seen) == 0ull;
            if (xc.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // (fenced form: buffer_inv sc0 sc1 behind the wait)
            if (!ok && lane == 0) {
                xc_st(&xc.ctl->error, 1u);
                unsigned* mir = reinterpret_cast<unsigned*>(xc.ctl->mirror);
                if (mir) xc_st(mir, 1u);
            }
        }
        // No acquire FENCE: the peers' panels are read with system-scope (sc0 sc1) loads, which bypass this CU's caches —
        // issued after the flag loads returned (the loop above consumed their values), and the producers drained their
        // system-scope stores before flagging.  A system-scope buffer_inv here cost ~1.7 us per exchange (the first build).
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const bool rowv = t * 16 + li < M;
            half4 pv[XC_MAX_WORLD];
#pragma unroll
            for (int q = 0; q < XC_MAX_WORLD; ++q) {
                pv[q] = part[t];
                if (ok && rowv && q < xc.world && q != xc.rank)
                    pv[q] = xc_ld8(xc.stage[q] + hb + (int64_t)(t * 16 + li) * ya.sm + y_off);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < XC_MAX_WORLD; ++q)
                    if (q < xc.world) acc += (float)pv[q][r];        // rank order, fp32, one rounding below
                S[t][r] = ok ? (float)(h16)acc : __builtin_nanf("");
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + li;
        float s[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[r] = S[t][r];
            s2[r] = S2[t][r];
        }
        if (MODE == SG_QKV) {
            // panel -> (section, head, 8-wide rotary block): q and k panels hold rows d0..d0+7 | d0+D/2..d0+D/2+7
            const int H = rp.H, D = rp.D, half = D >> 1, pph = D >> 4;
            const int sec = panel / (H * pph), hd = (panel / pph) % H, pp = panel % pph;
            h16 val[4], oth[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                val[r] = (h16)s[r];
                oth[r] = (h16)__shfl_xor((float)val[r], 32, 64);         // rotary partner: lane g <-> g ^ 2 (exact)
            }
            if (m >= M) continue;
            const int slot = (rp.slot0_dev ? *rp.slot0_dev : rp.slot0) + m;
            half4 out;
            if (sec == 2) {                                              // v: natural row order, plain copy
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = val[r];
                h16* dst = rp.v_cache + (int64_t)slot * rp.stride_t + (int64_t)hd * rp.stride_h + 16 * pp + 4 * g;
                *reinterpret_cast<half4*>(dst) = out;
                continue;
            }
            const bool hi = g >= 2;
            const int d = 8 * pp + 4 * (g & 1) + (hi ? half : 0);
            if (sec == 0 || rp.rotate_k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const h16 cs = rope_cs[t][r], sn = rope_sn[t][r];
                    // (x*cos) + (rotate_half(x)*sin) in fp16: rotate_half = -x2 for the low half, x1 for the high half
                    const h16 rh = hi ? oth[r] : (h16)(-(float)oth[r]);
                    out[r] = hadd_rn(hmul_rn(val[r], cs), hmul_rn(rh, sn));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = val[r];
            }
            h16* dst = (sec == 0) ? rp.q_out + ((int64_t)m * H + hd) * D + d
                                  : rp.k_cache + (int64_t)slot * rp.stride_t + (int64_t)hd * rp.stride_h + d;
            *reinterpret_cast<half4*>(dst) = out;
            continue;
        }
        const bool row_ok = m < M;
        if (!row_ok && !(MODE == SG_PLAIN && ss_out)) continue;
        if (!row_ok) {                               // keep the wave converged for the shuffles below
            s2[0] = s2[1] = s2[2] = s2[3] = 0.f;
        }
        if (row_ok) {
            if (MODE == SG_F32) {                    // logits.float(): fp16 GEMM, then cast; row-major fp32 rows
                float* dst = (float*)yv + (int64_t)m * ya.sm + panel * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = (float)(h16)s[r];
            } else {
                half4 o;
                if (GATEUP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const h16 gt = (h16)s[r], up = (h16)s2[r];
                        const float gf = (float)gt;
                        const h16 act = (h16)(gf / (1.0f + expf(-gf)));
                        o[r] = hmul_rn(act, up);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (h16)s[r];
                        if (resid)               // residual + hidden, fp16 add
                            o[r] = hadd_rn(res_early ? res_pre[t][r] : resid[(int64_t)m * ra.sm + r_off + r], o[r]);
                        s2[r] = (float)o[r];
                    }
                }
                h16* dst = (h16*)yv + (int64_t)m * ya.sm + y_off;
                if (((ya.sm | ya.sk) % 4) == 0 && (reinterpret_cast<uintptr_t>(yv) % 8) == 0) {
                    *reinterpret_cast<half4*>(dst) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r] = o[r];
                }
            }
        }
        if (MODE == SG_PLAIN && ss_out) {            // this panel's share of sum(y^2) per row, for the next norm prologue
            float q = s2[0] * s2[0] + s2[1] * s2[1] + s2[2] * s2[2] + s2[3] * s2[3];
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (g == 0) ss_out[(int64_t)panel * 32 + m] = q;
        }
    }
    if constexpr (XCHG) {
        if (lane == 0) xc_st(&xc.pepoch[panel], xepoch);      // this panel's next exchange (a later launch) is epoch + 1
    }
    SG_STAMP(6);
}

// ---- NARROW panels (round 5): 8 weight rows per workgroup, for the FEW-PANEL norm GEMMs of a tensor-parallel rank ------
// A rank's q|k|v / gate|up shard has 86-120 16-row panels (7B / 13B at 8 ranks): one workgroup per panel leaves 2/3 of the
// 256 CUs idle, and the K loop of such a workgroup is bound by what ONE CU can pull (~37 GB/s whatever is in flight:
// profiles/r04_gemm_phase_stamps.json, r04_gemm_deep_prefetch_ab.jsonl).  Splitting K across workgroups buys the CUs with a
// ~4.7 us cross-workgroup hand-off (SgKsplit above: no gain).  This form splits N instead — no hand-off at all: a workgroup
// owns 8 output rows and streams HALF the bytes; twice the workgroups.  The matrix core still does the K reduction: TWO
// 32-wide k-chunks are stacked on the 16 A rows (rows 0-7: chunk 2c, rows 8-15: chunk 2c+1 of the same 8 weight rows) and
// the matching two x chunks on the 16 B columns (columns 0-7: rows m of chunk 2c, columns 8-15: the same rows of chunk
// 2c+1), so the two DIAGONAL 8 x 8 blocks of the 16 x 16 result hold the even / odd halves of the K sum (the off-diagonal
// blocks are cross terms nobody reads); the epilogue adds the two.  Half the MFMA's flops are wasted — of a pipe that is
// ~10 % busy.  Weights are packed for it once (ops.pack_weight_n8): per 8-row panel and 64-wide super-chunk one contiguous
// KiB in operand order, so a wave's A load is still one 1 KiB run.  x rows come in tiles of 8 (MT tiles: <= 8 / 16 / 24
// rows).  Same rounding points as skinny_gemm_kernel (fp32 accumulate, one fp16 rounding, the fused epilogues' fp16
// arithmetic); the K sum is taken in another order (even | odd chunks, then the waves), so results agree with the 16-row
// form to fp32-summation noise, not bit for bit — both are checked against the oracle at the same tolerance.
// Every wave issues ALL loads of a batch up front; registers are not a constraint here (<= 1 workgroup per CU by design).
template <int MT, int MODE, bool NORM, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_n8_kernel(const half8* __restrict__ wp,
                                                                    const half8* __restrict__ wp_up,
                                                                    const h16* __restrict__ x,
                                                                    const float* __restrict__ ss_in,
                                                                    const h16* __restrict__ ln_w, int K, int M,
                                                                    int xa_sm, int xa_sk, float eps, void* yv, SgAct ya,
                                                                    SgRope rp) {
    static_assert(MODE == SG_GATEUP || MODE == SG_QKV, "narrow panels: the two norm GEMMs of a decoder layer");
    constexpr bool GATEUP = MODE == SG_GATEUP;
    constexpr int NA = GATEUP ? 2 : 1;
    constexpr int MT16 = (MT * 8 + 15) / 16;                    // 16-row tiles of the norm partials (producer's layout)
    const int panel = blockIdx.x;                               // 8 output rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4, lr = li & 7, par = li >> 3;
    const int nsc = K >> 6;                                     // 64-wide super-chunks
    const int cpw = (nsc + WAVES - 1) / WAVES;
    const int c0 = wave * cpw, c1 = min(nsc, c0 + cpw);

    __shared__ float sm[WAVES][NA][MT][64][4];
    __shared__ float red[MT16][WAVES * 4][16];
    __shared__ float sm_ss[WAVES][MT * 8];

    f32x4 acc[NA][MT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t pstride = (int64_t)nsc * 64;                  // half8 pieces per 8-row panel
    const half8* wa = wp + (int64_t)panel * pstride + lane;
    const half8* wu = GATEUP ? (wp_up + (int64_t)panel * pstride + lane) : wa;
    const int64_t sk = xa_sk, xcs = 8 * sk;                     // elements per super-chunk step of the B operand
    const h16* xr[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 8 + lr;
        xok[t] = m < M;
        xr[t] = x + (int64_t)(xok[t] ? m : 0) * xa_sm + (int64_t)(4 * par + g) * sk;
    }
    const h16* lnp = NORM ? (ln_w + 32 * par + 8 * g) : nullptr;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- prologue: every load of the norm partials and of the first batch in ONE basic block (see skinny_gemm_kernel) ----
    constexpr int G = WAVES * 4;
    constexpr int SSB = 2;
    const int nparts = K >> 4;
    const int pg = tid >> 4, m16 = tid & 15;
    float ssv[MT16][SSB][8], part0[MT16];
    half8 a[U][NA], b[U][MT], lw[U];
    const int last = max(c1 - 1, 0);
    {
        const float* ssp = (NORM && ss_in) ? ss_in : reinterpret_cast<const float*>(wp);
        if constexpr (NORM) {
#pragma unroll
            for (int t = 0; t < MT16; ++t)
#pragma unroll
                for (int bb = 0; bb < SSB; ++bb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int p = pg + bb * 8 * G + j * G;
                        const float v = ssp[(int64_t)min(p, nparts - 1) * 32 + t * 16 + m16];
                        ssv[t][bb][j] = (p < nparts) ? v : 0.f;
                    }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = min(c0 + u, last);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const half8 v = load_half8(xr[t] + xcs * cu);
                b[u][t] = xok[t] ? v : zero8;
            }
            if constexpr (NORM) lw[u] = load_half8(lnp + 64 * cu);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = min(c0 + u, last);
            a[u][0] = SG_LOAD(wa + (int64_t)cu * 64);
            if (GATEUP) a[u][NA - 1] = SG_LOAD(wu + (int64_t)cu * 64);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NORM) {
#pragma unroll
            for (int t = 0; t < MT16; ++t) {
                float part = 0.f;
#pragma unroll
                for (int bb = 0; bb < SSB; ++bb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += ssv[t][bb][j];
                asm volatile("" : "+v"(part));                  // the fold's first additions stay in front of everything else
                part0[t] = part;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    float inv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) inv[t] = 1.f;
    if constexpr (NORM) {
        float tot[MT];
        if (ss_in) {
            // per-panel sums of squares left by the producer of x (residual-epilogue GEMM / exchange): fold in a fixed order
#pragma unroll
            for (int t = 0; t < MT16; ++t) {
                float part = part0[t];
                for (int p0 = pg + SSB * 8 * G; p0 < nparts; p0 += 8 * G) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int p = p0 + j * G;
                        v[j] = (p < nparts) ? ss_in[(int64_t)p * 32 + t * 16 + m16] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += v[j];
                }
                red[t][pg][m16] = part;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int row = t * 8 + lr;
                tot[t] = 0.f;
                for (int j = 0; j < G; ++j) tot[t] += red[row >> 4][j][row & 15];
            }
        } else {
            // no hand-off (layer 0): this wave's share of sum(x^2) over ALL of K, then across the waves
            float ss[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) ss[t] = 0.f;
            for (int cc = c0; cc < c1; cc += 4) {
                half8 v[4][MT];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const half8 q = load_half8(xr[t] + xcs * min(cc + j, last));
                        v[j][t] = (xok[t] && cc + j < c1) ? q : zero8;
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[j][t][e];
                            ss[t] = fmaf(f, f, ss[t]);
                        }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                ss[t] += __shfl_xor(ss[t], 8, 64);              // the two stacked chunks
                ss[t] += __shfl_xor(ss[t], 16, 64);             // the four k-octets
                ss[t] += __shfl_xor(ss[t], 32, 64);
                if (lane < 8) sm_ss[wave][t * 8 + lane] = ss[t];
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                tot[t] = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) tot[t] += sm_ss[w][t * 8 + lr];
            }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) inv[t] = 1.0f / sqrtf(tot[t] / (float)K + eps);
    }

    // RoPE tables of this lane's output columns (positions -> table: two dependent loads, fetched under the K loop)
    const bool epi = wave == 0;
    half4 rope_cs[MT], rope_sn[MT];
    if (MODE == SG_QKV && epi) {
        const int H = rp.H, D = rp.D, pph = D >> 3;
        const int sec = panel / (H * pph), pp = panel % pph;
        const int d = 4 * pp + ((g & 1) ? (D >> 1) : 0);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = t * 8 + lr;
            rope_cs[t] = half4{0, 0, 0, 0};
            rope_sn[t] = half4{0, 0, 0, 0};
            if (m < M && sec != 2 && (sec == 0 || rp.rotate_k)) {
                const int64_t pos = rp.positions[m];
                rope_cs[t] = *reinterpret_cast<const half4*>(rp.cosb + pos * D + d);
                rope_sn[t] = *reinterpret_cast<const half4*>(rp.sinb + pos * D + d);
            }
        }
    }

    auto compute = [&](int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u < c1) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const half8 bn = NORM ? sg_normalise(b[u][t], lw[u], inv[t]) : b[u][t];
#pragma unroll
                    for (int aa = 0; aa < NA; ++aa)
                        acc[aa][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][aa], bn, acc[aa][t], 0, 0, 0);
                }
            }
        }
    };
    compute(c0);                                                // the batch the prologue issued (straight-line: no join)
    for (int c = c0 + U; c < c1; c += U) {
        // (loads past the end re-read the wave's last super-chunk instead of being predicated: a conditional load makes
        //  the waitcnt pass wait for everything; their MFMAs are skipped)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = min(c + u, last);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const half8 v = load_half8(xr[t] + xcs * cu);
                b[u][t] = xok[t] ? v : zero8;
            }
            if constexpr (NORM) lw[u] = load_half8(lnp + 64 * cu);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cu = min(c + u, last);
            a[u][0] = SG_LOAD(wa + (int64_t)cu * 64);
            if (GATEUP) a[u][NA - 1] = SG_LOAD(wu + (int64_t)cu * 64);
        }
        compute(c);
    }

    // merge across the waves; C layout: lane holds D[n = 4g + r][m = li].  Valid blocks: (g < 2, li < 8) = even chunks,
    // (g >= 2, li >= 8) = odd chunks of the same (weight row 4 (g & 1) + r, x row li & 7): lane + 40.
#pragma unroll
    for (int aa = 0; aa < NA; ++aa)
#pragma unroll
        for (int t = 0; t < MT; ++t) *reinterpret_cast<f32x4*>(&sm[wave][aa][t][lane][0]) = acc[aa][t];
    __syncthreads();
    if (!epi) return;
    const bool vl = g < 2 && li < 8;                            // lanes that own outputs (all 64 run the sums: no divergence)
    const int lo = vl ? lane : 0, hi_l = vl ? lane + 40 : 40;
    float S[MT][4], S2[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S[t][r] = 0.f;
            S2[t][r] = 0.f;
        }
#pragma unroll 2
        for (int w = 0; w < WAVES; ++w) {                       // (fully unrolled the reads of all waves and tiles are hoisted: spills)
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(&sm[w][0][t][lo][0]);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(&sm[w][0][t][hi_l][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[t][r] += e0[r] + o0[r];
            if (GATEUP) {
                const f32x4 e1 = *reinterpret_cast<const f32x4*>(&sm[w][NA - 1][t][lo][0]);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(&sm[w][NA - 1][t][hi_l][0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) S2[t][r] += e1[r] + o1[r];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 8 + lr;
        if (MODE == SG_QKV) {
            // 8-row panel -> (section, head, 4-wide rotary block): q and k panels hold rows d0..d0+3 | d0+D/2..d0+D/2+3
            const int H = rp.H, D = rp.D, half = D >> 1, pph = D >> 3;
            const int sec = panel / (H * pph), hd = (panel / pph) % H, pp = panel % pph;
            h16 val[4], oth[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                val[r] = (h16)S[t][r];
                oth[r] = (h16)__shfl_xor((float)val[r], 16, 64);         // rotary partner: lane g <-> g ^ 1 (exact)
            }
            if (!vl || m >= M) continue;
            const int slot = (rp.slot0_dev ? *rp.slot0_dev : rp.slot0) + m;
            half4 out;
            if (sec == 2) {                                              // v: natural row order
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = val[r];
                h16* dst = rp.v_cache + (int64_t)slot * rp.stride_t + (int64_t)hd * rp.stride_h + 8 * pp + 4 * g;
                *reinterpret_cast<half4*>(dst) = out;
                continue;
            }
            const bool hi = g == 1;
            const int d = 4 * pp + (hi ? half : 0);
            if (sec == 0 || rp.rotate_k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const h16 cs = rope_cs[t][r], sn = rope_sn[t][r];
                    const h16 rh = hi ? oth[r] : (h16)(-(float)oth[r]);
                    out[r] = hadd_rn(hmul_rn(val[r], cs), hmul_rn(rh, sn));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = val[r];
            }
            h16* dst = (sec == 0) ? rp.q_out + ((int64_t)m * H + hd) * D + d
                                  : rp.k_cache + (int64_t)slot * rp.stride_t + (int64_t)hd * rp.stride_h + d;
            *reinterpret_cast<half4*>(dst) = out;
        } else {
            if (!vl || m >= M) continue;
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const h16 gt = (h16)S[t][r], up = (h16)S2[t][r];
                const float gf = (float)gt;
                const h16 act = (h16)(gf / (1.0f + expf(-gf)));
                o[r] = hmul_rn(act, up);
            }
            h16* dst = (h16*)yv + (int64_t)m * ya.sm + (int64_t)panel * ya.sk + 4 * g;   // columns 8 panel + 4 g + r
            if (((ya.sm | ya.sk) % 4) == 0 && (reinterpret_cast<uintptr_t>(yv) % 8) == 0) {
                *reinterpret_cast<half4*>(dst) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = o[r];
            }
        }
    }
}

struct SgArgs {                       // one GEMM call: operands with their layouts
    const void *wp, *wp_up, *x, *ln_w, *resid;
    void* y;
    SgAct xa, ra, ya;
    float eps;
    int M, N, K;
    const float* ss_in;
    float* ss_out;
};

// Workgroups per panel group along K: only few-panel grids (see SgKsplit), only with a registered workspace that holds the
// partials, and only while every wave of every workgroup still gets >= 2 k-chunks.
template <int MT, int NA, int WAVES, int P>
static int sg_pick_ksplit(const SgArgs& a, SgKsplit& kx, hipStream_t st = nullptr) {
    kx = SgKsplit{nullptr, nullptr, nullptr};
#if SG_STAMPS
    {
        int d0 = 0;
        if (hipGetDevice(&d0) == hipSuccess && d0 >= 0 && d0 < 16 && g_sg_ws[d0] && g_sg_ws_bytes[d0] >= (6 << 20))
            kx.stamps = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(g_sg_ws[d0]) + (4 << 20));
    }
#endif
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !g_sg_ws[dev]) return 1;
    const int panels = a.N / 16, groups = panels / P, nchunks = a.K >> 5;
    if (panels > SG_TICKETS) return 1;
    int ks;
    if (g_sg_ksplit_force > 1) {                       // A/B (tf_sg_tune key 4): this many workgroups per panel, any grid
        ks = g_sg_ksplit_force;
    } else if (g_sg_ksplit_force == 1) {               // ... or never
        return 1;
    } else if (panels > 256 && panels <= 384 && nchunks >= 256) {
        // The one regime where the split pays (profiles/r04_gemm_ksplit_force_ab.jsonl): a grid a little larger than the
        // chip — 13B down_proj: 320 panels on 256 CUs, 64 CUs hold two workgroups and set the pace — with a LONG K
        // (432 k-chunks, 141.6 MB).  Three K-splits (960 workgroups, even): 41.8 -> 34.8 us at 17 rows, 32.4 -> 28.5 at 8,
        // 48.4 -> 39.8 at 32.  With a short K (13B o_proj, 160 k-chunks: 16.3 -> 19.2) or a grid that already fits
        // (7B: 256 panels) it loses.
        ks = 3;
    } else {
        if (groups >= g_sg_ksplit_max_groups) return 1;
        ks = (256 + groups - 1) / groups;
    }
    if (ks > SG_KSPLIT_MAX) ks = SG_KSPLIT_MAX;
    while (ks > 1 && nchunks / ks < 2 * WAVES) --ks;
    if (ks <= 1) return 1;
    const int64_t need = (int64_t)SG_TICKETS * 4 + (int64_t)panels * ks * NA * MT * 256 * 4;
    const int64_t slot_bytes = (g_sg_ws_bytes[dev] / SG_WS_SLOTS) & ~(int64_t)255;
    if (need > slot_bytes) return 1;
    const int slot = sg_slot_of(dev, st);
    if (slot < 0) return 1;                                   // more launch streams than slots: this one runs unsplit
    char* base = reinterpret_cast<char*>(g_sg_ws[dev]) + slot * slot_bytes;
    kx.tickets = reinterpret_cast<unsigned*>(base);
    kx.ws = reinterpret_cast<float*>(base + (int64_t)SG_TICKETS * 4);
    return ks;
}

template <int MT, int MODE, bool NORM, int WAVES, int P, int UX = 1>
static void launch_sg_w(const SgArgs& a, const SgRope& rp, hipStream_t st) {
    SgKsplit kx = {nullptr, nullptr, nullptr};
    if constexpr (UX > 1) {
        sg_pick_ksplit<MT, (MODE == SG_GATEUP ? 2 : 1), WAVES, P>(a, kx);      // (stamps pointer of the instrumented build)
        kx.ws = nullptr, kx.tickets = nullptr;
        hipLaunchKernelGGL((skinny_gemm_kernel<MT, MODE, NORM, WAVES, P, false, false, UX>), dim3(a.N / 16 / P), dim3(WAVES * 64),
                           0, st, (const half8*)a.wp, (const half8*)a.wp_up, (const h16*)a.x, a.ss_in, (const h16*)a.ln_w, a.K, a.M,
                           (int)a.xa.sm, (int)a.xa.sk, a.eps, a.N, (const h16*)a.resid, a.ra, a.y, a.ya, rp, a.ss_out, kx, SgXchg{});
        return;
    }
    if constexpr (P == 1 && MODE != SG_F32) {                     // few-panel grids only: never the P = 2 form, never lm_head
        const int ks = sg_pick_ksplit<MT, (MODE == SG_GATEUP ? 2 : 1), WAVES, P>(a, kx, st);
        if (ks > 1) {
            hipLaunchKernelGGL((skinny_gemm_kernel<MT, MODE, NORM, WAVES, P, true>), dim3(a.N / 16 / P, ks), dim3(WAVES * 64),
                               0, st, (const half8*)a.wp, (const half8*)a.wp_up, (const h16*)a.x, a.ss_in, (const h16*)a.ln_w, a.K, a.M,
                               (int)a.xa.sm, (int)a.xa.sk, a.eps, a.N, (const h16*)a.resid, a.ra, a.y, a.ya, rp, a.ss_out, kx,
                               SgXchg{});
            return;
        }
    }
    hipLaunchKernelGGL((skinny_gemm_kernel<MT, MODE, NORM, WAVES, P, false>), dim3(a.N / 16 / P), dim3(WAVES * 64), 0, st,
                       (const half8*)a.wp, (const half8*)a.wp_up, (const h16*)a.x, a.ss_in, (const h16*)a.ln_w, a.K, a.M,
                       (int)a.xa.sm, (int)a.xa.sk, a.eps, a.N, (const h16*)a.resid, a.ra, a.y, a.ya, rp, a.ss_out, kx, SgXchg{});
}

template <int MODE, bool NORM>
static int launch_sg(const SgArgs& a, const SgRope& rp, hipStream_t st) {
    const int panels = a.N / 16, nchunks = a.K >> 5;
    // two panels per wave (x read once per 2 KiB of weights): from g_sg_p2_rows activation rows up, while the halved
    // grid still covers every CU; the 8 waves of such a workgroup split K.
    const bool p2 = a.M >= g_sg_p2_rows && (panels % 2) == 0 && panels / 2 >= g_sg_p2_groups && nchunks >= 16;
    if (p2) {
        if (g_sg_p2_waves == 4) {
            if (a.M <= 16) launch_sg_w<1, MODE, NORM, 4, 2>(a, rp, st);
            else launch_sg_w<2, MODE, NORM, 4, 2>(a, rp, st);
        } else {
            if (a.M <= 16) launch_sg_w<1, MODE, NORM, 8, 2>(a, rp, st);
            else launch_sg_w<2, MODE, NORM, 8, 2>(a, rp, st);
        }
        TF_LAUNCH_CHECK();
        return TF_OK;
    }
    // Few-panel gate|up GEMMs at two row tiles (a tensor-parallel rank's shard at 17-32 rows: 108 panel pairs at 13B TP 8)
    // run 8 waves per panel like the other few-panel forms, not 4: 13B TP-8 retrieval verify 3 421 -> 3 215 us, target
    // verify 5 492 -> 5 278 (profiles/r04_tp_shard_waves_tail_ab.jsonl).  At ONE row tile more waves per panel were
    // measured and lose (16 waves: 7B TP-8 retrieval verify 1 877 -> 1 925 us; q|k|v 10.0 -> 10.9 us, gate|up 13.8 ->
    // 13.9): those launches are not bound by the length of a wave's K chain.  tf_sg_tune key 5 = largest panel count that
    // takes the form (0: never).
    constexpr bool CAN_DEEP = MODE != SG_F32;
    const bool deep = CAN_DEEP && panels <= g_sg_deep_panels && nchunks >= 64;    // few panels, long K: UX = 2 (see the kernel)
    if (MODE == SG_GATEUP && a.M > 16 && panels <= g_sg_few_panels && nchunks >= 32) {
        launch_sg_w<2, MODE, NORM, 8, 1>(a, rp, st);              // (UX = 2 would spill here: 256 registers + 4)
        TF_LAUNCH_CHECK();
        return TF_OK;
    }
    if (deep) {
        constexpr int DW = MODE == SG_GATEUP ? SG_WAVES : SG_WAVES_WIDE;
        if (a.M <= 16) launch_sg_w<1, MODE, NORM, DW, 1, (CAN_DEEP ? 2 : 1)>(a, rp, st);
        else launch_sg_w<2, MODE, NORM, SG_WAVES_WIDE, 1, ((CAN_DEEP && MODE != SG_GATEUP) ? 2 : 1)>(a, rp, st);
        TF_LAUNCH_CHECK();
        return TF_OK;
    }
    // wide variant: few panels and enough k-chunks that every wave still gets >= 2 of them
    constexpr bool CAN_WIDE = MODE != SG_GATEUP;
    const bool wide = CAN_WIDE && panels <= SG_WIDE_MAX_PANELS && nchunks >= 2 * SG_WAVES_WIDE;
    constexpr int WW = CAN_WIDE ? SG_WAVES_WIDE : SG_WAVES;
    if (a.M <= 16) {
        if (wide) launch_sg_w<1, MODE, NORM, WW, 1>(a, rp, st);
        else launch_sg_w<1, MODE, NORM, SG_WAVES, 1>(a, rp, st);
    } else {
        if (wide) launch_sg_w<2, MODE, NORM, WW, 1>(a, rp, st);
        else launch_sg_w<2, MODE, NORM, SG_WAVES, 1>(a, rp, st);
    }
    TF_LAUNCH_CHECK();
    return TF_OK;
}

static bool sg_act_ok(const SgAct& s) { return s.sm > 0 && s.sk > 0 && (s.sm % 8) == 0 && (s.sk % 8) == 0; }

static bool sg_shape_ok(int M, int N, int K, const SgAct& xa) {
    return M >= 1 && M <= 32 && N >= 16 && (N % 16) == 0 && K >= 32 && (K % 32) == 0 && sg_act_ok(xa) &&
           xa.sm <= 0x7fffffff && xa.sk <= 0x7fffffff;     // (the kernel takes x's two strides as 32-bit arguments)
}

// A/B knobs of the launch rule (tools/gemm_layout_ab.py): key 0 = rows from which two panels per wave are used (33:
// never), key 1 = waves per workgroup of that form (4 or 8), key 2 = smallest halved grid (panels / 2) that takes it,
// key 3 = panel-group count below which K is also split ACROSS workgroups (0 = never).  Returns the previous value, -1 for an unknown key.
extern "C" int tf_sg_tune(int key, int value) {
    int* slot = key == 0 ? &g_sg_p2_rows : key == 1 ? &g_sg_p2_waves : key == 2 ? &g_sg_p2_groups
                : key == 3 ? &g_sg_ksplit_max_groups : key == 4 ? &g_sg_ksplit_force : key == 5 ? &g_sg_few_panels
                : key == 6 ? &g_sg_deep_panels : key == 7 ? &g_sg_n8_u : nullptr;
    if (!slot) return -1;
    const int old = *slot;
    if (key == 1 && value != 4 && value != 8) return old;
    if (key == 7 && value != 0 && value != 5 && value != 8) return old;
    *slot = value;
    return old;
}

// Registers (ws != NULL) or removes the CURRENT device's split-K workspace: `bytes` of device memory, ZERO-filled, that
// stays allocated while GEMMs may run, cut into SG_WS_SLOTS slots — one per launch stream (see sg_slot_of) — of 16 KiB of
// per-panel tickets (left zero by every launch) + the partial sums of one launch at a time.  8 MiB (2 MiB per slot) covers
// every shape the rule splits.
extern "C" int tf_sg_workspace(void* ws, int64_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 16 || (ws && bytes < (int64_t)SG_TICKETS * 4 + 4096) || (reinterpret_cast<uintptr_t>(ws) % 16)) return TF_EINVAL;
    g_sg_ws[dev] = ws;
    g_sg_ws_bytes[dev] = ws ? bytes : 0;
    {
        std::lock_guard<std::mutex> lock(g_sg_slot_mutex);    // a new workspace: the stream -> slot table starts over
        g_sg_slots_used[dev] = 0;
    }
    return TF_OK;
}

extern "C" int tf_skinny_gemm_act(const void* w_packed, const void* x, int64_t xs_m, int64_t xs_k, const void* ln_w,
                                  float eps, const float* ss_in, const void* resid, int64_t rs_m, int64_t rs_k,
                                  float* ss_out, void* y, int64_t ys_m, int64_t ys_k, int M, int N, int K, int out_f32,
                                  void* stream) {
    SgArgs a = {};
    a.wp = w_packed, a.x = x, a.ln_w = ln_w, a.resid = resid, a.y = y;
    a.xa = SgAct{xs_m, xs_k}, a.ra = SgAct{rs_m, rs_k}, a.ya = SgAct{ys_m, ys_k};
    a.eps = eps, a.M = M, a.N = N, a.K = K, a.ss_in = ss_in, a.ss_out = ss_out;
    if (!w_packed || !x || !y || !sg_shape_ok(M, N, K, a.xa)) return TF_EINVAL;
    if ((out_f32 && (resid || ss_out)) || (ss_in && !ln_w)) return TF_EINVAL;
    if (ys_m <= 0 || (!out_f32 && ys_k <= 0) || (resid && (rs_m <= 0 || rs_k <= 0))) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SgRope rp = {};
    if (out_f32) return ln_w ? launch_sg<SG_F32, true>(a, rp, st) : launch_sg<SG_F32, false>(a, rp, st);
    return ln_w ? launch_sg<SG_PLAIN, true>(a, rp, st) : launch_sg<SG_PLAIN, false>(a, rp, st);
}

extern "C" int tf_skinny_gemm_ex(const void* w_packed, const void* x, int64_t ldx, const void* ln_w, float eps,
                                 const float* ss_in, const void* resid, int64_t ldr, float* ss_out, void* y,
                                 int64_t ldy, int M, int N, int K, int out_f32, void* stream) {
    return tf_skinny_gemm_act(w_packed, x, ldx, 8, ln_w, eps, ss_in, resid, ldr, 8, ss_out, y, ldy, 8, M, N, K, out_f32,
                              stream);
}

extern "C" int tf_skinny_gemm(const void* w_packed, const void* x, int64_t ldx, void* y, int64_t ldy, int M, int N,
                              int K, int out_f32, void* stream) {
    return tf_skinny_gemm_ex(w_packed, x, ldx, nullptr, 0.f, nullptr, nullptr, 0, nullptr, y, ldy, M, N, K, out_f32, stream);
}

extern "C" int tf_skinny_gemm_swiglu_act(const void* gate_packed, const void* up_packed, const void* x, int64_t xs_m,
                                         int64_t xs_k, const void* ln_w, float eps, const float* ss_in, void* act,
                                         int64_t ys_m, int64_t ys_k, int M, int I, int K, void* stream) {
    SgArgs a = {};
    a.wp = gate_packed, a.wp_up = up_packed, a.x = x, a.ln_w = ln_w, a.y = act;
    a.xa = SgAct{xs_m, xs_k}, a.ya = SgAct{ys_m, ys_k}, a.ra = SgAct{8, 8};
    a.eps = eps, a.M = M, a.N = I, a.K = K, a.ss_in = ss_in;
    if (!gate_packed || !up_packed || !x || !act || !sg_shape_ok(M, I, K, a.xa) || (ss_in && !ln_w)) return TF_EINVAL;
    if (ys_m <= 0 || ys_k <= 0) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SgRope rp = {};
    return ln_w ? launch_sg<SG_GATEUP, true>(a, rp, st) : launch_sg<SG_GATEUP, false>(a, rp, st);
}

extern "C" int tf_skinny_gemm_swiglu_ex(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx,
                                        const void* ln_w, float eps, const float* ss_in, void* act, int64_t ldy, int M,
                                        int I, int K, void* stream) {
    return tf_skinny_gemm_swiglu_act(gate_packed, up_packed, x, ldx, 8, ln_w, eps, ss_in, act, ldy, 8, M, I, K, stream);
}

extern "C" int tf_skinny_gemm_swiglu(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx,
                                     void* act, int64_t ldy, int M, int I, int K, void* stream) {
    return tf_skinny_gemm_swiglu_ex(gate_packed, up_packed, x, ldx, nullptr, 0.f, nullptr, act, ldy, M, I, K, stream);
}

extern "C" int tf_skinny_qkv_rope_act(const void* wqkv_packed, const void* x, int64_t xs_m, int64_t xs_k,
                                      const void* ln_w, float eps, const float* ss_in, const void* cosb,
                                      const void* sinb, const int64_t* positions, void* q_out, void* k_cache,
                                      void* v_cache, int64_t stride_t, int64_t stride_h, int slot0,
                                      const int32_t* slot0_dev, int M, int H, int D, int K, int rotate_k, void* stream) {
    if (!wqkv_packed || !x || !cosb || !sinb || !positions || !q_out || !k_cache || !v_cache) return TF_EINVAL;
    SgArgs a = {};
    a.wp = wqkv_packed, a.x = x, a.ln_w = ln_w;
    a.xa = SgAct{xs_m, xs_k}, a.ra = SgAct{8, 8}, a.ya = SgAct{8, 8};
    a.eps = eps, a.M = M, a.N = 3 * H * D, a.K = K, a.ss_in = ss_in;
    if (H < 1 || D < 32 || (D % 32) || !sg_shape_ok(M, a.N, K, a.xa) || (ss_in && !ln_w)) return TF_EINVAL;
    if ((stride_t % 4) || (stride_h % 4)) return TF_EINVAL;                          // 8-byte epilogue stores
    hipStream_t st = (hipStream_t)stream;
    SgRope rp;
    rp.cosb = (const h16*)cosb;
    rp.sinb = (const h16*)sinb;
    rp.positions = positions;
    rp.q_out = (h16*)q_out;
    rp.k_cache = (h16*)k_cache;
    rp.v_cache = (h16*)v_cache;
    rp.stride_t = stride_t;
    rp.stride_h = stride_h;
    rp.slot0_dev = slot0_dev;
    rp.slot0 = slot0;
    rp.H = H;
    rp.D = D;
    rp.rotate_k = rotate_k;
    return ln_w ? launch_sg<SG_QKV, true>(a, rp, st) : launch_sg<SG_QKV, false>(a, rp, st);
}

extern "C" int tf_skinny_qkv_rope(const void* wqkv_packed, const void* x, int64_t ldx, const void* ln_w, float eps,
                                  const float* ss_in, const void* cosb, const void* sinb, const int64_t* positions,
                                  void* q_out, void* k_cache, void* v_cache, int64_t stride_t, int64_t stride_h,
                                  int slot0, const int32_t* slot0_dev, int M, int H, int D, int K, int rotate_k,
                                  void* stream) {
    return tf_skinny_qkv_rope_act(wqkv_packed, x, ldx, 8, ln_w, eps, ss_in, cosb, sinb, positions, q_out, k_cache,
                                  v_cache, stride_t, stride_h, slot0, slot0_dev, M, H, D, K, rotate_k, stream);
}

// ---- narrow-panel entry points (skinny_gemm_n8_kernel) ----
// Weights packed by ops.pack_weight_n8: [N/8 panels][K/64 super-chunks][4 (g)][2 (chunk parity)][8 rows][8]; the q|k|v
// weight additionally in the 8-row rotary order (ops.rope_row_order_n8: every panel of the q and k sections holds rows
// d0..d0+3 and their partners d0+D/2..d0+D/2+3 of one head).  Applies to <= 24 rows with a norm prologue (ln_w) and
// K a multiple of 64 with >= 2 super-chunks per wave; anything else returns TF_EINVAL — the caller keeps the 16-row form.
#ifndef SG_N8_WAVES
#define SG_N8_WAVES 8          // waves per 8-row panel (A/B: variant build with SG_N8_WAVES=4)
#endif

static bool sg_n8_ok(int M, int N, int K, const SgAct& xa) {
    return M >= 1 && M <= 24 && N >= 8 && (N % 8) == 0 && K >= 64 * 2 * SG_N8_WAVES && (K % 64) == 0 && sg_act_ok(xa) &&
           xa.sm <= 0x7fffffff && xa.sk <= 0x7fffffff;
}

template <int MODE>
static int launch_sg_n8(const SgArgs& a, const SgRope& rp, hipStream_t st) {
    const int nsc = a.K >> 6, cpw = (nsc + SG_N8_WAVES - 1) / SG_N8_WAVES;
    // batch = super-chunks a wave keeps in flight (x NA weight streams): its whole share when that is <= 8 (7B: 8 — one
    // round trip per wave), else the even split of 10 (13B: 5 + 5)
    const bool u5 = g_sg_n8_u ? (g_sg_n8_u == 5) : (cpw > 8 && (cpw % 5) == 0);
#define N8_LAUNCH(MT_, U_)                                                                                                   \
    hipLaunchKernelGGL((skinny_gemm_n8_kernel<MT_, MODE, true, SG_N8_WAVES, U_>), dim3(a.N / 8), dim3(SG_N8_WAVES * 64), 0,  \
                       st, (const half8*)a.wp, (const half8*)a.wp_up, (const h16*)a.x, a.ss_in, (const h16*)a.ln_w, a.K,    \
                       a.M, (int)a.xa.sm, (int)a.xa.sk, a.eps, a.y, a.ya, rp)
    // (the gate|up form holds two weight streams: from two row tiles up a batch of 4 — or 13B's even 5 + 5 — keeps it inside 256 registers)
    if (a.M <= 8) {
        if (u5) N8_LAUNCH(1, 5); else N8_LAUNCH(1, 8);
    } else if constexpr (MODE == SG_GATEUP) {
        if (a.M <= 16) {
            if (u5) N8_LAUNCH(2, 5); else N8_LAUNCH(2, 4);
        } else {
            if (u5) N8_LAUNCH(3, 5); else N8_LAUNCH(3, 4);
        }
    } else if (a.M <= 16) {
        if (u5) N8_LAUNCH(2, 5); else N8_LAUNCH(2, 8);
    } else {
        N8_LAUNCH(3, 5);
    }
#undef N8_LAUNCH
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_skinny_gemm_swiglu_n8(const void* gate_n8, const void* up_n8, const void* x, int64_t xs_m, int64_t xs_k,
                                        const void* ln_w, float eps, const float* ss_in, void* act, int64_t ys_m,
                                        int64_t ys_k, int M, int I, int K, void* stream) {
    SgArgs a = {};
    a.wp = gate_n8, a.wp_up = up_n8, a.x = x, a.ln_w = ln_w, a.y = act;
    a.xa = SgAct{xs_m, xs_k}, a.ya = SgAct{ys_m, ys_k}, a.ra = SgAct{8, 8};
    a.eps = eps, a.M = M, a.N = I, a.K = K, a.ss_in = ss_in;
    if (!gate_n8 || !up_n8 || !x || !act || !ln_w || !sg_n8_ok(M, I, K, a.xa)) return TF_EINVAL;
    if (ys_m <= 0 || ys_k <= 0) return TF_EINVAL;
    const SgRope rp = {};
    return launch_sg_n8<SG_GATEUP>(a, rp, (hipStream_t)stream);
}

extern "C" int tf_skinny_qkv_rope_n8(const void* wqkv_n8, const void* x, int64_t xs_m, int64_t xs_k, const void* ln_w,
                                     float eps, const float* ss_in, const void* cosb, const void* sinb,
                                     const int64_t* positions, void* q_out, void* k_cache, void* v_cache, int64_t stride_t,
                                     int64_t stride_h, int slot0, const int32_t* slot0_dev, int M, int H, int D, int K,
                                     int rotate_k, void* stream) {
    if (!wqkv_n8 || !x || !ln_w || !cosb || !sinb || !positions || !q_out || !k_cache || !v_cache) return TF_EINVAL;
    SgArgs a = {};
    a.wp = wqkv_n8, a.x = x, a.ln_w = ln_w;
    a.xa = SgAct{xs_m, xs_k}, a.ra = SgAct{8, 8}, a.ya = SgAct{8, 8};
    a.eps = eps, a.M = M, a.N = 3 * H * D, a.K = K, a.ss_in = ss_in;
    if (H < 1 || D < 32 || (D % 32) || !sg_n8_ok(M, a.N, K, a.xa)) return TF_EINVAL;
    if ((stride_t % 4) || (stride_h % 4)) return TF_EINVAL;                          // 8-byte epilogue stores
    SgRope rp;
    rp.cosb = (const h16*)cosb;
    rp.sinb = (const h16*)sinb;
    rp.positions = positions;
    rp.q_out = (h16*)q_out;
    rp.k_cache = (h16*)k_cache;
    rp.v_cache = (h16*)v_cache;
    rp.stride_t = stride_t;
    rp.stride_h = stride_h;
    rp.slot0_dev = slot0_dev;
    rp.slot0 = slot0;
    rp.H = H;
    rp.D = D;
    rp.rotate_k = rotate_k;
    return launch_sg_n8<SG_QKV>(a, rp, (hipStream_t)stream);
}

// y = resid + all_reduce(x . W^T) over `world` ranks, GEMM and exchange in ONE launch (see SgXchg above): replaces
// ops.linear(a, w_o | w_down, out = staging) + tf_allreduce_oneshot_add_ss of the tensor-parallel decode layer
// (models/tensor_op.py:175-181,353-360).  peer_stage[r] / peer_ctl[r]: every rank's staging buffer (2 x half_elems fp16,
// fine-grained) and control buffer (tf_xchg_ctl_bytes(), fine-grained, zero-filled once), own entries included, peers'
// mapped through hipIpc.  The output block (out, its strides) and the staging halves share one activation layout; N / 16
// <= 512 panels, M * N <= half_elems.  out may alias resid.  ss_out: per-panel sums of squares of the result rows.
extern "C" int64_t tf_xchg_ctl_bytes(void) { return XC_PEPOCH_OFF + (int64_t)XC_MAXP * 4; }

extern "C" int tf_skinny_gemm_xchg(const void* w_packed, const void* x, int64_t xs_m, int64_t xs_k,
                                   void* const* peer_stage, void* const* peer_ctl, int rank, int world, int64_t half_elems,
                                   const void* resid, int64_t rs_m, int64_t rs_k, void* out, int64_t os_m, int64_t os_k,
                                   float* ss_out, int M, int N, int K, void* stream) {
    SgArgs a = {};
    a.wp = w_packed, a.x = x, a.resid = resid, a.y = out;
    a.xa = SgAct{xs_m, xs_k}, a.ra = SgAct{rs_m, rs_k}, a.ya = SgAct{os_m, os_k};
    a.M = M, a.N = N, a.K = K, a.ss_out = ss_out;
    if (!w_packed || !x || !out || !peer_stage || !peer_ctl || !sg_shape_ok(M, N, K, a.xa) || !sg_act_ok(a.ya)) return TF_EINVAL;
    if (world < 1 || world > XC_MAX_WORLD || rank < 0 || rank >= world || N / 16 > XC_MAXP) return TF_EINVAL;
    if (half_elems < 8 || (half_elems % 8) || (resid && !sg_act_ok(a.ra))) return TF_EINVAL;
    // the largest element offset of the block in its layout must stay inside a staging half
    const int64_t last = (int64_t)(M - 1) * os_m + (int64_t)(N / 8 - 1) * os_k + 7;
    if (last >= half_elems) return TF_EINVAL;
    SgXchg xc = {};
    for (int r = 0; r < world; ++r) {
        if (!peer_stage[r] || !peer_ctl[r]) return TF_EINVAL;
        xc.stage[r] = (h16*)peer_stage[r];
        xc.pf[r] = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(peer_ctl[r]) + 1024);
    }
    xc.ctl = reinterpret_cast<XcCtl*>(peer_ctl[rank]);
    xc.pepoch = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(peer_ctl[rank]) + XC_PEPOCH_OFF);
    xc.half_elems = half_elems, xc.rank = rank, xc.world = world;
    xc.fence = g_xc_fence;
    xc.timeout_ticks = (unsigned long long)(g_xc_timeout_ms > 0 ? g_xc_timeout_ms : 1) * (XC_WALL_HZ / 1000ull);
    if ((const void*)out == (const void*)xc.stage[rank] || (const void*)resid == (const void*)xc.stage[rank]) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SgRope rp = {};
    SgKsplit kx = {nullptr, nullptr, nullptr};
#if SG_STAMPS
    {
        SgKsplit tmp;
        SgArgs dummy = a;
        sg_pick_ksplit<1, 1, 8, 1>(dummy, tmp);
        kx.stamps = tmp.stamps;
    }
#endif
    const int nchunks = K >> 5;
    const bool wide = nchunks >= 2 * SG_WAVES_WIDE;
#define XCHG_LAUNCH(MT_, W_)                                                                                                  \
    hipLaunchKernelGGL((skinny_gemm_kernel<MT_, SG_PLAIN, false, W_, 1, false, true, 1>), dim3(N / 16), dim3(W_ * 64), 0, st, \
                       (const half8*)a.wp, (const half8*)nullptr, (const h16*)a.x, (const float*)nullptr,              \
                       (const h16*)nullptr, a.K, a.M, (int)a.xa.sm, (int)a.xa.sk, 0.f, a.N, (const h16*)a.resid, a.ra, a.y, a.ya, rp, a.ss_out, \
                       kx, xc)
    if (M <= 16) {
        if (wide) XCHG_LAUNCH(1, SG_WAVES_WIDE);
        else XCHG_LAUNCH(1, SG_WAVES);
    } else {
        if (wide) XCHG_LAUNCH(2, SG_WAVES_WIDE);
        else XCHG_LAUNCH(2, SG_WAVES);
    }
#undef XCHG_LAUNCH
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// Knobs of the fused exchange: key 0 = fenced form (0 / 1, see g_xc_fence), key 1 = wall-clock limit of one flag wait in
// milliseconds.  Returns the previous value, -1 for an unknown key.  Applies to launches (and captures) made afterwards.
extern "C" int tf_xchg_tune(int key, int value) {
    int* slot = key == 0 ? &g_xc_fence : key == 1 ? &g_xc_timeout_ms : nullptr;
    if (!slot) return -1;
    const int old = *slot;
    if (key == 0 && value != 0 && value != 1) return old;
    if (key == 1 && value < 1) return old;
    *slot = value;
    return old;
}

// Back to the state of a freshly allocated control buffer: every peer flag, this rank's per-panel exchange counts and the
// sticky error word zeroed (the host mirror pointer is kept, its word cleared).  COLLECTIVE by contract: every rank of the
// group resets between two barriers with no exchange in flight — the per-panel counts of the ranks must leave together
// (after a time-out they have drifted apart and nothing else can bring them back: advisor, round 4).  Blocking.
extern "C" int tf_xchg_reset(void* ctl) {
    if (!ctl) return TF_EINVAL;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    XcCtl c;
    e = hipMemcpy(&c, ctl, sizeof(c), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    const unsigned long long mirror = c.mirror;
    e = hipMemset(ctl, 0, (size_t)tf_xchg_ctl_bytes());
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(&reinterpret_cast<XcCtl*>(ctl)->mirror, &mirror, sizeof(mirror), hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
    if (mirror) *reinterpret_cast<volatile unsigned*>(mirror) = 0u;
    return (int)hipDeviceSynchronize();
}

// error word / epoch of an exchange control buffer (blocking host reads), fault injection, host mirror — the counterparts
// of tf_ar_error / tf_ar_epoch / tf_ar_inject_error / tf_ar_set_error_mirror for the fused form
extern "C" int tf_xchg_error(const void* ctl) {
    if (!ctl) return TF_EINVAL;
    XcCtl c;
    hipError_t e = hipMemcpy(&c, ctl, sizeof(c), hipMemcpyDeviceToHost);
    return e == hipSuccess ? (int)c.error : (int)e;
}
extern "C" int tf_xchg_set_error(void* ctl, int code, void* host_mirror, int set_mirror) {
    if (!ctl || code < 0) return TF_EINVAL;
    XcCtl* c = reinterpret_cast<XcCtl*>(ctl);
    const unsigned v = (unsigned)code;
    hipError_t e = hipMemcpy(&c->error, &v, sizeof(v), hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
    if (set_mirror) {
        const unsigned long long m = (unsigned long long)(uintptr_t)host_mirror;
        e = hipMemcpy(&c->mirror, &m, sizeof(m), hipMemcpyHostToDevice);
        if (e != hipSuccess) return (int)e;
    }
    unsigned long long m = 0;
    e = hipMemcpy(&m, &c->mirror, sizeof(m), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (m) *reinterpret_cast<volatile unsigned*>(m) = v;
    return TF_OK;
}
