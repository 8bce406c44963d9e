// Split-KV verify/decode attention for gfx950 (MI355X).
//
// Replaces flash_attn_with_kvcache at models/modeling_llama.py:240, models/tensor_op.py:168,316
// (target verify over the full KV, retrieval verify over the retrieval cache, q=1 decode) and,
// in the rope-on-read variant, models/modeling_llama_68m.py:151-190 (Llama-68M draft).
//
// Design (HBM-bound: q <= 32 rows against 4K..130K keys, arithmetic intensity ~ q flop/B):
//   * grid = (nsplit, H); a workgroup = 4 waves; the key range of a split is walked in tiles of
//     16 keys, wave w taking tiles w, w+4, ... so the 4 waves stream adjacent 4 KiB pieces.
//   * K and V tiles go global -> VGPR directly in MFMA operand shape (lane (i, g) = key i,
//     8 contiguous d at 32c+8g: one 16-B load), two tiles deep (8 KiB in flight per wave);
//     no LDS and no barrier in the main loop.
//   * S^T = K Q^T with v_mfma_f32_16x16x32_f16 ("swapped" QK^T): a lane then owns 4 keys of one
//     query, so the exponentiated tile is already the B operand of the PV MFMA — P never
//     leaves registers.
//   * V must be contracted over keys, but its memory order is d-contiguous.  The tile is
//     transposed on the matrix core itself: V_tile x Sel (a constant 0/1 selection operand)
//     returns V in C-layout = 4 consecutive keys of one d per lane = the A operand of
//     v_mfma_f32_16x16x16_f16.  The extra MFMAs are free (the kernel uses a few % of MFMA peak).
//   * online softmax in fp32 (running max / sum per query column), partial (m, l, O) per split
//     merged by a second tiny kernel.
#include "common.h"
#include "tree_mask.h"

#define NEG_BIG (-1.0e30f)
// The LDS block kernel reads a tree row's 8 visibility bits (8 consecutive keys per lane) with one 64-bit funnel shift
// (tree_mask.h) instead of 8 address computations and loads: 512-node Sequoia verify over a 125K prefix 3 140 -> 2 900 us
// (profiles/r02_gemm_pipeline_ab.jsonl, tune.py "attn_tree_verify_512"); 0 keeps the per-key form for bisecting.
#ifndef TF_TREE_MASK_FUNNEL
#define TF_TREE_MASK_FUNNEL 1
#endif
#define ATTN_SPLIT_BOUNDS __launch_bounds__(256)
// Waves per SIMD the two-q-tile form of the split-KV kernel (17..32 query rows: the gamma = 16 verifies) is compiled
// for.  Left to itself the compiler takes 298 registers for it (1 wave per SIMD, 4 KV tiles in flight per CU); held to
// 2 waves it fits 236 without scratch and streams 8-15 % faster at every shape of profiles/r02_nsplit_sweep.json
// (16 heads x 130 066 keys x 17 rows: 225 -> 206 us).  Measured and dropped in the same sweep: transposing each V
// fragment right before its PV MFMA to run the one-q-tile form at 3 waves per SIMD — no gain beyond run-to-run noise.
#ifndef TF_ATTN_QT2_OCC
#define TF_ATTN_QT2_OCC 2
#endif
// The LDS block kernel is VALU-issue-bound (≈270 non-MFMA instructions per 32 MFMAs with libm's exp2f): its softmax uses
// the bare v_exp_f32 (exp2f wraps it in a denormal-range fix-up: compare, 2 selects, ldexp) with the score scale folded
// into one fma: 580 -> 628 TF/s on a 1024-row chunk.  Results below 2^-126 flush to zero — probabilities of keys 2^126
// times below the row max.  (s_setprio around the PV MFMA cluster measured +0.8 %, inside the noise: not kept.)
#define COMBINE_GROUPS 8           // independent accumulation chains of the split merge (fixed: part of the arithmetic)
#define COMBINE_MAX_SPLITS 128
#define FUSED_MERGE_MAX_SPLITS 8   // = COMBINE_GROUPS: each split is then one chain, merged in registers by one workgroup
// Round 4: the in-launch merge also for MANY splits when the grid is small — the 4 / 5 heads of a tensor-parallel rank run
// 32-64 splits per head (one workgroup per CU), whose merge was a second launch: 6.0 us of attn_combine_kernel behind an
// 8.9 us split kernel (profiles/r04_tp8_7b_kernel_timeline_before.json).  First form — the last workgroup of a head folds
// all splits — measured 2x SLOWER than the merge kernel (retrieval verify 1 968 -> 2 147 us: one workgroup's serial fold
// against 28 x 1 024 threads).  Second form (this one): a RENDEZVOUS — all workgroups of a head wait for its last arrival,
// then each folds its slice of the output, 8 threads per element = attn_combine_kernel's 8 chains (bit-identical).  Only
// while every workgroup of the launch is resident: H * nsplit <= FUSED_MERGE_BIG_MAX_WGS, H <= 64.  Measured
// (profiles/r04_attn_rendezvous_merge_ab.jsonl): it loses too, by less — 7B TP-8 retrieval verify 1 654 -> 1 742 us (+2.8 us
// per layer), 7B TP 4 1 819 -> 2 040, 13B TP 8 3 007 -> 3 191: a poll of the head's counter across XCDs, the uncached
// read-back of the partials and the second counter cost ~9 us where the merge kernel costs 6 with its launch.  OFF by
// default (tf_attn_tune(0, 1) turns it on; the tests do, so the path stays checked).
#ifndef FUSED_MERGE_BIG_SPLITS
#define FUSED_MERGE_BIG_SPLITS 64
#endif
#ifndef FUSED_MERGE_BIG_MAX_WGS
#define FUSED_MERGE_BIG_MAX_WGS 256
#endif
#ifndef TF_ATTN_EAGER_TILES
#define TF_ATTN_EAGER_TILES 16 // splits of up to this many 16-key tiles per wave use the unconditional-prefetch loop
#endif

// Two deeper load pipelines, MEASURED AND REJECTED in round 3 (profiles/r03_attn_pipeline_ab.jsonl; cold-cache hipGraph
// chains, tools/attn_variants_ab.py) — both compile-time off, kept buildable (tools/ab_variants.py "deep8" / "ring4"):
//   TF_ATTN_DEEP_TILES = N   short streams (a wave owns <= 2 N tiles): attn_split_deep_kernel issues the loads of N
//       tiles (8 KiB each; one wave per SIMD = 512 registers) before the first MFMA and consumes them in order.  The
//       hypothesis — the 4 103-key retrieval verify is latency-bound in the two-deep loop, 4 dependent round trips per
//       wave — is wrong: N = 8 takes it from 20.8 to 25.4 us per launch, the draft-sized stream from 6.5 to 7.5.
//   TF_ATTN_RING_Q1 / _Q2 = N   long streams: a ring of N tiles in flight per wave, every load unconditional (so the
//       waitcnt pass keeps vmcnt(8 (N - 1)) in front of each tile — the two-q-tile form's conditional prefetch runs one
//       tile deep).  N = 4: 32 heads x 125K keys 346.9 -> 347.6 us, 16 heads x 17 rows x 130K keys 195.2 -> 204.3 us,
//       32 heads x 18 rows 370.0 -> 376.2 us.  More bytes in flight than the shipped loops keep buys nothing here.
// Both forms walk the same tiles in the same order per wave: bit-identical partials.
#ifndef TF_ATTN_DEEP_TILES
#define TF_ATTN_DEEP_TILES 0
#endif
#ifndef TF_ATTN_RING_Q1
#define TF_ATTN_RING_Q1 0
#endif
#ifndef TF_ATTN_RING_Q2
#define TF_ATTN_RING_Q2 0
#endif

// agent-scope relaxed accesses: global_store / global_load ... sc1 (write-through / L2-coherent across XCDs)
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// max over the 4 lane groups of a wave (lanes i, i^16, i^32, i^48) on the VALU: v_permlane{32,16}_swap + v_max instead of
// two ds_bpermute round trips through the LDS queue (and their lgkmcnt(0) waits in the middle of the softmax chain).
__device__ __forceinline__ float vmax_raw(float a, float b) {          // plain v_max_f32: no canonicalising pre-ops
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The asm max helpers above are invisible to the compiler's hazard recogniser: it inserts the wait states an MFMA result
// needs before a VALU instruction reads it (11 for an 8-pass v_mfma_f32_16x16x32_f16) in front of ITS OWN instructions,
// not in front of inline asm — in one build of the prefill slab the scheduler put a v_max3_f32 directly behind the MFMA
// that produced its operand, and the row maxima were garbage (NaN outputs).  Scores that come straight out of MFMAs pass
// through this block first: the values become outputs of an asm that waits 12 cycles, so the MFMAs are issued before it
// and every later reader — asm or not — sees settled registers.  12 cycles per (q-tile, sub-step).
__device__ __forceinline__ void mfma_settle8(float (&x)[8]) {
    asm volatile("s_nop 7\n\ts_nop 3"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ float group_max4(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// P = exp(S - m) goes to the PV MFMA as fp16.  Rounded ONCE (flash-attn's choice; TF_ATTN_P_SPLIT 0) that costs the
// attention output ~1 fp16 ulp on average: 36 % of the outputs of a 7 x 4 103-key retrieval verify differ from the
// exactly accumulated result, and a 7B-width layer's logits end up 1.4x further from it than the CPU oracle's (fp32 P)
// are (tests/test_gpu_configs.py, fp64 truth; profiles/r03_attn_pipeline_ab.jsonl).  With TF_ATTN_P_SPLIT 1 (default) P
// is fed as hi + lo — lo = fp16(p - hi), a possibly subnormal fp16 the matrix core takes at full precision — into the
// same accumulator: 8 more MFMAs per tile and q-tile on a matrix core that is ~9 % busy.  0.16 % of the outputs then
// differ from exact (mean error 0.25 ulp = the final rounding alone) for +0.9 % on the 125K-key stream.
#ifndef TF_ATTN_P_SPLIT
#define TF_ATTN_P_SPLIT 1
#endif
// The same hi + lo feed in the block / prefill / tree kernels (pair_softmax_pv, lds_softmax_*) and in the 68M draft's
// rope-on-read kernel: TF_BLOCK_P_SPLIT / TF_DRAFT_P_SPLIT.  Measured in round 4 (tools/prefill_psplit_ab.py,
// profiles/r04_psplit_block_ab.jsonl) — see the note there for what ships and why.
#ifndef TF_BLOCK_P_SPLIT
#define TF_BLOCK_P_SPLIT 0
#endif
#ifndef TF_DRAFT_P_SPLIT
#define TF_DRAFT_P_SPLIT 1
#endif


template <int D, int QT>
struct AttnState {
    static constexpr int NC = D / 32;
    static constexpr int NT = D / 16;
    half8 qf[QT][NC];
    f32x4 acc[QT][NT];
    float m[QT];
    float l[QT];
};

template <int D>
__device__ __forceinline__ void load_kv_tile(const h16* __restrict__ kbase, const h16* __restrict__ vbase,
                                             int64_t stride_t, int tile, int sk, int li, int g,
                                             half8 (&kf)[D / 32], half8 (&vf)[D / 32]) {
    int key = tile * 16 + li;
    key = key < sk ? key : sk - 1;                     // clamp: masked below, but must stay in-bounds
    const int64_t off = (int64_t)key * stride_t + 8 * g;
#pragma unroll
    for (int c = 0; c < D / 32; ++c) {
        kf[c] = load_half8_stream(kbase + off + 32 * c);
        vf[c] = load_half8_stream(vbase + off + 32 * c);
    }
}

// Round 6: the same tile fetched as FULL ROWS.  load_kv_tile above issues MFMA A-operand fragments straight from memory: every
// load instruction touches sixteen rows x 64 bytes (lane (li, g): key li, dims 32 c + 8 g), and a kernel that only reads with that
// pattern reaches 5.1 TB/s where 256-byte rows, four per instruction, reach 7.0 (tools/probes/hbm_read_probe.hip,
// profiles/r06_hbm_read_probe.jsonl).  Here lane l of instruction j reads 16 bytes of row RPI j + l / LPR at piece l % LPR
// (LPR = D / 8 lanes per row, RPI = 64 / LPR rows per instruction; as many instructions as before), and the consumer turns the
// rows into the fragments load_kv_tile would have produced through a wave-private LDS tile (rows padded by 16 bytes: the
// fragment reads of a 16-lane group then cover all banks).  Same values in the same registers: the kernel's output bits do not change.
#ifndef TF_ATTN_ROW_LOADS
#define TF_ATTN_ROW_LOADS 1
#endif
template <int D>
__device__ __forceinline__ void load_kv_rows(const h16* __restrict__ kbase, const h16* __restrict__ vbase, int64_t stride_t, int tile,
                                             int sk, int lane, half8 (&kr)[D / 32], half8 (&vr)[D / 32]) {
    constexpr int LPR = D / 8, RPI = 64 / LPR;
    const int rl = lane / LPR, q = lane % LPR;
#pragma unroll
    for (int j = 0; j < D / 32; ++j) {
        int key = tile * 16 + RPI * j + rl;
        key = key < sk ? key : sk - 1;                 // clamp: masked below, but must stay in-bounds
        const int64_t off = (int64_t)key * stride_t + 8 * q;
        kr[j] = load_half8_stream(kbase + off);
        vr[j] = load_half8_stream(vbase + off);
    }
}
// stage: this wave's [2][16][D + 8] halfs of LDS.  One wave writes and reads it (LDS serves a wave's operations in order): no barrier.
template <int D>
__device__ __forceinline__ void rows_to_frags(const half8 (&kr)[D / 32], const half8 (&vr)[D / 32], h16* stage, int lane, int li, int g,
                                              half8 (&kf)[D / 32], half8 (&vf)[D / 32]) {
    constexpr int LPR = D / 8, RPI = 64 / LPR, LDR = D + 8;
    const int rl = lane / LPR, q = lane % LPR;
#pragma unroll
    for (int j = 0; j < D / 32; ++j) {
        *reinterpret_cast<half8*>(stage + (RPI * j + rl) * LDR + 8 * q) = kr[j];
        *reinterpret_cast<half8*>(stage + (16 + RPI * j + rl) * LDR + 8 * q) = vr[j];
    }
#pragma unroll
    for (int c = 0; c < D / 32; ++c) {
        kf[c] = *reinterpret_cast<const half8*>(stage + li * LDR + 32 * c + 8 * g);
        vf[c] = *reinterpret_cast<const half8*>(stage + (16 + li) * LDR + 32 * c + 8 * g);
    }
}

struct TreeMask {                      // tree-attention visibility bits (block kernel, TREE = true)
    const uint32_t* rows;              // [n_rows][words] uint32, bit j of a row = tree key j visible
    int words, row0, start;            // words per row, first row of this launch, key index of tree key 0
};

// MASKED = false: every key of the tile is visible to every query row of the wave (all tiles but the last one or
// two of a causal stream; the unmasked prefix of a tree pass) — the per-key compare/select chain is dropped.
template <int D, int QT, bool TREE = false, bool MASKED = true>
__device__ __forceinline__ void attn_tile(AttnState<D, QT>& st, const half8 (&kf)[D / 32],
                                          const half8 (&vf)[D / 32], half8 sel0, half8 sel1, int tile,
                                          int sk, int sq, float scale, int li, int g, int qbase = 0,
                                          TreeMask tm = TreeMask{nullptr, 0, 0, 0}) {
    constexpr int NC = D / 32, NT = D / 16;
    // V tile -> key-contiguous fragments through the matrix core (exact: multiplies by 0/1)
    half4 va[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[t >> 1], (t & 1) ? sel1 : sel0, z, 0, 0, 0);
        va[t] = half4{(h16)r[0], (h16)r[1], (h16)r[2], (h16)r[3]};
    }
    half4 pb[QT];
#if TF_ATTN_P_SPLIT > 0
    half4 pl[QT];
#endif
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[c], st.qf[qt][c], s, 0, 0, 0);
        // lane holds S^T[key = tile*16 + 4g + r][q = qt*16 + li]
        const int qrow = qbase + qt * 16 + li;
        const int kmax = (qrow < sq) ? (sk - sq + qrow) : (sk - 1);   // bottom-right causal
        float x[4];
        bool ok[4];
        float tmax = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kidx = tile * 16 + 4 * g + r;
            if (!MASKED) {
                ok[r] = true;
            } else if (TREE) {
                const int j = kidx - tm.start;
                ok[r] = kidx < sk;
                if (j >= 0 && ok[r]) {
                    const int mrow = tm.row0 + min(qrow, sq - 1);
                    ok[r] = (tm.rows[(int64_t)mrow * tm.words + (j >> 5)] >> (j & 31)) & 1u;
                }
            } else {
                ok[r] = kidx <= kmax;
            }
            x[r] = s[r] * scale;
            tmax = ok[r] ? fmaxf(tmax, x[r]) : tmax;
        }
        tmax = group_max4(tmax);
        const float mnew = fmaxf(st.m[qt], tmax);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = ok[r] ? __expf(x[r] - mnew) : 0.f;
            psum += p;
            pb[qt][r] = (h16)p;
#if TF_ATTN_P_SPLIT > 0
            pl[qt][r] = (h16)(p - (float)pb[qt][r]);
#endif
        }
        // The running maximum settles after the first tiles of a stream; rescaling the 32 accumulator registers
        // (alpha == 1 exactly when no query column of the wave raised its maximum) is skipped wave-uniformly then.
        if (__builtin_amdgcn_ballot_w64(mnew != st.m[qt])) {
            const float alpha = __expf(st.m[qt] - mnew);
            st.l[qt] *= alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                st.acc[qt][t][0] *= alpha; st.acc[qt][t][1] *= alpha;
                st.acc[qt][t][2] *= alpha; st.acc[qt][t][3] *= alpha;
            }
            st.m[qt] = mnew;
        }
        st.l[qt] += psum;
    }
    // PV: one V fragment feeds the MFMAs of every q-tile
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(va[t], pb[qt], st.acc[qt][t], 0, 0, 0);
#if TF_ATTN_P_SPLIT > 0
            st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(va[t], pl[qt], st.acc[qt][t], 0, 0, 0);
#endif
        }
}


// ws layout: o[H][nsplit][QR][D] | m[H][nsplit][QR] | l[H][nsplit][QR],  QR = QT*16
template <int D, int QT, int DEEP = 0, int NW = 4>
__device__ __forceinline__ void attn_split_body(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, int64_t stride_t,
    int64_t stride_h, int sq, int sk_host, const int32_t* __restrict__ sk_dev, int H, float scale, int nsplit,
    float* __restrict__ ws, unsigned* __restrict__ tickets, h16* __restrict__ out, int64_t osm, int64_t osk) {
    constexpr int NC = D / 32, NT = D / 16, QR = QT * 16;
    const int split = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    int sk = sk_dev ? *sk_dev : sk_host;
    if (sk > sk_host) sk = sk_host;
    if (sk < 1) sk = 1;

    const int ntiles = (sk + 15) >> 4;
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int t_begin = split * tps;
    const int t_end = min(ntiles, t_begin + tps);

    AttnState<D, QT> st;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = qt * 16 + li;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            st.qf[qt][c] = (row < sq) ? load_half8(q + ((int64_t)row * H + h) * D + 32 * c + 8 * g) : z;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) st.acc[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.m[qt] = NEG_BIG;
        st.l[qt] = 0.f;
    }
    half8 sel0, sel1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sel0[e] = (8 * g + e == li) ? (h16)1.0f : (h16)0.0f;
        sel1[e] = (8 * g + e == 16 + li) ? (h16)1.0f : (h16)0.0f;
    }

    const h16* kbase = k + (int64_t)h * stride_h;
    const h16* vbase = v + (int64_t)h * stride_h;

    // One block of LDS: the row -> fragment staging tiles of the stream loop (TF_ATTN_ROW_LOADS, 4-wave forms), then — behind a
    // barrier — the merge of the waves' partial results
    constexpr int HALVES = NW > 4 ? 2 : 1, DH = D / HALVES, NTH = NT / HALVES;
    constexpr bool ROWS = TF_ATTN_ROW_LOADS > 0 && NW == 4;
    constexpr size_t MERGE_BYTES = sizeof(float) * ((size_t)NW * 16 * (DH + 1) + 2 * NW * 16);
    constexpr size_t STAGE_BYTES = ROWS ? (size_t)NW * 2 * 16 * (D + 8) * sizeof(h16) : 0;
    __shared__ __attribute__((aligned(16))) unsigned char sm_raw[MERGE_BYTES > STAGE_BYTES ? MERGE_BYTES : STAGE_BYTES];
    h16* stage = reinterpret_cast<h16*>(sm_raw) + (size_t)wave * 2 * 16 * (D + 8);

    // tiles whose 16 keys are all <= sk - sq are visible to every row: no mask arithmetic
#define ATTN_TILE_FRAGS(KF, VF, T)                                                                    \
    do {                                                                                                \
        if ((T) * 16 + 15 <= sk - sq) attn_tile<D, QT, false, false>(st, KF, VF, sel0, sel1, (T), sk, sq, scale, li, g); \
        else attn_tile<D, QT, false, true>(st, KF, VF, sel0, sel1, (T), sk, sq, scale, li, g);        \
    } while (0)
    // (ROWS: what was loaded are rows; the fragments are made here, right before their use)
#define ATTN_TILE_AUTO(KX, VX, T)                                                                     \
    do {                                                                                                \
        if constexpr (ROWS) {                                                                           \
            half8 kf_[NC], vf_[NC];                                                                     \
            rows_to_frags<D>(KX, VX, stage, lane, li, g, kf_, vf_);                                     \
            ATTN_TILE_FRAGS(kf_, vf_, T);                                                               \
        } else {                                                                                        \
            ATTN_TILE_FRAGS(KX, VX, T);                                                                 \
        }                                                                                               \
    } while (0)
#define ATTN_LOAD_TILE(T, KX, VX)                                                                     \
    do {                                                                                                \
        if constexpr (ROWS) load_kv_rows<D>(kbase, vbase, stride_t, (T), sk, lane, KX, VX);             \
        else load_kv_tile<D>(kbase, vbase, stride_t, (T), sk, li, g, KX, VX);                           \
    } while (0)
    // Two forms of the same two-tiles-deep loop.  A load under `if (t1 < t_end)` makes the compiler assume the worst
    // case at the use of the OLDER tile — "no younger load was issued" — so it waits vmcnt(7..0) there, i.e. for the
    // prefetch it has just issued: the wave runs one tile deep.  Issuing the run-ahead loads unconditionally (past the
    // last tile they re-read it) gives the intended vmcnt(15..8).  Measured (tools/attn_merge_ab.py): short streams gain
    // (7B retrieval verify 21.0 -> 20.2 us, 4-head TP shard 14.3 -> 13.6, draft-sized 6.9 -> 6.5) but the 125K-key
    // streams, already bandwidth-bound with 8 MB in flight chip-wide, LOSE 0.7 % (32 heads) to 4 % (16 heads x 17 rows)
    // with twice as much in flight — so the form is chosen by the length of the wave's stream.
    static_assert(NW == 4 || (DEEP == 0 && (QT == 1 ? TF_ATTN_RING_Q1 : TF_ATTN_RING_Q2) == 0), "deep / ring forms: 4 waves");
    int t = t_begin + wave;
    if constexpr (DEEP > 0) {
        // rounds of N tiles: all N loads issued (tiles past the end re-read the last one: no conditional load, so the
        // waitcnt pass keeps vmcnt(8 (N - 1 - i)) in front of tile i), then consumed in order
#define ATTN_DEEP_ROUND(N)                                                                                   \
    do {                                                                                                       \
        half8 kd[N][NC], vd[N][NC];                                                                            \
        const int tl = t_end - 1;                                                                              \
        _Pragma("unroll") for (int i = 0; i < (N); ++i)                                                        \
            ATTN_LOAD_TILE(min(t + 4 * i, tl), kd[i], vd[i]);                        \
        _Pragma("unroll") for (int i = 0; i < (N); ++i) {                                                      \
            const int ti = t + 4 * i;                                                                          \
            if (ti < t_end) ATTN_TILE_AUTO(kd[i], vd[i], ti);                                                  \
        }                                                                                                      \
        t += 4 * (N);                                                                                          \
    } while (0)
        while (t < t_end) {
            const int left = (t_end - t + 3) >> 2;               // tiles this wave still owns (wave-uniform)
            if (left > DEEP / 2) ATTN_DEEP_ROUND(DEEP);
            else if (left > 1) ATTN_DEEP_ROUND(DEEP / 2);
            else ATTN_DEEP_ROUND(1);
        }
#undef ATTN_DEEP_ROUND
    } else if constexpr ((QT == 1 ? TF_ATTN_RING_Q1 : TF_ATTN_RING_Q2) > 0) {
        constexpr int RING = QT == 1 ? TF_ATTN_RING_Q1 : TF_ATTN_RING_Q2;
        if (t < t_end) {
            half8 kr[RING][NC], vr[RING][NC];
            const int tl = t_end - 1;
#pragma unroll
            for (int s = 0; s < RING; ++s) ATTN_LOAD_TILE(min(t + 4 * s, tl), kr[s], vr[s]);
            while (t < t_end) {
#pragma unroll
                for (int s = 0; s < RING; ++s) {
                    const int ti = t + 4 * s;
                    if (ti < t_end) ATTN_TILE_AUTO(kr[s], vr[s], ti);
                    ATTN_LOAD_TILE(min(ti + 4 * RING, tl), kr[s], vr[s]);
                }
                t += 4 * RING;
            }
        }
    } else if (t < t_end) {
        half8 ka[NC], va_[NC], kb[NC], vb[NC];
        ATTN_LOAD_TILE(t, ka, va_);
        // (one-q-tile form only: with both loops the two-q-tile form no longer fits its 2-waves-per-SIMD register budget)
        if (TF_ATTN_EAGER_TILES > 0 && QT == 1 && t_end - t_begin <= NW * TF_ATTN_EAGER_TILES) {
            const int tl = t_end - 1;
            while (true) {
                ATTN_LOAD_TILE(min(t + NW, tl), kb, vb);
                ATTN_TILE_AUTO(ka, va_, t);
                if (t + NW >= t_end) break;
                ATTN_LOAD_TILE(min(t + 2 * NW, tl), ka, va_);
                ATTN_TILE_AUTO(kb, vb, t + NW);
                if (t + 2 * NW >= t_end) break;
                t += 2 * NW;
            }
        } else {
            while (t < t_end) {
                const int t1 = t + NW;
                if (t1 < t_end) ATTN_LOAD_TILE(t1, kb, vb);
                ATTN_TILE_AUTO(ka, va_, t);
                if (t1 >= t_end) break;
                const int t2 = t1 + NW;
                if (t2 < t_end) ATTN_LOAD_TILE(t2, ka, va_);
                ATTN_TILE_AUTO(kb, vb, t1);
                t = t2;
            }
        }
    }

#undef ATTN_TILE_AUTO
#undef ATTN_TILE_FRAGS
#undef ATTN_LOAD_TILE
    // ---- merge the NW waves of this split through LDS, one q-tile at a time (8 waves: one half of D at a time, so the
    // staging stays under the 64 KiB static limit) ----
    float (*sm_o)[16][DH + 1] = reinterpret_cast<float (*)[16][DH + 1]>(sm_raw);
    float (*sm_m)[16] = reinterpret_cast<float (*)[16]>(sm_raw + sizeof(float) * (size_t)NW * 16 * (DH + 1));
    float (*sm_l)[16] = sm_m + NW;
    if constexpr (ROWS) __syncthreads();               // another wave may still be reading its staging tile where this one is about to write
    float* ws_o = ws;
    float* ws_m = ws + (int64_t)H * nsplit * QR * D;
    float* ws_l = ws_m + (int64_t)H * nsplit * QR;
    const int64_t pbase = ((int64_t)h * nsplit + split) * QR;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lsum = st.l[qt];
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
#pragma unroll
        for (int hv = 0; hv < HALVES; ++hv) {
#pragma unroll
            for (int tt = 0; tt < NTH; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm_o[wave][li][16 * tt + 4 * g + r] = st.acc[qt][hv * NTH + tt][r];
            if (g == 0) {
                sm_m[wave][li] = st.m[qt];
                sm_l[wave][li] = lsum;
            }
            __syncthreads();
            for (int e = tid; e < 16 * DH; e += 64 * NW) {
                const int qq = e / DH, dl = e - qq * DH, d = hv * DH + dl;
                float mm = sm_m[0][qq];
#pragma unroll
                for (int w = 1; w < NW; ++w) mm = fmaxf(mm, sm_m[w][qq]);
                // sum over the waves in wave order (for NW = 4 the very expression of round 2: ((a0 w0 + a1 w1) + a2 w2) + a3 w3)
                float o = 0.f, lw = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const float ww = __expf(sm_m[w][qq] - mm);
                    o = (w == 0) ? sm_o[w][qq][dl] * ww : o + sm_o[w][qq][dl] * ww;
                    lw = (w == 0) ? sm_l[w][qq] * ww : lw + sm_l[w][qq] * ww;
                }
                if (tickets == nullptr) {
                    ws_o[(pbase + qt * 16 + qq) * D + d] = o;
                    if (d == 0) {
                        ws_m[pbase + qt * 16 + qq] = mm;
                        ws_l[pbase + qt * 16 + qq] = lw;
                    }
                } else {                                   // one-launch form: write-through (sc1) stores, see below
                    st_agent(&ws_o[(pbase + qt * 16 + qq) * D + d], o);
                    if (d == 0) {
                        st_agent(&ws_m[pbase + qt * 16 + qq], mm);
                        st_agent(&ws_l[pbase + qt * 16 + qq], lw);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (tickets == nullptr) return;                    // two-launch form: attn_combine_kernel merges the splits

    // ---- fused merge: the LAST workgroup of this head to finish folds the nsplit partials (same arithmetic, same
    // order as attn_combine_kernel: bit-identical output) — no second launch, no kernel boundary.  The partials cross
    // XCDs (one L2 each): they are written with agent-scope write-through stores and read back with agent-scope loads
    // (sc1), ordered by "own stores landed (vmcnt 0) -> ticket".  An agent-scope release FENCE instead writes the whole
    // L2 back: measured 21.7 -> 47.9 us on the 7B retrieval-verify shape.
    __shared__ int s_last;
    // Every thread drains ITS OWN write-through stores before the barrier that precedes the ticket: a workgroup-scope
    // release fence emits no s_waitcnt vmcnt(0) on gfx950 (the ISA was store -> s_barrier -> atomic), so the ticket
    // could overtake the partials and the last arriver — on another XCD — could fold the previous launch's values.
    // Inline asm: the compiler's waitcnt pass may not drop or move it (MI355X_MICROARCH.md, compiler hazard).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        s_last = __hip_atomic_fetch_add(&tickets[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nsplit - 1);
    __syncthreads();
    const int64_t hbase = (int64_t)h * nsplit * QR;
    if (D == 128 && nsplit > FUSED_MERGE_MAX_SPLITS) {
        // ---- many splits, small grid (host-checked: every workgroup of the launch is resident): RENDEZVOUS merge ----
        // All nsplit workgroups of the head wait for the head's last arrival, then each folds ITS slice of the output —
        // elements e = split, split + nsplit, ... of the sq x D/4 float4 items — 8 threads per element, one per
        // accumulation chain of attn_combine_kernel (chain g folds splits g, g + 8, ...; chains summed in order; the
        // normaliser summed over the splits in order): the same bits as the merge kernel, without its launch, and the
        // fold is spread over all the head's workgroups instead of one (the one-workgroup fold measured 2x slower than
        // the merge kernel: profiles/r04_tp_shard_structural_ab.jsonl).  A second counter (tickets[64 + h]) counts the
        // slices done; whoever finishes last leaves both words zero for the next launch.
        if (tid == 0) {
            for (unsigned spins = 0; spins < (1u << 24); ++spins) {
                if (__hip_atomic_load(&tickets[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nsplit) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        const int E = sq * (D / 4);
        const int cnt = (E - split + nsplit - 1) / nsplit;             // elements of this workgroup's slice (may be <= 0)
        const int grp = tid >> 3, gch = tid & 7;                       // 32 elements per pass x 8 chains
        for (int i0 = 0; i0 < cnt; i0 += 32) {
            const int i = i0 + grp;
            const bool on = i < cnt;
            const int e = on ? split + i * nsplit : 0;
            const int r = e / (D / 4), d4 = e - r * (D / 4);
            // every load of this chain issued up front: <= 8 splits x (4 + 2) values
            f32x4 px[FUSED_MERGE_BIG_SPLITS / 8];
            float pm[FUSED_MERGE_BIG_SPLITS / 8], pl[FUSED_MERGE_BIG_SPLITS / 8];
#pragma unroll
            for (int jj = 0; jj < FUSED_MERGE_BIG_SPLITS / 8; ++jj) {
                const int sp = gch + 8 * jj;
                if (on && sp < nsplit) {
                    const float* xp = ws_o + (hbase + (int64_t)sp * QR + r) * D + 4 * d4;
                    px[jj] = f32x4{ld_agent(xp), ld_agent(xp + 1), ld_agent(xp + 2), ld_agent(xp + 3)};
                    pm[jj] = ld_agent(&ws_m[hbase + (int64_t)sp * QR + r]);
                    pl[jj] = ld_agent(&ws_l[hbase + (int64_t)sp * QR + r]);
                } else {
                    px[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                    pm[jj] = NEG_BIG;
                    pl[jj] = 0.f;
                }
            }
            float mm = NEG_BIG;                                         // max over ALL splits: fmaxf is order-free
#pragma unroll
            for (int jj = 0; jj < FUSED_MERGE_BIG_SPLITS / 8; ++jj) mm = fmaxf(mm, pm[jj]);
            mm = fmaxf(mm, __shfl_xor(mm, 1, 64));
            mm = fmaxf(mm, __shfl_xor(mm, 2, 64));
            mm = fmaxf(mm, __shfl_xor(mm, 4, 64));
            f32x4 ch = {0.f, 0.f, 0.f, 0.f};
            float lw[FUSED_MERGE_BIG_SPLITS / 8];
#pragma unroll
            for (int jj = 0; jj < FUSED_MERGE_BIG_SPLITS / 8; ++jj) {
                const float w = (gch + 8 * jj < nsplit) ? __expf(pm[jj] - mm) : 0.f;
                lw[jj] = pl[jj] * w;                                    // attn_combine_kernel: sm_l[s] *= w
                if (gch + 8 * jj < nsplit) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ch[c] = fmaf(px[jj][c], w, ch[c]);
                }
            }
            // normaliser: l = sum over s = 0 .. nsplit-1 IN ORDER of l_s w_s (s = 8 jj + chain) — gathered by shuffles
            const int lane0 = (threadIdx.x & 63) & ~7;
            float l = 0.f;
#pragma unroll
            for (int jj = 0; jj < FUSED_MERGE_BIG_SPLITS / 8; ++jj)
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    const float v = __shfl(lw[jj], lane0 + gg, 64);
                    if (8 * jj + gg < nsplit) l += v;
                }
            // chains summed in chain order
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gg = 0; gg < 8; ++gg)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += __shfl(ch[c], lane0 + gg, 64);
            if (on && gch == 0) {
                half4 o4;
#pragma unroll
                for (int c = 0; c < 4; ++c) o4[c] = (h16)(acc[c] / l);
                *reinterpret_cast<half4*>(out + (int64_t)r * osm + (int64_t)((h * D + 4 * d4) >> 3) * osk + ((4 * d4) & 7)) = o4;
            }
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned dn = __hip_atomic_fetch_add(&tickets[64 + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dn == (unsigned)(nsplit - 1)) {
                __hip_atomic_store(&tickets[64 + h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&tickets[h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (!s_last) return;
    // nsplit <= FUSED_MERGE_MAX_SPLITS: every load of an output element is issued up front — one memory
    // latency — and each split is its own accumulation chain of attn_combine_kernel
    for (int e = tid; e < sq * (D / 4); e += 64 * NW) {
        const int r = e / (D / 4), d4 = e - r * (D / 4);
        float pm[FUSED_MERGE_MAX_SPLITS], pl[FUSED_MERGE_MAX_SPLITS];
        f32x4 px[FUSED_MERGE_MAX_SPLITS];
#pragma unroll
        for (int s = 0; s < FUSED_MERGE_MAX_SPLITS; ++s) {
            if (s < nsplit) {
                const float* xp = ws_o + (hbase + (int64_t)s * QR + r) * D + 4 * d4;
                px[s] = f32x4{ld_agent(xp), ld_agent(xp + 1), ld_agent(xp + 2), ld_agent(xp + 3)};
                pm[s] = ld_agent(&ws_m[hbase + (int64_t)s * QR + r]);
                pl[s] = ld_agent(&ws_l[hbase + (int64_t)s * QR + r]);
            } else {
                px[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                pm[s] = NEG_BIG;
                pl[s] = 0.f;
            }
        }
        float mm = NEG_BIG;
#pragma unroll
        for (int s = 0; s < FUSED_MERGE_MAX_SPLITS; ++s) mm = fmaxf(mm, pm[s]);
        float l = 0.f;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < FUSED_MERGE_MAX_SPLITS; ++s) {
            float w = 0.f;
            if (s < nsplit) {
                w = __expf(pm[s] - mm);
                l += pl[s] * w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += fmaf(px[s][c], w, 0.f);
        }
        half4 o4;
#pragma unroll
        for (int c = 0; c < 4; ++c) o4[c] = (h16)(acc[c] / l);
        // output element (row r, column h * D + 4 d4 ...) in the caller's activation layout (see tf_attn_decode_act):
        // row-major rows of H * D (osm = H * D, osk = 8) or k-octet-major (osm = 8, osk = 8 * R)
        *reinterpret_cast<half4*>(out + (int64_t)r * osm + (int64_t)((h * D + 4 * d4) >> 3) * osk + ((4 * d4) & 7)) = o4;
    }
    if (tid == 0) __hip_atomic_store(&tickets[h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch (graph replays included)
}

template <int D, int QT>
__global__ ATTN_SPLIT_BOUNDS void attn_split_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, const int32_t* __restrict__ sk_dev,
    int sq, int sk_host, int H, int nsplit, int stride_t, int stride_h, float scale,      // <- 14 dwords preloaded into SGPRs
    float* __restrict__ ws, unsigned* __restrict__ tickets, h16* __restrict__ out, int64_t osm, int64_t osk) {
    attn_split_body<D, QT>(q, k, v, (int64_t)stride_t, (int64_t)stride_h, sq, sk_host, sk_dev, H, scale, nsplit, ws, tickets, out,
                           osm, osk);
}

// The deep-prefetch form for short streams (one q-tile; see TF_ATTN_DEEP_TILES): one wave per SIMD, 512 registers.
#if TF_ATTN_DEEP_TILES > 0
template <int D>
__global__ __launch_bounds__(256, 1) void attn_split_deep_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, const int32_t* __restrict__ sk_dev,
    int sq, int sk_host, int H, int nsplit, int stride_t, int stride_h, float scale,      // <- 14 dwords preloaded into SGPRs
    float* __restrict__ ws, unsigned* __restrict__ tickets, h16* __restrict__ out, int64_t osm, int64_t osk) {
    attn_split_body<D, 1, TF_ATTN_DEEP_TILES>(q, k, v, (int64_t)stride_t, (int64_t)stride_h, sq, sk_host, sk_dev, H, scale, nsplit,
                                              ws, tickets, out, osm, osk);
}
#endif

// The two-q-tile form compiled for TF_ATTN_QT2_OCC waves per SIMD (see the note at the top of the file).
// TF_ATTN_Q2_WAVES: waves per workgroup of the two-q-tile form (4 = round 2; 8 = two waves per SIMD at the one-workgroup-
// per-CU grid, so one wave's softmax / MFMA work runs under the other's loads)
#ifndef TF_ATTN_Q2_WAVES
#define TF_ATTN_Q2_WAVES 4
#endif
#if TF_ATTN_QT2_OCC > 0
template <int D>
__global__ __launch_bounds__(64 * TF_ATTN_Q2_WAVES, TF_ATTN_Q2_WAVES == 8 ? 2 : TF_ATTN_QT2_OCC) void attn_split_q2_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, const int32_t* __restrict__ sk_dev,
    int sq, int sk_host, int H, int nsplit, int stride_t, int stride_h, float scale,      // <- 14 dwords preloaded into SGPRs
    float* __restrict__ ws, unsigned* __restrict__ tickets, h16* __restrict__ out, int64_t osm, int64_t osk) {
    attn_split_body<D, 2, 0, TF_ATTN_Q2_WAVES>(q, k, v, (int64_t)stride_t, (int64_t)stride_h, sq, sk_host, sk_dev, H, scale, nsplit, ws,
                                               tickets, out, osm, osk);
}
#endif

// ---- 32-key step of the block kernel: two 16-key tiles A, B per softmax update ----------------------------
// The QK^T products of both tiles give a lane 8 keys of one query (A keys 4g..4g+3, B keys 4g..4g+3); the V tiles
// transposed on the matrix core give a lane the SAME 8 keys of one d — so exp(S) and V^T are the B and A operands of
// ONE v_mfma_f32_16x16x32_f16 per 16 output columns: half the PV MFMAs and half the running-max bookkeeping of the
// 16-key step.  Scores are kept in the log2 domain (scale * log2 e folded into one multiply, v_exp_f32 is 2^x).
template <int D, int QT>
struct PairScores {
    f32x4 sa[QT], sb[QT];
    half8 va8[D / 16];
};

template <int D, int QT>
__device__ __forceinline__ void pair_qk(const AttnState<D, QT>& st, const half8 (&kfa)[D / 32], const half8 (&vfa)[D / 32],
                                        const half8 (&kfb)[D / 32], const half8 (&vfb)[D / 32], half8 sel0, half8 sel1,
                                        PairScores<D, QT>& ps) {
    constexpr int NC = D / 32, NT = D / 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 ra = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfa[t >> 1], (t & 1) ? sel1 : sel0, z, 0, 0, 0);
        const f32x4 rb = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfb[t >> 1], (t & 1) ? sel1 : sel0, z, 0, 0, 0);
        ps.va8[t] = half8{(h16)ra[0], (h16)ra[1], (h16)ra[2], (h16)ra[3], (h16)rb[0], (h16)rb[1], (h16)rb[2], (h16)rb[3]};
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfa[c], st.qf[qt][c], a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfb[c], st.qf[qt][c], b, 0, 0, 0);
        }
        ps.sa[qt] = a;
        ps.sb[qt] = b;
    }
}

// st.m is kept in the log2 domain by the callers of this function (converted back before the partials are stored)
template <int D, int QT, bool TREE, bool MASKED = true>
__device__ __forceinline__ void pair_softmax_pv(AttnState<D, QT>& st, const PairScores<D, QT>& ps, int tile_a, int tile_b,
                                                bool b_valid, int sk, int sq, float scale_log2, int li, int g, int qbase,
                                                TreeMask tm) {
    constexpr int NT = D / 16;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = qbase + qt * 16 + li;
        const int kmax = (qrow < sq) ? (sk - sq + qrow) : (sk - 1);   // bottom-right causal
        float x[8];
        bool ok[8];
        float tmax = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int kidx = (r < 4 ? tile_a : tile_b) * 16 + 4 * g + (r & 3);
            bool v = (r < 4) || b_valid;
            if (!MASKED) {
                v = true;
            } else if (TREE) {
                const int j = kidx - tm.start;
                v = v && kidx < sk;
                if (j >= 0 && v) {
                    const int mrow = tm.row0 + min(qrow, sq - 1);
                    v = (tm.rows[(int64_t)mrow * tm.words + (j >> 5)] >> (j & 31)) & 1u;
                }
            } else {
                v = v && kidx <= kmax;
            }
            ok[r] = v;
            x[r] = (r < 4 ? ps.sa[qt][r] : ps.sb[qt][r - 4]) * scale_log2;
            tmax = v ? fmaxf(tmax, x[r]) : tmax;
        }
        tmax = group_max4(tmax);
        const float mnew = fmaxf(st.m[qt], tmax);
        float psum = 0.f;
        half8 pb, pl;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float p = ok[r] ? exp2f(x[r] - mnew) : 0.f;
            psum += p;
            pb[r] = (h16)p;
            pl[r] = (h16)(p - (float)pb[r]);
        }
        if (__builtin_amdgcn_ballot_w64(mnew != st.m[qt])) {
            const float alpha = exp2f(st.m[qt] - mnew);
            st.l[qt] *= alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                st.acc[qt][t][0] *= alpha; st.acc[qt][t][1] *= alpha;
                st.acc[qt][t][2] *= alpha; st.acc[qt][t][3] *= alpha;
            }
            st.m[qt] = mnew;
        }
        st.l[qt] += psum;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ps.va8[t], pb, st.acc[qt][t], 0, 0, 0);
#if TF_BLOCK_P_SPLIT > 0
            st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ps.va8[t], pl, st.acc[qt][t], 0, 0, 0);
#endif
        }
    }
}

// Block variant: up to 128 query rows in one pass over the keys — the chunked prefill (graph_infer.py:30-37,
// TP_llama.py:246-250: q_len = 128) and the Sequoia tree passes (tensor_op.py:171,265: SDPA with a dense
// additive mask over [prefix | tree]).  The 128 rows are cut into `rg` row groups of 32 (two MFMA q-tiles per
// wave); wave w owns row group w % rg and walks the key tiles ti, ti + TI, ... of the split (ti = w / rg,
// TI = 4 / rg).  With rg = 4 every wave walks every tile, so a 128-token chunk streams the KV cache from HBM once
// (the waves' identical tile loads meet in L1/L2) instead of once per 32-row slab; with rg = 1 the 4 waves
// interleave tiles like the decode kernel.  Each wave writes its own partial (m, l, O) — the workspace holds
// nsplit * TI partials of QR = 32 * rg rows per head — and the same combine kernel folds them.
//
// TREE = false: bottom-right causal mask.  TREE = true: keys [0, tree_start) are visible to every row, key
// tree_start + j is visible to row i iff bit j of mask row (mask_row0 + i) is set (the reference's additive mask
// is 0 / fp16-min over the tree columns: SpecTree_TP.py:65-67,83-87,170; a 0/1 bit loses nothing).
#ifndef TF_BLOCK_NO_LDS
#define TF_BLOCK_NO_LDS 0       // 1: 65..128-row blocks use the register-only kernel instead of the LDS-shared one
#endif
#ifndef TF_BLOCK_OCC
#define TF_BLOCK_OCC 1          // waves per SIMD the block kernel is compiled for (A/B: 1 lets it use 512 registers)
#endif
template <int D, bool TREE>
__global__ __launch_bounds__(256, TF_BLOCK_OCC) void attn_block_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, int64_t stride_t,
    int64_t stride_h, int sq, int sk, int H, float scale, int nsplit, int rg, float* __restrict__ ws,
    const uint32_t* __restrict__ mask, int mask_words, int mask_row0, int tree_start) {
    constexpr int NC = D / 32, NT = D / 16, QT = 2;
    const int split = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int TI = 4 / rg, rgi = wave % rg, ti = wave / rg;
    const int QR = 32 * rg;
    const int qbase = rgi * 32;
    const int ntiles = (sk + 15) >> 4;
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int t_begin = split * tps;
    int t_end = min(ntiles, t_begin + tps);
    if (!TREE) {                                   // causal: rows of this wave see keys <= sk - sq + qbase + 31
        const int last_key = sk - sq + min(sq - 1, qbase + 31);
        t_end = min(t_end, (last_key >> 4) + 1);
    }
    const bool wave_active = qbase < sq;

    AttnState<D, QT> st;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = qbase + qt * 16 + li;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            st.qf[qt][c] = (row < sq) ? load_half8(q + ((int64_t)row * H + h) * D + 32 * c + 8 * g) : z;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) st.acc[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.m[qt] = NEG_BIG;
        st.l[qt] = 0.f;
    }
    half8 sel0, sel1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sel0[e] = (8 * g + e == li) ? (h16)1.0f : (h16)0.0f;
        sel1[e] = (8 * g + e == 16 + li) ? (h16)1.0f : (h16)0.0f;
    }
    const h16* kbase = k + (int64_t)h * stride_h;
    const h16* vbase = v + (int64_t)h * stride_h;
    TreeMask tm;
    tm.rows = mask;
    tm.words = mask_words;
    tm.row0 = mask_row0;
    tm.start = tree_start;
    if (wave_active) {
        // pairs of tiles (t, t + TI): both tiles are requested up front; as soon as their QK^T / transpose MFMAs have
        // consumed the K/V registers the next pair is requested, so its latency hides under the softmax + PV phase
        half8 ka[NC], va_[NC], kb[NC], vb[NC];
        const float scale_log2 = scale * 1.4426950408889634f;
        PairScores<D, QT> ps;
        int t = t_begin + ti;
        if (t < t_end) load_kv_tile<D>(kbase, vbase, stride_t, t, sk, li, g, ka, va_);
        if (t + TI < t_end) load_kv_tile<D>(kbase, vbase, stride_t, t + TI, sk, li, g, kb, vb);
        while (t < t_end) {
            const int ta = t, tb = t + TI;
            const bool b_valid = tb < t_end;
            if (!b_valid) {                                 // odd tail: B carries no keys (its registers may be stale)
#pragma unroll
                for (int c = 0; c < NC; ++c) { kb[c] = ka[c]; vb[c] = va_[c]; }
            }
            pair_qk<D, QT>(st, ka, va_, kb, vb, sel0, sel1, ps);
            t += 2 * TI;
            if (t < t_end) load_kv_tile<D>(kbase, vbase, stride_t, t, sk, li, g, ka, va_);
            if (t + TI < t_end) load_kv_tile<D>(kbase, vbase, stride_t, t + TI, sk, li, g, kb, vb);
            // both tiles entirely below this wave's first causal limit / inside the unmasked tree prefix: no mask
            const int last_key = max(ta, tb) * 16 + 15;
            const bool clear = b_valid && (TREE ? (last_key < tm.start && last_key < sk) : (last_key <= sk - sq + qbase));
            if (clear) pair_softmax_pv<D, QT, TREE, false>(st, ps, ta, tb, true, sk, sq, scale_log2, li, g, qbase, tm);
            else pair_softmax_pv<D, QT, TREE, true>(st, ps, ta, tb, b_valid, sk, sq, scale_log2, li, g, qbase, tm);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st.m[qt] *= 0.6931471805599453f;     // log2 domain -> natural log for the merge
    }
    const int np = nsplit * TI;                               // partials per head
    float* ws_o = ws;
    float* ws_m = ws + (int64_t)H * np * QR * D;
    float* ws_l = ws_m + (int64_t)H * np * QR;
    const int64_t pbase = ((int64_t)h * np + split * TI + ti) * QR;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lsum = st.l[qt];
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        const int64_t row = pbase + qbase + qt * 16 + li;      // lane holds O[q = li][d = 16t + 4g + r]
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) *reinterpret_cast<f32x4*>(ws_o + row * D + 16 * tt + 4 * g) = st.acc[qt][tt];
        if (g == 0) {
            ws_m[row] = st.m[qt];
            ws_l[row] = lsum;
        }
    }
}

// LDS-shared form of the block kernel for 65..128 query rows (rg = 4: every wave walks every key tile).  In the
// register-only form above the four waves of a workgroup each fetch the SAME K/V tiles from L2 and each transposes V
// on the matrix core.  Here the workgroup loads each 32-key K/V slab ONCE (global -> registers -> LDS, double-buffered,
// the next slab's loads in flight under the current slab's MFMAs):
//   * K is stored row-major (row stride D+8 halfs); a wave reads its QK^T A operands from rows 8(i>>2)+(i&3) (tile A)
//     and +4 (tile B), so that after the two MFMAs lane group g holds the scores of the 8 CONSECUTIVE keys 8g..8g+7;
//   * V is stored TRANSPOSED by the loader threads (sVt[d][key], 2-byte scatter writes), so the PV A operand
//     V^T[d][8g..8g+7] is one 16-byte LDS read — no transposing MFMAs, no V registers;
//   * exp(S) of those 8 keys is the B operand of ONE v_mfma_f32_16x16x32_f16 per 16 output columns.
// Per 32 keys and wave: 16 + 16 MFMAs (the register form needs 48); one barrier per 64-key slab.
// 128 query rows x 124 928 keys x 32 heads: 1143 us (register form) -> 608 us = 431 TF/s of QK^T + PV.
#ifndef BLK_SLAB
#define BLK_SLAB 64           // keys per load / barrier step of the LDS block kernel = two 32-key MFMA sub-steps
                              // (8 swizzle blocks -> 2-way write conflicts; 32: 4-way, 4 % slower)
// V^T image in LDS: element (d, key r) lives in row d at 8-key block ((r >> 3) ^ ((d >> 3) & (BLK_SLAB/8 - 1))).
// Without the XOR the loader's 2-byte scatter writes — 16 lanes with d = 8*lc + e, same key — all fall on ONE bank
// (row stride and the 8-row lane stride are both multiples of 32 dwords): a 16-way conflict on 16 writes per thread
// per slab, which bounded the kernel.  Reads stay one 16-byte access per (d, block).
#endif
#define BLK_VSWZ(d) (((d) >> 3) & (BLK_SLAB / 8 - 1))
// D = 128: no padding, no transposing store.  Rows of K and V are 256 B = one pass over the 64 banks:
//   K  16-byte chunk j of slab row r sits at chunk j ^ fK(r), fK(r) = (r & 3) | ((r >> 3) & 3) << 2.  The MFMA A-fragment
//      read (lane (li, g): row 8 (li >> 2) + (li & 3) [+4], chunk 4c + g) then lands at chunk (4c + g) ^ li: 16 distinct
//      chunks in each 16-lane service group of ds_read_b128 (the padded layout was 2-way conflicted there);
//   V  row-major like K, 32-byte unit t (= 16 output columns) of row r at unit t ^ fV(r), fV(r) = (r & 3) | ((r >> 3) & 1) << 2;
//      the PV A operand V^T[d][8 keys] comes from two ds_read_b64_tr_b16 (hardware 4 x 16 transpose across a 16-lane
//      group: lane i passes row i >> 2, columns 4 (i & 3)..+3 and receives column i of the 4 rows) — the 8 rows a
//      half-wave touches hit 8 distinct units.  Replaces 8 two-byte scatter stores per 16 bytes of V.
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the padded + transposed layout on a 1024-row prefill chunk: 0.43.
typedef __fp16 tr_fp16x4 __attribute__((vector_size(8)));
__device__ __forceinline__ half4 lds_read_tr4(const h16* p) {
    const tr_fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tr_fp16x4*)(p));
    return __builtin_bit_cast(half4, r);
}
template <int D>
struct BlkLayout {
    static constexpr bool TR = D == 128;
    static constexpr int RS = TR ? D : D + 8;                         // K row stride (halfs)
    static constexpr int VS = BLK_SLAB + 8;                           // V^T row stride (transposed layout only)
    static constexpr int K_HALFS = BLK_SLAB * RS;
    static constexpr int V_HALFS = TR ? BLK_SLAB * D : D * VS;
};
template <int D, int QT, bool TREE, bool MASKED>
__device__ __forceinline__ void lds_softmax_pv(AttnState<D, QT>& st, const f32x4 (&sa)[QT], const f32x4 (&sb)[QT],
                                               const h16* __restrict__ svt, int g0, int key0, int sk, int sq,
                                               float scale_log2, int li, int g, int qbase, TreeMask tm) {
    constexpr int NT = D / 16, VS = BLK_SLAB + 8;
    half8 pb[QT], pl[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = qbase + qt * 16 + li;
        const int kmax = (qrow < sq) ? (sk - sq + qrow) : (sk - 1);   // bottom-right causal
        float x[8];
        bool ok[8];
        float tmax = NEG_BIG;
#if TF_TREE_MASK_FUNNEL
        uint32_t vis8 = 0xFFu;                 // the lane's 8 keys are consecutive: 8 consecutive bits of the mask row
        if (MASKED && TREE) {
            const int kidx0 = key0 + 8 * g;
            vis8 = tf_tree_vis8(tm.rows + (int64_t)(tm.row0 + min(qrow, sq - 1)) * tm.words, tm.words,
                                kidx0 - tm.start, sk - kidx0);
        }
#endif
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int kidx = key0 + 8 * g + r;
            bool v = true;
            if (MASKED) {
                if (TREE) {
#if TF_TREE_MASK_FUNNEL
                    v = (vis8 >> r) & 1u;
#else
                    const int j = kidx - tm.start;
                    v = kidx < sk;
                    if (j >= 0 && v) {
                        const int mrow = tm.row0 + min(qrow, sq - 1);
                        v = (tm.rows[(int64_t)mrow * tm.words + (j >> 5)] >> (j & 31)) & 1u;
                    }
#endif
                } else {
                    v = kidx <= kmax;
                }
            }
            ok[r] = v;
            x[r] = (r < 4 ? sa[qt][r] : sb[qt][r - 4]);                 // raw score: the scale rides in the fma below
            if (MASKED) tmax = v ? fmaxf(tmax, x[r]) : tmax;
        }
        if (!MASKED) {
            mfma_settle8(x);
            tmax = vmax3_raw(vmax3_raw(x[0], x[1], x[2]), vmax3_raw(x[3], x[4], x[5]), vmax_raw(x[6], x[7]));
        }
        tmax = group_max4(tmax);
        const float mnew = fmaxf(st.m[qt], tmax * scale_log2);          // scale > 0: max commutes with it
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float p = ok[r] ? __builtin_amdgcn_exp2f(fmaf(x[r], scale_log2, -mnew)) : 0.f;
            psum += p;
            pb[qt][r] = (h16)p;
            pl[qt][r] = (h16)(p - (float)pb[qt][r]);
        }
        if (__builtin_amdgcn_ballot_w64(mnew != st.m[qt])) {
            const float alpha = __builtin_amdgcn_exp2f(st.m[qt] - mnew);
            st.l[qt] *= alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                st.acc[qt][t][0] *= alpha; st.acc[qt][t][1] *= alpha;
                st.acc[qt][t][2] *= alpha; st.acc[qt][t][3] *= alpha;
            }
            st.m[qt] = mnew;
        }
        st.l[qt] += psum;
    }
    // PV: every V^T fragment is read from LDS once and feeds the MFMAs of all q-tiles of the wave
    if constexpr (BlkLayout<D>::TR) {
        const int fv = (li >> 2) | ((g & 1) << 2);
        const h16* vrow = svt + (8 * (g0 + g) + (li >> 2)) * D + 4 * (li & 3);       // keys 8(g0+g) + {0..3}; +4 rows below
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int pos = (t ^ fv) * 16;
            const half4 lo = lds_read_tr4(vrow + pos), hi = lds_read_tr4(vrow + 4 * D + pos);
            const half8 vt = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pb[qt], st.acc[qt][t], 0, 0, 0);
#if TF_BLOCK_P_SPLIT > 0
                st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pl[qt], st.acc[qt][t], 0, 0, 0);
#endif
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int d = 16 * t + li;                                              // V^T[d][keys 8(g0+g) .. +7]
            const half8 vt = load_half8(svt + d * VS + 8 * ((g0 + g) ^ BLK_VSWZ(d)));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pb[qt], st.acc[qt][t], 0, 0, 0);
#if TF_BLOCK_P_SPLIT > 0
                st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pl[qt], st.acc[qt][t], 0, 0, 0);
#endif
            }
        }
    }
}

// ---- the fully visible slab (TF_BLOCK_PIPE) --------------------------------------------------------------------------
// A 64-key slab all of whose keys every row of the wave may see — every slab of a long prefix but the last one or two — is
// walked as ONE softmax step instead of two:  QK(0) QK(1) | max, exps | PV(0) PV(1).
//   1  both QK^T runs first, then the two sub-steps' softmax / PV as before (bit-identical to the alternating form):
//      1024-row chunk over 124 928 keys 722 -> 754 TF/s;
//   2  (default) ONE running-max update — and at most one accumulator rescale — per slab: both sub-steps' probabilities are
//      taken against max(m, slab max).  The same softmax (m only ever has to dominate the scores seen so far), half the
//      max / ballot / rescale bookkeeping, two long MFMA runs instead of four short ones: 722 -> 799 TF/s
//      (profiles/r03_prefill_slab_ab.jsonl; error against attention accumulated in fp64 unchanged: max 1.64e-5, mean
//      2.03e-6 on outputs of magnitude ~0.01).  Its max chain is plain fmaxf — measured equal to the asm v_max3 chain, and
//      the compiler's hazard recogniser then sees every reader of the MFMA results (see mfma_settle8).
//   0  the alternating form (also what masked slabs, the TREE form and D = 64 use).
#ifndef TF_BLOCK_PIPE
#define TF_BLOCK_PIPE 2
#endif
template <int D, int QT>
__device__ __forceinline__ void lds_softmax_clear(AttnState<D, QT>& st, const f32x4 (&sa)[QT], const f32x4 (&sb)[QT],
                                                  float scale_log2, half8 (&pb)[QT], half8 (&pl)[QT]) {
    constexpr int NT = D / 16;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (r < 4 ? sa[qt][r] : sb[qt][r - 4]);
        mfma_settle8(x);
        float tmax = vmax3_raw(vmax3_raw(x[0], x[1], x[2]), vmax3_raw(x[3], x[4], x[5]), vmax_raw(x[6], x[7]));
        tmax = group_max4(tmax);
        const float mnew = fmaxf(st.m[qt], tmax * scale_log2);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(x[r], scale_log2, -mnew));
            psum += p;
            pb[qt][r] = (h16)p;
            pl[qt][r] = (h16)(p - (float)pb[qt][r]);
        }
        if (__builtin_amdgcn_ballot_w64(mnew != st.m[qt])) {
            const float alpha = __builtin_amdgcn_exp2f(st.m[qt] - mnew);
            st.l[qt] *= alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                st.acc[qt][t][0] *= alpha; st.acc[qt][t][1] *= alpha;
                st.acc[qt][t][2] *= alpha; st.acc[qt][t][3] *= alpha;
            }
            st.m[qt] = mnew;
        }
        st.l[qt] += psum;
    }
}
template <int D, int QT>
__device__ __forceinline__ void lds_softmax_clear2(AttnState<D, QT>& st, const f32x4 (&s0a)[QT], const f32x4 (&s0b)[QT],
                                                   const f32x4 (&s1a)[QT], const f32x4 (&s1b)[QT], float scale_log2,
                                                   half8 (&p0)[QT], half8 (&p1)[QT], half8 (&l0)[QT], half8 (&l1)[QT]) {
    constexpr int NT = D / 16;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float x[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x[r] = s0a[qt][r]; x[4 + r] = s0b[qt][r]; x[8 + r] = s1a[qt][r]; x[12 + r] = s1b[qt][r];
        }
        float tmax = x[0];                               // plain fmaxf: every reader of the MFMA results is compiler-visible
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
        tmax = group_max4(tmax);
        const float mnew = fmaxf(st.m[qt], tmax * scale_log2);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(x[r], scale_log2, -mnew));
            psum += p;
            const h16 ph = (h16)p, pw = (h16)(p - (float)ph);
            if (r < 8) { p0[qt][r] = ph; l0[qt][r] = pw; }
            else { p1[qt][r - 8] = ph; l1[qt][r - 8] = pw; }
        }
        if (__builtin_amdgcn_ballot_w64(mnew != st.m[qt])) {
            const float alpha = __builtin_amdgcn_exp2f(st.m[qt] - mnew);
            st.l[qt] *= alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                st.acc[qt][t][0] *= alpha; st.acc[qt][t][1] *= alpha;
                st.acc[qt][t][2] *= alpha; st.acc[qt][t][3] *= alpha;
            }
            st.m[qt] = mnew;
        }
        st.l[qt] += psum;
    }
}
template <int D, int QT>
__device__ __forceinline__ void lds_pv_tr(AttnState<D, QT>& st, const half8 (&pb)[QT], const half8 (&pl)[QT],
                                          const h16* __restrict__ svt, int g0, int li, int g) {
    constexpr int NT = D / 16;
    const int fv = (li >> 2) | ((g & 1) << 2);
    const h16* vrow = svt + (8 * (g0 + g) + (li >> 2)) * D + 4 * (li & 3);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int pos = (t ^ fv) * 16;
        const half4 lo = lds_read_tr4(vrow + pos), hi = lds_read_tr4(vrow + 4 * D + pos);
        const half8 vt = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pb[qt], st.acc[qt][t], 0, 0, 0);
#if TF_BLOCK_P_SPLIT > 0
            st.acc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pl[qt], st.acc[qt][t], 0, 0, 0);
#endif
        }
    }
}

// K / V slab loads of the LDS block kernel by LDS-DMA instead of through 32 staging registers per lane (swizzled layout,
// D = 128): 1 = TREE form only, 2 = chain form too (default), 0 = register-staged.  The TREE form spilled 24 registers at
// its 2 waves per SIMD; without the staging registers it spills 2: 512-node Sequoia verify over a 124 928-token prefix
// 2 298 -> 1 905 us, 128-row chain block 465 -> 422 us, outputs bit-identical (profiles/r03_block_dma_ab.jsonl).
#ifndef TF_BLOCK_DMA
#define TF_BLOCK_DMA 2
#endif
#ifndef TF_BLOCK_TREE_OCC
#define TF_BLOCK_TREE_OCC 2      // waves per SIMD of the TREE form of the LDS block kernel.  At 2 it spills 46 registers; at 1
                                 // (512 registers, no spill) the 512-node Sequoia verify is 25 % SLOWER (2 890 -> 3 630 us): kept at 2
#endif
// One workgroup: the 128-row block ``q`` against slabs [s_begin, s_end) of head h; partial (m, l, O) to slot ``split``.
template <int D, bool TREE>
__device__ __forceinline__ void attn_block_lds_body(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, int64_t stride_t,
    int64_t stride_h, int sq, int sk, int H, float scale, int nsplit, float* __restrict__ ws,
    const uint32_t* __restrict__ mask, int mask_words, int mask_row0, int tree_start, int split, int h, int s_begin,
    int s_end) {
    constexpr int NC = D / 32, NT = D / 16, QT = 2, QR = 128;
    constexpr int SLAB = BLK_SLAB;                          // keys per load step: 64 -> 32 KiB (D=128) in flight per WG
    constexpr bool TR = BlkLayout<D>::TR;                   // swizzled K, row-major V + transpose reads (see above)
    constexpr int RS = BlkLayout<D>::RS;                    // K row stride in LDS (halfs)
    constexpr int VS = BlkLayout<D>::VS;                    // V^T row stride in LDS (halfs)
    constexpr int VPR = D / 8;                              // 16-byte vectors per row
    constexpr int RPP = 256 / VPR;                          // rows covered by one pass of the 256 threads
    constexpr int NPASS = SLAB / RPP;                       // passes per slab (4 for D=128, 2 for D=64)
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_smem[];
    h16* sK = reinterpret_cast<h16*>(blk_smem);             // [2][SLAB * RS]
    h16* sVt = sK + 2 * BlkLayout<D>::K_HALFS;              // [2][D * VS]  or  [2][SLAB * D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int qbase = wave * 32;

    AttnState<D, QT> st;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = qbase + qt * 16 + li;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            st.qf[qt][c] = (row < sq) ? load_half8(q + ((int64_t)row * H + h) * D + 32 * c + 8 * g) : z;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) st.acc[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.m[qt] = NEG_BIG;
        st.l[qt] = 0.f;
    }
    const h16* kbase = k + (int64_t)h * stride_h;
    const h16* vbase = v + (int64_t)h * stride_h;
    TreeMask tm;
    tm.rows = mask;
    tm.words = mask_words;
    tm.row0 = mask_row0;
    tm.start = tree_start;
    const float scale_log2 = scale * 1.4426950408889634f;
    const int lr = tid / VPR, lc = tid % VPR;               // this thread's (row, 16-byte column) in a load pass
    const int row_a = 8 * (li >> 2) + (li & 3);             // slab row behind MFMA row li of tile A (tile B: +4)

    // K / V slabs reach LDS either through staging registers (fetch -> stash: 32 registers per lane live across a slab's
    // MFMAs) or — TF_BLOCK_DMA, swizzled layout only — by LDS-DMA straight into the buffer the NEXT iteration reads (free
    // since the previous barrier), the XOR swizzle applied on the source side (see attn_prefill_ahead_body).
    constexpr bool DMA = TF_BLOCK_DMA && TR && (TREE || TF_BLOCK_DMA > 1);
    half8 gk[DMA ? 1 : NPASS], gv[DMA ? 1 : NPASS];
    auto fetch = [&](int slab, int buf) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            int key = slab * SLAB + p * RPP + lr;
            key = key < sk ? key : sk - 1;                  // clamp: masked below, but must stay in-bounds
            if constexpr (DMA) {
                const int r = p * RPP + lr;
                const int fk = (r & 3) | (((r >> 3) & 3) << 2), fv = (r & 3) | (((r >> 3) & 1) << 2);
                h16* dk = sK + buf * BlkLayout<D>::K_HALFS + (p * RPP + 4 * wave) * D;      // wave-uniform 1 KiB of this pass
                h16* dv = sVt + buf * BlkLayout<D>::V_HALFS + (p * RPP + 4 * wave) * D;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(kbase + (int64_t)key * stride_t + 8 * (lc ^ fk)),
                    (__attribute__((address_space(3))) void*)dk, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(vbase + (int64_t)key * stride_t +
                                                                     8 * ((((lc >> 1) ^ fv) << 1) | (lc & 1))),
                    (__attribute__((address_space(3))) void*)dv, 16, 0, 0);
            } else {
                gk[p] = load_half8_stream(kbase + (int64_t)key * stride_t + 8 * lc);
                gv[p] = load_half8_stream(vbase + (int64_t)key * stride_t + 8 * lc);
            }
        }
    };
    auto stash = [&](int buf) {
        if constexpr (DMA) return;
        h16* dk = sK + buf * BlkLayout<D>::K_HALFS;
        h16* dv = sVt + buf * BlkLayout<D>::V_HALFS;
#pragma unroll
        for (int p = 0; p < (DMA ? 0 : NPASS); ++p) {
            const int r = p * RPP + lr;
            if constexpr (TR) {
                const int fk = (r & 3) | (((r >> 3) & 3) << 2), fv = (r & 3) | (((r >> 3) & 1) << 2);
                store_half8(dk + r * RS + 8 * (lc ^ fk), gk[p]);
                store_half8(dv + r * D + 8 * ((((lc >> 1) ^ fv) << 1) | (lc & 1)), gv[p]);
            } else {
                store_half8(dk + r * RS + 8 * lc, gk[p]);
#pragma unroll
                for (int e = 0; e < 8; ++e)                // d = 8*lc + e -> swizzle key = lc
                    dv[(8 * lc + e) * VS + ((((r >> 3) ^ (lc & (SLAB / 8 - 1))) << 3) | (r & 7))] = gv[p][e];
            }
        }
    };

    if (s_begin < s_end) {
        fetch(s_begin, 0);
        stash(0);
    }
    // Every load issued so far — the Q fragments above all — has landed before the loop: without this the compiler,
    // which cannot tell the Q loads from the slab prefetch through the in-order vmcnt, waits vmcnt(0) right after issuing
    // each prefetch (the path that skips the stash above leaves Q pending at the loop header), i.e. serialises the HBM
    // latency of every slab with its MFMAs.
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();
    for (int sl = s_begin; sl < s_end; ++sl) {
        const int buf = (sl - s_begin) & 1;
        fetch(min(sl + 1, s_end - 1), buf ^ 1);             // in flight under this slab's MFMAs (unconditional: see attn_split_body)
        const h16* bk = sK + buf * BlkLayout<D>::K_HALFS;
        const h16* bv = sVt + buf * BlkLayout<D>::V_HALFS;
        bool piped = false;
        if constexpr (TF_BLOCK_PIPE && TR && !TREE && SLAB == 64) {
            const int last = sl * SLAB + SLAB - 1;
            if (last < sk && last <= sk - sq + qbase) {              // wave-uniform: both sub-steps fully visible
                piped = true;
                f32x4 s0a[QT], s0b[QT], s1a[QT], s1b[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s0a[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; s0b[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    s1a[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; s1b[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int kcol = 8 * ((4 * c + g) ^ li);
                    const half8 ka = load_half8(bk + row_a * RS + kcol);
                    const half8 kb = load_half8(bk + (row_a + 4) * RS + kcol);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        s0a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, st.qf[qt][c], s0a[qt], 0, 0, 0);
                        s0b[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, st.qf[qt][c], s0b[qt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int kcol = 8 * ((4 * c + g) ^ li);
                    const half8 ka = load_half8(bk + (32 + row_a) * RS + kcol);
                    const half8 kb = load_half8(bk + (32 + row_a + 4) * RS + kcol);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        s1a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, st.qf[qt][c], s1a[qt], 0, 0, 0);
                        s1b[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, st.qf[qt][c], s1b[qt], 0, 0, 0);
                    }
                }
                half8 p0[QT], p1[QT], l0[QT], l1[QT];
                if constexpr (TF_BLOCK_PIPE >= 2) {
                    lds_softmax_clear2<D, QT>(st, s0a, s0b, s1a, s1b, scale_log2, p0, p1, l0, l1);
                    lds_pv_tr<D, QT>(st, p0, l0, bv, 0, li, g);
                    lds_pv_tr<D, QT>(st, p1, l1, bv, 4, li, g);
                } else {
                    lds_softmax_clear<D, QT>(st, s0a, s0b, scale_log2, p0, l0);
                    lds_pv_tr<D, QT>(st, p0, l0, bv, 0, li, g);
                    lds_softmax_clear<D, QT>(st, s1a, s1b, scale_log2, p1, l1);
                    lds_pv_tr<D, QT>(st, p1, l1, bv, 4, li, g);
                }
            }
        }
#pragma unroll
        for (int sub = 0; sub < (piped ? 0 : SLAB / 32); ++sub) {
            const int key0 = sl * SLAB + 32 * sub;
            if (key0 >= sk) break;
            f32x4 sa[QT], sb[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                sa[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                sb[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int kcol = TR ? 8 * ((4 * c + g) ^ li) : 32 * c + 8 * g;
                const half8 ka = load_half8(bk + (32 * sub + row_a) * RS + kcol);
                const half8 kb = load_half8(bk + (32 * sub + row_a + 4) * RS + kcol);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    sa[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, st.qf[qt][c], sa[qt], 0, 0, 0);
                    sb[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, st.qf[qt][c], sb[qt], 0, 0, 0);
                }
            }
            const int last_key = key0 + 31;
            const bool clear = TREE ? (last_key < tm.start && last_key < sk) : (last_key <= sk - sq + qbase);
            if (clear)
                lds_softmax_pv<D, QT, TREE, false>(st, sa, sb, bv, 4 * sub, key0, sk, sq, scale_log2, li, g, qbase, tm);
            else
                lds_softmax_pv<D, QT, TREE, true>(st, sa, sb, bv, 4 * sub, key0, sk, sq, scale_log2, li, g, qbase, tm);
        }
        if (sl + 1 < s_end) stash(buf ^ 1);
        if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA writes are in LDS before the barrier
        __syncthreads();
    }

    float* ws_o = ws;
    float* ws_m = ws + (int64_t)H * nsplit * QR * D;
    float* ws_l = ws_m + (int64_t)H * nsplit * QR;
    const int64_t pbase = ((int64_t)h * nsplit + split) * QR;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lsum = st.l[qt];
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        const int64_t row = pbase + qbase + qt * 16 + li;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) *reinterpret_cast<f32x4*>(ws_o + row * D + 16 * tt + 4 * g) = st.acc[qt][tt];
        if (g == 0) {
            ws_m[row] = st.m[qt] * 0.6931471805599453f;     // log2 domain -> natural log for the merge
            ws_l[row] = lsum;
        }
    }
}

template <int D, bool TREE>
__global__ __launch_bounds__(256, TREE ? TF_BLOCK_TREE_OCC : 2) void attn_block_lds_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, int64_t stride_t,
    int64_t stride_h, int sq, int sk, int H, float scale, int nsplit, float* __restrict__ ws,
    const uint32_t* __restrict__ mask, int mask_words, int mask_row0, int tree_start) {
    const int split = blockIdx.x, h = blockIdx.y;
    const int nslabs = (sk + BLK_SLAB - 1) / BLK_SLAB;
    const int sps = (nslabs + nsplit - 1) / nsplit;
    const int s_begin = split * sps;
    attn_block_lds_body<D, TREE>(q, k, v, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, mask, mask_words, mask_row0,
                                 tree_start, split, h, s_begin, min(nslabs, s_begin + sps));
}

// ---- prefill body with the NEXT slab's QK^T issued ahead and K / V brought in by LDS-DMA (TF_PREFILL_AHEAD) -----------------
// The dependency chain of a slab is QK^T -> softmax -> PV: matrix core, VALU, matrix core.  K[s] is dead as soon as
// QK^T(s) has run, so with the SAME two K and two V buffers the loop can hold K one slab ahead of V: iteration s runs
// QK^T(s + 1) (from the K buffer this iteration's PV does not touch), then the softmax of slab s on scores computed one
// iteration earlier, then PV(s).  One barrier per slab as before: iteration s brings K[s + 2] in over K[s] (dead since the
// previous barrier) and V[s + 1] over V[s - 1].  The second set of score registers does not fit next to the 32 staging
// registers of the register-staged loads (256 per lane at two waves per SIMD: that build spills 34 registers and runs at
// 424 TF/s), so K and V come in by LDS-DMA — `global_load_lds_dwordx4`: one instruction of a wave moves 64 x 16 B straight
// into 1 KiB of LDS, lane-linear, no staging registers and no ds_write; the XOR swizzle of the layout is applied on the
// SOURCE side (the lane that fills slot j of row r loads chunk j ^ f(r): the swizzles are involutions).
// 1024-row chunk x 124 928 keys x 32 heads, same box (profiles/r03_prefill_slab_ab.jsonl): 792.7 -> 848.4 TF/s (+7 %; +10.5 %
// on a slower box).  Splitting the softmax so that the compiler interleaves the next slab's QK^T MFMAs with this slab's exps
// INSIDE one wave measured slower (820.5): what overlaps the matrix core with the VALU here is the other wave of the SIMD.
// D = 128, chain (non-tree) form only; 0 = the shared body (attn_block_lds_body).
#ifndef TF_PREFILL_AHEAD
#define TF_PREFILL_AHEAD 1
#endif
template <int D>
__device__ __forceinline__ void attn_prefill_ahead_body(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, int64_t stride_t,
    int64_t stride_h, int sq, int sk, int H, float scale, int nsplit, float* __restrict__ ws, int split, int h,
    int s_begin, int s_end) {
    static_assert(BlkLayout<D>::TR && BLK_SLAB == 64, "swizzled-K / transpose-read layout, 64-key slabs");
    constexpr int NC = D / 32, NT = D / 16, QT = 2, QR = 128, SLAB = BLK_SLAB, RS = BlkLayout<D>::RS;
    constexpr int VPR = D / 8, RPP = 256 / VPR, NPASS = SLAB / RPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_smem[];
    h16* sK = reinterpret_cast<h16*>(blk_smem);             // [2][SLAB * RS]
    h16* sVt = sK + 2 * BlkLayout<D>::K_HALFS;              // [2][SLAB * D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int qbase = wave * 32;

    AttnState<D, QT> st;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = qbase + qt * 16 + li;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            st.qf[qt][c] = (row < sq) ? load_half8(q + ((int64_t)row * H + h) * D + 32 * c + 8 * g) : z;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) st.acc[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.m[qt] = NEG_BIG;
        st.l[qt] = 0.f;
    }
    const h16* kbase = k + (int64_t)h * stride_h;
    const h16* vbase = v + (int64_t)h * stride_h;
    const TreeMask tm{nullptr, 0, 0, 0};
    const float scale_log2 = scale * 1.4426950408889634f;
    const int lr = tid / VPR, lc = tid % VPR;
    const int row_a = 8 * (li >> 2) + (li & 3);

    // one slab (64 rows x 256 B) = 4 passes; pass p, wave w fills rows 16 p + 4 w .. + 3 (1 KiB, lane-linear)
    auto dma_slab = [&](const h16* base, h16* dst_buf, int slab, bool is_v) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = p * RPP + lr;                                   // lr = tid / 16: row within the pass
            int key = slab * SLAB + r;
            key = key < sk ? key : sk - 1;
            int c;
            if (!is_v) {
                const int fk = (r & 3) | (((r >> 3) & 3) << 2);
                c = lc ^ fk;
            } else {
                const int fv = (r & 3) | (((r >> 3) & 1) << 2);
                c = (((lc >> 1) ^ fv) << 1) | (lc & 1);
            }
            const h16* src = base + (int64_t)key * stride_t + 8 * c;
            h16* dst = dst_buf + (p * RPP + 4 * wave) * D;                // wave-uniform 1-KiB destination of this pass
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto fetch_k = [&](int slab, int buf) { dma_slab(kbase, sK + buf * BlkLayout<D>::K_HALFS, slab, false); };
    auto fetch_v = [&](int slab, int buf) { dma_slab(vbase, sVt + buf * BlkLayout<D>::V_HALFS, slab, true); };
    // both 32-key sub-steps of one slab: scores S^T[key][q] of tiles A / B (keys 4g..4g+3 of either half) per q-tile
    auto qk_slab = [&](const h16* bk, f32x4 (&a0)[QT], f32x4 (&b0)[QT], f32x4 (&a1)[QT], f32x4 (&b1)[QT]) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            a0[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; b0[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            a1[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; b1[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int kcol = 8 * ((4 * c + g) ^ li);
                const half8 ka = load_half8(bk + (32 * sub + row_a) * RS + kcol);
                const half8 kb = load_half8(bk + (32 * sub + row_a + 4) * RS + kcol);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    if (sub == 0) {
                        a0[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, st.qf[qt][c], a0[qt], 0, 0, 0);
                        b0[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, st.qf[qt][c], b0[qt], 0, 0, 0);
                    } else {
                        a1[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, st.qf[qt][c], a1[qt], 0, 0, 0);
                        b1[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, st.qf[qt][c], b1[qt], 0, 0, 0);
                    }
                }
            }
    };

    if (s_begin < s_end) {
        f32x4 c0a[QT], c0b[QT], c1a[QT], c1b[QT];            // scores of the CURRENT slab
        fetch_k(s_begin, 0);
        fetch_v(s_begin, 0);
        fetch_k(min(s_begin + 1, s_end - 1), 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): Q fragments and the prologue slabs have landed
        __syncthreads();
        qk_slab(sK, c0a, c0b, c1a, c1b);
        __syncthreads();                                     // every wave is done with K[s_begin] before iteration 0 overwrites it
        for (int sl = s_begin; sl < s_end; ++sl) {
            const int i = sl - s_begin;
            fetch_k(min(sl + 2, s_end - 1), i & 1);          // K[sl + 2] over K[sl] (dead since the previous barrier), V[sl + 1]
            fetch_v(min(sl + 1, s_end - 1), (i + 1) & 1);    // over V[sl - 1]; unconditional: in flight under the MFMAs
            const h16* bk_next = sK + ((i + 1) & 1) * BlkLayout<D>::K_HALFS;
            const h16* bv = sVt + (i & 1) * BlkLayout<D>::V_HALFS;
            f32x4 n0a[QT], n0b[QT], n1a[QT], n1b[QT];
            const int last = sl * SLAB + SLAB - 1;
            if (last < sk && last <= sk - sq + qbase) {      // wave-uniform: the whole slab is visible to every row of the wave
                half8 p0[QT], p1[QT], l0[QT], l1[QT];
                qk_slab(bk_next, n0a, n0b, n1a, n1b);        // next slab's QK^T (past the end: finite garbage, never used)
                lds_softmax_clear2<D, QT>(st, c0a, c0b, c1a, c1b, scale_log2, p0, p1, l0, l1);
                lds_pv_tr<D, QT>(st, p0, l0, bv, 0, li, g);
                lds_pv_tr<D, QT>(st, p1, l1, bv, 4, li, g);
            } else {
                qk_slab(bk_next, n0a, n0b, n1a, n1b);
                const int key0 = sl * SLAB;
                lds_softmax_pv<D, QT, false, true>(st, c0a, c0b, bv, 0, key0, sk, sq, scale_log2, li, g, qbase, tm);
                if (key0 + 32 < sk)
                    lds_softmax_pv<D, QT, false, true>(st, c1a, c1b, bv, 4, key0 + 32, sk, sq, scale_log2, li, g, qbase, tm);
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                c0a[qt] = n0a[qt]; c0b[qt] = n0b[qt]; c1a[qt] = n1a[qt]; c1b[qt] = n1b[qt];
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): this wave's DMA writes are in LDS before the barrier
            __syncthreads();
        }
    }

    float* ws_o = ws;
    float* ws_m = ws + (int64_t)H * nsplit * QR * D;
    float* ws_l = ws_m + (int64_t)H * nsplit * QR;
    const int64_t pbase = ((int64_t)h * nsplit + split) * QR;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lsum = st.l[qt];
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        const int64_t row = pbase + qbase + qt * 16 + li;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) *reinterpret_cast<f32x4*>(ws_o + row * D + 16 * tt + 4 * g) = st.acc[qt][tt];
        if (g == 0) {
            ws_m[row] = st.m[qt] * 0.6931471805599453f;
            ws_l[row] = lsum;
        }
    }
}

// A whole causal prefill chunk (sq rows = nrb blocks of 128) in ONE launch.  Block rb sees keys [0, sk - sq + end_rb).
// Launching the blocks one by one streams the KV cache from HBM once per 128 rows (128 flop per byte: an 800 TF/s
// ceiling at 6.3 TB/s, and the 125K-token prefill ran at ~490).  Here the row blocks that read the SAME key range run at
// the same time on the SAME XCD, so all but one of them hit that XCD's L2: workgroup ids are laid out as
//   id % 8 = XCD (hardware round-robin),  (id / 8) % RBG = row block within a group of RBG <= 8,  id / (8 RBG) = the rest,
// and a (split, head) pair is pinned to XCD pair % 8.  Every row block uses the key partition of the full sk, so its
// splits cover the same slabs as its neighbours'; splits past a block's causal edge write a neutral partial.
template <int D>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, int64_t stride_t,
    int64_t stride_h, int sq, int sk, int H, float scale, int nsplit, float* __restrict__ ws, int nrb, int rbg) {
    const int id = blockIdx.x;
    const int pairs = H * nsplit;                                     // host guarantees pairs % 8 == 0
    const int xcd = id & 7, j = id >> 3;
    const int rb_lo = j % rbg, rest = j / rbg;
    const int pair = (rest % (pairs >> 3)) * 8 + xcd;
    const int rb = rb_lo + rbg * (rest / (pairs >> 3));
    if (rb >= nrb) return;
    const int h = pair / nsplit, split = pair - h * nsplit;
    const int r0 = rb * 128;
    const int rows = min(128, sq - r0);
    const int sk_eff = sk - (sq - (r0 + rows));                        // keys visible to the last row of this block
    const int nslabs_all = (sk + BLK_SLAB - 1) / BLK_SLAB;
    const int nslabs = (sk_eff + BLK_SLAB - 1) / BLK_SLAB;
    const int sps = (nslabs_all + nsplit - 1) / nsplit;
    const int s_begin = split * sps;
    float* ws_rb = ws + (int64_t)rb * H * nsplit * 128 * (D + 2);
    if constexpr (TF_PREFILL_AHEAD && BlkLayout<D>::TR)
        attn_prefill_ahead_body<D>(q + (int64_t)r0 * H * D, k, v, stride_t, stride_h, rows, sk_eff, H, scale, nsplit, ws_rb,
                                   split, h, s_begin, min(nslabs, s_begin + sps));
    else
        attn_block_lds_body<D, false>(q + (int64_t)r0 * H * D, k, v, stride_t, stride_h, rows, sk_eff, H, scale, nsplit, ws_rb,
                                      nullptr, 0, 0, 0, split, h, s_begin, min(nslabs, s_begin + sps));
}

template <int D>
static size_t blk_lds_bytes() {
    return (size_t)2 * (BlkLayout<D>::K_HALFS + BlkLayout<D>::V_HALFS) * sizeof(h16);
}

// Merge of the per-split partials.  grid (H, sq), block (D, CG): thread (d, g) folds the splits s == g (mod CG)
// with independent loads (the first version walked all splits serially per thread: 29 us at nsplit=32 — a
// dependent-latency chain, 8% on top of the 331 us split kernel); the CG partial sums meet in LDS.
template <int D>
__global__ __launch_bounds__(D * COMBINE_GROUPS) void attn_combine_kernel(const float* __restrict__ ws,
                                                                          h16* __restrict__ out, int sq, int H,
                                                                          int nsplit, int QR, int64_t osm, int64_t osk) {
    __shared__ float sm_w[COMBINE_MAX_SPLITS];
    __shared__ float sm_l[COMBINE_MAX_SPLITS];
    __shared__ float sm_o[COMBINE_GROUPS][D];
    const int h = blockIdx.x, qq = blockIdx.y, d = threadIdx.x, g = threadIdx.y;
    const int tid = g * D + d;
    const float* ws_o = ws;
    const float* ws_m = ws + (int64_t)H * nsplit * QR * D;
    const float* ws_l = ws_m + (int64_t)H * nsplit * QR;
    const int64_t base = (int64_t)h * nsplit * QR + qq;
    // this thread's partial outputs (splits g, g + 8, ...) are requested TOGETHER with the split maxima — they do not
    // depend on them; loaded after the weights were known they were a second dependent round trip (the partials come from
    // other XCDs' workgroups: Infinity-Cache latency) in a 6 us kernel.  Unconditional loads (clamped split index): a
    // predicated load makes the compiler branch and wait per load.
    constexpr int CPT = COMBINE_MAX_SPLITS / COMBINE_GROUPS;
    float ov[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int sc = min(g + i * COMBINE_GROUPS, nsplit - 1);
        ov[i] = ws_o[(base + (int64_t)sc * QR) * D + d];
    }
    if (tid < nsplit) {
        sm_w[tid] = ws_m[base + (int64_t)tid * QR];
        sm_l[tid] = ws_l[base + (int64_t)tid * QR];
    }
    __syncthreads();
    float mm = NEG_BIG;
    for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, sm_w[s]);
    __syncthreads();
    if (tid < nsplit) {
        const float w = __expf(sm_w[tid] - mm);
        sm_w[tid] = w;
        sm_l[tid] *= w;
    }
    __syncthreads();
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {                                  // same splits in the same order as `for s = g; s += 8`
        const int s = g + i * COMBINE_GROUPS;
        if (s < nsplit) o = fmaf(ov[i], sm_w[s], o);
    }
    sm_o[g][d] = o;
    __syncthreads();
    if (g == 0) {
        float l = 0.f;
        for (int s = 0; s < nsplit; ++s) l += sm_l[s];
        float acc = 0.f;
#pragma unroll
        for (int gg = 0; gg < COMBINE_GROUPS; ++gg) acc += sm_o[gg][d];
        out[(int64_t)qq * osm + (int64_t)((h * D + d) >> 3) * osk + (d & 7)] = (h16)(acc / l);
    }
}

// Merge for many query rows (block attention, > 32 rows): ONE WAVE per (row, head), no LDS, no barrier.  The grid of
// the kernel above is one 1024-thread workgroup with three barriers per (row, head) — 4096 of them for a 128-row
// prefill slab, 27 us per launch, 5 % of a 125K-token prefill.  Here lane s holds the weight of split s (wave max /
// wave sum), each lane owns D/64 consecutive output columns, and the nsplit partial rows are read with independent
// coalesced loads.
template <int D>
__global__ __launch_bounds__(256) void attn_combine_rows_kernel(const float* __restrict__ ws, h16* __restrict__ out,
                                                                int sq, int H, int nsplit, int QR, int block_rows) {
    constexpr int VPL = D / 64;
    static_assert(COMBINE_MAX_SPLITS <= 128, "two weights per lane");
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);           // pair = row * H + head (output order)
    if (pair >= sq * H) return;
    const int row = pair / H, h = pair - row * H;
    // block_rows > 0 (tf_attn_prefill): rows [rb * block_rows, ...) have their own workspace image
    const int rb = block_rows > 0 ? row / block_rows : 0;
    const int qq = row - rb * block_rows;
    ws += (int64_t)rb * H * nsplit * QR * (D + 2);
    const float* ws_o = ws;
    const float* ws_m = ws + (int64_t)H * nsplit * QR * D;
    const float* ws_l = ws_m + (int64_t)H * nsplit * QR;
    const int64_t base = (int64_t)h * nsplit * QR + qq;
    const bool has0 = lane < nsplit, has1 = lane + 64 < nsplit;
    const float m0 = has0 ? ws_m[base + (int64_t)lane * QR] : NEG_BIG;
    const float m1 = has1 ? ws_m[base + (int64_t)(lane + 64) * QR] : NEG_BIG;
    const float l0 = has0 ? ws_l[base + (int64_t)lane * QR] : 0.f;
    const float l1 = has1 ? ws_l[base + (int64_t)(lane + 64) * QR] : 0.f;
    const float mm = wave_max(fmaxf(m0, m1));
    const float w0 = has0 ? __expf(m0 - mm) : 0.f;
    const float w1 = has1 ? __expf(m1 - mm) : 0.f;
    const float l = wave_sum(w0 * l0 + w1 * l1);
    float acc[VPL];
#pragma unroll
    for (int e = 0; e < VPL; ++e) acc[e] = 0.f;
    const float* po = ws_o + base * D + lane * VPL;
    const int64_t sstride = (int64_t)QR * D;
    for (int half = 0; half < 2; ++half) {                          // splits [0, 64) weigh by w0, [64, 128) by w1
        const float wreg = half ? w1 : w0;
        const int cnt = min(max(nsplit - 64 * half, 0), 64);
        const float* ph = po + (int64_t)(64 * half) * sstride;
        int s = 0;
        for (; s + 4 <= cnt; s += 4) {                              // four independent loads in flight per lane
            float v[4][VPL];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < VPL; ++e) v[j][e] = ph[(int64_t)(s + j) * sstride + e];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = __shfl(wreg, s + j, 64);
#pragma unroll
                for (int e = 0; e < VPL; ++e) acc[e] = fmaf(v[j][e], w, acc[e]);
            }
        }
        for (; s < cnt; ++s) {
            const float w = __shfl(wreg, s, 64);
#pragma unroll
            for (int e = 0; e < VPL; ++e) acc[e] = fmaf(ph[(int64_t)s * sstride + e], w, acc[e]);
        }
    }
    const float inv = 1.0f / l;
    h16* o = out + (int64_t)pair * D + lane * VPL;
#pragma unroll
    for (int e = 0; e < VPL; ++e) o[e] = (h16)(acc[e] * inv);
}

// ------------------------------------------------------------------------------------------
// Draft attention, RoPE applied to cached keys on read (modeling_llama_68m.py:151-190).
// Tiny problem (12 heads, <=259 keys, D=64): one wave per (head, query row); latency-bound.
// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_rope_on_read_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, const h16* __restrict__ cosb,
    const h16* __restrict__ sinb, h16* __restrict__ out, int64_t stride_t, int64_t stride_h, int sq, int kv_len,
    int H, float scale) {
    constexpr int NV = D / 8, HV = NV / 2;          // half8 vectors per row / per half row
    extern __shared__ float smem[];                 // [4 waves][kv_len] scores -> probabilities
    const int h = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qrow = blockIdx.y * 4 + wave;
    const bool active = qrow < sq;
    float* pw = smem + (size_t)wave * kv_len;
    const int kmax = active ? (kv_len - sq + qrow) : -1;    // last visible key (bottom-right causal)
    float lsum = 0.f;
    if (active) {
        const h16* qp = q + ((int64_t)qrow * H + h) * D;
        half8 qv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) qv[i] = load_half8(qp + 8 * i);
        const h16* kb = k + (int64_t)h * stride_h;
        float mloc = NEG_BIG;
        for (int j = lane; j <= kmax; j += 64) {
            const h16* kp = kb + (int64_t)j * stride_t;
            const h16* cp = cosb + (int64_t)j * D;
            const h16* sp = sinb + (int64_t)j * D;
            half8 kv[NV], cv[NV], sv[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                kv[i] = load_half8(kp + 8 * i);
                cv[i] = load_half8(cp + 8 * i);
                sv[i] = load_half8(sp + 8 * i);
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < HV; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const h16 x1 = kv[i][e], x2 = kv[i + HV][e];
                    // (x*cos) + (rotate_half(x)*sin), every op rounded to fp16 like the reference
                    const h16 lo = hadd_rn(hmul_rn(x1, cv[i][e]), hmul_rn((h16)(-(float)x2), sv[i][e]));
                    const h16 hi = hadd_rn(hmul_rn(x2, cv[i + HV][e]), hmul_rn(x1, sv[i + HV][e]));
                    s = fmaf((float)qv[i][e], (float)lo, s);
                    s = fmaf((float)qv[i + HV][e], (float)hi, s);
                }
            s *= scale;
            pw[j] = s;
            mloc = fmaxf(mloc, s);
        }
        const float mx = wave_max(mloc);
        for (int j = lane; j <= kmax; j += 64) {
            const float p = __expf(pw[j] - mx);     // same lane wrote pw[j]
            pw[j] = p;
            lsum += p;
        }
        lsum = wave_sum(lsum);
    }
    __syncthreads();
    if (active) {
        const h16* vb = v + (int64_t)h * stride_h;
        for (int d = lane; d < D; d += 64) {
            float o = 0.f;
            for (int j = 0; j <= kmax; ++j) o = fmaf(pw[j], (float)vb[(int64_t)j * stride_t + d], o);
            out[((int64_t)qrow * H + h) * D + d] = (h16)(o / lsum);
        }
    }
}

// Fast path of the draft attention (kv_len <= DRAFT_LDS_MAX_KEYS): one workgroup per (head, 16-query block).
//   phase 1  all 4 waves rotate the head's cached keys ONCE into LDS (fp16, the reference's rounding points)
//            and stage V transposed (Vt[d][key]) so that both MFMA B operands are 16-byte LDS reads;
//   phase 2  S = Q K^T on the matrix core (v_mfma_f32_16x16x32_f16, wave w takes key tiles w, w+4, ...);
//   phase 3  fp32 softmax, wave w owns query rows 4w..4w+3, P rounded to fp16 like flash-attn;
//   phase 4  O = P V on the matrix core, wave w owns the 16 output columns 16w..16w+15.
// The first version (kept below as the large-kv fallback) re-rotated every key for every query row and walked
// the PV sum serially per lane: 68 us per call in the 125K-token draft prefill, 39 us in a decode step.
#define DRAFT_LDS_MAX_KEYS 384
#define DRAFT_KPAD 8                       // halfs of row padding: 144-B K rows / (kv+8)-half P,Vt rows
template <int D>
__global__ __launch_bounds__(256) void attn_rope_on_read_mfma_kernel(
    const h16* __restrict__ q, const h16* __restrict__ k, const h16* __restrict__ v, const h16* __restrict__ cosb,
    const h16* __restrict__ sinb, h16* __restrict__ out, int64_t stride_t, int64_t stride_h, int sq, int kv_len,
    int H, float scale) {
    static_assert(D == 64, "draft head_dim");
    constexpr int KS = D + DRAFT_KPAD;                // K row stride (halfs)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int kvp = (kv_len + 31) & ~31;              // keys padded to the MFMA k-step
    const int PS = kvp + DRAFT_KPAD;                  // P / Vt row stride (halfs)
    h16* sK = reinterpret_cast<h16*>(smem_raw);       // [kvp][KS]
    h16* sVt = sK + (size_t)kvp * KS;                 // [D][PS]
    h16* sP = sVt + (size_t)D * PS;                   // [16][PS]
#if TF_DRAFT_P_SPLIT > 0
    h16* sPl = sP + (size_t)16 * PS;                  // [16][PS]  low-order parts of P
    float* sS = reinterpret_cast<float*>(sPl + (size_t)16 * PS);  // [16][kvp]
#else
    float* sS = reinterpret_cast<float*>(sP + (size_t)16 * PS);   // [16][kvp]
#endif
    float* sL = sS + (size_t)16 * kvp;                // [16]

    const int h = blockIdx.x, q0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const h16* kb = k + (int64_t)h * stride_h;
    const h16* vb = v + (int64_t)h * stride_h;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- phase 1: rotate K -> sK, transpose V -> sVt ----
    for (int e = tid; e < kvp * 4; e += 256) {
        const int j = e >> 2, i = e & 3;              // key j, 8-wide vector i of the low half (d = 8i..8i+7)
        half8 lo = zero8, hi = zero8;
        if (j < kv_len) {
            const h16* kp = kb + (int64_t)j * stride_t;
            const half8 x1 = load_half8(kp + 8 * i), x2 = load_half8(kp + 8 * i + 32);
            const half8 c1 = load_half8(cosb + (int64_t)j * D + 8 * i), c2 = load_half8(cosb + (int64_t)j * D + 8 * i + 32);
            const half8 s1 = load_half8(sinb + (int64_t)j * D + 8 * i), s2 = load_half8(sinb + (int64_t)j * D + 8 * i + 32);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                // (x*cos) + (rotate_half(x)*sin), every op rounded to fp16 like the reference
                lo[t] = hadd_rn(hmul_rn(x1[t], c1[t]), hmul_rn((h16)(-(float)x2[t]), s1[t]));
                hi[t] = hadd_rn(hmul_rn(x2[t], c2[t]), hmul_rn(x1[t], s2[t]));
            }
        }
        store_half8(sK + (size_t)j * KS + 8 * i, lo);
        store_half8(sK + (size_t)j * KS + 8 * i + 32, hi);
    }
    for (int e = tid; e < kvp * 8; e += 256) {
        const int j = e >> 3, i = e & 7;
        const half8 x = (j < kv_len) ? load_half8(vb + (int64_t)j * stride_t + 8 * i) : zero8;
#pragma unroll
        for (int t = 0; t < 8; ++t) sVt[(size_t)(8 * i + t) * PS + j] = x[t];
    }
    half8 qf[D / 32];
    {
        const int row = q0 + li;
#pragma unroll
        for (int c = 0; c < D / 32; ++c)
            qf[c] = (row < sq) ? load_half8(q + ((int64_t)row * H + h) * D + 32 * c + 8 * g) : zero8;
    }
    __syncthreads();

    // ---- phase 2: S[q][key] = scale * Q K^T ----
    for (int t = wave; t < kvp / 16; t += 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < D / 32; ++c) {
            const half8 bk = load_half8(sK + (size_t)(t * 16 + li) * KS + 32 * c + 8 * g);
            s = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[c], bk, s, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[(size_t)(4 * g + r) * kvp + t * 16 + li] = s[r] * scale;   // C: row 4g+r, col li
    }
    __syncthreads();

    // ---- phase 3: softmax, wave owns rows 4w..4w+3 ----
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr, qrow = q0 + row;
        const int kmax = (qrow < sq) ? (kv_len - sq + qrow) : -1;       // bottom-right causal
        float mloc = NEG_BIG;
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
#if TF_DRAFT_P_SPLIT > 0
            sPl[(size_t)row * PS + j] = (h16)(p - (float)ph);          // low-order part of P: its own A operand below
#endif
        }
        lsum = wave_sum(lsum);
        if (lane == 0) sL[row] = lsum;
    }
    __syncthreads();

    // ---- phase 4: O[:, 16w..16w+15] = P V ----
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < kvp / 32; ++c) {
        const half8 ap = load_half8(sP + (size_t)li * PS + 32 * c + 8 * g);
        const half8 bv = load_half8(sVt + (size_t)(16 * wave + li) * PS + 32 * c + 8 * g);
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap, bv, o, 0, 0, 0);
#if TF_DRAFT_P_SPLIT > 0
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(load_half8(sPl + (size_t)li * PS + 32 * c + 8 * g), bv, o, 0, 0, 0);
#endif
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r, qrow = q0 + row;
        if (qrow < sq) out[((int64_t)qrow * H + h) * D + 16 * wave + li] = (h16)(o[r] / sL[row]);
    }
}

static size_t draft_mfma_lds_bytes(int kv_len) {
    const size_t kvp = (size_t)((kv_len + 31) & ~31), PS = kvp + DRAFT_KPAD;
    return kvp * (64 + DRAFT_KPAD) * 2 + 64 * PS * 2 + (TF_DRAFT_P_SPLIT > 0 ? 2 : 1) * 16 * PS * 2 + 16 * kvp * 4 + 16 * 4;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int64_t tf_attn_decode_ws_floats(int H, int sq, int D, int nsplit) {
    const int QR = ((sq + 15) / 16) * 16;
    return (int64_t)H * nsplit * QR * (D + 2);
}

// Split count of the split-KV kernel.  Measured over heads-per-rank 4..40 x 4K..130K keys x one / two q-tiles
// (tools/nsplit_sweep.py, profiles/r02_nsplit_sweep.json): the fastest grid is ONE workgroup per CU — nsplit * H ~ 256 —
// as long as a workgroup keeps >= 8 key tiles (2 per wave); more splits only add partials for the merge kernel to
// re-read, fewer leave CUs idle.  (The first rule — up to 4 workgroups per CU, >= 32 tiles each — was 5-25 % slower
// on the TP-shard shapes: 16 heads x 130K keys 237 vs 206 us, 32 heads x 12 305 keys 66 vs 54 us.)
// A/B switch of the rendezvous merge (many splits on a small grid, see FUSED_MERGE_BIG_SPLITS): 1 on, 0 (shipped: it
// measured slower) the two-launch merge
static int g_attn_rendezvous = 0;
extern "C" int tf_attn_tune(int key, int value) {
    if (key != 0) return -1;
    const int old = g_attn_rendezvous;
    g_attn_rendezvous = value ? 1 : 0;
    return old;
}

extern "C" int tf_attn_decode_pick_nsplit(int H, int sk) {
    const int tiles = (sk + 15) / 16;
    if (H < 1) H = 1;
    int by_grid = 256 / H;
    // Head counts that leave >= 16 CUs without a workgroup at one workgroup per CU (40 and 20 heads: 240 of 256) take TWO
    // workgroups per CU when the stream is long: 13B / 130K keys / 18 rows 513 -> 484 us at 12 splits; at 12 305 keys the
    // 6-split one-launch merge stays ahead, 61.4 vs 63.4 us; 7, 8 or 13 splits — a second workgroup on SOME CUs — cost
    // 30-50 %: a CU's stream rate, not the chip's, bounds this kernel (profiles/r04_attn_nsplit_g16.jsonl)
    if (by_grid >= 1 && 256 - by_grid * H >= 16 && 512 / H >= 2 * by_grid && tiles / (512 / H) >= 256) by_grid = 512 / H;
    int by_work = tiles / 8;
    int n = by_work < by_grid ? by_work : by_grid;
    if (n < 1) n = 1;
    if (n > 128) n = 128;
    return n;
}

template <int D, int QT>
static int launch_attn(const void* q, const void* k, const void* v, void* out, int64_t osm, int64_t osk,
                       int64_t stride_t, int64_t stride_h, int sq, int sk, const int32_t* sk_dev, int H, float scale,
                       int nsplit, float* ws, unsigned* tickets, hipStream_t st) {
    dim3 grid(nsplit, H), block(256);
    if (stride_t > 0x7fffffff || stride_h > 0x7fffffff) return TF_EINVAL;   // (the kernels take the two strides as 32-bit arguments)
    // many splits on a big grid: the parallel merge kernel wins; on a small grid (few heads) the last workgroup of a head
    // folds them inside the launch (see FUSED_MERGE_BIG_SPLITS)
    if (tickets && nsplit > FUSED_MERGE_MAX_SPLITS &&
        (D != 128 || H > 64 || nsplit > FUSED_MERGE_BIG_SPLITS || (int64_t)nsplit * H > FUSED_MERGE_BIG_MAX_WGS ||
         TF_ATTN_Q2_WAVES > 4 || !g_attn_rendezvous))
        tickets = nullptr;
    bool launched = false;
#if TF_ATTN_DEEP_TILES > 0
    if constexpr (QT == 1) {
        // tiles one wave owns at the HOST key count (a device-side count can only be smaller)
        const int ntiles = (sk + 15) / 16, tps = (ntiles + nsplit - 1) / nsplit, per_wave = (tps + 3) / 4;
        if (per_wave <= 2 * TF_ATTN_DEEP_TILES && (int64_t)nsplit * H <= 320) {      // short streams, <= ~1 workgroup per CU
            hipLaunchKernelGGL((attn_split_deep_kernel<D>), grid, block, 0, st, (const h16*)q, (const h16*)k, (const h16*)v,
                               sk_dev, sq, sk, H, nsplit, (int)stride_t, (int)stride_h, scale, ws, tickets, (h16*)out, osm, osk);
            launched = true;
        }
    }
#endif
#if TF_ATTN_QT2_OCC > 0
    if constexpr (QT == 2)
        hipLaunchKernelGGL((attn_split_q2_kernel<D>), grid, dim3(64 * TF_ATTN_Q2_WAVES), 0, st, (const h16*)q, (const h16*)k, (const h16*)v,
                           sk_dev, sq, sk, H, nsplit, (int)stride_t, (int)stride_h, scale, ws, tickets, (h16*)out, osm, osk);
    else
#endif
    if (!launched)
        hipLaunchKernelGGL((attn_split_kernel<D, QT>), grid, block, 0, st, (const h16*)q, (const h16*)k, (const h16*)v,
                           sk_dev, sq, sk, H, nsplit, (int)stride_t, (int)stride_h, scale, ws, tickets, (h16*)out, osm, osk);
    TF_LAUNCH_CHECK();
    if (tickets) return TF_OK;
    hipLaunchKernelGGL((attn_combine_kernel<D>), dim3(H, sq), dim3(D, COMBINE_GROUPS), 0, st, (const float*)ws,
                       (h16*)out, sq, H, nsplit, QT * 16, osm, osk);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

static int attn_decode_any(const void* q, const void* k, const void* v, void* out, int64_t osm, int64_t osk,
                           int64_t stride_t, int64_t stride_h, int sq, int sk, const int32_t* sk_dev, int H, int D,
                           float scale, int nsplit, float* ws, int64_t ws_floats, unsigned* tickets, void* stream) {
    if (!q || !k || !v || !out || !ws) return TF_EINVAL;
    if (sq < 1 || sq > 32 || sk < 1 || H < 1 || nsplit < 1 || nsplit > COMBINE_MAX_SPLITS) return TF_EINVAL;
    if ((stride_t % 8) || (stride_h % 8)) return TF_EINVAL;          // 16-B loads
    if (osm < 8 || osk < 8 || (osm % 4) || (osk % 4)) return TF_EINVAL;   // 8-byte output stores
    if (ws_floats < tf_attn_decode_ws_floats(H, sq, D, nsplit)) return TF_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const int QT = (sq + 15) / 16;
    if (D == 128 && QT == 1) return launch_attn<128, 1>(q, k, v, out, osm, osk, stride_t, stride_h, sq, sk, sk_dev, H, scale, nsplit, ws, tickets, st);
    if (D == 128 && QT == 2) return launch_attn<128, 2>(q, k, v, out, osm, osk, stride_t, stride_h, sq, sk, sk_dev, H, scale, nsplit, ws, tickets, st);
    if (D == 64 && QT == 1) return launch_attn<64, 1>(q, k, v, out, osm, osk, stride_t, stride_h, sq, sk, sk_dev, H, scale, nsplit, ws, tickets, st);
    if (D == 64 && QT == 2) return launch_attn<64, 2>(q, k, v, out, osm, osk, stride_t, stride_h, sq, sk, sk_dev, H, scale, nsplit, ws, tickets, st);
    return TF_EINVAL;
}

extern "C" int tf_attn_decode(const void* q, const void* k, const void* v, void* out, int64_t stride_t,
                              int64_t stride_h, int sq, int sk, const int32_t* sk_dev, int H, int D, float scale,
                              int nsplit, float* ws, int64_t ws_floats, void* stream) {
    return attn_decode_any(q, k, v, out, (int64_t)H * D, 8, stride_t, stride_h, sq, sk, sk_dev, H, D, scale, nsplit, ws,
                           ws_floats, nullptr, stream);
}

// The same call as ONE launch: the last workgroup of each head to finish merges that head's splits (bit-identical to
// the two-launch form).  ``tickets``: H uint32 words of device memory, zero before the first call; every call leaves
// them zero.  One ticket array serves one stream: two calls that may run concurrently need two arrays.
extern "C" int tf_attn_decode_fused(const void* q, const void* k, const void* v, void* out, int64_t stride_t,
                                    int64_t stride_h, int sq, int sk, const int32_t* sk_dev, int H, int D, float scale,
                                    int nsplit, float* ws, int64_t ws_floats, uint32_t* tickets, void* stream) {
    if (!tickets) return TF_EINVAL;
    return attn_decode_any(q, k, v, out, (int64_t)H * D, 8, stride_t, stride_h, sq, sk, sk_dev, H, D, scale, nsplit, ws,
                           ws_floats, tickets, stream);
}

// Either form with the OUTPUT in an explicit activation layout (include/triforce_hip.h, tf_skinny_gemm_act): element
// (row, column c = h * D + d) at out[row * out_sm + (c / 8) * out_sk + c % 8] — the o_proj GEMM that consumes it then
// reads its B operand in k-octet-major form.  tickets == NULL: two launches; else the one-launch merge.
extern "C" int tf_attn_decode_act(const void* q, const void* k, const void* v, void* out, int64_t out_sm, int64_t out_sk,
                                  int64_t stride_t, int64_t stride_h, int sq, int sk, const int32_t* sk_dev, int H, int D,
                                  float scale, int nsplit, float* ws, int64_t ws_floats, uint32_t* tickets, void* stream) {
    return attn_decode_any(q, k, v, out, out_sm, out_sk, stride_t, stride_h, sq, sk, sk_dev, H, D, scale, nsplit, ws,
                           ws_floats, tickets, stream);
}

extern "C" int64_t tf_attn_block_ws_floats(int H, int D, int nsplit) { return (int64_t)H * nsplit * 128 * (D + 2); }

extern "C" int tf_attn_block_pick_nsplit(int H, int sq, int sk) {
    const int tiles = (sk + 15) / 16;
    const int rg = sq <= 32 ? 1 : (sq <= 64 ? 2 : 4);
    int by_work = tiles / (sq <= 32 ? 32 : 16);      // >= 8 key tiles per wave (rg = 1) / 16 per workgroup
    // rg = 4 is matrix-core bound with ~2 resident workgroups per CU: 512 workgroups fill the chip, and every
    // extra split costs 128 rows x (D+2) floats of partials per head for the combine kernel to re-read
    int by_grid = (rg == 4 ? 512 : 1024) / (H > 0 ? H : 1);
    int n = by_work < by_grid ? by_work : by_grid;
    const int cap = COMBINE_MAX_SPLITS / (4 / rg);   // partials per head = nsplit * (4 / rg)
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    return n;
}

template <int D, bool TREE>
static int launch_block(const void* q, const void* k, const void* v, void* out, int64_t stride_t, int64_t stride_h,
                        int sq, int sk, int H, float scale, int nsplit, float* ws, const uint32_t* mask,
                        int mask_words, int mask_row0, int tree_start, hipStream_t st) {
    const int rg = sq <= 32 ? 1 : (sq <= 64 ? 2 : 4);
    if (rg == 4 && !TF_BLOCK_NO_LDS) {
        static bool attr_set = false;                     // 70 KiB of dynamic LDS: above the 64 KiB default limit
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_block_lds_kernel<D, TREE>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)blk_lds_bytes<D>());
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_block_lds_kernel<D, TREE>), dim3(nsplit, H), dim3(256), blk_lds_bytes<D>(), st,
                           (const h16*)q, (const h16*)k, (const h16*)v, stride_t, stride_h, sq, sk, H, scale, nsplit, ws,
                           mask, mask_words, mask_row0, tree_start);
    } else
        hipLaunchKernelGGL((attn_block_kernel<D, TREE>), dim3(nsplit, H), dim3(256), 0, st, (const h16*)q,
                           (const h16*)k, (const h16*)v, stride_t, stride_h, sq, sk, H, scale, nsplit, rg, ws, mask,
                           mask_words, mask_row0, tree_start);
    TF_LAUNCH_CHECK();
    if (sq > 32)
        hipLaunchKernelGGL((attn_combine_rows_kernel<D>), dim3((sq * H + 3) / 4), dim3(256), 0, st, (const float*)ws,
                           (h16*)out, sq, H, nsplit * (4 / rg), 32 * rg, 0);
    else
        hipLaunchKernelGGL((attn_combine_kernel<D>), dim3(H, sq), dim3(D, COMBINE_GROUPS), 0, st, (const float*)ws,
                           (h16*)out, sq, H, nsplit * (4 / rg), 32 * rg, (int64_t)H * D, (int64_t)8);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// ---- whole-chunk causal prefill: tf_attn_prefill ---------------------------------------------------------------
#define PREFILL_MAX_ROWS 4096
extern "C" int tf_attn_prefill_pick_nsplit(int H, int sq, int sk) {
    const int nrb = (sq + 127) / 128;
    const int slabs = (sk + BLK_SLAB - 1) / BLK_SLAB;
    int n = 1024 / ((H > 0 ? H : 1) * nrb);          // ~4 workgroups per CU over the launch
    const int by_work = slabs / 16;                   // >= 16 slabs (1024 keys) per workgroup
    if (n > by_work) n = by_work;
    if (n > COMBINE_MAX_SPLITS) n = COMBINE_MAX_SPLITS;
    if (n < 1) n = 1;
    while ((H * n) % 8) ++n;                          // (split, head) pairs are dealt to the 8 XCDs
    return n;
}

extern "C" int64_t tf_attn_prefill_ws_floats(int H, int sq, int D, int nsplit) {
    return (int64_t)((sq + 127) / 128) * H * nsplit * 128 * (D + 2);
}

template <int D>
static int launch_prefill(const void* q, const void* k, const void* v, void* out, int64_t stride_t, int64_t stride_h,
                          int sq, int sk, int H, float scale, int nsplit, float* ws, hipStream_t st) {
    static bool attr_set = false;                         // 70 KiB of dynamic LDS: above the 64 KiB default limit
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_prefill_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)blk_lds_bytes<D>());
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nrb = (sq + 127) / 128, rbg = nrb < 8 ? nrb : 8;
    const int groups = (nrb + rbg - 1) / rbg;
    const int64_t blocks = (int64_t)H * nsplit * rbg * groups;
    hipLaunchKernelGGL((attn_prefill_kernel<D>), dim3((unsigned)blocks), dim3(256), blk_lds_bytes<D>(), st, (const h16*)q,
                       (const h16*)k, (const h16*)v, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, nrb, rbg);
    TF_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_combine_rows_kernel<D>), dim3((sq * H + 3) / 4), dim3(256), 0, st, (const float*)ws,
                       (h16*)out, sq, H, nsplit, 128, 128);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// Causal attention of a prefill chunk of 129..4096 query rows in one launch (bottom-right aligned: row i sees keys
// [0, sk - sq + i]); replaces one flash_attn_with_kvcache call per 128-row piece of the chunk.  (H * nsplit) % 8 == 0.
extern "C" int tf_attn_prefill(const void* q, const void* k, const void* v, void* out, int64_t stride_t,
                               int64_t stride_h, int sq, int sk, int H, int D, float scale, int nsplit, float* ws,
                               int64_t ws_floats, void* stream) {
    if (!q || !k || !v || !out || !ws) return TF_EINVAL;
    if (sq < 1 || sq > PREFILL_MAX_ROWS || sk < sq || H < 1 || nsplit < 1 || nsplit > COMBINE_MAX_SPLITS) return TF_EINVAL;
    if ((H * nsplit) % 8) return TF_EINVAL;
    if ((stride_t % 8) || (stride_h % 8)) return TF_EINVAL;
    if (ws_floats < tf_attn_prefill_ws_floats(H, sq, D, nsplit)) return TF_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    if (D == 128) return launch_prefill<128>(q, k, v, out, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, st);
    if (D == 64) return launch_prefill<64>(q, k, v, out, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, st);
    return TF_EINVAL;
}

// Attention of a block of 1..128 query rows against the cache, D = 128 or 64.  mask == NULL: bottom-right causal
// (a prefill chunk).  mask != NULL: tree attention — keys [0, tree_start) visible to all rows, key tree_start + j
// visible to row i iff bit j of mask[(mask_row0 + i) * mask_words + j / 32] is set.
extern "C" int tf_attn_block(const void* q, const void* k, const void* v, void* out, int64_t stride_t,
                             int64_t stride_h, int sq, int sk, int H, int D, float scale, int nsplit, float* ws,
                             int64_t ws_floats, const uint32_t* mask, int mask_words, int mask_row0, int tree_start,
                             void* stream) {
    if (!q || !k || !v || !out || !ws) return TF_EINVAL;
    if (sq < 1 || sq > 128 || sk < 1 || H < 1 || nsplit < 1) return TF_EINVAL;
    if (!mask && sk < sq) return TF_EINVAL;
    if (mask && (mask_words < 1 || mask_row0 < 0 || tree_start < 0 || tree_start > sk ||
                 sk - tree_start > 32 * mask_words)) return TF_EINVAL;
    const int rg = sq <= 32 ? 1 : (sq <= 64 ? 2 : 4);
    if (nsplit * (4 / rg) > COMBINE_MAX_SPLITS) return TF_EINVAL;
    if ((stride_t % 8) || (stride_h % 8)) return TF_EINVAL;
    if (ws_floats < tf_attn_block_ws_floats(H, D, nsplit)) return TF_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    if (D == 128)
        return mask ? launch_block<128, true>(q, k, v, out, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, mask,
                                              mask_words, mask_row0, tree_start, st)
                    : launch_block<128, false>(q, k, v, out, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, nullptr,
                                               0, 0, 0, st);
    if (D == 64)
        return mask ? launch_block<64, true>(q, k, v, out, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, mask,
                                             mask_words, mask_row0, tree_start, st)
                    : launch_block<64, false>(q, k, v, out, stride_t, stride_h, sq, sk, H, scale, nsplit, ws, nullptr,
                                              0, 0, 0, st);
    return TF_EINVAL;
}

extern "C" int tf_attn_rope_on_read(const void* q, const void* k, const void* v, const void* cosb, const void* sinb,
                                    void* out, int64_t stride_t, int64_t stride_h, int sq, int kv_len, int H, int D,
                                    float scale, void* stream) {
    if (!q || !k || !v || !cosb || !sinb || !out) return TF_EINVAL;
    if (D != 64 || sq < 1 || kv_len < sq || H < 1) return TF_EINVAL;
    if (kv_len <= DRAFT_LDS_MAX_KEYS) {
        static bool attr_set = false;                     // > 64 KiB of dynamic LDS needs the opt-in once
        const size_t lds = draft_mfma_lds_bytes(kv_len);
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_rope_on_read_mfma_kernel<64>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)draft_mfma_lds_bytes(DRAFT_LDS_MAX_KEYS));
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        dim3 grid(H, (sq + 15) / 16), block(256);
        hipLaunchKernelGGL((attn_rope_on_read_mfma_kernel<64>), grid, block, lds, (hipStream_t)stream, (const h16*)q,
                           (const h16*)k, (const h16*)v, (const h16*)cosb, (const h16*)sinb, (h16*)out, stride_t,
                           stride_h, sq, kv_len, H, scale);
        TF_LAUNCH_CHECK();
        return TF_OK;
    }
    const size_t lds = (size_t)4 * kv_len * sizeof(float);
    if (lds > 64 * 1024) return TF_ERANGE;
    dim3 grid(H, (sq + 3) / 4), block(256);
    hipLaunchKernelGGL((attn_rope_on_read_kernel<64>), grid, block, lds, (hipStream_t)stream, (const h16*)q,
                       (const h16*)k, (const h16*)v, (const h16*)cosb, (const h16*)sinb, (h16*)out, stride_t, stride_h,
                       sq, kv_len, H, scale);
    TF_LAUNCH_CHECK();
    return TF_OK;
}
