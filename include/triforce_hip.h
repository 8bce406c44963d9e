/*
 * triforce_hip.h — C ABI of libtriforce_hip.so (hand-written gfx950 / CDNA4 kernels for the
 * TriForce hierarchical draft/verify decode path).
 *
 * The reference (Infini-AI-Lab/TriForce @ 2024-10-08) has no FFI layer: its native boundary is
 * the third-party op flash_attn.flash_attn_with_kvcache plus torch kernels called from Python
 * (SURVEY.md §8b, B4).  Each entry point below replaces one such native call site; the citation
 * on every declaration is the reference file:line it stands in for.  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - fp16 tensors are passed as `const void*` / `void*` (IEEE binary16, last dim contiguous);
 *   - KV caches are addressed as base + token*stride_t + head*stride_h + d  (strides in ELEMENTS),
 *     so both the reference layout [T][H][D] and the head-major layout [H][T][D] used by
 *     triforce_amd are valid inputs;
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*), allocates nothing,
 *     never synchronises and is hipGraph-capturable; workspaces are caller-provided;
 *   - return value: 0 on success, a negative errno-style code on bad arguments (TF_E*), or the
 *     positive hipError_t of a failed launch.
 */
#ifndef TRIFORCE_HIP_H
#define TRIFORCE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_OK        0
#define TF_EINVAL  (-22)   /* bad argument (shape / alignment / unsupported head_dim)        */
#define TF_ENOSPC  (-28)   /* caller-provided workspace too small                           */
#define TF_ERANGE  (-34)   /* size exceeds what the kernel supports (e.g. chunks > 32768)    */

/* ABI version; bumped on any signature change. */
int tf_abi_version(void);

/* -------------------------------------------------------------------------------------------
 * Verify / decode attention  (replaces flash_attn_with_kvcache(q, k_cache, v_cache,
 * softmax_scale, causal=True): models/modeling_llama.py:240, models/tensor_op.py:168,316)
 *
 * Split-KV attention of sq <= 32 query rows against sk cached keys per head, bottom-right
 * aligned causal mask (query i sees keys [0, sk - sq + i]).  QK^T and PV run on MFMA
 * (v_mfma_f32_16x16x32_f16 / 16x16x16_f16), softmax in fp32, output fp16.
 *   q        [sq][H][D] fp16 (post-RoPE)
 *   k, v     layer base of the cache, see stride convention above; keys [0, sk) are read
 *   out      [sq][H*D] fp16
 *   sk_dev   optional device int32: if non-NULL the key count is read from it at run time (lets a
 *            captured hipGraph follow a growing cache); `sk` is then the upper bound used to size
 *            the launch
 *   ws       float workspace, at least tf_attn_decode_ws_floats(H, sq, D, nsplit) floats
 *   nsplit   number of KV splits per head (>=1); tf_attn_decode_pick_nsplit gives a default
 * D must be 64 or 128.
 * ------------------------------------------------------------------------------------------- */
int64_t tf_attn_decode_ws_floats(int H, int sq, int D, int nsplit);
int tf_attn_decode_pick_nsplit(int H, int sk);
int tf_attn_decode(const void* q, const void* k, const void* v, void* out,
                   int64_t stride_t, int64_t stride_h,
                   int sq, int sk, const int32_t* sk_dev, int H, int D, float scale,
                   int nsplit, float* ws, int64_t ws_floats, void* stream);
/* The same call as ONE launch (tf_attn_decode is split kernel + merge kernel): the last workgroup of each head to
 * finish merges that head's splits inside the split kernel — same arithmetic in the same order, bit-identical output.
 *   tickets  H uint32 words of device memory, ZERO before the first call; every call leaves them zero.  One ticket
 *            array serves one stream: calls that may run concurrently need separate arrays.
 * Falls back to the two-launch form when 3 * nsplit * 16*ceil(sq/16) floats do not fit the kernel's 64*(D+1)-float
 * merge scratch (e.g. D = 128: nsplit > 172 at sq <= 16). */
int tf_attn_decode_fused(const void* q, const void* k, const void* v, void* out,
                         int64_t stride_t, int64_t stride_h,
                         int sq, int sk, const int32_t* sk_dev, int H, int D, float scale,
                         int nsplit, float* ws, int64_t ws_floats, uint32_t* tickets, void* stream);
/* tf_attn_decode / tf_attn_decode_fused (tickets != NULL) with the OUTPUT in an explicit activation layout: element (row,
 * column c = h * D + d) at out[row * out_sm + (c / 8) * out_sk + c % 8] (see tf_skinny_gemm_act; row-major = H * D, 8). */
int tf_attn_decode_act(const void* q, const void* k, const void* v, void* out, int64_t out_sm, int64_t out_sk,
                       int64_t stride_t, int64_t stride_h, int sq, int sk, const int32_t* sk_dev, int H, int D,
                       float scale, int nsplit, float* ws, int64_t ws_floats, uint32_t* tickets, void* stream);
/* tf_attn_tune(0, v): v = 0 (default) keeps the split merge of > 8 splits as a second launch; 1 lets small grids (H * nsplit
 * <= 256, H <= 64, D = 128) merge inside the launch by a rendezvous of the head's workgroups (csrc/attn.hip; measured
 * slower, profiles/r04_attn_rendezvous_merge_ab.jsonl).  Returns the previous value, -1 for an unknown key.  Bit-identical
 * either way. */
int tf_attn_tune(int key, int value);

/* -------------------------------------------------------------------------------------------
 * Block attention: 1 <= sq <= 128 query rows in ONE pass over the keys.
 *   mask == NULL  bottom-right causal — flash_attn_with_kvcache with q_len = 128 in the chunked prefill
 *                 (utils/graph_infer.py:30-37 -> models/modeling_llama.py:240; models/TP_llama.py:246-250 ->
 *                 models/tensor_op.py:168).
 *   mask != NULL  tree attention — F.scaled_dot_product_attention with the dense additive [prefix | tree] mask
 *                 of the Sequoia path (models/tensor_op.py:171,265; masks built at utils/SpecTree_TP.py:65-67,
 *                 83-87,170): keys [0, tree_start) are visible to every row, key tree_start + j is visible to
 *                 query row i iff bit (j % 32) of mask[(mask_row0 + i) * mask_words + j / 32] is set; keys
 *                 [tree_start, sk) must fit 32 * mask_words bits.
 * Each workgroup's waves own 32 query rows apiece and walk the same key tiles, so the KV cache is streamed
 * from HBM once per call.  ws: at least tf_attn_block_ws_floats(H, D, nsplit) floats.
 * ------------------------------------------------------------------------------------------- */
int64_t tf_attn_block_ws_floats(int H, int D, int nsplit);
int tf_attn_block_pick_nsplit(int H, int sq, int sk);
int tf_attn_block(const void* q, const void* k, const void* v, void* out,
                  int64_t stride_t, int64_t stride_h, int sq, int sk, int H, int D, float scale,
                  int nsplit, float* ws, int64_t ws_floats,
                  const uint32_t* mask, int mask_words, int mask_row0, int tree_start, void* stream);

/* Whole-chunk causal prefill: 1 <= sq <= 4096 query rows (bottom-right aligned: row i sees keys [0, sk - sq + i]) in
 * ONE launch — the chunked prefill's attention (utils/graph_infer.py:30-37 -> models/modeling_llama.py:240 with
 * q_len = the chunk).  The 128-row blocks of the chunk that read the same key range run together on one XCD and share
 * its L2, so the KV cache is streamed from HBM about once per chunk instead of once per 128 rows.
 * (H * nsplit) % 8 must be 0 (tf_attn_prefill_pick_nsplit returns such a value); sk >= sq. */
int tf_attn_prefill_pick_nsplit(int H, int sq, int sk);
int64_t tf_attn_prefill_ws_floats(int H, int sq, int D, int nsplit);
int tf_attn_prefill(const void* q, const void* k, const void* v, void* out,
                    int64_t stride_t, int64_t stride_h, int sq, int sk, int H, int D, float scale,
                    int nsplit, float* ws, int64_t ws_floats, void* stream);

/* -------------------------------------------------------------------------------------------
 * Draft (Llama-68M) attention with RoPE applied to the cached keys on read
 * (models/modeling_llama_68m.py:151-190): keys are cached UN-rotated and rotated with
 * cache-relative positions 0..kv_len-1 at every call; q arrives already rotated.
 *   q [sq][H][D] fp16, k/v cache rows [0,kv_len), cos/sin [max_pos][D] fp16, out [sq][H*D].
 * D must be 64.  Bottom-right causal.
 * ------------------------------------------------------------------------------------------- */
int tf_attn_rope_on_read(const void* q, const void* k, const void* v, const void* cos, const void* sin,
                         void* out, int64_t stride_t, int64_t stride_h,
                         int sq, int kv_len, int H, int D, float scale, void* stream);

/* -------------------------------------------------------------------------------------------
 * The Llama-68M draft forward as one entry point (models/modeling_llama_68m.py:129-190 with the StreamingLLM cache of
 * models/cache.py:266-330; utils/graph_infer.py:52-57 `draft_run`): n <= 32 token ids -> fp32 logits [n][vocab] and,
 * when probs_out != NULL, softmax(top_p(logits[n-1] / temperature)) [vocab] (utils/sampling.py:43-60).
 * Native host code: the call issues the whole launch chain (embedding + positions, per layer RMSNorm + q|k|v + RoPE +
 * KV append / rope-on-read attention / o_proj + residual / RMSNorm + gate|up + SwiGLU / down_proj + residual, final
 * RMSNorm + lm_head, top-p) on `stream` through the kernels of the entry points above — bit-identical to calling them
 * one by one — allocates nothing and is graph-capturable.
 *   model  device pointers: embed [vocab][hidden] fp16; per layer ln1 / ln2 [hidden], wqkv packed in ROTARY-PAIR row
 *          order (tf_skinny_qkv_rope), wo / wgate / wup / wdown packed (tf_skinny_gemm); norm, lm_head packed;
 *          cos / sin [max_pos][head_dim] fp16.  head_dim must be 64 (tf_attn_rope_on_read).
 *   cache  per layer K / V base pointers, stride convention of the top of this file; keys are stored UN-rotated
 *   slot0  cache row of ids[0] (the rows [slot0, slot0 + n) are written); kv_len >= slot0 + n keys are attended
 *   ws     >= tf_draft_forward_ws_bytes(model, n) bytes of device scratch
 * ------------------------------------------------------------------------------------------- */
#define TF_DRAFT_MAX_LAYERS 8
typedef struct TfDraftModel {
    const void* embed;
    const void* ln1[TF_DRAFT_MAX_LAYERS];
    const void* wqkv[TF_DRAFT_MAX_LAYERS];
    const void* wo[TF_DRAFT_MAX_LAYERS];
    const void* ln2[TF_DRAFT_MAX_LAYERS];
    const void* wgate[TF_DRAFT_MAX_LAYERS];
    const void* wup[TF_DRAFT_MAX_LAYERS];
    const void* wdown[TF_DRAFT_MAX_LAYERS];
    const void* norm;
    const void* lm_head;
    const void* cos;
    const void* sin;
    int32_t layers, hidden, heads, head_dim, inter, vocab;
    float eps, scale;
} TfDraftModel;
typedef struct TfDraftCache {
    void* k[TF_DRAFT_MAX_LAYERS];
    void* v[TF_DRAFT_MAX_LAYERS];
    int64_t stride_t, stride_h;
} TfDraftCache;
int64_t tf_draft_forward_ws_bytes(const TfDraftModel* model, int n);
int tf_draft_forward_68m(const TfDraftModel* model, const TfDraftCache* cache, const int64_t* ids, int n, int slot0,
                         int kv_len, float* logits_out, float* probs_out, float temperature, float top_p,
                         void* ws, int64_t ws_bytes, void* stream);

/* The same forward as ONE LAUNCH (csrc/draft_persist.hip; replaces the ~40 kernels of models/modeling_llama_68m.py:129-190
 * + utils/sampling.py:5-27,43-60 per draft step, and the 13 launches of tf_draft_forward_68m): 256 co-resident workgroups,
 * the stages meet through per-edge arrival counters in `ctl` (write-through stores, agent-scope loads, no fences, no
 * cooperative launch).  Outputs — logits, the probability row, the K / V rows written — are BIT-IDENTICAL to
 * tf_draft_forward_68m.  Shapes: hidden 768, 12 heads of 64, inter 3072, 1-2 layers, vocab % 16 == 0 and <= 32768,
 * 1 <= n <= 16, kv_len <= 384, a device with >= 256 compute units (tf_draft_persist_supported: 0 = taken, TF_EINVAL /
 * TF_ERANGE otherwise — the caller keeps tf_draft_forward_68m).
 *   ws    >= tf_draft_persist_ws_bytes(model) bytes of device scratch, 256-byte aligned, owned by this control block's launches
 *   ctl   tf_draft_persist_ctl_bytes() bytes of device memory, 64-byte aligned, ZERO-filled once (tf_draft_persist_reset);
 *         launches on one ctl must be stream-ordered (graph replays of one engine are).
 * Every wait is bounded by wall-clock time (tf_draft_persist_tune key 0, ms, default 2000; frozen into captured graphs).
 * A time-out sets the sticky error word of ctl (+ the optional pinned host mirror): the outputs of that and of every later
 * launch are NaN until tf_draft_persist_reset.  tf_draft_persist_error: 0 or 1 + index of the edge whose wait timed out
 * (blocking read).  tf_draft_persist_tune key 1 = fault injection for tests (edge index + 1 whose first producer loses its
 * arrival; 0 off).  tf_draft_persist_stamps: device buffer of 256 x (return value) u64 100-MHz wall-clock stamps per
 * workgroup written by the following launches (NULL: off) — tools/draft_persist_stamps.py. */
int64_t tf_draft_persist_ctl_bytes(void);
int64_t tf_draft_persist_ws_bytes(const TfDraftModel* model);
int tf_draft_persist_supported(const TfDraftModel* model, int n, int kv_len);
int tf_draft_forward_68m_persist(const TfDraftModel* model, const TfDraftCache* cache, const int64_t* ids, int n, int slot0,
                                 int kv_len, float* logits_out, float* probs_out, float temperature, float top_p,
                                 void* ws, int64_t ws_bytes, void* ctl, void* stream);
int tf_draft_persist_tune(int key, int value);
int tf_draft_persist_stamps(void* buf);
int tf_draft_persist_error(const void* ctl);
int tf_draft_persist_reset(void* ctl, void* host_mirror, int set_mirror);

/* -------------------------------------------------------------------------------------------
 * Retrieval-cache build (models/cache.py:146-178 == :517-556): chunk-mean scoring, per-head
 * top-k, chunk gather.
 * tf_retrieval_score : scores[h][c] = fp16( q[h] . fp16(mean_{t in chunk c} K[t][h]) ), un-scaled
 *                      (cache.py:154-157).  k rows [0, C*chunk) are read once.
 * tf_retrieval_topk  : idx[h][0] = 0, idx[h][1..sets-1] = the sets-1 best chunks of [1,C),
 *                      descending score, ascending chunk index among equal scores
 *                      (cache.py:159-162; torch.topk leaves the tie order undefined).  C <= 32768.
 * tf_retrieval_gather: dst slot j of head h <- chunk idx[h][j] (chunk rows of D), K and V
 *                      (cache.py:163-175).
 * ------------------------------------------------------------------------------------------- */
int tf_retrieval_score(const void* k, int64_t stride_t, int64_t stride_h, const void* q,
                       void* scores, int C, int chunk, int H, int D, void* stream);
int tf_retrieval_topk(const void* scores, int32_t* idx, int C, int sets, int H, void* stream);
int tf_retrieval_gather(const void* k_src, const void* v_src, int64_t src_stride_t, int64_t src_stride_h,
                        const int32_t* idx, void* k_dst, void* v_dst, int64_t dst_stride_t,
                        int64_t dst_stride_h, int sets, int chunk, int H, int D, void* stream);

/* -------------------------------------------------------------------------------------------
 * KV row movement.
 * tf_kv_copy_rows : dst[l][h][dst_t0+i] = src[l][h][src_t0+i], i<n, for L layers x H heads of
 *                   D elements — RetrievalCache.update_graph_cache (cache.py:180-182, :566-575).
 *                   Source and destination must not overlap.
 * tf_kv_shift_rows: in-place move of rows [src_t0, src_t0+n) to [dst_t0, dst_t0+n), dst_t0 <=
 *                   src_t0, overlap allowed — StreamingLLMEvictionCache.evict_for_spec /
 *                   evict_prefill (cache.py:252-265).
 * ------------------------------------------------------------------------------------------- */
int tf_kv_copy_rows(const void* src, int64_t src_stride_l, int64_t src_stride_t, int64_t src_stride_h,
                    void* dst, int64_t dst_stride_l, int64_t dst_stride_t, int64_t dst_stride_h,
                    int src_t0, int dst_t0, int n, int L, int H, int D, void* stream);
int tf_kv_shift_rows(void* cache, int64_t stride_l, int64_t stride_t, int64_t stride_h,
                     int src_t0, int dst_t0, int n, int L, int H, int D, void* stream);
/* The same two movements for the K AND V tensors of one cache (equal strides and row ranges) in one launch — the two
 * tail copies and the two window shifts every decode step ends with (cache.py:180-182,252-265). */
int tf_kv_copy_rows_pair(const void* src_k, const void* src_v, int64_t src_stride_l, int64_t src_stride_t,
                         int64_t src_stride_h, void* dst_k, void* dst_v, int64_t dst_stride_l, int64_t dst_stride_t,
                         int64_t dst_stride_h, int src_t0, int dst_t0, int n, int L, int H, int D, void* stream);
int tf_kv_shift_rows_pair(void* k_cache, void* v_cache, int64_t stride_l, int64_t stride_t, int64_t stride_h,
                          int src_t0, int dst_t0, int n, int L, int H, int D, void* stream);

/* tf_kv_gather_rows: rows offset+idx[j] -> offset+j (j < n) of K and V for L layers x H heads — the compaction of
 *                   the accepted tree nodes, DistributedSimpleCache.gather_kv_incremental (cache.py:333-343).
 *                   idx (device int32) must be strictly increasing (a root-to-leaf path of the tree). */
int tf_kv_gather_rows(void* k_cache, void* v_cache, int64_t stride_l, int64_t stride_t, int64_t stride_h,
                      int offset, const int32_t* idx, int n, int L, int H, int D, void* stream);

/* -------------------------------------------------------------------------------------------
 * Dense-block glue (models/modeling_llama.py:132-159,221-238; models/tensor_op.py:25-64).
 * tf_rmsnorm    : optional fused residual add: s = fp16(x + res) (written to sum_out when
 *                 non-NULL), y = w * fp16(s * rsqrt(mean(s^2)+eps))  — the cast precedes the
 *                 weight multiply exactly as modeling_llama.py:138-143.
 * tf_rope_append: split a fused qkv row [q | k | v] (each H*D), rotate q (and k unless
 *                 rotate_k==0) at positions[i] with fp16 tables, write q to q_out [rows][H][D]
 *                 and the k/v rows into the cache at token slot0+i (slot0 read from slot0_dev
 *                 when non-NULL).  fp16 arithmetic order = (x*cos) + (rotate_half(x)*sin).
 * tf_silu_mul   : out = fp16(silu(gate)) * up for a fused [gate | up] row of 2*I.
 * ------------------------------------------------------------------------------------------- */
int tf_rmsnorm(const void* x, const void* res, const void* w, void* y, void* sum_out,
               int rows, int hidden, float eps, void* stream);
int tf_rope_append(const void* qkv, int64_t qkv_row_stride, const void* cos, const void* sin,
                   const int64_t* positions, void* q_out, void* k_cache, void* v_cache,
                   int64_t stride_t, int64_t stride_h, int slot0, const int32_t* slot0_dev,
                   int rows, int H, int D, int rotate_k, void* stream);
int tf_silu_mul(const void* gate_up, void* out, int rows, int I, void* stream);
/* tf_embed_rows: x = embed_tokens(input_ids) (models/modeling_llama.py:342, TP_llama.py:206) for the n <= 32 rows of a
 * decode forward, written in an explicit activation layout (see tf_skinny_gemm_act): element (m, k) at
 * out[m * out_sm + (k / 8) * out_sk + k % 8].  embed [vocab][hidden] fp16, ids [n] int64 (clamped to the vocabulary). */
int tf_embed_rows(const void* embed, const int64_t* ids, void* out, int64_t out_sm, int64_t out_sk, int n, int hidden,
                  int vocab, void* stream);
/* tf_set_tokens: the token ids, positions and length scalars of one decode forward in ONE launch; the ids (known on the
 * host: the loop has just read its accept record) travel as kernel arguments.  dst[i] = host_vals[i] for i < n_vals,
 * pad for n_vals <= i < n_dst (utils/decoding.py:94,177 — the verify / pass token rows padded with a filler id);
 * pos[i] = pos0 + i for i < n_pos (position_ids, decoding.py:178); *slot = pos0, *sk = sk_val when given (the device-side
 * lengths of a captured full-cache forward).  HOST pointer: host_vals; device pointers: dst, pos, slot, sk (each optional
 * with its count 0 / NULL).  n_dst, n_vals <= 32, n_pos <= 64. */
int tf_set_tokens(int64_t* dst, int n_dst, const int64_t* host_vals, int n_vals, int64_t pad, int64_t* pos, int n_pos,
                  int64_t pos0, int32_t* slot, int32_t* sk, int32_t sk_val, void* stream);

/* -------------------------------------------------------------------------------------------
 * Skinny (decode) GEMMs — nn.Linear / F.linear with M <= 32 activation rows (models/modeling_llama.py:
 * 156-159,212-214,243,408; models/tensor_op.py:140-142,175,353-357): y = x . W^T, fp16 in, fp32 MFMA accumulate.
 * W is passed PRE-PACKED in MFMA-operand order: [N/16 panels][K/32 chunks][64 pieces][8 fp16], piece (g*16+i)
 * of a tile = W[n0+i][k0+8g .. +7]  (triforce_amd.ops.pack_weight); N % 16 == 0, K % 32 == 0.
 * tf_skinny_gemm        : y [M][ldy] fp16, or fp32 when out_f32 != 0 (the fp16 result cast to float — `logits.float()`).
 * tf_skinny_gemm_swiglu : act [M][I] = fp16(silu(fp16(x.Wg^T))) * fp16(x.Wu^T)  — gate/up GEMMs + SwiGLU fused.
 * ------------------------------------------------------------------------------------------- */
int tf_skinny_gemm(const void* w_packed, const void* x, int64_t ldx, void* y, int64_t ldy, int M, int N, int K,
                   int out_f32, void* stream);
int tf_skinny_gemm_swiglu(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx, void* act,
                          int64_t ldy, int M, int I, int K, void* stream);

/* Fused forms of the same kernel — what sits between the GEMMs of a decoder layer folded into the GEMM that consumes
 * or produces it, with the reference's rounding points (each was a 4-5 us launch of its own at <= 18 rows):
 *   ln_w != NULL : RMSNorm prologue on the input rows, h = ln_w * fp16(x * rsqrt(mean(x^2) + eps))
 *                  (models/modeling_llama.py:138-143, tensor_op.py:52-64); x is then the residual stream.
 *   resid != NULL: y = fp16(resid + fp16(x . W^T))  (hidden_states = residual + ..., modeling_llama.py:278,284);
 *                  y may alias resid (every element is read and written by the same lane).
 *   ss_out != NULL (with a fp16 output): every panel also writes sum(y[m][n]^2) over its 16 columns to
 *                  ss_out[panel * 32 + m]; ss_in != NULL (with ln_w): the norm prologue folds those N/16 partials
 *                  (of the GEMM that produced x, so K/16 of them) instead of re-reading x — the sum of squares of the
 *                  residual stream is handed from GEMM to GEMM and x is read exactly once per GEMM.
 * tf_skinny_qkv_rope: fused q|k|v projection + RoPE + KV append (modeling_llama.py:212-238 / tensor_op.py:140-160,
 *                  the work of tf_rope_append in the GEMM epilogue).  The weight must be packed in ROTARY-PAIR row
 *                  order: within the q and k sections every 16-row panel = rows d0..d0+7 and d0+D/2..d0+D/2+7 of one
 *                  head (triforce_amd.ops.rope_row_order); v rows keep their order.  q -> q_out [M][H][D]; k (rotated
 *                  unless rotate_k == 0) and v rows -> cache slot slot0 + m (slot0 read from slot0_dev if non-NULL). */
int tf_skinny_gemm_ex(const void* w_packed, const void* x, int64_t ldx, const void* ln_w, float eps,
                      const float* ss_in, const void* resid, int64_t ldr, float* ss_out, void* y, int64_t ldy,
                      int M, int N, int K, int out_f32, void* stream);
int tf_skinny_gemm_swiglu_ex(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx,
                             const void* ln_w, float eps, const float* ss_in, void* act, int64_t ldy,
                             int M, int I, int K, void* stream);
int tf_skinny_qkv_rope(const void* wqkv_packed, const void* x, int64_t ldx, const void* ln_w, float eps,
                       const float* ss_in, const void* cos, const void* sin, const int64_t* positions, void* q_out,
                       void* k_cache, void* v_cache, int64_t stride_t, int64_t stride_h, int slot0,
                       const int32_t* slot0_dev, int M, int H, int D, int K, int rotate_k, void* stream);

/* The same three kernels with an explicit ACTIVATION LAYOUT per operand (round 4).  Element (m, k) of an activation
 * block lives at base[m * s_m + (k / 8) * s_k + (k % 8)]  (element strides, multiples of 8):
 *   row-major [M][ld] (the reference's tensors)            : s_m = ld, s_k = 8        — what the entry points above pass;
 *   k-octet-major, R >= M rows (triforce_amd.ops.ActBlock) : s_m = 8,  s_k = 8 * R    — the 16-byte pieces of one
 *     k-octet of all rows are contiguous, so the B operand of a 16-row MFMA tile is 4 runs of 256 B instead of 16 row
 *     fragments of 64 B.  The decode layer keeps its residual stream, attention output and SwiGLU output in this form
 *     when a block has more than 16 rows (the gamma = 16 verifies of offloading_TP.py: tensor_op.py:276-328,346-360).
 * The fp32 logits of out_f32 != 0 stay row-major (ys_m = row stride, ys_k ignored); q_out of the q|k|v form stays
 * [M][H][D].  tf_sg_tune: A/B knobs of the launch rule (key 0: rows from which a wave multiplies TWO weight panels
 * against one B operand, 33 = never; key 1: waves per workgroup of that form, 4 or 8; key 2: smallest halved grid,
 * panels / 2, that takes it); returns the previous value, -1 for an unknown key. */
int tf_skinny_gemm_act(const void* w_packed, const void* x, int64_t xs_m, int64_t xs_k, const void* ln_w, float eps,
                       const float* ss_in, const void* resid, int64_t rs_m, int64_t rs_k, float* ss_out, void* y,
                       int64_t ys_m, int64_t ys_k, int M, int N, int K, int out_f32, void* stream);
int tf_skinny_gemm_swiglu_act(const void* gate_packed, const void* up_packed, const void* x, int64_t xs_m,
                              int64_t xs_k, const void* ln_w, float eps, const float* ss_in, void* act, int64_t ys_m,
                              int64_t ys_k, int M, int I, int K, void* stream);
int tf_skinny_qkv_rope_act(const void* wqkv_packed, const void* x, int64_t xs_m, int64_t xs_k, const void* ln_w,
                           float eps, const float* ss_in, const void* cos, const void* sin, const int64_t* positions,
                           void* q_out, void* k_cache, void* v_cache, int64_t stride_t, int64_t stride_h, int slot0,
                           const int32_t* slot0_dev, int M, int H, int D, int K, int rotate_k, void* stream);
int tf_sg_tune(int key, int value);
/* NARROW-PANEL forms of the two norm GEMMs (round 5; csrc/gemv.hip skinny_gemm_n8_kernel) for the FEW-PANEL shards of a
 * tensor-parallel rank (q|k|v: 3 * H/W * D rows, gate|up: I/W rows — 86-120 16-row panels at 8 ranks, on 256 CUs;
 * models/tensor_op.py:140-142,353-357 on the shards of models/TP_layers.py:126-147): a workgroup owns 8 output rows, two
 * 32-wide k-chunks stacked on the MFMA's 16 A rows and the matching x chunks on its 16 B columns, the two diagonal 8 x 8
 * blocks summed — twice the workgroups, half the bytes each, no cross-workgroup hand-off.  Weights packed by
 * triforce_amd.ops.pack_weight_n8 ([N/8][K/64][4][2][8][8]; q|k|v in triforce_amd.ops.rope_row_order_n8: every 8-row
 * panel of the q and k sections = rows d0..d0+3 and d0+D/2..d0+D/2+3 of one head).  Same operands, layouts and rounding
 * points as tf_skinny_gemm_swiglu_act / tf_skinny_qkv_rope_act; requires ln_w (norm prologue), M <= 24, K % 64 == 0,
 * K >= 1024 — otherwise -EINVAL and the caller keeps the 16-row form.  Results agree with the 16-row form to fp32
 * summation order (the K sum is taken even | odd chunk first), not bit for bit.  tf_sg_tune key 7: super-chunks per batch
 * (0 = rule, 5, 8). */
int tf_skinny_gemm_swiglu_n8(const void* gate_n8, const void* up_n8, const void* x, int64_t xs_m, int64_t xs_k,
                             const void* ln_w, float eps, const float* ss_in, void* act, int64_t ys_m, int64_t ys_k,
                             int M, int I, int K, void* stream);
int tf_skinny_qkv_rope_n8(const void* wqkv_n8, const void* x, int64_t xs_m, int64_t xs_k, const void* ln_w, float eps,
                          const float* ss_in, const void* cos, const void* sin, const int64_t* positions, void* q_out,
                          void* k_cache, void* v_cache, int64_t stride_t, int64_t stride_h, int slot0,
                          const int32_t* slot0_dev, int M, int H, int D, int K, int rotate_k, void* stream);
/* Split-K workspace of the CURRENT device (csrc/gemv.hip, SgKsplit): GEMMs with few output panels — the q|k|v and
 * gate|up shards of a tensor-parallel rank — split K across up to 4 workgroups per panel; their partial sums meet in
 * `ws` (zero-filled device memory, first 16 KiB = per-panel tickets, left zero by every launch; 8 MiB covers every shape
 * the rule splits), summed in split order by the last workgroup to arrive.  Without a registered workspace no GEMM is
 * split.  The workspace is cut into 4 equal slots, one per LAUNCH STREAM (first come, first served): split GEMMs on different
 * streams of one device may run concurrently; a fifth stream's GEMMs run unsplit.  ws = NULL removes it.  (tf_sg_tune key
 * 3: panel-group count below which the split applies, 0 = never.) */
int tf_sg_workspace(void* ws, int64_t bytes);

/* -------------------------------------------------------------------------------------------
 * Sampling / accept-rollback (utils/sampling.py:63-75, utils/decoding.py:97-134,190-220).
 * tf_sample_inverse_cdf: token = first index whose inclusive cumulative sum of probs exceeds
 *                        u * sum(probs)  (stand-in for torch.multinomial with an explicit uniform).
 * tf_accept_chain      : sequential speculative accept test of g2 drafted tokens against the
 *                        target distribution rows p[0..g2] and the drafting rows q[0..g2-1]:
 *                        flag_i = u_i < min(1, p_i[x_i]/q_i[x_i])  (<= when inclusive), count =
 *                        accepted prefix (wave ballot + ffs), stop at an accepted eos; then the
 *                        correction token: residual max_fn(p_count - q_count) on rejection, bonus
 *                        p_g2 when everything passed.  The sample uses uniforms[examined].
 *                        out[0]=count, out[1]=next token, out[2]=reason (0 rejected, 1 all
 *                        accepted, 2 eos accepted), out[3]=number of uniforms consumed.
 * tf_middle_accept     : one inner TriForce step (decoding.py:190-220): test drafted token d
 *                        (= tokens[n+1]) with p[n], q_d; sample the follow-up token from p[n+acc].
 *                        out[0]=accepted(0/1), out[1]=follow-up token.  Also writes the follow-up
 *                        token into tokens[n+1+acc] when that index is <= gamma.
 * ------------------------------------------------------------------------------------------- */
/* tf_topp_probs: probs = softmax(top_p_filter(logits / temperature)) for `rows` rows of V <= 32768 fp32 logits —
 * utils/sampling.py:43-60 (norm_logits with top_k = -1) without the vocabulary sort: threshold search on the
 * fp32 bit pattern + index-ordered tie scan (== a stable descending sort; ties -> lowest token id). */
int tf_topp_probs(const float* logits, float* probs, int rows, int V, float temperature, float top_p, void* stream);
int tf_sample_inverse_cdf(const float* probs, const float* u, int64_t* token_out, int V, void* stream);
int tf_accept_chain(const float* p, const float* q, const int64_t* tokens, const float* uniforms,
                    int g2, int V, int inclusive, int64_t eos_token_id, int64_t* out, void* stream);
int tf_middle_accept(const float* p, const float* q_d, int64_t* tokens, const float* uniforms,
                     int n, int gamma, int V, int64_t* out, void* stream);
/* tf_mid_record_tokens: tokens[n + 1] (, tokens[n + 2]) as a tf_middle_accept record (accepted, follow-up, drafted) implies —
 * the tensor-parallel loop applies RANK 0's broadcast record (utils/decoding.py:452-470) with it in one launch. */
int tf_mid_record_tokens(const int64_t* rec, int64_t* tokens, int tokens_len, int n, void* stream);
/* tf_topp_probs_multi (csrc/topp_multi.hip; round 6): tf_topp_probs with every row spread over 16 (8 from 17 rows up) workgroups of
 * ONE launch — every workgroup takes the row maximum from the whole row (tune key 3 = 0: the slices exchange their maxima through
 * an in-launch edge instead), the slices of a row meet through an in-launch edge (mass sums + compacted candidates), then every
 * workgroup of the row runs the exact-integer select over the row's candidates and writes its own entries.  Probabilities are
 * BIT-IDENTICAL to tf_topp_probs (utils/sampling.py:5-27,43-60).  rows <= 32, V % 4 == 0, 64 <= V <= 32768, rows x slices <= the
 * device's CU count (TF_ERANGE otherwise: the caller keeps tf_topp_probs).
 *   panel_max  NULL, or [V / 16][32] fp32: per-16-column-panel row maxima of the logits as the lm_head GEMM's epilogue leaves them
 *              (tf_skinny_gemm_act with out_f32 = 1 and ss_out != NULL): the first edge is skipped
 *   ctl        tf_topp_multi_ctl_bytes() bytes, 64-byte aligned, ZERO-filled once; launches on one ctl must not overlap
 *   ws         >= tf_topp_multi_ws_bytes(rows, V) bytes of device scratch, 256-byte aligned
 * Bounded waits (tf_topp_multi_tune key 0 = ms, default 2000; key 1 = fault injection for tests; key 2 = entries of the LDS
 * candidate list; key 3 = where the row maximum comes from): a time-out sets the sticky
 * error word (tf_topp_multi_error) and NaN-fills the rows until tf_topp_multi_reset. */
int64_t tf_topp_multi_ctl_bytes(void);
int64_t tf_topp_multi_ws_bytes(int rows, int V);
int tf_topp_probs_multi(const float* logits, const float* panel_max, float* probs, int rows, int V, float temperature,
                        float top_p, void* ctl, void* ws, int64_t ws_bytes, void* stream);
int tf_topp_multi_tune(int key, int value);
int tf_topp_multi_error(const void* ctl);
int tf_topp_multi_reset(void* ctl);

/* The same three kernels with their uniforms behind a DEVICE CURSOR (round 5): u_k = ubuf[*cursor + k] — the form that
 * can sit INSIDE a captured hipGraph (frozen arguments, fresh numbers at every replay), which is how the inner loop of
 * utils/decoding.py:163-223 becomes ONE graph launch per iteration (draft forward, draw, retrieval verify, accept test).
 *   tf_sample_inverse_cdf_cur: u = ubuf[*cursor + off]; the cursor is left alone.
 *   tf_middle_accept_cur     : accept test with ubuf[*cursor + 1], follow-up sample with ubuf[*cursor + 2]; *cursor += 3;
 *                              out[3] = the cursor value the decision started from (the host checks its mirror).
 *   tf_accept_chain_cur      : the chain over ubuf[*cursor ...]; *cursor += out[3].
 * In ALL forms (cursor or not) what the kernel leaves for the next chain of launches on the device — the token id in
 * `tokens`, the cursor — is stored write-through (agent scope) and drained BEFORE the record `out` is stored.  CONTRACT: a
 * host that polls `out` in pinned memory may enqueue the next chain the moment it sees the record ONLY ON THE SAME STREAM
 * (stream order then also orders the chain behind this kernel's completion).  A chain on another stream is ordered by the
 * record alone, which the runtime knows nothing about: its first kernel would need an explicit agent-scope acquire of what it
 * reads, and none of the entry points does that (the two-stream loop of round 5 was removed for this reason). */
int tf_sample_inverse_cdf_cur(const float* probs, const float* ubuf, const int64_t* cursor, int off, int64_t* token_out,
                              int V, void* stream);
int tf_middle_accept_cur(const float* p, const float* q_d, int64_t* tokens, int tokens_len, const float* ubuf,
                         int64_t* cursor, int n, int gamma, int V, int64_t* out, void* stream);
int tf_accept_chain_cur(const float* p, const float* q, const int64_t* tokens, const float* ubuf, int64_t* cursor, int g2,
                        int V, int inclusive, int64_t eos_token_id, int64_t* out, void* stream);
/* tf_middle_accept_cur: a `tokens` buffer of >= gamma + 2 entries (the engine's shared token buffer) also receives the
 *   follow-up token of the LAST position (index gamma + 1), so that after the inner loop it holds all of [next, t_1 .. t_g2].
 * tf_accept_chain_step: tf_accept_chain_cur over tok_buf[1 ..] that also leaves on the device what follows its record
 *   (utils/decoding.py:124,137): tok_buf[0 .. g2 + 1] = the PASS TOKENS of the catch-up draft forward ([next, accepted ...,
 *   resampled | bonus token, pad ...]; pad in its place after an accepted eos), and for nsets <= 2 captured target-verify lengths
 *   the positions (n_pos entries from S'), append slot (S') and key count (S' + qlen) of the NEXT verify, S' = *s_src + count + 1
 *   = the cache length after the roll-back (s_src: the append slot of the verify just run).  Neither the catch-up draft nor the
 *   next target verify then needs a set-up launch behind a record read.  tok_len >= g2 + 2; g2 <= 62. */
int tf_accept_chain_step(const float* p, const float* q, int64_t* tok_buf, int tok_len, const float* ubuf, int64_t* cursor,
                         int g2, int V, int inclusive, int64_t eos_token_id, int64_t pad, const int32_t* s_src, int nsets,
                         int64_t* pos_a, int n_pos_a, int32_t* slot_a, int32_t* sk_a, int qlen_a, int64_t* pos_b, int n_pos_b,
                         int32_t* slot_b, int32_t* sk_b, int qlen_b, int64_t* out, void* stream);

/* -------------------------------------------------------------------------------------------
 * Sequoia tree verification (utils/SpecTree_TP.py:147-199: accept_step + the walk in verify()).
 *   p_rows        [N][V] fp32 target probabilities of every tree node (top-p filtered softmax)
 *   draft_logits  [N][V] fp32 draft logits of every node (q = softmax(draft_logits / temperature), with the
 *                 rejected sibling tokens removed, as :159-166)
 *   tokens        [N] int64 token of every node;  succ_off [N+1], succ: CSR of grow_map["Successors"]
 *   uniforms      fp32 stream: one per examined child, then one for the residual sample
 *   out int64 [4 + 60]: out[0] = len(accept_list) incl. the root, out[1] = next token, out[2] = terminal
 *                 (accepted token 0 / 2, or a NaN residual), out[3] = uniforms consumed, out[4+j] = accept_list[j]
 * One single-workgroup launch replaces the reference's host loop (one device sync per examined child).  V <= 32768.
 * ------------------------------------------------------------------------------------------- */
int tf_tree_accept(const float* p_rows, const float* draft_logits, const int64_t* tokens,
                   const int32_t* succ_off, const int32_t* succ, const float* uniforms, int V,
                   float temperature, int64_t* out, void* stream);

/* Sampling WITHOUT replacement for the tree growth (test/offloading_seqouia.py:29-39):
 * out[row][0..k) = the k token ids with the largest log(rand[row][i]) / softmax(logits[row] / temperature)[i], in
 * descending order (= `(rand.log() / q).topk(k).indices`); rand is fp16, logits fp32 [rows][V]; k <= 16, V <= 32768.
 * One kernel instead of softmax + log + div + multi-block top-k (which also hangs when replayed from a hipGraph). */
int tf_sample_without_replacement(const float* logits, const void* rand_f16, int64_t* out, int rows, int V, int k,
                                  float temperature, void* stream);

/* -------------------------------------------------------------------------------------------
 * Offloading tier (models/cache.py:345-351 copy_back_from_buffer, :372-376 copy_kv, :573-575):
 * asynchronous pinned-host <-> device copies of a head-major KV block = H rows of width_elems
 * contiguous fp16 (n tokens x D) at the given pitches (in elements).  Enqueued on `copy_stream`
 * (hipMemcpy2DAsync); the caller orders them against compute with events — no device-wide syncs.
 * *_host pointers are HOST pointers to pinned memory.
 * ------------------------------------------------------------------------------------------- */
int tf_kv_h2d_async(void* dst_dev, int64_t dst_pitch_elems, const void* src_host, int64_t src_pitch_elems,
                    int64_t width_elems, int H, void* copy_stream);
int tf_kv_d2h_async(void* dst_host, int64_t dst_pitch_elems, const void* src_dev, int64_t src_pitch_elems,
                    int64_t width_elems, int H, void* copy_stream);

/* -------------------------------------------------------------------------------------------
 * One-shot all-reduce over peer-mapped buffers (xGMI) — replaces dist.all_reduce at models/tensor_op.py:179,326,359
 * for the 57..184 KB decode messages of the tensor-parallel path (2 per layer and forward; a ring collective pays
 * 2(W-1) latency hops on each).  Every rank stages its fp16 partial in a FINE-GRAINED buffer its peers have mapped
 * (tf_ar_alloc + tf_ar_get_ipc_handle / tf_ar_open_ipc_handle), exchanges one READY flag, reads all partials and adds
 * them in rank order with fp32 accumulation (bit-identical on every rank), then exchanges one DONE flag so the staging
 * buffer can be reused when the kernel ends.  The epoch lives in the control block: capturable in a hipGraph.  All
 * spins are bounded (tf_ar_error != 0 afterwards = a peer never arrived; from then on every call fills `out` with NaN
 * and returns at once — the host polls tf_ar_error once per decode step).
 *   peer_data / peer_flags : `world` device-visible pointers (own entry included); this rank's partial must already be
 *                            in peer_data[rank], written by an earlier kernel of `stream`; n % 8 == 0 fp16 elements;
 *                            `out` is ordinary device memory, not the staging buffer.
 * ------------------------------------------------------------------------------------------- */
int tf_ar_flags_bytes(void);
int tf_ar_ipc_handle_bytes(void);
int tf_ar_alloc(int64_t bytes, void** ptr);
int tf_ar_free(void* ptr);
int tf_ar_get_ipc_handle(void* ptr, void* handle_out);
int tf_ar_open_ipc_handle(const void* handle, void** ptr_out);
int tf_ar_close_ipc_handle(void* ptr);
int tf_allreduce_oneshot(void* const* peer_data, void* const* peer_flags, int rank, int world, void* out, int64_t n,
                         void* stream);
/* out = resid + all_reduce(partials): `hidden_states = residual + all_reduce(...)` (tensor_op.py:179-181,359-360) in the
 * same launch; the residual is added in fp16 to the ROUNDED sum (the reference's two rounding points); resid may be
 * `out` itself (in-place residual stream), NULL = plain all-reduce. */
int tf_allreduce_oneshot_add(void* const* peer_data, void* const* peer_flags, int rank, int world, const void* resid,
                             void* out, int64_t n, void* stream);
/* ... plus the per-panel sums of squares of the result rows (out = [n / hidden][hidden], <= 32 rows):
 * ss_out[panel * 32 + row], the hand-off tf_skinny_gemm_ex folds as ss_in — the RMSNorm after an all-reduce
 * (tensor_op.py:52-64 after :179 / :359) then runs in the prologue of the next GEMM without re-reading the rows. */
int tf_allreduce_oneshot_add_ss(void* const* peer_data, void* const* peer_flags, int rank, int world, const void* resid,
                                void* out, int64_t n, int hidden, float* ss_out, void* stream);
/* Alternating form: every staging buffer holds 2 * half_elems values and exchange e (the control block's epoch, 1 for
 * the first exchange after tf_ar_alloc) uses the half e & 1, which removes the closing DONE handshake: a rank cannot
 * overwrite a half before every peer has signalled READY for the exchange in between (csrc/allreduce.hip header).
 * `expect_half` = the half this rank's producer wrote; if the epoch selects the other one the call fills `out` with NaN
 * and sets the sticky error 3.  resid and ss_out may be NULL (hidden is read only with ss_out).  One form per control
 * block for its lifetime, the same on every rank. */
int tf_allreduce_oneshot_alt(void* const* peer_data, void* const* peer_flags, int rank, int world, const void* resid,
                             void* out, int64_t n, int hidden, float* ss_out, int64_t half_elems, int expect_half,
                             void* stream);
/* tf_allreduce_oneshot_add_ss / _alt for a residual stream kept k-octet-major (tf_skinny_gemm_act): n = pack_rows *
 * hidden elements, piece i = (k-octet i / pack_rows, row i % pack_rows); ss_out[panel * 32 + row] as above (required).
 * half_elems == 0: the READY / reduce / DONE form; > 0: alternating halves with expect_half. */
int tf_allreduce_oneshot_act(void* const* peer_data, void* const* peer_flags, int rank, int world, const void* resid,
                             void* out, int64_t n, int hidden, int pack_rows, float* ss_out, int64_t half_elems,
                             int expect_half, void* stream);
/* GEMM + all-reduce in ONE launch (csrc/gemv.hip, SgXchg) — o_proj / down_proj of the tensor-parallel decode layer with
 * their exchange in the GEMM epilogue: out = resid + sum over ranks of fp16(x . W^T)  (models/tensor_op.py:175-181,
 * 353-360; the arithmetic of tf_skinny_gemm_act(out = staging) + tf_allreduce_oneshot_add_ss).  Every workgroup publishes
 * its 16-column panel in this rank's staging half (half = device epoch & 1), flags the same panel on every peer, waits for
 * the peers' flags of that panel, reads their panels and finishes — no grid-wide wait, no DONE phase, no host-side half.
 * peer_stage[r] / peer_ctl[r], r < world: every rank's staging buffer (2 * half_elems fp16) and control buffer
 * (tf_xchg_ctl_bytes() bytes, zero-filled once), fine-grained memory (tf_ar_alloc), own entries included, peers' mapped
 * through hipIpc (tf_ar_get_ipc_handle / tf_ar_open_ipc_handle).  The result block `out` (may alias resid) and the
 * staging halves share the activation layout (os_m, os_k); N / 16 <= 512; the block must fit a half.  ss_out: per-panel
 * sums of squares of the result rows (may be NULL).  Bounded spins: a time-out NaN-fills the panel and sets the sticky
 * error word (tf_xchg_error; tf_xchg_set_error(ctl, code, host_mirror, set_mirror) injects / clears it and registers a
 * pinned host word that mirrors it). */
int64_t tf_xchg_ctl_bytes(void);
int tf_skinny_gemm_xchg(const void* w_packed, const void* x, int64_t xs_m, int64_t xs_k, void* const* peer_stage,
                        void* const* peer_ctl, int rank, int world, int64_t half_elems, const void* resid, int64_t rs_m,
                        int64_t rs_k, void* out, int64_t os_m, int64_t os_k, float* ss_out, int M, int N, int K,
                        void* stream);
int tf_xchg_error(const void* ctl);
int tf_xchg_set_error(void* ctl, int code, void* host_mirror, int set_mirror);
/* tf_xchg_tune: key 0 = FENCED form of the exchange (1: system-scope release before the flag stores and acquire behind the
 *   flag wait, the pair the first build had, ~1.7 us each; 0, default: s_waitcnt vmcnt(0) + system-scope accesses in issue
 *   order) — selected by the engine's start-up litmus on the real group (utils/oneshot_ar.GemmExchange.litmus) when the
 *   fence-free form shows a single mismatch, or forced with TRIFORCE_XCHG_FENCE=1; key 1 = wall-clock limit of one flag wait
 *   in milliseconds (default 5000: a dead or starved peer is reported in seconds, not after 2^27 polls).  Returns the
 *   previous value (a value outside the key's range only reads), -1 for an unknown key.
 * tf_xchg_reset: control block back to its freshly allocated state (peer flags, per-panel exchange counts, error word; the
 *   host mirror stays registered).  COLLECTIVE by contract: all ranks, between two barriers, nothing in flight. */
int tf_xchg_tune(int key, int value);
int tf_xchg_reset(void* ctl);
/* Message-passing litmus around the exchange (tools/xgmi_litmus.py): tf_ar_litmus_stage advances the device counter
 * *it_dev and then writes, with PLAIN stores from an ordinary kernel (the role of the o_proj / down_proj epilogue), the
 * pattern v_rank[i] = (7 i + 13 it + 101 rank) mod 509 into `staging` (n fp16 values); after an all-reduce of those
 * partials tf_ar_litmus_check adds to *bad (uint64, device) the number of elements of `out` that differ from
 * fp16(sum over ranks of v_r[i]) for the same iteration.  Both are capturable. */
int tf_ar_litmus_stage(void* staging, int64_t n, int rank, uint32_t* it_dev, void* stream);
int tf_ar_litmus_check(const void* out, int64_t n, int world, const uint32_t* it_dev, uint64_t* bad, void* stream);
int tf_ar_error(const void* flags_local);
/* completed exchanges of this control block (device-side epoch), blocking host read: lets a caller that counts its
 * exchanges (expect_half) resynchronise after a failed launch / capture; TF_EINVAL for NULL, -(hip error) on failure */
int64_t tf_ar_epoch(const void* flags_local);
/* host_word: pinned, device-mapped host memory (4 bytes) that receives every error code the kernels set from now on (and
 * the current one at once) — the per-step health poll then reads host memory instead of copying the control block;
 * NULL removes the mirror.  The word must outlive the control block. */
int tf_ar_set_error_mirror(void* flags_local, void* host_word);
/* Fault injection for tests: sets (code > 0) or clears (0) the sticky error word of a control block from the host. */
int tf_ar_inject_error(void* flags_local, int code);

#ifdef __cplusplus
}
#endif
#endif /* TRIFORCE_HIP_H */
