// tf_draft_forward_68m: one decode call of the Llama-68M draft model as ONE C entry point
// (models/modeling_llama_68m.py:129-190 forward + utils/graph_infer.py:52-57 draft_run; utils/sampling.py:43-60 for the
// optional probability row).  The host side is native: the entry point issues the whole launch chain on the given
// stream — embedding gather + positions, per layer {RMSNorm + q|k|v + RoPE(q) + KV append, rope-on-read attention,
// o_proj + residual, RMSNorm + gate|up + SwiGLU, down_proj + residual}, final RMSNorm + lm_head, top-p — through the
// same kernels as the individual entry points, so its output is bit-identical to calling them one by one, and the whole
// call is graph-capturable (no allocation, no synchronisation; intermediates live in the caller's workspace).
#include <string.h>

#include "common.h"

// ids -> residual stream rows and their positions (replaces an index-select and an arange launch)
__global__ __launch_bounds__(256) void draft_embed_kernel(const h16* __restrict__ embed, const int64_t* __restrict__ ids,
                                                          h16* __restrict__ x, int64_t* __restrict__ pos, int n,
                                                          int hidden, int vocab, int pos0) {
    const int row = blockIdx.x;
    int64_t id = ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const half8* src = reinterpret_cast<const half8*>(embed + id * hidden);
    half8* dst = reinterpret_cast<half8*>(x + (int64_t)row * hidden);
    for (int i = threadIdx.x; i < hidden / 8; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) pos[row] = pos0 + row;
}

static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

extern "C" int64_t tf_draft_forward_ws_bytes(const TfDraftModel* m, int n) {
    if (!m || n < 1) return 0;
    const int64_t hid = m->hidden, I = m->inter;
    return align256((int64_t)n * hid * 2)            // x    residual stream
         + align256((int64_t)n * 8)                  // pos
         + align256((int64_t)n * hid * 2)            // q    [n][H][D]
         + align256((int64_t)n * hid * 2)            // a    attention output
         + align256((int64_t)n * I * 2)              // act  SwiGLU output
         + align256((hid / 16) * 32 * 4);            // ss   sum-of-squares hand-off
}

extern "C" int tf_draft_forward_68m(const TfDraftModel* m, const TfDraftCache* c, const int64_t* ids, int n, int slot0,
                                    int kv_len, float* logits_out, float* probs_out, float temperature, float top_p,
                                    void* ws, int64_t ws_bytes, void* stream) {
    if (!m || !c || !ids || !logits_out || !ws) return TF_EINVAL;
    if (n < 1 || n > 32 || m->layers < 1 || m->layers > TF_DRAFT_MAX_LAYERS) return TF_EINVAL;
    if (m->head_dim != 64 || m->heads * m->head_dim != m->hidden || m->hidden % 32 || m->inter % 32 || m->vocab % 16)
        return TF_EINVAL;
    if (slot0 < 0 || kv_len < slot0 + n) return TF_EINVAL;
    if (ws_bytes < tf_draft_forward_ws_bytes(m, n)) return TF_ENOSPC;
    const int hid = m->hidden, H = m->heads, D = m->head_dim, I = m->inter, V = m->vocab;
    char* p = static_cast<char*>(ws);
    h16* x = reinterpret_cast<h16*>(p);             p += align256((int64_t)n * hid * 2);
    int64_t* pos = reinterpret_cast<int64_t*>(p);   p += align256((int64_t)n * 8);
    h16* q = reinterpret_cast<h16*>(p);             p += align256((int64_t)n * hid * 2);
    h16* a = reinterpret_cast<h16*>(p);             p += align256((int64_t)n * hid * 2);
    h16* act = reinterpret_cast<h16*>(p);           p += align256((int64_t)n * I * 2);
    float* ss = reinterpret_cast<float*>(p);
    hipStream_t st = (hipStream_t)stream;

    hipLaunchKernelGGL(draft_embed_kernel, dim3(n), dim3(96), 0, st, (const h16*)m->embed, ids, x, pos, n, hid, V, slot0);
    TF_LAUNCH_CHECK();
    int rc;
    for (int l = 0; l < m->layers; ++l) {
        // keys are cached UN-rotated (rotate_k = 0) and rotated on read with cache-relative positions (68m.py:151-178)
        rc = tf_skinny_qkv_rope(m->wqkv[l], x, hid, m->ln1[l], m->eps, l > 0 ? ss : nullptr, m->cos, m->sin, pos, q,
                                c->k[l], c->v[l], c->stride_t, c->stride_h, slot0, nullptr, n, H, D, hid, 0, stream);
        if (rc) return rc;
        rc = tf_attn_rope_on_read(q, c->k[l], c->v[l], m->cos, m->sin, a, c->stride_t, c->stride_h, n, kv_len, H, D,
                                  m->scale, stream);
        if (rc) return rc;
        rc = tf_skinny_gemm_ex(m->wo[l], a, hid, nullptr, 0.f, nullptr, x, hid, ss, x, hid, n, hid, hid, 0, stream);
        if (rc) return rc;
        rc = tf_skinny_gemm_swiglu_ex(m->wgate[l], m->wup[l], x, hid, m->ln2[l], m->eps, ss, act, I, n, I, hid, stream);
        if (rc) return rc;
        rc = tf_skinny_gemm_ex(m->wdown[l], act, I, nullptr, 0.f, nullptr, x, hid, ss, x, hid, n, hid, I, 0, stream);
        if (rc) return rc;
    }
    rc = tf_skinny_gemm_ex(m->lm_head, x, hid, m->norm, m->eps, ss, nullptr, 0, nullptr, logits_out, V, n, V, hid, 1,
                           stream);
    if (rc) return rc;
    if (probs_out)                                   // only the last row is sampled from (graph_infer.py:57)
        return tf_topp_probs(logits_out + (int64_t)(n - 1) * V, probs_out, 1, V, temperature, top_p, stream);
    return TF_OK;
}
