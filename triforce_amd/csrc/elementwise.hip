// Dense-block glue kernels for gfx950: RMSNorm (+fused residual add), RoPE + KV append, SwiGLU.
// Rounding points follow the reference exactly:
//   RMSNorm  models/modeling_llama.py:138-143 == tensor_op.py:52-64  (fp32 normalise -> fp16 -> * weight)
//   RoPE     models/tensor_op.py:25-50, modeling_llama_68m.py:30-38   ((x*cos) + (rotate_half(x)*sin) in fp16)
//   SwiGLU   models/modeling_llama.py:156-159                         (fp16 silu, fp16 product)
// All are memory-bound one-pass kernels with 16-byte accesses.
#include "common.h"

// One workgroup per row.  hidden % 8 == 0.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const h16* __restrict__ x, const h16* __restrict__ res,
                                                      const h16* __restrict__ w, h16* __restrict__ y,
                                                      h16* __restrict__ sum_out, int hidden, float eps) {
    extern __shared__ float red[];                  // [4] partial sums + row cache of fp32 values is not needed
    const int row = blockIdx.x, tid = threadIdx.x;
    const h16* xr = x + (int64_t)row * hidden;
    const h16* rr = res ? res + (int64_t)row * hidden : nullptr;
    h16* so = sum_out ? sum_out + (int64_t)row * hidden : nullptr;
    const int nvec = hidden / 8;
    float ss = 0.f;
    for (int i = tid; i < nvec; i += 256) {
        half8 v = load_half8(xr + 8 * i);
        if (rr) {
            const half8 r = load_half8(rr + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = hadd_rn(v[e], r[e]);       // residual + hidden in fp16
            if (so) store_half8(so + 8 * i, v);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            ss = fmaf(f, f, ss);
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sqrtf(tot / (float)hidden + eps);
    const h16* src = (rr && so) ? so : xr;
    for (int i = tid; i < nvec; i += 256) {
        half8 v = load_half8(src + 8 * i);
        if (rr && !so) {
            const half8 r = load_half8(rr + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = hadd_rn(v[e], r[e]);
        }
        const half8 wv = load_half8(w + 8 * i);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const h16 n = (h16)((float)v[e] * inv);                        // cast BEFORE the weight multiply
            o[e] = hmul_rn(wv[e], n);
        }
        store_half8(y + (int64_t)row * hidden + 8 * i, o);
    }
}

// grid (rows, H), block D/2 threads: thread d handles the pair (d, d + D/2) of q and k, and two v's.
__global__ void rope_append_kernel(const h16* __restrict__ qkv, int64_t row_stride, const h16* __restrict__ cosb,
                                   const h16* __restrict__ sinb, const int64_t* __restrict__ positions,
                                   h16* __restrict__ q_out, h16* __restrict__ k_cache, h16* __restrict__ v_cache,
                                   int64_t stride_t, int64_t stride_h, int slot0, const int32_t* __restrict__ slot0_dev,
                                   int H, int D, int rotate_k) {
    const int row = blockIdx.x, h = blockIdx.y, d = threadIdx.x, half = D >> 1;
    const int64_t pos = positions[row];
    const int slot = (slot0_dev ? *slot0_dev : slot0) + row;
    const h16* base = qkv + (int64_t)row * row_stride + (int64_t)h * D;
    const h16* qp = base;
    const h16* kp = base + (int64_t)H * D;
    const h16* vp = base + (int64_t)2 * H * D;
    const h16 c1 = cosb[pos * D + d], c2 = cosb[pos * D + d + half];
    const h16 s1 = sinb[pos * D + d], s2 = sinb[pos * D + d + half];
    {
        const h16 x1 = qp[d], x2 = qp[d + half];
        h16* qo = q_out + ((int64_t)row * H + h) * D;
        qo[d] = hadd_rn(hmul_rn(x1, c1), hmul_rn((h16)(-(float)x2), s1));
        qo[d + half] = hadd_rn(hmul_rn(x2, c2), hmul_rn(x1, s2));
    }
    h16* kc = k_cache + (int64_t)slot * stride_t + (int64_t)h * stride_h;
    h16* vc = v_cache + (int64_t)slot * stride_t + (int64_t)h * stride_h;
    {
        const h16 x1 = kp[d], x2 = kp[d + half];
        if (rotate_k) {
            kc[d] = hadd_rn(hmul_rn(x1, c1), hmul_rn((h16)(-(float)x2), s1));
            kc[d + half] = hadd_rn(hmul_rn(x2, c2), hmul_rn(x1, s2));
        } else {
            kc[d] = x1;
            kc[d + half] = x2;
        }
    }
    vc[d] = vp[d];
    vc[d + half] = vp[d + half];
}

__global__ __launch_bounds__(256) void silu_mul_kernel(const h16* __restrict__ gu, h16* __restrict__ out, int I,
                                                       int64_t total_vec) {
    const int vpr = I / 8;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_vec; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / vpr;
        const int c = (int)(e - row * vpr);
        const half8 g = load_half8(gu + row * 2 * I + 8 * c);
        const half8 u = load_half8(gu + row * 2 * I + I + 8 * c);
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gf = (float)g[i];
            const h16 s = (h16)(gf / (1.0f + expf(-gf)));               // silu rounded to fp16
            o[i] = hmul_rn(s, u[i]);
        }
        store_half8(out + row * I + 8 * c, o);
    }
}

extern "C" int tf_rmsnorm(const void* x, const void* res, const void* w, void* y, void* sum_out, int rows,
                          int hidden, float eps, void* stream) {
    if (!x || !w || !y || rows < 1 || hidden < 8 || (hidden % 8)) return TF_EINVAL;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 4 * sizeof(float), (hipStream_t)stream, (const h16*)x,
                       (const h16*)res, (const h16*)w, (h16*)y, (h16*)sum_out, hidden, eps);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_rope_append(const void* qkv, int64_t qkv_row_stride, const void* cosb, const void* sinb,
                              const int64_t* positions, void* q_out, void* k_cache, void* v_cache, int64_t stride_t,
                              int64_t stride_h, int slot0, const int32_t* slot0_dev, int rows, int H, int D,
                              int rotate_k, void* stream) {
    if (!qkv || !cosb || !sinb || !positions || !q_out || !k_cache || !v_cache) return TF_EINVAL;
    if (rows < 1 || H < 1 || D < 2 || (D & 1) || D > 2048) return TF_EINVAL;
    hipLaunchKernelGGL(rope_append_kernel, dim3(rows, H), dim3(D / 2), 0, (hipStream_t)stream, (const h16*)qkv,
                       qkv_row_stride, (const h16*)cosb, (const h16*)sinb, positions, (h16*)q_out, (h16*)k_cache,
                       (h16*)v_cache, stride_t, stride_h, slot0, slot0_dev, H, D, rotate_k);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_silu_mul(const void* gate_up, void* out, int rows, int I, void* stream) {
    if (!gate_up || !out || rows < 1 || I < 8 || (I % 8)) return TF_EINVAL;
    const int64_t total = (int64_t)rows * (I / 8);
    int64_t gx = (total + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const h16*)gate_up,
                       (h16*)out, I, total);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// ---- embedding rows -> activation block in an explicit layout (tf_skinny_gemm_act) -------------------------------------
// x = embed_tokens(input_ids) (models/modeling_llama.py:342, TP_llama.py:206) for the <= 32 rows of a decode forward,
// written straight in the layout the decode layer's GEMMs read: element (m, k) at out[m * sm + (k / 8) * sk + k % 8].
// Thread p handles the 16-byte piece (m = p % n, k8 = p / n): in the k-octet-major form consecutive threads write
// consecutive pieces.  Out-of-range ids are clamped (the reference's index op would fault).
__global__ __launch_bounds__(256) void embed_rows_kernel(const h16* __restrict__ embed, const int64_t* __restrict__ ids,
                                                         h16* __restrict__ out, int64_t sm, int64_t sk, int n, int hidden,
                                                         int vocab) {
    const int64_t total = (int64_t)n * (hidden / 8);
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
        const int m = (int)(p % n);
        const int64_t k8 = p / n;
        int64_t id = ids[m];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        store_half8(out + (int64_t)m * sm + k8 * sk, load_half8(embed + id * hidden + 8 * k8));
    }
}

extern "C" int tf_embed_rows(const void* embed, const int64_t* ids, void* out, int64_t out_sm, int64_t out_sk, int n,
                             int hidden, int vocab, void* stream) {
    if (!embed || !ids || !out || n < 1 || n > 32 || hidden < 8 || (hidden % 8) || vocab < 1) return TF_EINVAL;
    if (out_sm < 8 || out_sk < 8 || (out_sm % 8) || (out_sk % 8)) return TF_EINVAL;
    const int64_t total = (int64_t)n * (hidden / 8);
    int64_t gx = (total + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const h16*)embed, ids,
                       (h16*)out, out_sm, out_sk, n, hidden, vocab);
    TF_LAUNCH_CHECK();
    return TF_OK;
}

// ---- token / position / length scalars of a decode forward in ONE launch -----------------------------------------------
// The decode loop knows a step's token ids on the host (it has just read the accept record) and used to ship them as
// tensor -> pinned staging -> device row -> static graph input, then set the positions and the two length scalars of the
// captured forward with three more launches: six ~10 us host-bound launches in front of every target verify
// (profiles/r04_gap_analysis_decode_steps.txt).  Here the ids travel as kernel arguments.
struct TokArgs { int64_t v[32]; };
__global__ __launch_bounds__(64) void set_tokens_kernel(int64_t* __restrict__ dst, int n_dst, TokArgs vals, int n_vals,
                                                        int64_t pad, int64_t* __restrict__ pos, int n_pos, int64_t pos0,
                                                        int32_t* __restrict__ slot, int32_t* __restrict__ sk, int32_t sk_val) {
    const int i = threadIdx.x;
    if (i < n_dst) dst[i] = i < n_vals ? vals.v[i] : pad;
    if (pos && i < n_pos) pos[i] = pos0 + i;
    if (i == 0) {
        if (slot) slot[0] = (int32_t)pos0;
        if (sk) sk[0] = sk_val;
    }
}

extern "C" int tf_set_tokens(int64_t* dst, int n_dst, const int64_t* host_vals, int n_vals, int64_t pad, int64_t* pos,
                             int n_pos, int64_t pos0, int32_t* slot, int32_t* sk, int32_t sk_val, void* stream) {
    if (n_dst < 0 || n_dst > 32 || n_vals < 0 || n_vals > 32 || n_pos < 0 || n_pos > 64) return TF_EINVAL;
    if ((n_dst > 0 && !dst) || (n_vals > 0 && !host_vals) || (n_pos > 0 && !pos)) return TF_EINVAL;
    if (n_dst == 0 && n_pos == 0 && !slot && !sk) return TF_OK;
    TokArgs a;
    for (int i = 0; i < 32; ++i) a.v[i] = i < n_vals ? host_vals[i] : pad;
    hipLaunchKernelGGL(set_tokens_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, n_dst, a, n_vals, pad,
                       n_pos > 0 ? pos : nullptr, n_pos, pos0, slot, sk, sk_val);
    TF_LAUNCH_CHECK();
    return TF_OK;
}
