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
#ifndef TF_SG_RES_EARLY
#define TF_SG_RES_EARLY 1   // residual epilogue operands fetched before the weight loop (0: at the tail, round 2's form)
#endif

enum { SG_PLAIN = 0, SG_GATEUP = 1, SG_F32 = 2, SG_QKV = 3 };

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

template <int MT, int MODE, bool NORM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(const half8* __restrict__ wp,
                                                                 const half8* __restrict__ wp_up,
                                                                 const h16* __restrict__ x, int64_t ldx,
                                                                 const h16* __restrict__ ln_w, float eps,
                                                                 const h16* resid, int64_t ldr, void* yv, int64_t ldy,
                                                                 int M, int N, int K, SgRope rp,
                                                                 const float* __restrict__ ss_in,
                                                                 float* __restrict__ ss_out) {
    constexpr bool GATEUP = MODE == SG_GATEUP;
    const int panel = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int nchunks = K >> 5;
    const int cpw = (nchunks + WAVES - 1) / WAVES;
    const int c0 = wave * cpw, c1 = min(nchunks, c0 + cpw);

    __shared__ float sm[WAVES][GATEUP ? 2 : 1][MT][64][4];
    __shared__ float sm_ss[NORM ? WAVES : 1][MT][16];

    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const half8* wa = wp + ((int64_t)panel * nchunks) * 64 + lane;
    const half8* wu = GATEUP ? (wp_up + ((int64_t)panel * nchunks) * 64 + lane) : nullptr;
    const h16* xr[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + li;
        xok[t] = m < M;
        xr[t] = x + (int64_t)(xok[t] ? m : 0) * ldx + 8 * g;
    }
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int U = SG_U;

    // first weight chunks of this wave: issued before the norm prologue so HBM is already streaming under it
    half8 a_pre[U], a2_pre[U];
    const bool pre = NORM && (c0 + U <= c1);
    if (pre) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a_pre[u] = SG_LOAD(wa + (int64_t)(c0 + u) * 64);
            if (GATEUP) a2_pre[u] = SG_LOAD(wu + (int64_t)(c0 + u) * 64);
        }
    }

    float inv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) inv[t] = 1.f;
    if (NORM) {
        float tot[MT];
        if (ss_in) {
            // The producer of x (a residual-epilogue GEMM over K/16 panels) left the per-panel sums of squares of
            // every row: ss_in[panel][32 rows].  Fold them in a fixed order — one round trip instead of a second
            // pass over x.  Thread (pg, m): partials of panels pg, pg + G, ... of row m; G = threads / 16.
            constexpr int G = WAVES * 4;
            const int nparts = K >> 4;
            const int pg = tid >> 4, m16 = tid & 15;
            float* red = &sm[0][0][0][0][0];                            // reuse the merge buffer: [MT][G][16] floats
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float part = 0.f;
                for (int p0 = pg; p0 < nparts; p0 += 8 * G) {
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
            int cc = c0;
            for (; cc + 8 <= c1; cc += 8) {                             // 8 independent loads per row tile in flight
                half8 v[8][MT];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < MT; ++t) v[j][t] = xok[t] ? load_half8(xr[t] + 32 * (cc + j)) : zero8;
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
            for (; cc < c1; ++cc) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const half8 v = xok[t] ? load_half8(xr[t] + 32 * cc) : zero8;
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

    // RoPE epilogue operands (wave 0 only): cos / sin of this lane's 4 output columns, fetched now so that the
    // positions -> table dependent loads do not sit at the tail of the kernel
    half4 rope_cs[MT], rope_sn[MT];
    if (MODE == SG_QKV && wave == 0) {
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

    // residual epilogue operands (wave 0 only): this lane's 4 residual values per row tile, fetched now — at the tail
    // they would be one more exposed memory round trip between the split-K merge and the store.  (resid may alias y:
    // every element is read and written by the same lane only.)
    half4 res_pre[MT];
    const bool res_early = TF_SG_RES_EARLY && MODE == SG_PLAIN && resid != nullptr && wave == 0 && (ldr % 4) == 0 &&
                           (reinterpret_cast<uintptr_t>(resid) % 8) == 0;
    if (res_early) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = t * 16 + li;
            res_pre[t] = half4{0, 0, 0, 0};
            if (m < M) res_pre[t] = *reinterpret_cast<const half4*>(resid + (int64_t)m * ldr + panel * 16 + 4 * g);
        }
    }

    int c = c0;
    for (; c + U <= c1; c += U) {
        half8 a[U], a2[U], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NORM && c == c0 && pre) {
                a[u] = a_pre[u];
                if (GATEUP) a2[u] = a2_pre[u];
            } else {
                a[u] = SG_LOAD(wa + (int64_t)(c + u) * 64);
                if (GATEUP) a2[u] = SG_LOAD(wu + (int64_t)(c + u) * 64);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                b[u][t] = xok[t] ? load_half8(xr[t] + 32 * (c + u)) : zero8;
                if (NORM) b[u][t] = sg_normalise(b[u][t], load_half8(ln_w + 32 * (c + u) + 8 * g), inv[t]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u][t], acc[t], 0, 0, 0);
                if (GATEUP) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[u], b[u][t], acc2[t], 0, 0, 0);
            }
    }
    for (; c < c1; ++c) {
        const half8 a = SG_LOAD(wa + (int64_t)c * 64);
        half8 a2 = zero8;
        if (GATEUP) a2 = SG_LOAD(wu + (int64_t)c * 64);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            half8 b = xok[t] ? load_half8(xr[t] + 32 * c) : zero8;
            if (NORM) b = sg_normalise(b, load_half8(ln_w + 32 * c + 8 * g), inv[t]);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
            if (GATEUP) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, acc2[t], 0, 0, 0);
        }
    }

    // split-K merge across the waves; C layout: lane holds D[n = 4g + r][m = li]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sm[wave][0][t][lane][r] = acc[t][r];
            if (GATEUP) sm[wave][GATEUP ? 1 : 0][t][lane][r] = acc2[t][r];
        }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + li;
        float s[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[r] = 0.f;
            s2[r] = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                s[r] += sm[w][0][t][lane][r];
                if (GATEUP) s2[r] += sm[w][GATEUP ? 1 : 0][t][lane][r];
            }
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!row_ok) break;
            const int n = panel * 16 + 4 * g + r;
            if (GATEUP) {
                const h16 gt = (h16)s[r], up = (h16)s2[r];
                const float gf = (float)gt;
                const h16 act = (h16)(gf / (1.0f + expf(-gf)));
                ((h16*)yv)[(int64_t)m * ldy + n] = hmul_rn(act, up);
            } else if (MODE == SG_F32) {
                ((float*)yv)[(int64_t)m * ldy + n] = (float)(h16)s[r];          // logits.float(): fp16 GEMM, then cast
            } else {
                h16 o = (h16)s[r];
                if (resid) o = hadd_rn(res_early ? res_pre[t][r] : resid[(int64_t)m * ldr + n], o);   // residual + hidden, fp16 add
                ((h16*)yv)[(int64_t)m * ldy + n] = o;
                s2[r] = (float)o;
            }
        }
        if (MODE == SG_PLAIN && ss_out) {            // this panel's share of sum(y^2) per row, for the next norm prologue
            float q = s2[0] * s2[0] + s2[1] * s2[1] + s2[2] * s2[2] + s2[3] * s2[3];
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (g == 0) ss_out[(int64_t)panel * 32 + m] = q;
        }
    }
}

template <int MT, int MODE, bool NORM, int WAVES>
static void launch_sg_w(const void* wp, const void* wp_up, const void* x, int64_t ldx, const void* ln_w, float eps,
                        const void* resid, int64_t ldr, void* y, int64_t ldy, int M, int N, int K, const SgRope& rp,
                        const float* ss_in, float* ss_out, hipStream_t st) {
    hipLaunchKernelGGL((skinny_gemm_kernel<MT, MODE, NORM, WAVES>), dim3(N / 16), dim3(WAVES * 64), 0, st,
                       (const half8*)wp, (const half8*)wp_up, (const h16*)x, ldx, (const h16*)ln_w, eps,
                       (const h16*)resid, ldr, y, ldy, M, N, K, rp, ss_in, ss_out);
}

template <int MODE, bool NORM>
static int launch_sg(const void* wp, const void* wp_up, const void* x, int64_t ldx, const void* ln_w, float eps,
                     const void* resid, int64_t ldr, void* y, int64_t ldy, int M, int N, int K, const SgRope& rp,
                     hipStream_t st, const float* ss_in = nullptr, float* ss_out = nullptr) {
    // wide variant: few panels and enough k-chunks that every wave still gets >= 2 of them
    constexpr bool CAN_WIDE = MODE != SG_GATEUP;
    const bool wide = CAN_WIDE && (N / 16) <= SG_WIDE_MAX_PANELS && (K >> 5) >= 2 * SG_WAVES_WIDE;
    constexpr int WW = CAN_WIDE ? SG_WAVES_WIDE : SG_WAVES;
    if (M <= 16) {
        if (wide) launch_sg_w<1, MODE, NORM, WW>(wp, wp_up, x, ldx, ln_w, eps, resid, ldr, y, ldy, M, N, K, rp, ss_in, ss_out, st);
        else launch_sg_w<1, MODE, NORM, SG_WAVES>(wp, wp_up, x, ldx, ln_w, eps, resid, ldr, y, ldy, M, N, K, rp, ss_in, ss_out, st);
    } else {
        if (wide) launch_sg_w<2, MODE, NORM, WW>(wp, wp_up, x, ldx, ln_w, eps, resid, ldr, y, ldy, M, N, K, rp, ss_in, ss_out, st);
        else launch_sg_w<2, MODE, NORM, SG_WAVES>(wp, wp_up, x, ldx, ln_w, eps, resid, ldr, y, ldy, M, N, K, rp, ss_in, ss_out, st);
    }
    TF_LAUNCH_CHECK();
    return TF_OK;
}

static bool sg_shape_ok(int M, int N, int K, int64_t ldx) {
    return M >= 1 && M <= 32 && N >= 16 && (N % 16) == 0 && K >= 32 && (K % 32) == 0 && (ldx % 8) == 0;
}

extern "C" int tf_skinny_gemm_ex(const void* w_packed, const void* x, int64_t ldx, const void* ln_w, float eps,
                                 const float* ss_in, const void* resid, int64_t ldr, float* ss_out, void* y,
                                 int64_t ldy, int M, int N, int K, int out_f32, void* stream) {
    if (!w_packed || !x || !y || !sg_shape_ok(M, N, K, ldx)) return TF_EINVAL;
    if ((out_f32 && (resid || ss_out)) || (ss_in && !ln_w)) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SgRope rp = {};
    if (out_f32)
        return ln_w ? launch_sg<SG_F32, true>(w_packed, nullptr, x, ldx, ln_w, eps, nullptr, 0, y, ldy, M, N, K, rp, st, ss_in)
                    : launch_sg<SG_F32, false>(w_packed, nullptr, x, ldx, nullptr, 0.f, nullptr, 0, y, ldy, M, N, K, rp, st);
    return ln_w ? launch_sg<SG_PLAIN, true>(w_packed, nullptr, x, ldx, ln_w, eps, resid, ldr, y, ldy, M, N, K, rp, st, ss_in,
                                            ss_out)
                : launch_sg<SG_PLAIN, false>(w_packed, nullptr, x, ldx, nullptr, 0.f, resid, ldr, y, ldy, M, N, K, rp, st,
                                             nullptr, ss_out);
}

extern "C" int tf_skinny_gemm(const void* w_packed, const void* x, int64_t ldx, void* y, int64_t ldy, int M, int N,
                              int K, int out_f32, void* stream) {
    return tf_skinny_gemm_ex(w_packed, x, ldx, nullptr, 0.f, nullptr, nullptr, 0, nullptr, y, ldy, M, N, K, out_f32, stream);
}

extern "C" int tf_skinny_gemm_swiglu_ex(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx,
                                        const void* ln_w, float eps, const float* ss_in, void* act, int64_t ldy, int M,
                                        int I, int K, void* stream) {
    if (!gate_packed || !up_packed || !x || !act || !sg_shape_ok(M, I, K, ldx) || (ss_in && !ln_w)) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SgRope rp = {};
    return ln_w ? launch_sg<SG_GATEUP, true>(gate_packed, up_packed, x, ldx, ln_w, eps, nullptr, 0, act, ldy, M, I, K, rp, st,
                                             ss_in)
                : launch_sg<SG_GATEUP, false>(gate_packed, up_packed, x, ldx, nullptr, 0.f, nullptr, 0, act, ldy, M, I, K,
                                              rp, st);
}

extern "C" int tf_skinny_gemm_swiglu(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx,
                                     void* act, int64_t ldy, int M, int I, int K, void* stream) {
    return tf_skinny_gemm_swiglu_ex(gate_packed, up_packed, x, ldx, nullptr, 0.f, nullptr, act, ldy, M, I, K, stream);
}

extern "C" int tf_skinny_qkv_rope(const void* wqkv_packed, const void* x, int64_t ldx, const void* ln_w, float eps,
                                  const float* ss_in, const void* cosb, const void* sinb, const int64_t* positions,
                                  void* q_out, void* k_cache, void* v_cache, int64_t stride_t, int64_t stride_h,
                                  int slot0, const int32_t* slot0_dev, int M, int H, int D, int K, int rotate_k,
                                  void* stream) {
    if (!wqkv_packed || !x || !cosb || !sinb || !positions || !q_out || !k_cache || !v_cache) return TF_EINVAL;
    if (H < 1 || D < 32 || (D % 32) || !sg_shape_ok(M, 3 * H * D, K, ldx) || (ss_in && !ln_w)) return TF_EINVAL;
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
    const int N = 3 * H * D;
    return ln_w ? launch_sg<SG_QKV, true>(wqkv_packed, nullptr, x, ldx, ln_w, eps, nullptr, 0, nullptr, 0, M, N, K, rp, st,
                                          ss_in)
                : launch_sg<SG_QKV, false>(wqkv_packed, nullptr, x, ldx, nullptr, 0.f, nullptr, 0, nullptr, 0, M, N, K, rp,
                                           st);
}
