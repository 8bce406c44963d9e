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
//   * grid = N/16 panels, 4 waves per workgroup split the panel's K range; partial 16x16 accumulators meet in
//     LDS.  x (<= 32 x K fp16, L2-resident) is read straight into the B operand.
//   * Variant GATEUP computes the gate and the up panel of the same columns and applies SwiGLU in the epilogue
//     (act = fp16(silu(fp16 gate)) * fp16 up — the reference's rounding points), saving a kernel and a round trip.
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

template <int MT, bool GATEUP, bool OUT_F32, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(const half8* __restrict__ wp,
                                                                    const half8* __restrict__ wp_up,
                                                                    const h16* __restrict__ x, int64_t ldx,
                                                                    void* __restrict__ yv, int64_t ldy, int M, int N,
                                                                    int K) {
    const int panel = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int nchunks = K >> 5;
    const int cpw = (nchunks + WAVES - 1) / WAVES;
    const int c0 = wave * cpw, c1 = min(nchunks, c0 + cpw);

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
    int c = c0;
    for (; c + U <= c1; c += U) {
        half8 a[U], a2[U], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = SG_LOAD(wa + (int64_t)(c + u) * 64);
            if (GATEUP) a2[u] = SG_LOAD(wu + (int64_t)(c + u) * 64);
#pragma unroll
            for (int t = 0; t < MT; ++t) b[u][t] = xok[t] ? load_half8(xr[t] + 32 * (c + u)) : zero8;
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
            const half8 b = xok[t] ? load_half8(xr[t] + 32 * c) : zero8;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
            if (GATEUP) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, acc2[t], 0, 0, 0);
        }
    }

    // split-K merge across the 4 waves; C layout: lane holds D[n = 4g + r][m = li]
    __shared__ float sm[WAVES][GATEUP ? 2 : 1][MT][64][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sm[wave][0][t][lane][r] = acc[t][r];
            if (GATEUP) sm[wave][GATEUP ? 1 : 0][t][lane][r] = acc2[t][r];
        }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = t * 16 + li;
            if (m >= M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) s += sm[w][0][t][lane][r];
                const int n = panel * 16 + 4 * g + r;
                if (GATEUP) {
                    float s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) s2 += sm[w][GATEUP ? 1 : 0][t][lane][r];
                    const h16 gt = (h16)s, up = (h16)s2;
                    const float gf = (float)gt;
                    const h16 act = (h16)(gf / (1.0f + expf(-gf)));
                    ((h16*)yv)[(int64_t)m * ldy + n] = hmul_rn(act, up);
                } else if (OUT_F32) {
                    ((float*)yv)[(int64_t)m * ldy + n] = (float)(h16)s;          // logits.float(): fp16 GEMM, then cast
                } else {
                    ((h16*)yv)[(int64_t)m * ldy + n] = (h16)s;
                }
            }
        }
    }
}

template <int MT, bool GATEUP, bool OUT_F32, int WAVES>
static void launch_sg_w(const void* wp, const void* wp_up, const void* x, int64_t ldx, void* y, int64_t ldy, int M,
                        int N, int K, hipStream_t st) {
    hipLaunchKernelGGL((skinny_gemm_kernel<MT, GATEUP, OUT_F32, WAVES>), dim3(N / 16), dim3(WAVES * 64), 0, st,
                       (const half8*)wp, (const half8*)wp_up, (const h16*)x, ldx, y, ldy, M, N, K);
}

template <bool GATEUP, bool OUT_F32>
static int launch_sg(const void* wp, const void* wp_up, const void* x, int64_t ldx, void* y, int64_t ldy, int M, int N,
                     int K, hipStream_t st) {
    // wide variant: few panels and enough k-chunks that every wave still gets >= 2 of them
    const bool wide = !GATEUP && (N / 16) <= SG_WIDE_MAX_PANELS && (K >> 5) >= 2 * SG_WAVES_WIDE;
    if (M <= 16) {
        if (wide) launch_sg_w<1, GATEUP, OUT_F32, GATEUP ? SG_WAVES : SG_WAVES_WIDE>(wp, wp_up, x, ldx, y, ldy, M, N, K, st);
        else launch_sg_w<1, GATEUP, OUT_F32, SG_WAVES>(wp, wp_up, x, ldx, y, ldy, M, N, K, st);
    } else {
        if (wide) launch_sg_w<2, GATEUP, OUT_F32, GATEUP ? SG_WAVES : SG_WAVES_WIDE>(wp, wp_up, x, ldx, y, ldy, M, N, K, st);
        else launch_sg_w<2, GATEUP, OUT_F32, SG_WAVES>(wp, wp_up, x, ldx, y, ldy, M, N, K, st);
    }
    TF_LAUNCH_CHECK();
    return TF_OK;
}

extern "C" int tf_skinny_gemm(const void* w_packed, const void* x, int64_t ldx, void* y, int64_t ldy, int M, int N,
                              int K, int out_f32, void* stream) {
    if (!w_packed || !x || !y || M < 1 || M > 32 || N < 16 || (N % 16) || K < 32 || (K % 32)) return TF_EINVAL;
    if (ldx % 8) return TF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return out_f32 ? launch_sg<false, true>(w_packed, nullptr, x, ldx, y, ldy, M, N, K, st)
                   : launch_sg<false, false>(w_packed, nullptr, x, ldx, y, ldy, M, N, K, st);
}

extern "C" int tf_skinny_gemm_swiglu(const void* gate_packed, const void* up_packed, const void* x, int64_t ldx,
                                     void* act, int64_t ldy, int M, int I, int K, void* stream) {
    if (!gate_packed || !up_packed || !x || !act || M < 1 || M > 32 || I < 16 || (I % 16) || K < 32 || (K % 32))
        return TF_EINVAL;
    if (ldx % 8) return TF_EINVAL;
    return launch_sg<true, false>(gate_packed, up_packed, x, ldx, act, ldy, M, I, K, (hipStream_t)stream);
}
