// What would the GEMM chain of a TP-8 rank's decode layer cost as ONE persistent launch with round 6's hand-offs?
//
// The 7B TP-8 rank's layer is six launches around ~12 us of HBM time (profiles/r06_tp8_7b_kernel_timeline_final.json: q|k|v 7.4,
// attention 8.0 + merge 5.4, o_proj 5.6, gate|up 9.0, down 6.7 us).  Round 3's seam probe (persist_seam_probe.hip) measured a
// persistent form at the SINGLE-GPU widths with a fenced counter barrier and lost; round 6 built different machinery for the 68M
// draft (csrc/draft_persist.hip: fixed workgroup roles, per-edge arrival counter + READY flags on their own lines, write-through
// stores / agent-scope loads instead of fences, the role's weights requested BEFORE its wait) and won 0.67x there.  This probe puts
// that machinery on the TP-8 rank's shapes, in isolation:
//
//   per layer, 8 rows:   q|k|v  [8][4096] x W[1536][4096]^T   (12.6 MB)      96 panels of 16 columns
//                        o_proj [8][512]  x W[4096][512]^T    ( 4.2 MB)     256 panels   (reads the first 512 columns of q|k|v: the
//                                                                                          attention between them is not part of the probe)
//                        gate|up + SwiGLU [8][4096] x 2 W[1376][4096]^T (22.5 MB)  86 panels (waves 0-3 gate, 4-7 up)
//                        down   [8][1376] x W[4096][1376]^T   (11.3 MB)     256 panels
//
//   chain     : four launches per layer (grid = the stage's panels, 512 threads, 8 waves split K), as the product's chain does
//   persist   : ONE launch of 256 workgroups x 512 threads for all layers; a stage is a range of workgroups; 4 edges per layer;
//               every role requests its first weight chunks before it waits for its input
//   persist_np: the same with the weights requested behind the wait (what the edges alone cost)
//
// Weights rotate through 12 distinct copies (607 MB: cold L2 / Infinity Cache); 32 layers per launch / hipGraph replay; medians of
// 10 replays; all spins bounded (wall clock).  Packed weights as in the product: [panel][K / 32 chunks][64 lanes] half8.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/tp8_layer_seam_probe.hip -o /tmp/tp8_seam && /tmp/tp8_seam
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

constexpr int HID = 4096, QKV = 1536, OK_ = 512, INTER = 1376, M = 8, WAVES = 8, THREADS = 512, G = 256, LAYERS = 32, COPIES = 12;
constexpr int EDGES = 4 * LAYERS;

struct Ctl {
    unsigned epoch, failed, pad[14];
    unsigned cnt[EDGES][9][16];              // one 64-byte line per counter: 8 shards (256 producers: 32 each) + the top / the only one
    unsigned flag[EDGES][8][16];             // 8 READY copies per edge, one line each
};
struct LayerW { const half8 *qkv, *o, *gate, *up, *down; };
struct Params {
    const LayerW* w;                          // [COPIES], device memory (a by-value table indexed by the layer would live in scratch)
    h16 *x, *qkv_out, *attn_out, *h1, *z, *h2;   // [8][HID], [8][QKV], -, [8][HID], [8][INTER], [8][HID]
    Ctl* ctl;
    u64 timeout_ticks;
    u64* stamps;                              // [G][8]: layer 16 of the persistent launch — after q|k|v's wait, its end, o_proj's wait, its end, ...
};

// (-DROLE_NOINLINE: the roles as real functions — the fully inlined persistent kernel sits at 256 registers with spills)
#ifdef ROLE_NOINLINE
#define ROLE_FN __device__ __attribute__((noinline))
#else
#define ROLE_FN __device__ __forceinline__
#endif

// ---- loads / stores of activations that cross workgroups inside one launch: agent scope (sc1), no fences ----
template <bool SC1> __device__ __forceinline__ half8 ld_act(const h16* p) {
    if (SC1) {
        union { u64 q[2]; half8 h; } v;
        v.q[0] = __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.q[1] = __hip_atomic_load(reinterpret_cast<const u64*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v.h;
    }
    return *reinterpret_cast<const half8*>(p);
}
template <bool SC1> __device__ __forceinline__ void st_act(h16* p, half4 v) {
    if (SC1) {
        union { u64 q; half4 h; } u;
        u.h = v;
        __hip_atomic_store(reinterpret_cast<u64*>(p), u.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<half4*>(p) = v;
    }
}

__device__ __forceinline__ void arrive(Ctl* c, int edge, unsigned epoch, unsigned producers) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last;
        if (producers == (unsigned)G) {
            // 256 arrivals on one line queue up for ~2.5 us: 8 shards of 32, the shard's last arriver moves on to the top counter
            const unsigned old = __hip_atomic_fetch_add(&c->cnt[edge][blockIdx.x & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = false;
            if (old + 1u == (epoch + 1u) * (G / 8)) {
                const unsigned top = __hip_atomic_fetch_add(&c->cnt[edge][8][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = top + 1u == (epoch + 1u) * 8u;
            }
        } else {
            const unsigned old = __hip_atomic_fetch_add(&c->cnt[edge][8][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old + 1u == (epoch + 1u) * producers;
        }
        if (last) {
#pragma unroll
            for (int k = 0; k < 8; ++k) __hip_atomic_store(&c->flag[edge][k][0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ bool wait(const Params& P, int edge, unsigned epoch) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const unsigned* slot = &P.ctl->flag[edge][blockIdx.x & 7][0];
        const u64 t0 = wall_clock64();
        int good = 1;
        for (unsigned spins = 0;; ++spins) {
            if ((int)(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (epoch + 1u)) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63u) == 63u) {
                if (__hip_atomic_load(&P.ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { good = 0; break; }
                if (wall_clock64() - t0 > P.timeout_ticks) {
                    __hip_atomic_store(&P.ctl->failed, (unsigned)(edge + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// One wave's share of a panel: chunks [c0, c1) (at most CP) of the panel's K / 32.  The weights of the first PRE chunks are requested
// before `before()` (the wait of the persistent form), the rest together with x; x ([8][K]) is fetched ONCE per workgroup — all its
// loads in flight together — and staged in LDS (a cross-workgroup activation comes through the fabric: a load per chunk and wave
// measured ~3.7 us per batch of 8 chunks).  Returns the wave's partial D tile.
template <int K, int CP, int PRE, bool SC1, bool GUARD, typename F>
__device__ __forceinline__ f32x4 wave_gemm(const half8* wpanel, int c0, int c1, const h16* x, h16* sx, F before) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));            // (opaque per call: keeps the per-thread addresses of four inlined roles from being hoisted out of the layer loop — 70 spilled values)
    const int lane = tid & 63, li = lane & 15, g = lane >> 4;
    const half8* wa = wpanel + (int64_t)c0 * 64;        // (c0 is wave-uniform: a scalar base, the chunk offsets are scalar adds, the lane is the only vector part)
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int LDX = K + 8, PIECES = K, NX = (PIECES + THREADS - 1) / THREADS;    // 16-byte pieces of x: 8 rows x K / 8
    half8 a[CP];
#pragma unroll
    for (int u = 0; u < PRE; ++u) a[u] = (!GUARD || c0 + u < c1) ? __builtin_nontemporal_load(wa + u * 64 + lane) : zero8;
    before();
    half8 xr[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int pc = tid + THREADS * i;
        xr[i] = pc < PIECES ? ld_act<SC1>(x + (int64_t)(pc / (K / 8)) * K + 8 * (pc % (K / 8))) : zero8;
    }
#pragma unroll
    for (int u = PRE; u < CP; ++u) a[u] = (!GUARD || c0 + u < c1) ? __builtin_nontemporal_load(wa + u * 64 + lane) : zero8;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int pc = tid + THREADS * i;
        if (pc < PIECES) *reinterpret_cast<half8*>(sx + (pc / (K / 8)) * LDX + 8 * (pc % (K / 8))) = xr[i];
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CP; ++u) {
        if (!GUARD || c0 + u < c1) {
            const half8 bq = li < M ? *reinterpret_cast<const half8*>(sx + li * LDX + 32 * (c0 + u) + 8 * g) : zero8;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], bq, acc, 0, 0, 0);
        }
    }
    return acc;
}

// a plain panel: 8 waves split K; out[row][panel * 16 + col] (fp16)
template <int K, int PRE, bool SC1, typename F>
ROLE_FN void role_plain(const half8* w, int panel, const h16* x, h16* out, int out_ld, float (*sm)[64][4], h16* sx, F before) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    constexpr int nchunks = K >> 5, cpw = (nchunks + WAVES - 1) / WAVES;
    const int c0 = min(nchunks, wave * cpw), c1 = min(nchunks, c0 + cpw);
    const f32x4 acc = wave_gemm<K, cpw, (PRE < cpw ? PRE : cpw), SC1, (nchunks % WAVES != 0)>(w + (int64_t)panel * nchunks * 64, c0, c1, x, sx, before);
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[wave][lane][r] = acc[r];
    __syncthreads();
    if (wave == 0 && li < M) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < WAVES; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += sm[k][lane][r];
        // (a crude stand-in for the norms of the real layer: keeps 32 chained GEMMs finite and non-zero)
        st_act<SC1>(out + (int64_t)li * out_ld + panel * 16 + 4 * g, half4{(h16)tanhf(s[0]), (h16)tanhf(s[1]), (h16)tanhf(s[2]), (h16)tanhf(s[3])});
    }
    __syncthreads();
}

// gate | up + SwiGLU: waves 0-3 split K of the gate panel, waves 4-7 of the up panel
template <int PRE, bool SC1, typename F>
ROLE_FN void role_gateup(const half8* wg, const half8* wu, int panel, const h16* x, h16* z, float (*sm)[64][4], h16* sx, F before) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    constexpr int nchunks = HID >> 5, cpw = nchunks / 4;
    const int kw = wave & 3;
    const half8* w = (wave < 4 ? wg : wu) + (int64_t)panel * nchunks * 64;
    const f32x4 acc = wave_gemm<HID, cpw, (PRE < cpw ? PRE : cpw), SC1, false>(w, kw * cpw, kw * cpw + cpw, x, sx, before);
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[wave][lane][r] = acc[r];
    __syncthreads();
    if (wave == 0 && li < M) {
        f32x4 sg = {0.f, 0.f, 0.f, 0.f}, su = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sg[r] += sm[k][lane][r];
                su[r] += sm[4 + k][lane][r];
            }
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gt = (float)(h16)sg[r];
            o[r] = (h16)((float)(h16)(gt / (1.f + __expf(-gt))) * (float)(h16)su[r]);
        }
        st_act<SC1>(z + (int64_t)li * INTER + panel * 16 + 4 * g, o);
    }
    __syncthreads();
}

constexpr size_t SMEM_BYTES = sizeof(float) * WAVES * 64 * 4 + (size_t)M * (HID + 8) * 2;
#define SMEM_DECL extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[]; \
    float (*sm)[64][4] = reinterpret_cast<float (*)[64][4]>(smem_raw); h16* sx = reinterpret_cast<h16*>(smem_raw + sizeof(float) * WAVES * 64 * 4);

// ---- the chain: one launch per stage ----
__global__ __launch_bounds__(THREADS) void k_qkv(const half8* w, const h16* x, h16* out) {
    SMEM_DECL
    role_plain<HID, 0, false>(w, blockIdx.x, x, out, QKV, sm, sx, [] {});
}
__global__ __launch_bounds__(THREADS) void k_o(const half8* w, const h16* x, h16* out) {
    SMEM_DECL
    role_plain<OK_, 0, false>(w, blockIdx.x, x, out, HID, sm, sx, [] {});
}
__global__ __launch_bounds__(THREADS) void k_gu(const half8* wg, const half8* wu, const h16* x, h16* z) {
    SMEM_DECL
    role_gateup<0, false>(wg, wu, blockIdx.x, x, z, sm, sx, [] {});
}
__global__ __launch_bounds__(THREADS) void k_down(const half8* w, const h16* x, h16* out) {
    SMEM_DECL
    role_plain<INTER, 0, false>(w, blockIdx.x, x, out, HID, sm, sx, [] {});
}

// ---- one persistent launch for all layers ----
// workgroups: q|k|v 0..95, o_proj all 256, gate|up 160..245, down all 256.  Edge 4 l + {0: q|k|v -> o_proj, 1: o_proj -> gate|up,
// 2: gate|up -> down, 3: down -> next layer's q|k|v}
template <bool PREFETCH>
__global__ __launch_bounds__(THREADS) void k_persist(Params P) {
    SMEM_DECL
    constexpr int PQ = PREFETCH ? 16 : 0, PO = PREFETCH ? 2 : 0, PG = PREFETCH ? 16 : 0, PD = PREFETCH ? 6 : 0;
    const int w = blockIdx.x;
    const unsigned epoch = __hip_atomic_load(&P.ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_load(&P.ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    bool alive = true;
    auto stamp = [&](int l, int k) {
        if (l == 16 && threadIdx.x == 0) P.stamps[w * 8 + k] = wall_clock64();
    };
#pragma unroll 1
    for (int l = 0; l < LAYERS && alive; ++l) {
        const LayerW L = P.w[l % COPIES];
        const h16* xin = (l == 0) ? P.x : P.h2;
        if (w < QKV / 16) {
            role_plain<HID, PQ, true>(L.qkv, w, xin, P.qkv_out, QKV, sm, sx, [&] { if (l > 0) alive = wait(P, 4 * (l - 1) + 3, epoch); stamp(l, 0); });
            if (!alive) break;
            stamp(l, 1);
            arrive(P.ctl, 4 * l + 0, epoch, QKV / 16);
        }
        role_plain<OK_, PO, true>(L.o, w, P.qkv_out, P.h1, HID, sm, sx, [&] { alive = wait(P, 4 * l + 0, epoch); stamp(l, 2); });
        if (!alive) break;
        stamp(l, 3);
        arrive(P.ctl, 4 * l + 1, epoch, G);
        if (w >= 160 && w < 160 + INTER / 16) {
            role_gateup<PG, true>(L.gate, L.up, w - 160, P.h1, P.z, sm, sx, [&] { alive = wait(P, 4 * l + 1, epoch); stamp(l, 4); });
            if (!alive) break;
            stamp(l, 5);
            arrive(P.ctl, 4 * l + 2, epoch, INTER / 16);
        }
        role_plain<INTER, PD, true>(L.down, w, P.z, P.h2, HID, sm, sx, [&] { alive = wait(P, 4 * l + 2, epoch); stamp(l, 6); });
        if (!alive) break;
        stamp(l, 7);
        arrive(P.ctl, 4 * l + 3, epoch, G);
    }
    // everyone has passed its last wait once the last edge is complete: workgroup 0 waits for it and opens the next epoch
    if (w == 0 && alive) {
        if (wait(P, 4 * (LAYERS - 1) + 3, epoch) && threadIdx.x == 0)
            __hip_atomic_store(&P.ctl->epoch, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static void fill_random(std::vector<h16>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& e : v) {
        s = s * 1664525u + 1013904223u;
        e = (h16)(((int)(s >> 9) % 2001 - 1000) * (scale / 1000.f));
    }
}

int main() {
    const int REPS = 10;
    const size_t nq = (size_t)QKV * HID, no = (size_t)HID * OK_, ng = (size_t)INTER * HID, nd = (size_t)HID * INTER;
    std::vector<h16> hq(nq), ho(no), hg(ng), hu(ng), hd(nd), hx((size_t)M * HID);
    fill_random(hq, 1, 0.03f); fill_random(ho, 2, 0.09f); fill_random(hg, 3, 0.05f); fill_random(hu, 4, 0.05f);
    fill_random(hd, 5, 0.10f); fill_random(hx, 6, 1.0f);      // (scales keep the 32-layer chain of un-normalised GEMMs finite and non-zero)
    Params P{};
    LayerW hw[COPIES];
    for (int i = 0; i < COPIES; ++i) {
        h16 *q, *o, *g, *u, *d;
        CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&o, no * 2)); CK(hipMalloc(&g, ng * 2)); CK(hipMalloc(&u, ng * 2)); CK(hipMalloc(&d, nd * 2));
        CK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(o, ho.data(), no * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(g, hg.data(), ng * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(u, hu.data(), ng * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d, hd.data(), nd * 2, hipMemcpyHostToDevice));
        hw[i] = LayerW{(const half8*)q, (const half8*)o, (const half8*)g, (const half8*)u, (const half8*)d};
    }
    LayerW* dw;
    CK(hipMalloc(&dw, sizeof(hw)));
    CK(hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice));
    P.w = dw;
    CK(hipMalloc(&P.x, (size_t)M * HID * 2)); CK(hipMalloc(&P.qkv_out, (size_t)M * QKV * 2)); CK(hipMalloc(&P.h1, (size_t)M * HID * 2));
    CK(hipMalloc(&P.z, (size_t)M * INTER * 2)); CK(hipMalloc(&P.h2, (size_t)M * HID * 2));
    CK(hipMemcpy(P.x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&P.stamps, (size_t)G * 8 * 8));
    CK(hipMemset(P.stamps, 0, (size_t)G * 8 * 8));
    CK(hipMalloc(&P.ctl, sizeof(Ctl)));
    CK(hipMemset(P.ctl, 0, sizeof(Ctl)));
    P.timeout_ticks = 100000000ull / 2;        // 0.5 s of the 100 MHz wall clock
    h16* ref;
    CK(hipMalloc(&ref, (size_t)M * HID * 2));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (const void* f : {(const void*)k_qkv, (const void*)k_o, (const void*)k_gu, (const void*)k_down, (const void*)k_persist<true>, (const void*)k_persist<false>})
        CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BYTES));

    auto run_variant = [&](int variant, const char* name) {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        if (variant == 0) {
            for (int l = 0; l < LAYERS; ++l) {
                const LayerW& L = hw[l % COPIES];
                hipLaunchKernelGGL(k_qkv, dim3(QKV / 16), dim3(THREADS), SMEM_BYTES, st, L.qkv, l == 0 ? P.x : P.h2, P.qkv_out);
                hipLaunchKernelGGL(k_o, dim3(HID / 16), dim3(THREADS), SMEM_BYTES, st, L.o, P.qkv_out, P.h1);
                hipLaunchKernelGGL(k_gu, dim3(INTER / 16), dim3(THREADS), SMEM_BYTES, st, L.gate, L.up, P.h1, P.z);
                hipLaunchKernelGGL(k_down, dim3(HID / 16), dim3(THREADS), SMEM_BYTES, st, L.down, P.z, P.h2);
            }
        } else if (variant == 1) {
            hipLaunchKernelGGL(k_persist<true>, dim3(G), dim3(THREADS), SMEM_BYTES, st, P);
        } else {
            hipLaunchKernelGGL(k_persist<false>, dim3(G), dim3(THREADS), SMEM_BYTES, st, P);
        }
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        std::vector<float> ts;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < REPS + 2; ++r) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(exec, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) ts.push_back(ms * 1e3f / LAYERS);
        }
        std::sort(ts.begin(), ts.end());
        Ctl head;
        CK(hipMemcpy(&head, P.ctl, 64, hipMemcpyDeviceToHost));
        if (variant == 0) CK(hipMemcpy(ref, P.h2, (size_t)M * HID * 2, hipMemcpyDeviceToDevice));
        std::vector<h16> a((size_t)M * HID), b((size_t)M * HID);
        CK(hipMemcpy(a.data(), P.h2, a.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), ref, b.size() * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, nonzero = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            diff += (a[i] != b[i]);
            nonzero += ((float)a[i] != 0.f);
        }
        printf("{\"variant\": \"%s\", \"us_per_layer_median\": %.2f, \"us_per_layer_min\": %.2f, \"layers_per_launch\": %d, \"wait_timeouts_edge\": %u, "
               "\"outputs_differing_from_chain\": %zu, \"nonzero_outputs\": %zu}\n", name, ts[ts.size() / 2], ts[0], LAYERS, head.failed, diff, nonzero);
        if (variant != 0) {
            std::vector<u64> hs((size_t)G * 8);
            CK(hipMemcpy(hs.data(), P.stamps, hs.size() * 8, hipMemcpyDeviceToHost));
            u64 t0 = ~0ull;
            for (int w = 0; w < QKV / 16; ++w) t0 = std::min(t0, hs[w * 8 + 0]);
            printf("{\"stamps_layer_16_us_since_first_qkv_wait_done\": {");
            const char* names[8] = {"qkv_wait_done", "qkv_done", "o_wait_done", "o_done", "gateup_wait_done", "gateup_done", "down_wait_done", "down_done"};
            for (int k = 0; k < 8; ++k) {
                std::vector<double> v;
                for (int w = 0; w < G; ++w)
                    if (hs[w * 8 + k] != 0ull) v.push_back((double)(hs[w * 8 + k] - t0) / 100.0);
                std::sort(v.begin(), v.end());
                printf("%s\"%s\": [%.2f, %.2f, %.2f]", k ? ", " : "", names[k], v.empty() ? 0.0 : v[0], v.empty() ? 0.0 : v[v.size() / 2], v.empty() ? 0.0 : v.back());
            }
            printf("}, \"columns\": \"min, median, max over the role's workgroups\"}\n");
            CK(hipMemset(P.stamps, 0, (size_t)G * 8 * 8));
        }
        fflush(stdout);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    };
    printf("{\"probe\": \"GEMM chain of a 7B TP-8 rank's decode layer at 8 rows: q|k|v 12.6 MB -> o_proj 4.2 MB -> gate|up 22.5 MB -> down 11.3 MB = 50.6 MB per layer, "
           "32 layers, weights rotate through 12 copies (607 MB)\"}\n");
    run_variant(0, "chain: four launches per layer (128 per graph)");
    run_variant(2, "one persistent launch, 256 workgroups, 4 edges per layer, weights requested BEHIND the wait");
    run_variant(1, "one persistent launch, 256 workgroups, 4 edges per layer, first weight chunks requested BEFORE the wait");
    run_variant(0, "chain (again)");
    return 0;
}
