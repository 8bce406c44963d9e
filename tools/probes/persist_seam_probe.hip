// What does keeping a GEMM -> GEMM seam of the decode layer INSIDE one launch buy on MI355X?
//
// The round-2 verdict asked for a persistent per-layer launch of the retrieval-verify forward.  This probe measures the
// seam such a kernel consists of, in isolation and with the product's streaming structure (packed 16 x 32 weight tiles,
// one contiguous KiB per wave load, 4 waves of a workgroup splitting K, MFMA K-reduction, LDS merge):
//
//   phase A   y[8][4096]   = x[8][4096]  . Wa^T      (o_proj:   256 panels,  33.5 MB of weights)
//   phase B   z[8][11008]  = silu(y Wg^T) * (y Wu^T) (gate|up: 688 panels, 180.4 MB) — every workgroup needs ALL of y
//
//   two      : two launches (A, then B), as the product does today
//   fused    : ONE launch of 256 persistent workgroups (one per CU): A, grid barrier, B — with the first weight
//              chunks of B's first panel already loaded into registers BEFORE the barrier (the only thing a
//              persistent kernel can do that a launch boundary cannot: keep HBM streaming across the seam)
//   fused_np : the same without the prefetch (what the barrier alone costs)
//
// Weights rotate through > 600 MB of distinct copies (cold L2 / Infinity Cache, like a 13 GB forward); 32 seams per
// hipGraph replay; medians of 10 replays.  All spins are bounded.  Build and run (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/persist_seam_probe.hip -o /tmp/persist_seam_probe && /tmp/persist_seam_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

constexpr int HID = 4096, INTER = 11008, M = 8, WAVES = 4, U = 4;

// one 16-row panel of W (packed: [panel][K/32 chunks][64 lanes] half8) against x[M<=16][K]; returns wave 0's merged D tile
// (panel < 0: this team has no panel in this round — it still takes part in the workgroup barriers)
__device__ __forceinline__ f32x4 panel_gemm(const half8* __restrict__ wp, int panel, const h16* __restrict__ x, int K,
                                            float (*sm)[64][4], const half8* pre, int npre) {
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & (WAVES - 1), li = lane & 15, g = lane >> 4;
    if (panel < 0) {
        __syncthreads();
        __syncthreads();
        return f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nchunks = K >> 5, cpw = (nchunks + WAVES - 1) / WAVES;
    const int c0 = wave * cpw, c1 = min(nchunks, c0 + cpw);
    const half8* wa = wp + ((int64_t)panel * nchunks) * 64 + lane;
    const h16* xr = x + (int64_t)(li < M ? li : 0) * K + 8 * g;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int c = c0;
    for (; c + U <= c1; c += U) {
        half8 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = (c == c0 && u < npre) ? pre[u] : __builtin_nontemporal_load(wa + (int64_t)(c + u) * 64);
            b[u] = li < M ? *reinterpret_cast<const half8*>(xr + 32 * (c + u)) : zero8;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u], acc, 0, 0, 0);
    }
    for (; c < c1; ++c) {
        const half8 a = __builtin_nontemporal_load(wa + (int64_t)c * 64);
        const half8 b = li < M ? *reinterpret_cast<const half8*>(xr + 32 * c) : zero8;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[wave][lane][r] = acc[r];
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += sm[w][lane][r];
    }
    __syncthreads();
    return s;
}

__device__ __forceinline__ void phase_a(const half8* wa, const h16* x, h16* y, int panel, float (*sm)[64][4]) {
    const f32x4 s = panel_gemm(wa, panel, x, HID, sm, nullptr, 0);
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    if (panel >= 0 && ((threadIdx.x >> 6) & (WAVES - 1)) == 0 && li < M)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[(int64_t)li * HID + panel * 16 + 4 * g + r] = (h16)s[r];
}

__device__ __forceinline__ void phase_b(const half8* wg, const half8* wu, const h16* y, h16* z, int panel, float (*sm)[64][4],
                                        const half8* pre, int npre) {
    const f32x4 sg = panel_gemm(wg, panel, y, HID, sm, pre, npre);
    const f32x4 su = panel_gemm(wu, panel, y, HID, sm, nullptr, 0);
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    if (panel >= 0 && ((threadIdx.x >> 6) & (WAVES - 1)) == 0 && li < M)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gt = (float)(h16)sg[r];
            z[(int64_t)li * INTER + panel * 16 + 4 * g + r] = (h16)((float)(h16)(gt / (1.f + __expf(-gt))) * (float)(h16)su[r]);
        }
}

__global__ __launch_bounds__(256) void kernel_a(const half8* wa, const h16* x, h16* y) {
    __shared__ float sm[WAVES][64][4];
    phase_a(wa, x, y, blockIdx.x, sm);
}
__global__ __launch_bounds__(256) void kernel_b(const half8* wg, const half8* wu, const h16* y, h16* z) {
    __shared__ float sm[WAVES][64][4];
    phase_b(wg, wu, y, z, blockIdx.x, sm, nullptr, 0);
}

// counter barrier, one workgroup per CU: release fence -> arrive -> bounded poll (relaxed sc1 loads) -> acquire fence
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (threadIdx.x != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every CU reader drops its stale L1 lines
    return ok != 0;
}

template <bool PREFETCH>
__global__ __launch_bounds__(256) void kernel_fused(const half8* wa, const half8* wg, const half8* wu, const h16* x, h16* y,
                                                    h16* z, unsigned* counter, unsigned target, unsigned* failed) {
    __shared__ float sm[WAVES][64][4];
    const int G = gridDim.x;
    for (int p = blockIdx.x; p < HID / 16; p += G) phase_a(wa, x, y, p, sm);
    // B's first panel: this wave's first U weight chunks of the gate matrix, in flight across the barrier
    half8 pre[U];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = HID >> 5, cpw = (nchunks + WAVES - 1) / WAVES;
    const bool has_b = (int)blockIdx.x < INTER / 16;                 // more workgroups than gate|up panels: nothing to prefetch
    if (PREFETCH && has_b) {
        const half8* w0 = wg + ((int64_t)blockIdx.x * nchunks + wave * cpw) * 64 + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) pre[u] = __builtin_nontemporal_load(w0 + (int64_t)u * 64);
    }
    if (!grid_barrier(counter, target)) {
        if (threadIdx.x == 0) atomicAdd(failed, 1u);
        return;
    }
    bool first = true;
    for (int p = blockIdx.x; p < INTER / 16; p += G) {
        phase_b(wg, wu, y, z, p, sm, pre, (PREFETCH && first && has_b) ? U : 0);
        first = false;
    }
}

// The same with TEAMS x 4 waves per workgroup (one workgroup per CU: 256 barrier participants, the occupancy of the launch
// form): team t of workgroup b takes panels (b * TEAMS + t), + G * TEAMS, ...
template <int TEAMS, bool PREFETCH>
__global__ __launch_bounds__(256 * TEAMS) void kernel_fused_teams(const half8* wa, const half8* wg, const half8* wu, const h16* x,
                                                                  h16* y, h16* z, unsigned* counter, unsigned target,
                                                                  unsigned* failed) {
    __shared__ float sm_all[TEAMS][WAVES][64][4];
    const int team = threadIdx.x >> 8, stride = gridDim.x * TEAMS, first_p = blockIdx.x * TEAMS + team;
    float (*sm)[64][4] = sm_all[team];
    const int rounds_a = (HID / 16 + stride - 1) / stride, rounds_b = (INTER / 16 + stride - 1) / stride;
    for (int r = 0; r < rounds_a; ++r) {
        const int p = first_p + r * stride;
        phase_a(wa, x, y, p < HID / 16 ? p : -1, sm);
    }
    half8 pre[U];
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & (WAVES - 1);
    const int nchunks = HID >> 5, cpw = (nchunks + WAVES - 1) / WAVES;
    const bool has_b = first_p < INTER / 16;
    if (PREFETCH && has_b) {
        const half8* w0 = wg + ((int64_t)first_p * nchunks + wave * cpw) * 64 + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) pre[u] = __builtin_nontemporal_load(w0 + (int64_t)u * 64);
    }
    if (!grid_barrier(counter, target)) {
        if (threadIdx.x == 0) atomicAdd(failed, 1u);
        return;
    }
    for (int r = 0; r < rounds_b; ++r) {
        const int p = first_p + r * stride;
        phase_b(wg, wu, y, z, p < INTER / 16 ? p : -1, sm, pre, (PREFETCH && r == 0 && has_b) ? U : 0);
    }
}

static void fill_random(std::vector<h16>& v, unsigned seed, float scale) {
    unsigned s = seed;
    for (auto& e : v) {
        s = s * 1664525u + 1013904223u;
        e = (h16)(((int)(s >> 9) % 2001 - 1000) * (scale / 1000.f));
    }
}

int main(int argc, char** argv) {
    const int COPIES = 4, SEAMS = 32, REPS = 10;
    const size_t na = (size_t)HID * HID, nb = (size_t)INTER * HID;
    std::vector<h16> ha(na), hb(nb), hx((size_t)M * HID);
    fill_random(ha, 1, 0.02f); fill_random(hb, 2, 0.02f); fill_random(hx, 3, 1.0f);
    h16 *wa[COPIES], *wg[COPIES], *wu[COPIES], *x, *y, *z, *z_ref;
    for (int i = 0; i < COPIES; ++i) {
        CK(hipMalloc(&wa[i], na * 2)); CK(hipMalloc(&wg[i], nb * 2)); CK(hipMalloc(&wu[i], nb * 2));
        CK(hipMemcpy(wa[i], ha.data(), na * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(wg[i], hb.data(), nb * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(wu[i], hb.data(), nb * 2, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&y, (size_t)M * HID * 2));
    CK(hipMalloc(&z, (size_t)M * INTER * 2)); CK(hipMalloc(&z_ref, (size_t)M * INTER * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    unsigned *counter, *failed;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&failed, 4));
    CK(hipMemset(failed, 0, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int G = 256;                          // persistent workgroups of the fused forms (256 = one per CU; argv: a list to sweep)

    auto run_variant = [&](int variant, const char* name) {
        CK(hipMemsetAsync(counter, 0, 4, st));
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(counter, 0, 4, st));
        for (int s = 0; s < SEAMS; ++s) {
            const int c = s % COPIES;
            if (variant == 0) {
                hipLaunchKernelGGL(kernel_a, dim3(HID / 16), dim3(256), 0, st, (const half8*)wa[c], x, y);
                hipLaunchKernelGGL(kernel_b, dim3(INTER / 16), dim3(256), 0, st, (const half8*)wg[c], (const half8*)wu[c], y, z);
            } else if (variant == 1) {
                hipLaunchKernelGGL(kernel_fused<true>, dim3(G), dim3(256), 0, st, (const half8*)wa[c], (const half8*)wg[c],
                                   (const half8*)wu[c], x, y, z, counter, (unsigned)(G * (s + 1)), failed);
            } else if (variant == 2) {
                hipLaunchKernelGGL(kernel_fused<false>, dim3(G), dim3(256), 0, st, (const half8*)wa[c], (const half8*)wg[c],
                                   (const half8*)wu[c], x, y, z, counter, (unsigned)(G * (s + 1)), failed);
            } else if (variant == 3) {
                hipLaunchKernelGGL((kernel_fused_teams<3, false>), dim3(G), dim3(768), 0, st, (const half8*)wa[c],
                                   (const half8*)wg[c], (const half8*)wu[c], x, y, z, counter, (unsigned)(G * (s + 1)), failed);
            } else {
                hipLaunchKernelGGL((kernel_fused_teams<3, true>), dim3(G), dim3(768), 0, st, (const half8*)wa[c],
                                   (const half8*)wg[c], (const half8*)wu[c], x, y, z, counter, (unsigned)(G * (s + 1)), failed);
            }
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
            if (r >= 2) ts.push_back(ms * 1e3f / SEAMS);
        }
        std::sort(ts.begin(), ts.end());
        unsigned f = 0;
        CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
        if (variant == 0) CK(hipMemcpy(z_ref, z, (size_t)M * INTER * 2, hipMemcpyDeviceToDevice));
        std::vector<h16> a((size_t)M * INTER), b((size_t)M * INTER);
        CK(hipMemcpy(a.data(), z, a.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), z_ref, b.size() * 2, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < a.size(); ++i) diff += (a[i] != b[i]);
        printf("{\"variant\": \"%s\", \"persistent_workgroups\": %d, \"us_per_seam_median\": %.2f, \"us_min\": %.2f, "
               "\"barrier_timeouts\": %u, \"outputs_differing_from_two_launch\": %zu}\n", name, variant == 0 ? 0 : G,
               ts[ts.size() / 2], ts[0], f, diff);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    };
    printf("{\"probe\": \"o_proj (33.5 MB) -> gate|up (180.4 MB) seam at 8 rows, 7B widths; bytes per seam 213.9 MB\"}\n");
    run_variant(0, "two launches (o_proj kernel, gate|up kernel)");
    std::vector<int> gs = {256, 512};
    if (argc > 1) {
        gs.clear();
        for (int i = 1; i < argc; ++i) gs.push_back(atoi(argv[i]));
    }
    for (int g : gs) {
        G = g;
        run_variant(2, "one launch: A, grid barrier, B; no prefetch");
        run_variant(1, "one launch: A, grid barrier, B; first chunks of B prefetched across the barrier");
    }
    G = 256;
    run_variant(3, "one launch, 12 waves per workgroup (3 teams of 4), one workgroup per CU: A, grid barrier, B; no prefetch");
    run_variant(4, "one launch, 12 waves per workgroup, with the prefetch across the barrier");
    run_variant(0, "two launches (again)");
    return 0;
}
