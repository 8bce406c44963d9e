// What does a plain streaming read reach on this MI355X?  The roofline fractions of DESIGN are quoted against the 8 TB/s
// datasheet peak; this probe measures the ceiling a kernel that does nothing but read can reach, so that "0.75 of peak" can be
// read against it.  A 2 GiB buffer (far beyond L2 / Infinity Cache) is read once per launch by a grid-stride loop of 16-byte
// non-temporal loads, U loads in flight per thread, summed into one value per thread (kept alive by a never-true store);
// variants: workgroups per CU x threads x U.  Median of 10 launches each, HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/hbm_read_probe.hip -o /tmp/hbm_read && /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int U>
__global__ void read_kernel(const f32x4* __restrict__ src, size_t n16, float* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(src + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];      // never true for the zero-filled buffer
}

// The decode attention's pattern on the reference's [token][head][128] cache: workgroup (split, head) walks ITS head's 256-byte rows of a
// contiguous token range — rows 8 KiB apart (32 heads) — 16 lanes per row, 4 rows per wave load, U loads in flight per wave, 4 waves.
// All 32 heads of a split move through the same tokens at the same time, so the chip as a whole still sweeps the buffer once.
template <int U>
__global__ void read_rows_kernel(const f32x4* __restrict__ src, int tokens, int heads, int nsplit, float* sink) {
    const int split = blockIdx.x % nsplit, head = blockIdx.x / nsplit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int t0 = (int)((long long)tokens * split / nsplit), t1 = (int)((long long)tokens * (split + 1) / nsplit);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int rows_per_step = 4 * U * waves;
    for (int t = t0 + 4 * U * wave; t < t1; t += rows_per_step) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = t + 4 * u + (lane >> 4);
            v[u] = row < t1 ? __builtin_nontemporal_load(src + ((size_t)row * heads + head) * 16 + (lane & 15)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];
}

// The same walk over TWO buffers (K rows and V rows of the same tokens, as the attention kernel reads them): U / 2 loads of either
// per wave in flight, 16-key tiles
template <int U>
__global__ void read_rows_kv_kernel(const f32x4* __restrict__ ksrc, const f32x4* __restrict__ vsrc, int tokens, int heads, int nsplit, float* sink) {
    const int split = blockIdx.x % nsplit, head = blockIdx.x / nsplit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int t0 = (int)((long long)tokens * split / nsplit), t1 = (int)((long long)tokens * (split + 1) / nsplit);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int H2 = U / 2;
    const int rows_per_step = 4 * H2 * waves;
    for (int t = t0 + 4 * H2 * wave; t < t1; t += rows_per_step) {
        f32x4 a[H2], b[H2];
#pragma unroll
        for (int u = 0; u < H2; ++u) {
            const int row = t + 4 * u + (lane >> 4);
            const size_t at = ((size_t)row * heads + head) * 16 + (lane & 15);
            a[u] = row < t1 ? __builtin_nontemporal_load(ksrc + at) : f32x4{0.f, 0.f, 0.f, 0.f};
            b[u] = row < t1 ? __builtin_nontemporal_load(vsrc + at) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < H2; ++u) acc += a[u] + b[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];
}

// The attention kernel's ACTUAL mapping (csrc/attn.hip load_kv_tile): a load instruction is an MFMA A-operand fragment — lane (li, g)
// reads 16 bytes of key 16 tile + li at dims 32 c + 8 g: SIXTEEN rows x 64 bytes per instruction, four instructions (c) per 256-byte
// row set, K and V; wave w owns tiles w, w + 4, ...; TILES tiles (8 loads each) in flight per wave.
template <int TILES>
__global__ void read_frag_kv_kernel(const f32x4* __restrict__ ksrc, const f32x4* __restrict__ vsrc, int tokens, int heads, int nsplit, float* sink) {
    const int split = blockIdx.x % nsplit, head = blockIdx.x / nsplit;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6, li = lane & 15, g = lane >> 4;
    const int ntiles = tokens / 16, tl0 = (int)((long long)ntiles * split / nsplit), tl1 = (int)((long long)ntiles * (split + 1) / nsplit);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = tl0 + wave; t < tl1; t += waves * TILES) {
        f32x4 a[TILES][4], b[TILES][4];
#pragma unroll
        for (int i = 0; i < TILES; ++i) {
            const int tt = min(t + waves * i, tl1 - 1);
            const size_t at = ((size_t)(tt * 16 + li) * heads + head) * 16 + g;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[i][c] = __builtin_nontemporal_load(ksrc + at + 4 * c);
                b[i][c] = __builtin_nontemporal_load(vsrc + at + 4 * c);
            }
        }
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += a[i][c] + b[i][c];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];
}

template <int TILES>
static void run_frag_kv(const f32x4* src, size_t n16, float* sink, int heads, int nsplit, int threads, hipStream_t st) {
    const int tokens = (int)(n16 / 2 / 16 / heads) & ~15;
    const f32x4* vsrc = src + (size_t)tokens * heads * 16;
    std::vector<float> ts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_frag_kv_kernel<TILES>, dim3(heads * nsplit), dim3(threads), 0, st, src, vsrc, tokens, heads, nsplit, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = 2.0 * tokens * heads * 256.0;
    printf("{\"pattern\": \"MFMA-fragment loads as csrc/attn.hip issues them: 16 rows x 64 bytes per instruction, K + V, [token][%d heads][128 halfs]\", \"splits\": %d, \"workgroups\": %d, "
           "\"threads\": %d, \"tiles_in_flight_per_wave\": %d, \"loads_in_flight_per_wave\": %d, \"ms_median\": %.3f, \"GBps_median\": %.0f, \"GBps_best\": %.0f, \"frac_of_8TBps\": %.3f}\n",
           heads, nsplit, heads * nsplit, threads, TILES, 8 * TILES, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
}

// The skinny GEMMs' weight stream (csrc/gemv.hip): packed panels [panel][K / 32 chunks][64 lanes] x 16 bytes — a wave-load is 1 KiB
// contiguous; a workgroup of 4 waves takes a (gate, up) panel pair of K = 4096 (2 x 128 KiB), wave w the chunks 32 w .. 32 w + 31 of
// either; U loads per wave in flight; the grid is the 688 pairs of the 7B gate|up GEMM, or fewer workgroups striding over them.
template <int U>
__global__ void read_panels_kernel(const f32x4* __restrict__ src, int pairs, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = blockIdx.x; p < pairs; p += gridDim.x) {
        const f32x4* g = src + ((size_t)(2 * p) * 128 + 32 * wave) * 64 + lane;
        const f32x4* u = src + ((size_t)(2 * p + 1) * 128 + 32 * wave) * 64 + lane;
        for (int c = 0; c < 32; c += U / 2) {
            f32x4 a[U / 2], b[U / 2];
#pragma unroll
            for (int i = 0; i < U / 2; ++i) {
                a[i] = __builtin_nontemporal_load(g + (size_t)(c + i) * 64);
                b[i] = __builtin_nontemporal_load(u + (size_t)(c + i) * 64);
            }
#pragma unroll
            for (int i = 0; i < U / 2; ++i) acc += a[i] + b[i];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];
}

// The same work with the weights laid out CHUNK-major ([K / 32 chunks][panel][64 lanes]): workgroups that run side by side read
// neighbouring KiBs at every step — one moving window across the chip instead of thousands of private 32 KiB streams.
template <int U>
__global__ void read_panels_tm_kernel(const f32x4* __restrict__ src, int pairs, int group, float* sink) {
    // `group` consecutive pairs form one chunk-major block (a GEMM of that many panel pairs): block stride = group * 2 * 128 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = blockIdx.x; p < pairs; p += gridDim.x) {
        const int blk = p / group, pl = p % group, np = 2 * group;
        const f32x4* base = src + (size_t)blk * np * 128 * 64 + lane;
        for (int c = 0; c < 32; c += U / 2) {
            f32x4 a[U / 2], b[U / 2];
#pragma unroll
            for (int i = 0; i < U / 2; ++i) {
                const size_t ch = (size_t)(32 * wave + c + i) * np;
                a[i] = __builtin_nontemporal_load(base + (ch + 2 * pl) * 64);
                b[i] = __builtin_nontemporal_load(base + (ch + 2 * pl + 1) * 64);
            }
#pragma unroll
            for (int i = 0; i < U / 2; ++i) acc += a[i] + b[i];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];
}

template <int U>
static void run_panels_tm(const f32x4* src, float* sink, int grid, hipStream_t st) {
    const int pairs = 688 * 8;
    std::vector<float> ts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_panels_tm_kernel<U>, dim3(grid), dim3(256), 0, st, src, pairs, 688, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)pairs * 2 * 128 * 1024;
    printf("{\"pattern\": \"the same panels laid out chunk-major ([chunk][panel][lane]: neighbours read neighbouring KiBs)\", \"workgroups\": %d, \"panel_pairs\": %d, "
           "\"loads_in_flight_per_wave\": %d, \"ms_median\": %.3f, \"GBps_median\": %.0f, \"GBps_best\": %.0f, \"frac_of_8TBps\": %.3f}\n", grid, pairs, U,
           ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
}

template <int U>
static void run_panels(const f32x4* src, float* sink, int grid, hipStream_t st) {
    const int pairs = 688 * 8;                           // eight different copies of the 180 MB stream per launch: 1.44 GB, cold
    std::vector<float> ts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_panels_kernel<U>, dim3(grid), dim3(256), 0, st, src, pairs, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)pairs * 2 * 128 * 1024;
    printf("{\"pattern\": \"packed GEMM panels (1 KiB per wave-load, a 2 x 128 KiB panel pair per workgroup, 4 waves split K)\", \"workgroups\": %d, \"panel_pairs\": %d, "
           "\"loads_in_flight_per_wave\": %d, \"ms_median\": %.3f, \"GBps_median\": %.0f, \"GBps_best\": %.0f, \"frac_of_8TBps\": %.3f}\n", grid, pairs, U,
           ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
}

template <int U>
static void run_rows_kv(const f32x4* src, size_t n16, float* sink, int heads, int nsplit, int threads, hipStream_t st) {
    const int tokens = (int)(n16 / 2 / 16 / heads);                     // K in the first half of the buffer, V in the second
    const f32x4* vsrc = src + (size_t)tokens * heads * 16;
    std::vector<float> ts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_rows_kv_kernel<U>, dim3(heads * nsplit), dim3(threads), 0, st, src, vsrc, tokens, heads, nsplit, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = 2.0 * tokens * heads * 256.0;
    printf("{\"pattern\": \"K rows + V rows of one head per workgroup (two buffers), [token][%d heads][128 halfs]\", \"splits\": %d, \"workgroups\": %d, \"threads\": %d, "
           "\"loads_in_flight_per_wave\": %d, \"ms_median\": %.3f, \"GBps_median\": %.0f, \"GBps_best\": %.0f, \"frac_of_8TBps\": %.3f}\n", heads, nsplit,
           heads * nsplit, threads, U, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
}

template <int U>
static void run_rows(const f32x4* src, size_t n16, float* sink, int heads, int nsplit, int threads, hipStream_t st) {
    const int tokens = (int)(n16 / 16 / heads);
    std::vector<float> ts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_rows_kernel<U>, dim3(heads * nsplit), dim3(threads), 0, st, src, tokens, heads, nsplit, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)tokens * heads * 256.0;
    printf("{\"pattern\": \"256-byte rows of one head per workgroup, [token][%d heads][128 halfs]\", \"splits\": %d, \"workgroups\": %d, \"threads\": %d, "
           "\"loads_in_flight_per_wave\": %d, \"ms_median\": %.3f, \"GBps_median\": %.0f, \"GBps_best\": %.0f, \"frac_of_8TBps\": %.3f}\n", heads, nsplit,
           heads * nsplit, threads, U, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
}

template <int U>
static void run(const f32x4* src, size_t n16, float* sink, int cus, int wg_per_cu, int threads, hipStream_t st) {
    std::vector<float> ts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_kernel<U>, dim3(cus * wg_per_cu), dim3(threads), 0, st, src, n16, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)n16 * 16.0;
    printf("{\"workgroups_per_cu\": %d, \"threads\": %d, \"loads_in_flight_per_thread\": %d, \"ms_median\": %.3f, \"GBps_median\": %.0f, \"GBps_best\": %.0f, "
           "\"frac_of_8TBps\": %.3f}\n", wg_per_cu, threads, U, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6,
           bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    f32x4* src;
    float* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 0, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("{\"probe\": \"streaming read of a 2 GiB buffer, 16-byte non-temporal loads, grid-stride\", \"device\": \"%s\", \"compute_units\": %d}\n", prop.name, cus);
    for (int wg : {1, 2, 4, 8}) {
        run<4>(src, n16, sink, cus, wg, 256, st);
        run<8>(src, n16, sink, cus, wg, 256, st);
        run<16>(src, n16, sink, cus, wg, 256, st);
    }
    run<8>(src, n16, sink, cus, 2, 512, st);
    run<8>(src, n16, sink, cus, 2, 1024, st);
    run<16>(src, n16, sink, cus, 1, 1024, st);
    for (int nsplit : {8, 16}) {
        run_rows<4>(src, n16, sink, 32, nsplit, 256, st);
        run_rows<8>(src, n16, sink, 32, nsplit, 256, st);
        run_rows<16>(src, n16, sink, 32, nsplit, 256, st);
    }
    run_rows<8>(src, n16, sink, 4, 64, 256, st);         // a TP-8 rank: 4 heads, 64 splits
    run_rows_kv<4>(src, n16, sink, 32, 8, 256, st);
    run_rows_kv<8>(src, n16, sink, 32, 8, 256, st);
    run_rows_kv<16>(src, n16, sink, 32, 8, 256, st);
    run_rows_kv<8>(src, n16, sink, 32, 16, 256, st);
    run_frag_kv<1>(src, n16, sink, 32, 8, 256, st);
    run_frag_kv<2>(src, n16, sink, 32, 8, 256, st);
    run_frag_kv<1>(src, n16, sink, 32, 16, 256, st);
    for (int grid : {5504, 1024, 768, 512, 256}) {
        run_panels<4>(src, sink, grid, st);
        run_panels<8>(src, sink, grid, st);
        run_panels<16>(src, sink, grid, st);
    }
    for (int grid : {5504, 768, 512, 256}) {
        run_panels_tm<4>(src, sink, grid, st);
        run_panels_tm<8>(src, sink, grid, st);
        run_panels_tm<16>(src, sink, grid, st);
    }
    return 0;
}
