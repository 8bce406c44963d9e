// Probe of ds_read_b64_tr_b16 on gfx950: every lane passes the address of 4 consecutive halfs; LDS holds its own index.
// Prints, per lane, the 4 LDS indices the instruction returned.  hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 fp16x4 __attribute__((vector_size(8)));
__global__ void k(float* out, int mode) {
    __shared__ __fp16 sm[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) sm[i] = (__fp16)(float)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off;
    if (mode == 0) off = l * 4;                                            // lane-linear
    else off = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16;         // [4 rows][stride 64] blocks, group g at col 16g
    fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(sm + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)r[j];
}
int main() {
    float* d; float h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4.0f %4.0f %4.0f %4.0f\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    return 0;
}
