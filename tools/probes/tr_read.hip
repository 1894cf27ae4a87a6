// Probe of ds_read_b64_tr_b16 semantics on gfx950: prints, for every lane, which LDS elements it received.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;  // element (row, col) = row*64 + col
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    // group g reads the block rows 4g..4g+3, columns 0..15; lane 4q+p points at row q, columns 4p..4p+3
    const short* addr = lds + (4 * g + q) * 64 + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
