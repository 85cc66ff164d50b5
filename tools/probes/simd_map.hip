// Which SIMD does wave w of a 512-thread workgroup land on? (HW_REG_HW_ID.simd_id, bits [5:4])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
    __shared__ char big[140 * 1024];  // force one workgroup per CU like the GEMM
    big[threadIdx.x] = 0;
    const int wave = threadIdx.x >> 6;
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // whole HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (int)hw;
}
int main() {
    int* d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 0, 0, d);
    int h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) {
        printf("wg %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd=%d waveslot=%d cu=%d", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
