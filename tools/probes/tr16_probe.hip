// Lane <-> element map of ds_read_b64_tr_b16 (gfx950): LDS holds shorts 0..1023; lane l issues the read at byte address 8*l (where a plain
// ds_read_b64 would return elements 4l..4l+3). Prints what each lane receives.
// Result on MI355X (round 2): the 16 lanes of a group address 64 shorts; lane i of the group receives shorts {i, 16+i, 32+i, 48+i} of those 64,
// i.e. its element e comes from lane 4e + i/4 of the group, position i%4: a 4x16 -> 16x4 transpose per 16-lane group (groups 0-15, 16-31, ...).   build: hipcc --offload-arch=gfx950 -O3 -o tr16_probe tr16_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* o) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
    for (int e = 0; e < 4; ++e) o[threadIdx.x * 4 + e] = v[e];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
