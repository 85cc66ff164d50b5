// LDS bank behaviour of ds_read_b64_tr_b16 for the attention V tile read as the PV MFMA's A operand (V^T fragment) straight from a
// [64 keys][64 d] bf16 row-major tile (128-byte rows, filled by LDS-DMA from the q|k|v GEMM's output): which 16-byte-chunk swizzle makes
// the 16 reads of a tile conflict-free? Each variant: 8 waves x ITER x 16 tr reads (2 per (d-half, 16-key step) fragment), cycles by
// s_memtime, plus a correctness check of the fragment values against the index pattern stored in the tile.
//   lane (l31, lh), fragment (dd, J), half hf: reads 4 shorts at row key = 16J + 8lh + 4hf + ((l31 & 15) >> 2), column d = 32dd + 16*((l31 >> 4) & 1) + 4*(l31 & 3);
//   receives V[16J + 8lh + 4hf + e][32dd + l31] for e = 0..3 (tools/probes/tr16_probe.hip: per 16-lane group a 4x16 -> 16x4 transpose).
// build: hipcc --offload-arch=gfx950 -O3 -o tr16_layout_probe tr16_layout_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int swz(int variant, int key) {  // XOR on the 16-byte chunk index (0..7) of row `key`
    switch (variant) {
        case 0: return 0;
        case 1: return ((key >> 1) & 1) << 2;          // swap the 64-byte halves on key bit 1
        case 2: return (key & 3) << 1;
        case 3: return key & 7;
        case 4: return (key >> 1) & 7;                 // the K / V^T tiles' swizzle of the current kernel
        case 5: return ((key >> 1) & 1) << 2 | ((key >> 2) & 1) << 1;
        case 6: return ((key & 1) << 2) | (((key >> 1) & 1) << 1);
        default: return ((key >> 2) & 1) << 2;
    }
}

template <int VARIANT, bool TR>
__global__ __launch_bounds__(512) void k(unsigned long long* cyc, int* bad, int iters) {
    __shared__ __attribute__((aligned(16))) short tile[64 * 64 + 64 * 64];  // [key][d] image | V^T [d][key] image for the b128 baseline
    const int tid = threadIdx.x, lane = tid & 63;
    const int l31 = lane & 31, lh = lane >> 5;
    // fill: element (key, d) = key * 64 + d, stored at chunk (d / 8) ^ swz(key)
    for (int i = tid; i < 64 * 64; i += 512) {
        const int key = i >> 6, d = i & 63;
        tile[key * 64 + ((((d >> 3) ^ swz(VARIANT, key)) << 3) | (d & 7))] = (short)(key * 64 + d);
        // V^T image with the current kernel's swizzle: row d, chunk (key/8) ^ ((d>>1)&7)
        tile[4096 + d * 64 + ((((key >> 3) ^ ((d >> 1) & 7)) << 3) | (key & 7))] = (short)(key * 64 + d);
    }
    __syncthreads();
    int acc = 0, wrong = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                s8 f;
                if (TR) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int key = 16 * J + 8 * lh + 4 * hf + ((l31 & 15) >> 2);
                        const int d = 32 * dd + 16 * ((l31 >> 4) & 1) + 4 * (l31 & 3);
                        const short* a = tile + key * 64 + ((((d >> 3) ^ swz(VARIANT, key)) << 3) | (d & 7));
                        const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a);
                        f[4 * hf + 0] = v[0]; f[4 * hf + 1] = v[1]; f[4 * hf + 2] = v[2]; f[4 * hf + 3] = v[3];
                    }
                } else {
                    const int d = 32 * dd + l31;
                    f = *(const s8*)(tile + 4096 + d * 64 + ((((2 * J + lh) ^ ((d >> 1) & 7)) << 3)));
                }
                if (it == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) wrong += (f[e] != (short)((16 * J + 8 * lh + e) * 64 + 32 * dd + l31));
                }
                acc ^= f[0] ^ f[1] ^ f[2] ^ f[3] ^ f[4] ^ f[5] ^ f[6] ^ f[7];
                asm volatile("" : "+v"(acc));
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
    atomicAdd(bad, wrong + (acc == 0x7fffffff));
}

template <int V, bool TR>
void run(const char* name) {
    unsigned long long* d; int* bad;
    const int blocks = 256, iters = 2000;
    hipMalloc(&d, blocks * 8 * sizeof(unsigned long long)); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k<V, TR>), dim3(blocks), dim3(512), 0, 0, d, bad, iters);
    hipLaunchKernelGGL((k<V, TR>), dim3(blocks), dim3(512), 0, 0, d, bad, iters);
    hipDeviceSynchronize();
    unsigned long long h[blocks * 8]; int hb;
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks * 8; ++i) s += (double)h[i];
    const double per_wave = s / (blocks * 8);                       // cycles for iters * 8 fragments (16 KB read per wave and iteration)
    const double bytes = (double)iters * 8 * 1024;                   // per wave
    printf("%-34s wrong %6d   %8.0f cyc/wave   %.1f B/clk/CU (8 waves)   %.2f cyc per 1 KiB fragment and wave\n", name, hb / 2, per_wave, 8 * bytes / per_wave,
           per_wave / (iters * 8));
    hipFree(d); hipFree(bad);
}

int main() {
    run<0, false>("b128 V^T image (current kernel)");
    run<0, true>("tr, no swizzle");
    run<1, true>("tr, chunk ^= ((key>>1)&1)<<2");
    run<2, true>("tr, chunk ^= (key&3)<<1");
    run<3, true>("tr, chunk ^= key&7");
    run<4, true>("tr, chunk ^= (key>>1)&7");
    run<5, true>("tr, chunk ^= k1<<2 | k2<<1");
    run<6, true>("tr, chunk ^= k0<<2 | k1<<1");
    run<7, true>("tr, chunk ^= k2<<2");
    return 0;
}
