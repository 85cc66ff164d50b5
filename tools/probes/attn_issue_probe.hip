// Issue-rate model of ONE software-pipelined attention "unit" (gfx950): what does a SIMD sustain when a wave interleaves the 8 MFMAs of a
// (32-key x 32-query) unit -- 4 score MFMAs of the NEXT unit chained on one accumulator, 4 PV MFMAs of the PREVIOUS unit on two -- with the
// softmax VALU work of the CURRENT unit (16 v_exp_f32, 8 v_cvt_pk_bf16_f32, 16 v_add_f32 = 5 fillers per MFMA gap), registers only or with the
// unit's LDS fragment reads, at one or two waves per SIMD?  d = 64 attention has 160 VALU per 32 MFMAs, exactly the 5-per-gap budget the
// guide quotes for a one-wave-per-SIMD stream (MI355X_MICROARCH.md "Per-instruction cycle constants").
//   build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o attn_issue_probe attn_issue_probe.hip ; run: ./attn_issue_probe
// (launch bound 2 waves per SIMD = a 256-register budget: the compiler then keeps the score accumulators in arch VGPRs, where v_exp can read them;
//  with the 512-register budget of a declared one-wave-per-SIMD kernel it parks them in AGPRs and adds a v_accvgpr_read per score)
// Output per variant: ns and shader cycles (s_memtime) per MFMA per SIMD; 32 cycles = the matrix pipe's own rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ uint32_t pk(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// FILL: VALU fillers per MFMA gap (0, 4, 5, 6, 7); LDS: 0 = fragments in registers, 1 = per unit 2 ds_read_b128 (K) + 4 ds_read_b64 (V) as the
// real loop would issue them (one read after each of six MFMAs); WPS: waves per SIMD (block = 256 * WPS threads, one block per CU)
template <int FILL, int LDS, int WPS>
__global__ __launch_bounds__(256 * WPS, 2) void unit_probe(float* out, unsigned long long* cyc, int units) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((float*)smem)[i] = 1e-3f * (i & 63);
    __syncthreads();
    bf16x8 kf[2][4], qf[4], vf[2][4];   // [h]: the fragment set unit h computes from; with LDS, unit h's reads fill the set of unit h ^ 1
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 8; ++r) { kf[h][i][r] = (short)(0x3c00 + lane + i); qf[i][r] = (short)(0x3b00 + r); vf[h][i][r] = (short)(0x3a00 + 2 * r + i); }
    f32x16 s[2], o[2];
    for (int r = 0; r < 16; ++r) { s[0][r] = 1e-3f * r; s[1][r] = 2e-3f * r; o[0][r] = 0.f; o[1][r] = 0.f; }
    uint32_t p[2][8];
    for (int i = 0; i < 8; ++i) { p[0][i] = 0x3c003c00u; p[1][i] = 0x3c003c00u; }
    float sum0 = 0.f, sum1 = 0.f, extra = 1.f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* lk = smem + lane * 16;
    const char* lv = smem + 16384 + lane * 8;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int u = 0; u < units; u += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // unit u + h: scores of the next unit into s[h ^ 1], softmax of s[h] into p[h], PV from p[h ^ 1]
            f32x16& sc = s[h];
            f32x16& sn = s[h ^ 1];
            uint32_t* pc = p[h];
            const uint32_t* pp = p[h ^ 1];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                // ---- the MFMA of this gap: even = score k-step g/2 (chained on sn), odd = PV (d = (g>>1)&1, slot = g>>2) ----
                if ((g & 1) == 0) {
                    const int ks = g >> 1;
                    sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[h][ks], qf[ks], ks == 0 ? zero : sn, 0, 0, 0);
                } else {
                    const int d = (g >> 1) & 1, sl = g >> 2;
                    bf16x8 pf;
                    const uint32_t* q4 = pp + 4 * sl;
                    pf[0] = (short)q4[0]; pf[1] = (short)(q4[0] >> 16); pf[2] = (short)q4[1]; pf[3] = (short)(q4[1] >> 16);
                    pf[4] = (short)q4[2]; pf[5] = (short)(q4[2] >> 16); pf[6] = (short)q4[3]; pf[7] = (short)(q4[3] >> 16);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[h][2 * d + sl], pf, o[d], 0, 0, 0);
                }
                // ---- fillers: 2 exp + 1 cvt + 2 add on the two scores 2g, 2g + 1 of the current unit ----
                if (FILL >= 4) {
                    float e0 = __builtin_amdgcn_exp2f(sc[2 * g]), e1 = __builtin_amdgcn_exp2f(sc[2 * g + 1]);
                    pc[g] = pk(e0, e1);
                    sum0 += e0;
                    if (FILL >= 5) sum1 += e1;
                    if (FILL >= 6) extra = fmaf(extra, 1.0001f, e0);
                    if (FILL >= 7) extra = fmaf(extra, 0.9999f, e1);
                }
                if (LDS) {   // the real loop's reads per PAIR of units: 4 ds_read_b128 (K fragments of a 32-key block) + 8 ds_read_b64 (V^T halves)
                    typedef short s4 __attribute__((ext_vector_type(4)));
                    if (h == 0 && g < 4) kf[1][g] = *(const bf16x8*)(lk + (u & 6) * 1024 + g * 4096);
                    else if (h == 0 || g < 4) {
                        const int f = g & 3, half = (h == 0) ? 0 : 4;
                        const s4 w = *(const s4*)(lv + (u & 6) * 512 + f * 2048 + half * 256);
                        vf[h ^ 1][f][half + 0] = w[0]; vf[h ^ 1][f][half + 1] = w[1]; vf[h ^ 1][f][half + 2] = w[2]; vf[h ^ 1][f][half + 3] = w[3];
                    }
                }
                SB();
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = sum0 + sum1 + extra;
    for (int r = 0; r < 16; ++r) acc += o[0][r] + o[1][r] + s[0][r] + s[1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FILL, int LDS, int WPS>
void run(const char* name, float* out, unsigned long long* cyc, unsigned long long* hcyc, int per_cu = 1) {
    const int units = 20000, grid = 256 * per_cu;
    unit_probe<FILL, LDS, WPS><<<grid, 256 * WPS>>>(out, cyc, 64);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    unit_probe<FILL, LDS, WPS><<<grid, 256 * WPS>>>(out, cyc, units);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hcyc, cyc, grid * sizeof(*hcyc), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < grid; ++i) c += (double)hcyc[i];
    c /= grid;
    const double mf = (double)units * 8 * WPS * per_cu;   // MFMAs per SIMD
    printf("%-58s %7.2f ns/MFMA/SIMD  %6.1f memtime-ticks per MFMA of one wave  (%.3f ms, %.0f TFLOP/s-equivalent)\n", name, ms * 1e6 / mf, c / ((double)units * 8), ms,
           mf * 4 * 256 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; unsigned long long *cyc, *hcyc;
    hipMalloc(&out, 512 * 512 * sizeof(float));
    hipMalloc(&cyc, 512 * sizeof(*cyc));
    hcyc = (unsigned long long*)malloc(512 * sizeof(*hcyc));
    run<0, 0, 1>("MFMA only, 1 wave/SIMD", out, cyc, hcyc);
    run<0, 0, 2>("MFMA only, 2 waves/SIMD", out, cyc, hcyc);
    run<4, 0, 1>("4 fillers/gap (2 exp, cvt, add), regs, 1 wave/SIMD", out, cyc, hcyc);
    run<5, 0, 1>("5 fillers/gap (2 exp, cvt, 2 add), regs, 1 wave/SIMD", out, cyc, hcyc);
    run<6, 0, 1>("6 fillers/gap, regs, 1 wave/SIMD", out, cyc, hcyc);
    run<7, 0, 1>("7 fillers/gap, regs, 1 wave/SIMD", out, cyc, hcyc);
    run<5, 1, 1>("5 fillers/gap + LDS fragment reads, 1 wave/SIMD", out, cyc, hcyc);
    run<5, 0, 2>("5 fillers/gap, regs, 2 waves/SIMD", out, cyc, hcyc);
    run<5, 1, 2>("5 fillers/gap + LDS fragment reads, 2 waves/SIMD", out, cyc, hcyc);
    run<7, 1, 2>("7 fillers/gap + LDS fragment reads, 2 waves/SIMD", out, cyc, hcyc);
    run<5, 1, 1>("5 fillers/gap + LDS reads, 2 four-wave workgroups per CU", out, cyc, hcyc, 2);
    run<6, 1, 1>("6 fillers/gap + LDS fragment reads, 1 wave/SIMD", out, cyc, hcyc);
    run<6, 1, 2>("6 fillers/gap + LDS fragment reads, 2 waves/SIMD", out, cyc, hcyc);
    return 0;
}
