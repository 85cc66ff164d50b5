// Issue-rate probe for the VALU / transcendental / MFMA instructions the spatial attention loop is made of (gfx950).
// Every kernel runs ITERS x 8 independent copies of one instruction (or one mix) per wave; WAVES waves per SIMD; all 256 CUs.
// Output: cycles per instruction per wave at the measured shader clock (calibrated on v_fma_f32 = 4 cycles per wave64, asserted
// by the v_mfma line: 32x32x16 bf16 = 8 passes = 32 cycles).   build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 0.5f + i; }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    bf16x8 fa, fb;
    for (int r = 0; r < 8; ++r) { fa[r] = (__bf16)(threadIdx.x * 1e-3f); fb[r] = (__bf16)(0.25f); }
    float c2[8][2];
    for (int i = 0; i < 8; ++i) { c2[i][0] = a[i]; c2[i][1] = b[i]; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f2{a[i], b[i]};
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 1) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP8(S)
#undef S
        } else if (OP == 2) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            REP8(S)
#undef S
        } else if (OP == 3) {
#define S(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 4) {
#define S(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 5) {
#define S(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 6) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
            REP8(S)
#undef S
        } else if (OP == 7) {  // 2 independent MFMA chains x 4
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
            }
        } else if (OP == 13) {  // ONE dependent MFMA chain (8 per iteration)
            for (int j = 0; j < 8; ++j) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        } else if (OP == 8) {  // exp and fma interleaved in ONE wave: serial (sum) or co-executed (max)?
#define S(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(a[i]), "+v"(c2[i][0]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 9) {  // 8 MFMA + 8 exp + 8 fma in one wave, MFMAs first
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
            }
#define S(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(a[i]), "+v"(c2[i][0]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 10) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(S)
#undef S
        } else if (OP == 11) {
#define S(i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            REP8(S)
#undef S
        } else if (OP == 12) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1\n\tv_ldexp_f32 %0, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(it));
            REP8(S)
#undef S
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + c2[i][0] + p[i].x + p[i].y;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

typedef void (*kern_t)(float*, int);

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const char* names[14] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_dot2c_f32_bf16", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_pk_add_f32",
                           "v_mfma_f32_32x32x16_bf16", "exp+fma pairs (per pair)", "8 mfma + 8 exp + 8 fma (per group of 3)", "v_add_f32",
                           "v_exp_f16", "v_mul+v_ldexp (per pair)", "v_mfma one dependent chain"};
    kern_t ks[] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>, probe<11>, probe<12>, probe<13>};
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double fma_ns[5] = {0};
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        for (int k = 0; k < 14; ++k) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(ks[k], dim3(256), dim3(256 * wps), 0, 0, out, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double ns_per_inst = best * 1e6 / ((double)iters * 8 * wps);  // per instruction per wave, waves on a SIMD serialised
            if (k == 0) fma_ns[wps] = ns_per_inst;
            printf("waves/SIMD %d  %-42s %8.3f ns/inst/wave  = %6.2f cycles (v_fma_f32 := 4)\n", wps, names[k], ns_per_inst,
                   4.0 * ns_per_inst / fma_ns[wps]);
        }
    }
    // sustained MFMA rate: the same 2-chain MFMA loop, 4 waves per SIMD, for longer and longer launches (does the clock hold?)
    for (int it = 20000; it <= 20000000; it *= 10) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(ks[7], dim3(256), dim3(1024), 0, 0, out, it);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("sustained mfma: %9d iters  %9.3f ms  %7.3f ns per MFMA per SIMD  -> %6.1f TFLOP/s chip\n", it, ms, ms * 1e6 / ((double)it * 8 * 4),
               (double)it * 8 * 4 * 1024 * 32768.0 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
