// Probe: operand layout and raw rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3) on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// D[n][m] = sum_k A[n][k] * B[m][k]; A, B: [32][64] fp8 bytes row-major. Hypothesis: lane l holds row l&31, k in [32*(l>>5), +32).
extern "C" __global__ void probe_layout(const uint8_t* A, const uint8_t* B, float* D) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    i32x8_t a = *(const i32x8_t*)(A + r * 64 + kh * 32);
    i32x8_t b = *(const i32x8_t*)(B + r * 64 + kh * 32);
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    // C/D map (same as bf16 32x32): col = lane&31 (B row), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (A row)
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        D[row * 32 + r] = c[reg];
    }
}

extern "C" __global__ __launch_bounds__(256) void probe_rate(float* out, int iters) {
    i32x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x30303030 + i; }
    f32x16_t c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// quantisation builtin check: f32 -> fp8 e4m3 (OCP on gfx950), two values per call
extern "C" __global__ void probe_cvt(const float* x, uint8_t* q, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        q[2 * i] = packed & 0xff;
        q[2 * i + 1] = (packed >> 8) & 0xff;
    }
}
