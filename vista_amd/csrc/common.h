// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the Vista denoising hot path.
// Everything here is written for MI355X only: 64-lane wavefronts, MFMA 32x32x16 bf16, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // one MFMA A/B operand (8 bf16, 4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

#define VK_WAVE 64

// ---- the 16-bit STORAGE type of the build ----------------------------------------------------------------------------------------------------
// Default build (libvista_hip.so): every activation and weight is bf16 (BASELINE config 2 names bf16). -DVK_F16=1 (libvista_hip_f16.so, round 6):
// the same kernels with IEEE fp16 storage and v_mfma_f32_32x32x16_f16 -- the reference's own autocast width (sample_utils.py:301-303), 3 more
// mantissa bits at the same MFMA rate and the same instruction counts (v_cvt_pk_f16_f32 / v_cvt_f32_f16 / v_dot2c_f32_f16 exist on gfx950).
// The helpers below keep their historical "bf16" names: they pack / unpack THE STORAGE TYPE. What stays bf16 in both builds has its own names
// (*_bf16x): the softmax numerators P and the V operand of the attention P.V product -- the zero-base softmax needs bf16's exponent range
// (attention.hip), so the q|k|v projections of an fp16 build write their V column block as bf16 (VkGemmDesc.alt_cols_from).
#ifndef VK_F16
#define VK_F16 0
#endif
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

__device__ __forceinline__ float bf16x_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16x_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// two fp32 -> packed bf16x2 (round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x(float a, float b) {
    f32x2_t v = {a, b};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ f32x16_t vk_mfma_bf16x(bf16x8_t a, bf16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

#if VK_F16
__device__ __forceinline__ float bf16_lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[0]; }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[1]; }
__device__ __forceinline__ float bf16_to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
// two fp32 -> packed fp16x2 (round-to-nearest-even: v_cvt_pk_f16_f32; values beyond 65504 become inf -- the reference's autocast does the same)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2_t v = {a, b};
    f16x2_t r = __builtin_convertvector(v, f16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ f32x16_t vk_mfma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// sum += x.lo * y.lo + x.hi * y.hi on packed pairs of the storage type (v_dot2c_f32_f16)
__device__ __forceinline__ float vk_dot2(uint32_t x, uint32_t y, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, x), __builtin_bit_cast(f16x2_t, y), acc, false);
}
#define VK_ONE_PAIR 0x3c003c00u   // (1, 1) in the storage type
#else
__device__ __forceinline__ float bf16_lo(uint32_t u) { return bf16x_lo(u); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return bf16x_hi(u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t u) { return __builtin_bit_cast(float, ((uint32_t)u) << 16); }
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return pack_bf16x(a, b); }
__device__ __forceinline__ f32x16_t vk_mfma(bf16x8_t a, bf16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float vk_dot2(uint32_t x, uint32_t y, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x), __builtin_bit_cast(bf16x2_t, y), acc, false);
}
#define VK_ONE_PAIR 0x3f803f80u
#endif
__device__ __forceinline__ uint16_t f32_to_bf16(float a) { return (uint16_t)(pack_bf16(a, 0.f) & 0xffffu); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x);
    f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z);
    f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16(f[0], f[1]); v.y = pack_bf16(f[2], f[3]);
    v.z = pack_bf16(f[4], f[5]); v.w = pack_bf16(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// erf-form GELU (F.gelu default, reference vwm/modules/attention.py:92): x * Phi(x) with the normal CDF as a logistic of an odd quintic,
// Phi(x) ~= 1 / (1 + exp(-x (a + b x^2 + c x^4))), a = 1.59501577, b = 7.40112920e-2, c = -7.03033577e-4 (minimax fit of x * Phi(x) over
// |x| <= 12; x^2 clamped at 50 where the quintic would turn around and Phi is 0 / 1 in fp32 anyway). |x Phi(x) - gelu(x)| <= 2.6e-5
// everywhere in fp32 (tests/test_host_cpu.py evaluates this very formula) -- a hundred times below the bf16 resolution of the values the
// GEGLU epilogue stores. 7 plain VALU ops + 2 transcendentals per gate, against 15 + 2 for the erfc form (Abramowitz-Stegun 7.1.26) it
// replaces: the GEGLU epilogue is VALU-bound with every wave of the CU in it at once, and the gelu was half of its instructions.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float u = fminf(x * x, 50.f);
    const float t = fmaf(u, fmaf(u, 0.0010142630f, -0.10677572f), -2.3011212f);  // -(a + b u + c u^2) * log2(e)
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * t));
}

// XCD-aware, bijective block remap (8 XCDs, block b is observed on XCD b%8): gives each XCD a
// contiguous range of logical tile ids so neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int nx = 8;
    int xcd = bid % nx, slot = bid / nx;
    int q = nblk / nx, r = nblk % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// error codes returned by the C-ABI launchers
#define VK_OK 0
#define VK_EINVAL (-22)
#define VK_ELAUNCH (-5)

#define VK_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return VK_ELAUNCH; } while (0)
