// Probe: block-scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3, E8M0 scales per 32 K-elements) on gfx950.
// Data operands: lane l holds row l&31, k-bytes [32*(l>>5), +32) (tools/probes/fp8_mfma_probe.hip). This probe feeds PER-LANE scale values
// (SA[l], SB[l] in byte OPSEL of the lane's scale VGPR) so that the host can find out which lane's scale reaches which (row, K-block).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int OPSEL>
__device__ void run(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    i32x8_t a = *(const i32x8_t*)(A + r * 64 + kh * 32);
    i32x8_t b = *(const i32x8_t*)(B + r * 64 + kh * 32);
    const int sa = (0x7f7f7f7f & ~(0xff << (8 * OPSEL))) | ((int)SA[lane] << (8 * OPSEL));
    const int sb = (0x7f7f7f7f & ~(0xff << (8 * OPSEL))) | ((int)SB[lane] << (8 * OPSEL));
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, sa, OPSEL, sb);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        D[row * 32 + r] = c[reg];
    }
}
// same with a hand-placed instruction: destination NOT overlapping any source (early clobber), wait states before it
extern "C" __global__ void probe_scale_asm(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    i32x8_t a = *(const i32x8_t*)(A + r * 64 + kh * 32);
    i32x8_t b = *(const i32x8_t*)(B + r * 64 + kh * 32);
    const int sa = 0x7f7f7f00 | (int)SA[lane];
    const int sb = 0x7f7f7f00 | (int)SB[lane];
    f32x16_t c;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        D[row * 32 + r] = c[reg];
    }
}
// ONE scale VGPR carrying the A scale in byte 0 and the B scale in byte 1, passed as BOTH scale operands with op_sel 0 / 1
extern "C" __global__ void probe_scale_packed(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    i32x8_t a = *(const i32x8_t*)(A + r * 64 + kh * 32);
    i32x8_t b = *(const i32x8_t*)(B + r * 64 + kh * 32);
    const int s = 0x7f7f0000 | (int)SA[lane] | ((int)SB[lane] << 8);
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s, 1, s);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        D[row * 32 + r] = c[reg];
    }
}
// every byte of the FIRST scale operand set to SA[lane]; the second operand a true unit vector; all op_sel values for the first
template <int OA>
__device__ void run_a(const uint8_t* A, const uint8_t* B, const uint8_t* SA, float* D) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    i32x8_t a = *(const i32x8_t*)(A + r * 64 + kh * 32);
    i32x8_t b = *(const i32x8_t*)(B + r * 64 + kh * 32);
    const int sa = 0x01010101 * (int)SA[lane];
    const int sb = 0x7f7f7f7f;
    f32x16_t c;
    if (OA == 0) asm volatile("s_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\ts_nop 7" : "=&v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    if (OA == 1) asm volatile("s_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel:[1,0,0] op_sel_hi:[0,0,0]\n\ts_nop 7" : "=&v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    if (OA == 2) asm volatile("s_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\ts_nop 7" : "=&v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    if (OA == 3) asm volatile("s_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 7" : "=&v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        D[row * 32 + r] = c[reg];
    }
}
extern "C" __global__ void probe_a0(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) { run_a<0>(A, B, SA, D); }
extern "C" __global__ void probe_a1(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) { run_a<1>(A, B, SA, D); }
extern "C" __global__ void probe_a2(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) { run_a<2>(A, B, SA, D); }
extern "C" __global__ void probe_a3(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) { run_a<3>(A, B, SA, D); }
// scale VGPRs swapped in the instruction: SB in the FIRST scale slot, SA in the second
extern "C" __global__ void probe_swapped(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) {
    const int lane = threadIdx.x;
    const int r = lane & 31, kh = lane >> 5;
    i32x8_t a = *(const i32x8_t*)(A + r * 64 + kh * 32);
    i32x8_t b = *(const i32x8_t*)(B + r * 64 + kh * 32);
    const int sa = 0x01010101 * (int)SA[lane], sb = 0x01010101 * (int)SB[lane];
    f32x16_t c;
    asm volatile("s_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %4, %3 op_sel_hi:[0,0,0]\n\ts_nop 7" : "=&v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        D[row * 32 + r] = c[reg];
    }
}
extern "C" __global__ void probe_scale0(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) { run<0>(A, B, SA, SB, D); }
extern "C" __global__ void probe_scale2(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) { run<2>(A, B, SA, SB, D); }
