// fp8 (OCP e4m3) dense GEMM for the UNet's 1x1 / Linear projections: BASELINE config 5 ("fp8 1x1 conv-as-GEMM path").
//
//   out[m][n] = epilogue( a_scale[m] * w_scale[n] * sum_k Aq[m][k] * Wq[n][k] )
//
// Aq: activations quantised per ROW (token) by vk_quantize_rows_fp8, Wq: weights quantised per OUTPUT CHANNEL at pack time.
// Same structure as gemm.hip (LDS-DMA 2-stage pipeline, 128-byte swizzled LDS rows, swapped MFMA orientation, shared fused
// epilogue) with 128 fp8 K-elements per row and v_mfma_scale_f32_32x32x64_f8f6f4 (scale operands 0 = unscaled fp8 x fp8):
// per K-step a wave issues the same number of MFMA cycles as the bf16 kernel for twice the FLOPs, and stages half the bytes
// per FLOP through the LDS-DMA path that bounds the bf16 kernel.
// Operand layout of the 32x32x64 instruction (probed, tools/probes/fp8_mfma_probe.hip, tools/probes/mx_scale_probe.hip): lane l holds row
// l&31; its VGPRs 0-3 are 16 k-bytes of the instruction's FIRST 32-element K-block and VGPRs 4-7 are 16 k-bytes of the SECOND (lanes r and
// r+32 together hold both halves of each block); C/D layout is the bf16 32x32 one. Block scales (MX, E8M0 = 2^(e-127) per 32 K-elements):
// the scale of (row r, block b) is byte op_sel of the scale VGPR of lane r + 32b. The fragment loader below therefore hands lane half
// h the 16-B chunks (h, 2 + h) of each 64-byte K-slab: chunks (0,1) = block 0 and (2,3) = block 1 are 32 CONSECUTIVE k-bytes in memory.
//
// Three ways to scale the activations: per ROW (a_scale, applied to the accumulators after the K-loop: inputs quantised by
// vk_quantize_rows_fp8 / vk_layernorm_quant_fp8), per n consecutive rows (a_div: the per-image scales of vk_groupnorm_silu_fp8, for the
// implicit-GEMM convolution loaders below -- a conv pixel only sums taps of its own image, so the scale factors out of the K-sum) or per
// (row, 32-element block) in the MFMA itself (a_mx: the producer GEMM's epilogue quantised its own output locally -- GEGLU with mx_out --
// so no quantisation pass exists between the two FeedForward GEMMs).
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ uint4 g_zero16_f8;

struct F8Args {
    const float* a_scale;  // per-row activation scale [M] (a_div <= 1) or one per a_div consecutive rows [ceil(M / a_div)], or NULL (then a_mx)
    int a_div;
    const float* w_scale;  // [N padded] per-output-channel weight scale (GEGLU: in packed row order)
    int K;                 // real K (multiple of 16); p.K = weight row stride in bytes (multiple of 128, zero-filled past K)
    const uint8_t* a_mx;   // E8M0 block scales of the activations, [M][ld_mx] bytes, one per 32 K-elements, or NULL
    int ld_mx;
    uint8_t* mx_out;       // EPI_GEGLU: E8M0 block scales of the fp8 output written to p.out ([M][ld_mx_out], one per 32 output columns), or NULL
    int ld_mx_out;
};

// lower lane: the 16 output columns of fragment A; upper lane: those of fragment B (each lane holds columns {0-3, 8-11} + 4*lh of both)
__device__ __forceinline__ uint4 widen_quads_fp8(uint32_t a_g0, uint32_t a_g1, uint32_t b_g0, uint32_t b_g1) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(a_g0, b_g0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a_g1, b_g1, false, false);
    return make_uint4(r0[0], r0[1], r1[0], r1[1]);
}

template <int AMODE, int EPI, bool OUT_F32, bool AMX, int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 16) ? 4 : 2) void gemm_fp8_kernel(const VkGemmDesc p, const F8Args q) {
    constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;
    constexpr int AP = BM / RPP, WP = (BN + RPP - 1) / RPP;  // the last W pass may be partial (320 rows, 128 per pass)
    static_assert(BM % RPP == 0 && BN % 8 == 0, "A rows must be a multiple of the staging pass, W rows of a wave's 8-row slice");
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    constexpr int FX = FN, FY = FM;  // X = weights (MFMA row operand), Y = activations
    // two tile stages, then the LDS-staged epilogue's vectors (gemm_common.h) and the tile's BN per-channel weight scales
    constexpr int EV_OFF = 2 * STAGE_BYTES, WS_OFF = EV_OFF + epi_vec_floats(BN) * 4;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + epi_vec_floats(BN) * 4 + BN * 4];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tilesN = (p.N + BN - 1) / BN;
    const int tilesM = (p.M + BM - 1) / BM;
    const int logical = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int tn = logical % tilesN, tm = logical / tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const uint8_t* __restrict__ Ag = (const uint8_t*)p.A;
    const uint8_t* __restrict__ Wg = (const uint8_t*)p.Wt;
    const uint8_t* zsrc = (const uint8_t*)&g_zero16_f8;  // formed once and opaque: see gemm.hip
    asm volatile("" : "+s"(zsrc));

    const int lc = tid & 7, lr = tid >> 3;
    const int lsrc = lc ^ ((lr >> 1) & 7);  // logical 16-B chunk this lane fetches so that the swizzled image lands (gemm.hip)
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    // 32-bit byte offsets from the (uniform) tensor bases: half the address registers of per-lane pointers (the entry point checks that
    // both tensors are smaller than 4 GiB)
    uint32_t woff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) woff[i] = (uint32_t)(n0 + lr + RPP * i) * (uint32_t)p.K + lsrc * 16;
    // A-row state. Implicit-GEMM loaders (AMODE_CONV3X3 / AMODE_TEMPORAL3: the ResBlock convolutions on fp8 GroupNorm output): the K axis is
    // the bf16 kernels' [Cin/64][tap][64] order (gemm.hip), in bytes; a 128-byte K-step covers TWO (slab, tap) units, and since a lane always
    // stages the same 16-byte chunk of a row, the unit it works on (first or second of the step) and its 16-channel offset inside the unit
    // are per-lane constants: each lane steps its own (tap, slab) pair by two units per K-step.
    constexpr int NTAPS = (AMODE == AMODE_CONV3X3) ? 9 : (AMODE == AMODE_TEMPORAL3) ? 3 : 1;
    uint32_t aoff[AP];
    int a_y0[AP], a_x0[AP];  // conv: (row, column) of tap (0, 0) / frame index; rows past M get coordinates no tap can bring into range
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int m = m0 + lr + RPP * i;
        const bool row_ok = m < p.M;
        if (!row_ok) m = p.M - 1;
        if (AMODE == AMODE_DENSE) {
            aoff[i] = (uint32_t)m * (uint32_t)p.lda + lsrc * 16;
            a_y0[i] = a_x0[i] = 0;
        } else if (AMODE == AMODE_CONV3X3) {  // stride 1, pad 1
            const int hw = p.H * p.Wd;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int oy = rem / p.Wd;
            a_y0[i] = row_ok ? oy - 1 : -(1 << 20);
            a_x0[i] = rem - oy * p.Wd - 1;
            aoff[i] = (uint32_t)img * (uint32_t)(hw * p.Cin) + (lsrc & 3) * 16;
        } else {  // TEMPORAL3: m = (b*T + t)*S + s, zero padding at the clip ends
            a_y0[i] = row_ok ? (m / p.S) % p.T : -(1 << 20);
            a_x0[i] = 0;
            aoff[i] = (uint32_t)m * (uint32_t)p.Cin + (lsrc & 3) * 16;
        }
    }
    int tap_l = lsrc >> 2, cbase_l = 0;  // this lane's next (tap, channel base); units past the last slab (odd unit count) read zeros

    auto dma_tile = [&](int kt, int stage) {
        const int k0 = kt * 128;
        char* sA = smem + stage * STAGE_BYTES + wave_u * 1024;
        char* sW = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            if (RPP * (i + 1) <= BN || RPP * i + 8 * wave_u < BN)  // wave-uniform: waves past the tile's last row skip the partial pass
                __builtin_amdgcn_global_load_lds((gptr_t)(Wg + (woff[i] + (uint32_t)k0)), (lptr_t)(sW + i * RPP * 128), 16, 0, 0);
        if (AMODE == AMODE_DENSE) {
            const bool inside = k0 + lsrc * 16 < q.K;  // K tail (K = 320: 2.5 K-steps): chunks past the row end are zero-filled
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const uint8_t* src = inside ? Ag + (aoff[i] + (uint32_t)k0) : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        } else if (AMODE == AMODE_CONV3X3) {
            const bool inside = cbase_l < p.Cin;
            const int ky = tap_l / 3, kx = tap_l - ky * 3;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int iy = a_y0[i] + ky, ix = a_x0[i] + kx;
                const bool ok = inside && iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
                const uint8_t* src = ok ? Ag + (aoff[i] + (uint32_t)((iy * p.Wd + ix) * p.Cin + cbase_l)) : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        } else {
            const bool inside = cbase_l < p.Cin;
            const int dt = tap_l - 1;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int t = a_y0[i] + dt;
                const bool ok = inside && t >= 0 && t < p.T;
                const uint8_t* src = ok ? Ag + (aoff[i] + (uint32_t)(dt * p.S * p.Cin + cbase_l)) : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        }
        if (AMODE != AMODE_DENSE) {
            tap_l += 2;
            if (tap_l >= NTAPS) { tap_l -= NTAPS; cbase_l += 64; }
        }
    };

    f32x16_t acc[FX][FY];
#pragma unroll
    for (int i = 0; i < FX; ++i)
#pragma unroll
        for (int j = 0; j < FY; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int MW = FM * 32, NW = FN * 32;
    const int xoff = wn * NW, yoff = wm * MW;
    const int sw = (l31 >> 1) & 7;
    const int xrow_off = A_BYTES + (xoff + l31) * 128, yrow_off = (yoff + l31) * 128;

    auto load_frag = [&](const char* rowp, int ks) {
        const int c = ks * 4 + lh;  // lane half h holds chunk h of K-block 0 (VGPRs 0-3) and chunk 2+h of K-block 1 (VGPRs 4-7) of the slab
        const i32x4_t lo = *(const i32x4_t*)(rowp + ((c ^ sw) << 4));
        const i32x4_t hi = *(const i32x4_t*)(rowp + (((c + 2) ^ sw) << 4));
        i32x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    // MX activation scales (AMX): per K-step one dword per activation row = the 4 block scales of its 128 k-bytes; a lane needs blocks
    // (2*ks + lh), so the dword is pre-shifted by 8*lh and the instruction picks byte 0 (ks = 0) / byte 2 (ks = 1) through op_sel. The
    // dword of K-step kt+1 is fetched while kt computes. Weights carry per-channel fp32 scales applied after the loop: unit block scales.
    int unit_scale = 0x7f7f7f7f;
    asm volatile("" : "+v"(unit_scale));  // opaque: a literal scale operand would be re-interpreted by the compiler as an fp32 constant
    uint32_t mxoff[FY];
    int ysc[FY];
    if constexpr (AMX) {
#pragma unroll
        for (int fj = 0; fj < FY; ++fj) {
            const int m = m0 + wm * MW + fj * 32 + l31;
            mxoff[fj] = (uint32_t)(m < p.M ? m : p.M - 1) * (uint32_t)q.ld_mx;
            ysc[fj] = (int)(*(const uint32_t*)(q.a_mx + mxoff[fj]) >> (8 * lh));
        }
    }
    auto compute = [&](int stage, int ktn) {
        const char* sb = smem + stage * STAGE_BYTES;
        int ynext[FY];
        if constexpr (AMX) {
#pragma unroll
            for (int fj = 0; fj < FY; ++fj) ynext[fj] = (int)(*(const uint32_t*)(q.a_mx + mxoff[fj] + ktn * 4) >> (8 * lh));
        }
        if constexpr (WM * WN == 16 && FX * FY > 4) {
            // sixteen 32x160 wave tiles (N = 320 convolutions): 5 + 1 eight-register fragments next to 80 accumulators do not fit 128
            // VGPRs at once -- the activation fragment stays, the weight fragments pass through one at a time
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                i32x8_t yf[FY];
#pragma unroll
                for (int f = 0; f < FY; ++f) yf[f] = load_frag(sb + yrow_off + f * 32 * 128, ks);
#pragma unroll
                for (int fi = 0; fi < FX; ++fi) {
                    const i32x8_t xf = load_frag(sb + xrow_off + fi * 32 * 128, ks);
#pragma unroll
                    for (int fj = 0; fj < FY; ++fj)
                        acc[fi][fj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf, yf[fj], acc[fi][fj], 0, 0, 0, 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x8_t xf[FX], yf[FY];
#pragma unroll
            for (int f = 0; f < FX; ++f) xf[f] = load_frag(sb + xrow_off + f * 32 * 128, ks);
#pragma unroll
            for (int f = 0; f < FY; ++f) yf[f] = load_frag(sb + yrow_off + f * 32 * 128, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fi = 0; fi < FX; ++fi)
#pragma unroll
                for (int fj = 0; fj < FY; ++fj) {
                    if constexpr (!AMX) acc[fi][fj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[fi], yf[fj], acc[fi][fj], 0, 0, 0, 0, 0, 0);
                    else if (ks == 0) acc[fi][fj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[fi], yf[fj], acc[fi][fj], 0, 0, 0, unit_scale, 0, ysc[fj]);
                    else acc[fi][fj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[fi], yf[fj], acc[fi][fj], 0, 0, 0, unit_scale, 2, ysc[fj]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (AMX) {
#pragma unroll
            for (int fj = 0; fj < FY; ++fj) ysc[fj] = ynext[fj];
        }
    };

    const int nk = p.K / 128;
    float* const epi_vec = (float*)(smem + EV_OFF);
    float* const ws_lds = (float*)(smem + WS_OFF);
    const EpiPlan eplan = epi_plan<EPI, OUT_F32, BM, BN>(p, m0, n0);
    dma_tile(0, 0);
    if (eplan.fast) epi_stage_vectors<BN, NT>(p, epi_vec, n0, eplan, tid);  // under the first tile's DMA
    for (int i = tid; i < BN / 4; i += NT) *(float4*)(ws_lds + 4 * i) = *(const float4*)(q.w_scale + n0 + 4 * i);  // (padded to the tile)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) dma_tile(kt + 1, stage ^ 1);
        compute(stage, kt + 1 < nk ? kt + 1 : kt);
        __syncthreads();
    }

    // dequantise in registers (row scale of the lane's output row x channel scales of its column quads), then the shared epilogue
#pragma unroll
    for (int fj = 0; fj < FY; ++fj) {
        const int m = m0 + wm * MW + fj * 32 + l31;
        const int mc = m < p.M ? m : p.M - 1;
        const float sa = AMX ? 1.f : q.a_scale[AMODE != AMODE_DENSE ? mc / q.a_div : mc];  // convolutions: one scale per image group  // MX activations were scaled inside the MFMA
#pragma unroll
        for (int fi = 0; fi < FX; ++fi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 s4 = *(const float4*)(ws_lds + wn * NW + fi * 32 + 8 * g + 4 * lh);
                acc[fi][fj][4 * g + 0] *= sa * s4.x;
                acc[fi][fj][4 * g + 1] *= sa * s4.y;
                acc[fi][fj][4 * g + 2] *= sa * s4.z;
                acc[fi][fj][4 * g + 3] *= sa * s4.w;
            }
    }
    if constexpr (EPI == EPI_GEGLU && (FX % 2) == 0) {
        if (q.mx_out != nullptr) {
            // GEGLU with MX fp8 output: value*gelu(gate) as in the shared epilogue, then every 32 output columns of a row (= two
            // fragments x {this lane, lane^32}) get their own power-of-two scale 2^e >= max|h| / 448 and leave as e4m3 bytes: the
            // FeedForward's second GEMM consumes them through a_mx with no quantisation pass in between.
            const float* __restrict__ bias = p.bias;
            const int nout = p.N >> 1;
#pragma unroll
            for (int fj = 0; fj < FY; ++fj) {
                const int m = m0 + wm * MW + fj * 32 + l31;
                if (m >= p.M) continue;  // both lanes of a row leave together
#pragma unroll
                for (int fp = 0; fp < FX; fp += 2) {
                    float h[2][2][4];
                    float amax = 0.f;
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int nfrag = n0 + wn * NW + (fp + f) * 32;
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const int np = nfrag + 8 * g + 4 * lh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float a = acc[fp + f][fj][4 * g + e], gt = acc[fp + f][fj][4 * (g + 2) + e];
                                if (eplan.fast) { a += epi_vec[np - n0 + e]; gt += epi_vec[np - n0 + 16 + e]; }  // bias staged in LDS (zeros if none)
                                else if (bias) { a += bias[np + e]; gt += bias[np + 16 + e]; }
                                const float v = a * gelu_erf_f(gt);
                                h[f][g][e] = v;
                                amax = fmaxf(amax, fabsf(v));
                            }
                        }
                    }
                    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                    int ex = 0;
                    if (amax > 0.f) frexpf(amax * (1.f / 448.f), &ex);  // amax/448 = f * 2^ex, f in [0.5, 1)  ->  2^ex >= amax/448
                    else ex = -127;
                    ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
                    const float inv = ldexpf(1.f, -ex);
                    uint32_t qd[2][2];
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
                            qd[f][g] = pack_fp8x4(h[f][g][0] * inv, h[f][g][1] * inv, h[f][g][2] * inv, h[f][g][3] * inv);
                    const uint4 w = widen_quads_fp8(qd[0][0], qd[0][1], qd[1][0], qd[1][1]);
                    const int nc0 = (n0 + wn * NW + fp * 32) >> 1;  // first output column of the 32-column block
                    if (nc0 + 16 * lh < nout) *(uint4*)((uint8_t*)p.out + (size_t)m * p.ldc + nc0 + 16 * lh) = w;
                    if (lh == 0 && nc0 < nout) q.mx_out[(size_t)m * q.ld_mx_out + (nc0 >> 5)] = (uint8_t)(ex + 127);
                }
            }
            return;
        }
    }
    constexpr int CAP = (WM * WN == 16) ? 128 : 256;
    if constexpr (!OUT_F32 && EPI == EPI_LINEAR) {
        if (eplan.fast) { gemm_epilogue_linear_lds<FX, FY, FM, FN, BN, CAP>(p, acc, m0, n0, wm, wn, l31, lh, tn * WN + wn, nullptr, epi_vec, eplan.img0); return; }
    } else if constexpr (EPI == EPI_GEGLU) {
        if (eplan.fast) { gemm_epilogue_geglu_lds<FX, FY, FM, FN, BN>(p, acc, m0, n0, wm, wn, l31, lh, nullptr, epi_vec); return; }
    }
    gemm_epilogue<EPI, OUT_F32, FX, FY, FM, FN>(p, acc, m0, n0, wm, wn, l31, lh, tn * WN + wn);
}

template <int AMODE, int EPI, bool OUT_F32, bool AMX, int WM, int WN, int FM, int FN>
int launch_cfg(const VkGemmDesc* d, const F8Args& q, hipStream_t stream) {
    constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    const int tiles = ((d->N + BN - 1) / BN) * ((d->M + BM - 1) / BM);
    VkGemmDesc desc = *d;   // the shared epilogues (gemm_common.h) bound their rows by m_end (ABI v5 row ranges: not offered by the fp8 GEMM, all rows)
    desc.m_begin = 0;
    desc.m_end = d->M;
    hipLaunchKernelGGL((gemm_fp8_kernel<AMODE, EPI, OUT_F32, AMX, WM, WN, FM, FN>), dim3(tiles), dim3(WM * WN * 64), 0, stream, desc, q);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

int fp8_cfg(const VkGemmDesc* d);

template <int EPI, bool OUT_F32, bool AMX>
int launch(const VkGemmDesc* d, const F8Args& q, hipStream_t stream) {
    const int cfg = fp8_cfg(d);
    if (cfg == 4 && EPI != EPI_GEGLU) return launch_cfg<AMODE_DENSE, EPI, OUT_F32, AMX, 4, 2, 2, 5>(d, q, stream);
    if (cfg >= 3) return launch_cfg<AMODE_DENSE, EPI, OUT_F32, AMX, 4, 4, 2, 2>(d, q, stream);  // sixteen 64x64 wave tiles, 4 waves per SIMD (as in gemm.hip)
    return launch_cfg<AMODE_DENSE, EPI, OUT_F32, AMX, 2, 2, 2, 2>(d, q, stream);
}

// implicit-GEMM convolutions: 256x320 block tiles as eight 64x160 wave tiles where Cout is a multiple of 320 (every UNet level; the
// sixteen-wave 32x160 form of the bf16 kernels does not fit eight-register fp8 fragments into 128 VGPRs), else 256x256
template <int AMODE>
int launch_conv(const VkGemmDesc* d, const F8Args& q, hipStream_t stream) {
    if (d->N % 320 == 0) return launch_cfg<AMODE, EPI_LINEAR, false, false, 4, 2, 2, 5>(d, q, stream);
    return launch_cfg<AMODE, EPI_LINEAR, false, false, 4, 4, 2, 2>(d, q, stream);
}

// one wave per row: amax -> scale = amax / 448, q = e4m3(x / scale)
__global__ __launch_bounds__(256) void quantize_rows_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ qo, float* __restrict__ scale,
                                                            int M, int K, long long ldx, long long ldq) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    constexpr int MAXC = 10;  // 8-element chunks per lane: K <= 5120
    const int nch = K >> 3;
    uint4 v[MAXC];
    float amax = 0.f;
    const uint16_t* xr = x + (size_t)row * ldx;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            v[i] = *(const uint4*)(xr + c * 8);
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) scale[row] = sc;
    uint8_t* qr = qo + (size_t)row * ldq;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * inv, -448.f), 448.f);  // the cvt yields NaN above 448, it does not saturate
            int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            *(int2*)(qr + c * 8) = make_int2(lo, hi);
        }
    }
}

}  // namespace

namespace {
int fp8_cfg(const VkGemmDesc* d) {  // the tile variant launch<>() picks (1 = 128x128, 3 = 256x256, 4 = 256x320)
    int cfg = d->tile_cfg & 7;
    if (cfg == 0) {
        auto wgs = [&](int bm, int bn) { return (long long)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn); };
        const int n256 = (d->N + 255) / 256 * 256;
        if (d->epi != EPI_GEGLU && d->N % 320 == 0 && wgs(256, 320) >= 192) cfg = 4;
        else if (n256 * 10 <= d->N * 11 && wgs(256, 256) >= 192) cfg = 3;
        else cfg = 1;
    }
    return (cfg == 4 || cfg == 3) ? cfg : 1;
}

int gemm_fp8_entry(const VkGemmDesc* d, const F8Args& q, hipStream_t stream) {
    if (!d || !d->A || !d->Wt || !d->out || (!q.a_scale && !q.a_mx) || (q.a_scale && q.a_mx) || !q.w_scale) return VK_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % 128) != 0 || (d->N % 4) != 0 || q.K <= 0 || q.K > d->K || (q.K % 16) != 0 ||
        d->tile_cfg < 0 || (d->tile_cfg & 7) > 4 || d->tile_cfg > 7 || q.a_div < 0)
        return VK_EINVAL;
    const bool conv = d->amode == AMODE_CONV3X3 || d->amode == AMODE_TEMPORAL3;
    if ((long long)(d->N + 320) * d->K >= (1LL << 32)) return VK_EINVAL;  // 32-bit offsets in the loaders
    if (conv ? (q.a_div < 1 || (long long)d->M * d->Cin >= (1LL << 32)) : (q.a_div > 1 || (long long)d->M * d->lda >= (1LL << 32))) return VK_EINVAL;
    if (!conv && (d->amode != AMODE_DENSE || d->lda < q.K || (d->lda % 16) != 0)) return VK_EINVAL;
    if (conv) {  // fp8 GroupNorm output -> ResBlock convolution: stride 1, pad 1, bf16 output, per-image-group activation scales
        const int nt = d->amode == AMODE_CONV3X3 ? 9 : 3;
        if (d->Cin <= 0 || (d->Cin % 64) != 0 || q.K != nt * d->Cin || q.a_mx || q.mx_out || d->epi != EPI_LINEAR || d->out_f32 || d->rowstat_out)
            return VK_EINVAL;
        if (d->amode == AMODE_CONV3X3 && (d->H <= 0 || d->Wd <= 0 || d->stride != 1 || d->ups != 1 || d->asym_pad || d->Hout != d->H ||
                                          d->Wout != d->Wd || d->M % (d->H * d->Wd) != 0))
            return VK_EINVAL;
        if (d->amode == AMODE_TEMPORAL3 && (d->T <= 0 || d->S <= 0 || d->halo_prev || d->halo_next || d->M % (d->T * d->S) != 0)) return VK_EINVAL;
    }
    if ((d->rowvec || d->rowvec2) && d->rows_per_vec <= 0) return VK_EINVAL;
    if (d->ln_stats || d->A2 || (d->rowvec2 && !d->res2)) return VK_EINVAL;  // bf16-GEMM-only features
    if (d->rowstat_out && (d->epi != EPI_LINEAR || d->out_f32)) return VK_EINVAL;
    if (q.a_mx && (q.ld_mx * 32 < d->K || (q.ld_mx % 4) != 0 || (q.K % 32) != 0)) return VK_EINVAL;  // one dword of block scales per 128-byte K-step
    if (q.mx_out && (d->epi != EPI_GEGLU || ((d->N >> 1) % 32) != 0 || (d->ldc % 16) != 0 || q.ld_mx_out * 32 < (d->N >> 1))) return VK_EINVAL;
    const bool f32 = d->out_f32 != 0;
    if (d->amode == AMODE_CONV3X3) return launch_conv<AMODE_CONV3X3>(d, q, stream);
    if (d->amode == AMODE_TEMPORAL3) return launch_conv<AMODE_TEMPORAL3>(d, q, stream);
    if (q.a_mx) return (d->epi == EPI_LINEAR && !f32) ? launch<EPI_LINEAR, false, true>(d, q, stream) : VK_EINVAL;  // the FF-out GEMM
    if (d->epi == EPI_LINEAR) return f32 ? launch<EPI_LINEAR, true, false>(d, q, stream) : launch<EPI_LINEAR, false, false>(d, q, stream);
    if (d->epi == EPI_GEGLU && !f32 && (d->N % 32) == 0) return launch<EPI_GEGLU, false, false>(d, q, stream);
    return VK_EINVAL;
}
}  // namespace

extern "C" int vk_gemm_fp8(const VkGemmDesc* d, const float* a_scale, const float* w_scale, int32_t k_real, void* stream_) {
#if VK_F16
    return VK_EINVAL;   // BASELINE config 5 (fp8) exists in the bf16 build only (include/vista_hip.h: vk_act_dtype)
#endif
    const F8Args q{a_scale, 0, w_scale, k_real, nullptr, 0, nullptr, 0};
    return gemm_fp8_entry(d, q, (hipStream_t)stream_);
}

extern "C" int vk_gemm_fp8_mx(const VkGemmDesc* d, const VkFp8Args* a, void* stream_) {
#if VK_F16
    return VK_EINVAL;   // BASELINE config 5 (fp8) exists in the bf16 build only (include/vista_hip.h: vk_act_dtype)
#endif
    if (!a) return VK_EINVAL;
    const F8Args q{a->a_scale, a->a_scale_rows, a->w_scale, a->k_real, (const uint8_t*)a->a_mx, a->ld_mx, (uint8_t*)a->mx_out, a->ld_mx_out};
    return gemm_fp8_entry(d, q, (hipStream_t)stream_);
}

extern "C" int vk_gemm_fp8_rowstat_parts(const VkGemmDesc* d) {
    if (!d || d->epi != EPI_LINEAR || d->out_f32 || d->N <= 0) return VK_EINVAL;
    const int cfg = fp8_cfg(d);
    const int bn = cfg == 4 ? 320 : (cfg == 3 ? 256 : 128), wn = cfg == 4 ? 2 : (cfg == 3 ? 4 : 2);
    return ((d->N + bn - 1) / bn) * wn;
}

extern "C" int vk_quantize_rows_fp8(const void* x, void* q, float* scale, int32_t M, int32_t K, int64_t ldx, int64_t ldq, void* stream_) {
    if (!x || !q || !scale || M <= 0 || K <= 0 || (K % 8) != 0 || K > 5120 || ldx < K || ldq < K || (ldx % 8) != 0 || (ldq % 8) != 0) return VK_EINVAL;
    hipLaunchKernelGGL(quantize_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream_, (const uint16_t*)x, (uint8_t*)q, scale, M, K,
                       (long long)ldx, (long long)ldq);
    VK_CHECK_LAUNCH();
    return VK_OK;
}
