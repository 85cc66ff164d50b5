// Pieces shared by the bf16 (gemm.hip) and fp8 (gemm_fp8.hip) implicit-GEMM kernels: operand-mode / epilogue enums and the fused
// epilogue. Included inside each translation unit's own anonymous namespace scope.
#pragma once
#include "common.h"
#include "vista_hip.h"

namespace {

enum { AMODE_DENSE = 0, AMODE_CONV3X3 = 1, AMODE_TEMPORAL3 = 2, AMODE_CONV3D = 3 };
enum { EPI_LINEAR = 0, EPI_GEGLU = 1, EPI_TRANS = 2 };

constexpr int BK = 64;


// Fused epilogue shared by the GEMM kernels. Accumulator element r = 4*g + e of tile (fi,fj): X-row = 32*fi + 8*g + 4*lh + e,
// Y-row = 32*fj + l31 (X = weights / Y = activations, swapped for EPI_TRANS). MW/NW = wave-tile extents along m / n.
// Lane l31 of the lower half-wave (lh = 0) holds columns c..c+3 of accumulator quad g and lane l31 of the upper half the next
// four, for the SAME output row. Swapping the upper half of quad g with the lower half of quad g+1 (v_permlane32_swap) leaves the
// lower lane with 8 contiguous bf16 of quad g and the upper lane with 8 contiguous bf16 of quad g+1: one 16-byte store instead
// of two 8-byte ones. The epilogue of these GEMMs is store-ISSUE bound (16-40 stores per lane), so halving the count matters most
// where tiles are short (K = 320: 5 K-steps per tile).
__device__ __forceinline__ uint4 widen_pair(uint2 qa, uint2 qb) {
    const auto rx = __builtin_amdgcn_permlane32_swap(qa.x, qb.x, false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(qa.y, qb.y, false, false);
    return make_uint4(rx[0], ry[0], rx[1], ry[1]);
}

// The inverse, for loads: the lower lane read the 8 contiguous bf16 of quad g (its own 4 and its partner's 4), the upper lane those of
// quad g+1; two swaps hand every lane its own 4 values of both quads.
__device__ __forceinline__ void unwiden_pair(const uint4& w, uint2& qa, uint2& qb) {
    const auto rx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
    qa = make_uint2(rx[0], ry[0]);
    qb = make_uint2(rx[1], ry[1]);
}

// LayerNorm fold (VkGemmDesc.ln_*): (mean, rstd) of activation row m from the producer's partial sums, summed in slab order. The GEMM
// kernel evaluates this ONCE per tile row at kernel start (the HBM latency of the slab reads hides behind the first tile's DMA) and
// parks the pairs in LDS; the epilogues read them back with ds_read -- reading the slabs from the epilogue itself exposed ~2 us of
// HBM latency per tile with nothing to overlap it (one workgroup per CU): +15-20 % on the level-0 GEGLU / q|k|v kernels.
__device__ __forceinline__ float2 ln_row_stats(const VkGemmDesc& p, int m) {
    const float2* __restrict__ st = (const float2*)p.ln_stats;
    float s = 0.f, q = 0.f;
    for (int i = 0; i < p.ln_parts; ++i) {
        const float2 v = st[(size_t)i * p.M + m];
        s += v.x;
        q += v.y;
    }
    const float inv_c = 1.f / (float)p.K;
    const float mu = s * inv_c;
    return make_float2(mu, rsqrtf(fmaxf(q * inv_c - mu * mu, 0.f) + p.ln_eps));
}

template <int EPI, bool OUT_F32, int FX, int FY, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const VkGemmDesc& p, f32x16_t (&acc)[FX][FY], int m0, int n0, int wm, int wn, int l31, int lh,
                                              int stat_part = 0, const float2* lnrow = nullptr) {
    // lnrow: LDS table of (mean, rstd) for the tile's rows m0 .. m0 + BM (NULL = no LayerNorm fold)
    constexpr int MW = FM * 32, NW = FN * 32;
    const float* __restrict__ lncs = p.ln_colsum;
    if (EPI == EPI_LINEAR) {
        const float* __restrict__ bias = p.bias;
        const float* __restrict__ rowvec = p.rowvec;
        const float* __restrict__ rowvec2 = p.rowvec2;
        const uint16_t* __restrict__ res1 = (const uint16_t*)p.res1;
        const uint16_t* __restrict__ res2 = (const uint16_t*)p.res2;
        // 16-byte stores need whole 16-column groups and 16-byte aligned rows (always true for the UNet's shapes)
        const bool wide = !OUT_F32 && (p.N % 16 == 0) && (p.ldc % 8 == 0) && (((size_t)p.out & 15) == 0);
        // ... and so do 16-byte RESIDUAL loads (unwiden_pair): a lane's natural residual access is 8 bytes of a 640-byte-strided row, 20 of
        // them per residual and wave tile, each touching 32 cache lines for 16 bytes apiece; pairs of quads read as one 16-byte load per
        // lane halve the instructions and double the bytes used per line touched
        const bool wide1 = res1 && (p.N % 16 == 0) && (p.ld_res1 % 8 == 0) && (((size_t)res1 & 15) == 0);
        const bool wide2 = res2 && (p.N % 16 == 0) && (p.ld_res2 % 8 == 0) && (((size_t)res2 & 15) == 0);
#pragma unroll
        for (int fj = 0; fj < FY; ++fj) {
            const int m = m0 + wm * MW + fj * 32 + l31;
            if (m >= p.M) continue;  // both lanes of a row (l31, l31 + 32) leave together
            const float* rv = rowvec ? rowvec + (size_t)(m / p.rows_per_vec) * p.ldv : nullptr;
            const float* rv2 = rowvec2 ? rowvec2 + (size_t)(m / p.rows_per_vec) * p.ldv : nullptr;
            float nrm = 0.f, rs = 1.f;
            if (lnrow) { const float2 t = lnrow[m - m0]; rs = t.y; nrm = -t.x * t.y; }
            float ssum = 0.f, qsum = 0.f;  // row sums of the bf16-rounded outputs this lane stores (rowstat_out)
#pragma unroll
            for (int fi = 0; fi < FX; ++fi) {
                uint2 packed[4];
                uint2 r1q[4], r2q[4];  // the residuals' quads of this fragment
#pragma unroll
                for (int g = 0; g < 4; ++g) { r1q[g] = make_uint2(0, 0); r2q[g] = make_uint2(0, 0); }
                if (res1) {
                    if (wide1) {
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const int nb = n0 + wn * NW + fi * 32 + 16 * gp;
                            const uint4 w = nb < p.N ? *(const uint4*)(res1 + (size_t)m * p.ld_res1 + nb + 8 * lh) : make_uint4(0, 0, 0, 0);
                            unwiden_pair(w, r1q[2 * gp], r1q[2 * gp + 1]);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = n0 + wn * NW + fi * 32 + 8 * g + 4 * lh;
                            if (n < p.N) r1q[g] = *(const uint2*)(res1 + (size_t)m * p.ld_res1 + n);
                        }
                    }
                }
                if (res2) {
                    if (wide2) {
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const int nb = n0 + wn * NW + fi * 32 + 16 * gp;
                            const uint4 w = nb < p.N ? *(const uint4*)(res2 + (size_t)m * p.ld_res2 + nb + 8 * lh) : make_uint4(0, 0, 0, 0);
                            unwiden_pair(w, r2q[2 * gp], r2q[2 * gp + 1]);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = n0 + wn * NW + fi * 32 + 8 * g + 4 * lh;
                            if (n < p.N) r2q[g] = *(const uint2*)(res2 + (size_t)m * p.ld_res2 + n);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * NW + fi * 32 + 8 * g + 4 * lh;
                    packed[g] = make_uint2(0, 0);
                    if (n >= p.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[fi][fj][4 * g + e];
                    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bias) b = *(const float4*)(bias + n);
                    if (lnrow) {  // rstd*(acc - mean*colsum) + bias as two fmas per value: nrm = -rstd*mean is per row
                        const float4 c = *(const float4*)(lncs + n);
                        v[0] = fmaf(rs, v[0], fmaf(nrm, c.x, b.x)); v[1] = fmaf(rs, v[1], fmaf(nrm, c.y, b.y));
                        v[2] = fmaf(rs, v[2], fmaf(nrm, c.z, b.z)); v[3] = fmaf(rs, v[3], fmaf(nrm, c.w, b.w));
                    } else if (bias) {
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (rv) {
                        const float4 b = *(const float4*)(rv + n);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (res1) {
                        const uint2 r = r1q[g];
                        v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                    if (res2) {
                        const uint2 r = r2q[g];
                        float t[4] = {bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y)};
                        if (rv2) {
                            const float4 b = *(const float4*)(rv2 + n);
                            t[0] += b.x; t[1] += b.y; t[2] += b.z; t[3] += b.w;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += p.beta * t[e];
                    }
                    if (OUT_F32) {
                        *(float4*)((float*)p.out + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        packed[g].x = pack_bf16(v[0], v[1]);
                        packed[g].y = pack_bf16(v[2], v[3]);
                        if (p.rowstat_out) {
                            const float a0 = bf16_lo(packed[g].x), a1 = bf16_hi(packed[g].x), a2 = bf16_lo(packed[g].y), a3 = bf16_hi(packed[g].y);
                            ssum += (a0 + a1) + (a2 + a3);
                            qsum = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, qsum))));
                        }
                        if (!wide) *(uint2*)((uint16_t*)p.out + (size_t)m * p.ldc + n) = packed[g];
                    }
                }
                if (!OUT_F32 && wide) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int nb = n0 + wn * NW + fi * 32 + 16 * gp;  // both lanes of a row share the group's validity
                        const uint4 w = widen_pair(packed[2 * gp], packed[2 * gp + 1]);
                        if (nb < p.N) *(uint4*)((uint16_t*)p.out + (size_t)m * p.ldc + nb + 8 * lh) = w;
                    }
                }
            }
            if (!OUT_F32 && p.rowstat_out) {  // the row's NW columns of this wave tile live in lanes l31 and l31 + 32: fixed-order pair sum
                const float so = __shfl_xor(ssum, 32, 64), qo = __shfl_xor(qsum, 32, 64);
                if (lh == 0) ((float2*)p.rowstat_out)[(size_t)stat_part * p.M + m] = make_float2(ssum + so, qsum + qo);
            }
        }
    } else if (EPI == EPI_GEGLU) {
        // packed weight rows: every 32-row fragment = [16 value rows | 16 gate rows] of the SAME 16 output columns, so the value quad g
        // (g = 0, 1) and its gate quad g + 2 sit in the same lane whatever the wave tile width -- any block tile can run GEGLU.
        const float* __restrict__ bias = p.bias;
        const int nout = p.N >> 1;
        const bool wide = (nout % 16 == 0) && (p.ldc % 8 == 0) && (((size_t)p.out & 15) == 0);
#pragma unroll
        for (int fj = 0; fj < FY; ++fj) {
            const int m = m0 + wm * MW + fj * 32 + l31;
            if (m >= p.M) continue;
            float nrm = 0.f, rs = 1.f;
            if (lnrow) { const float2 t = lnrow[m - m0]; rs = t.y; nrm = -t.x * t.y; }
#pragma unroll
            for (int fi = 0; fi < FX; ++fi) {
                const int nfrag = n0 + wn * NW + fi * 32;  // first packed row of the fragment
                uint2 packed[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int np = nfrag + 8 * g + 4 * lh;           // packed row of the value part; its gate is 16 rows further
                    const int nc = (nfrag >> 1) + 8 * g + 4 * lh;    // output column
                    packed[g] = make_uint2(0, 0);
                    if (nc >= nout) continue;
                    float a[4], gt[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = acc[fi][fj][4 * g + e]; gt[e] = acc[fi][fj][4 * (g + 2) + e]; }
                    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bg = ba;
                    if (bias) {
                        ba = *(const float4*)(bias + np);
                        bg = *(const float4*)(bias + np + 16);
                    }
                    if (lnrow) {  // rstd*(acc - mean*colsum) + bias as two fmas per value
                        const float4 ca = *(const float4*)(lncs + np);
                        const float4 cg = *(const float4*)(lncs + np + 16);
                        a[0] = fmaf(rs, a[0], fmaf(nrm, ca.x, ba.x)); a[1] = fmaf(rs, a[1], fmaf(nrm, ca.y, ba.y));
                        a[2] = fmaf(rs, a[2], fmaf(nrm, ca.z, ba.z)); a[3] = fmaf(rs, a[3], fmaf(nrm, ca.w, ba.w));
                        gt[0] = fmaf(rs, gt[0], fmaf(nrm, cg.x, bg.x)); gt[1] = fmaf(rs, gt[1], fmaf(nrm, cg.y, bg.y));
                        gt[2] = fmaf(rs, gt[2], fmaf(nrm, cg.z, bg.z)); gt[3] = fmaf(rs, gt[3], fmaf(nrm, cg.w, bg.w));
                    } else if (bias) {
                        a[0] += ba.x; a[1] += ba.y; a[2] += ba.z; a[3] += ba.w;
                        gt[0] += bg.x; gt[1] += bg.y; gt[2] += bg.z; gt[3] += bg.w;
                    }
                    packed[g].x = pack_bf16(a[0] * gelu_erf_f(gt[0]), a[1] * gelu_erf_f(gt[1]));
                    packed[g].y = pack_bf16(a[2] * gelu_erf_f(gt[2]), a[3] * gelu_erf_f(gt[3]));
                    if (!wide) *(uint2*)((uint16_t*)p.out + (size_t)m * p.ldc + nc) = packed[g];
                }
                if (wide) {
                    const int nb = nfrag >> 1;  // 16 output columns of this fragment: 8 per lane half after the swap
                    const uint4 w = widen_pair(packed[0], packed[1]);
                    if (nb < nout) *(uint4*)((uint16_t*)p.out + (size_t)m * p.ldc + nb + 8 * lh) = w;
                }
            }
        }
    } else {  // EPI_TRANS: out[img][n][key], key = m % S contiguous
        const bool wide = (p.S % 16 == 0) && (p.M % 16 == 0) && (((size_t)p.out & 15) == 0);
#pragma unroll
        for (int fj = 0; fj < FY; ++fj) {
            const int n = n0 + wn * NW + fj * 32 + l31;
            if (n >= p.N) continue;
            const float bn = p.bias ? p.bias[n] : 0.f;  // the lane's output channel (VAE AttnBlock v projection carries a bias)
            const float csn = lnrow ? lncs[n] : 0.f;
#pragma unroll
            for (int fi = 0; fi < FX; ++fi) {
                uint2 packed[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[fi][fj][4 * g + e];
                    if (lnrow) {  // the lane's 4 consecutive keys are 4 activation rows: one (mean, rstd) pair each, 32 contiguous LDS bytes
                        const float4* t = (const float4*)(lnrow + wm * MW + fi * 32 + 8 * g + 4 * lh);
                        const float4 t01 = t[0], t23 = t[1];
                        v[0] = t01.y * (v[0] - t01.x * csn); v[1] = t01.w * (v[1] - t01.z * csn);
                        v[2] = t23.y * (v[2] - t23.x * csn); v[3] = t23.w * (v[3] - t23.z * csn);
                    }
                    packed[g].x = pack_bf16(v[0] + bn, v[1] + bn);
                    packed[g].y = pack_bf16(v[2] + bn, v[3] + bn);
                    if (!wide) {
                        const int m = m0 + wm * MW + fi * 32 + 8 * g + 4 * lh;
                        if (m >= p.M) continue;
                        const int img = m / p.S;
                        const int key = m - img * p.S;
                        *(uint2*)((uint16_t*)p.out + ((size_t)img * p.N + n) * p.S + key) = packed[g];
                    }
                }
                if (wide) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int mb = m0 + wm * MW + fi * 32 + 16 * gp;  // 16 consecutive keys of one image (S % 16 == 0)
                        const uint4 w = widen_pair(packed[2 * gp], packed[2 * gp + 1]);
                        if (mb < p.M) {
                            const int img = mb / p.S;
                            const int key = mb - img * p.S + 8 * lh;
                            *(uint4*)((uint16_t*)p.out + ((size_t)img * p.N + n) * p.S + key) = w;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace
