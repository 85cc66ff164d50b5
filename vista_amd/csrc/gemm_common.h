// Pieces shared by the bf16 (gemm.hip) and fp8 (gemm_fp8.hip) implicit-GEMM kernels: operand-mode / epilogue enums and the fused
// epilogue. Included inside each translation unit's own anonymous namespace scope.
#pragma once
#include "common.h"
#include "vista_hip.h"

namespace {

enum { AMODE_DENSE = 0, AMODE_CONV3X3 = 1, AMODE_TEMPORAL3 = 2, AMODE_CONV3D = 3 };
enum { EPI_LINEAR = 0, EPI_GEGLU = 1, EPI_TRANS = 2 };

constexpr int BK = 64;

// Row range of a launch (VkGemmDesc.m_begin / m_end, ABI v5): every launcher passes its kernels a desc whose m_end is the exclusive row bound
// (0 -> M). false = an invalid range.
inline bool norm_row_range(VkGemmDesc& d) {
    if (d.m_end == 0) d.m_end = d.M;
    return d.m_begin >= 0 && d.m_begin < d.m_end && d.m_end <= d.M;
}


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

// 16-byte output store of the LDS-staged epilogues. -DVK_EPI_NT_STORES=1 builds them with the non-temporal hint (an A/B build: tools/build_variant.sh;
// the GroupNorm apply pass gained 20 % from it on tensors that do not fit the Infinity Cache, profiles/r05_gn_nontemporal_ab.txt).
#ifndef VK_EPI_NT_STORES
#define VK_EPI_NT_STORES 0
#endif
__device__ __forceinline__ void epi_st16(void* p, const uint4 v) {
#if VK_EPI_NT_STORES
    typedef unsigned epi_u32x4_t __attribute__((ext_vector_type(4)));
    const epi_u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (epi_u32x4_t*)p);
#else
    *(uint4*)p = v;
#endif
}

// The inverse, for loads: the lower lane read the 8 contiguous bf16 of quad g (its own 4 and its partner's 4), the upper lane those of
// quad g+1; two swaps hand every lane its own 4 values of both quads.
__device__ __forceinline__ void unwiden_pair(const uint4& w, uint2& qa, uint2& qb) {
    const auto rx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
    qa = make_uint2(rx[0], ry[0]);
    qb = make_uint2(rx[1], ry[1]);
}

__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}

// MX-fp8 quantisation of ONE 32-column block of an output row held as the four accumulator quads of a 32x32 fragment: lane half lh holds
// columns 8g + 4lh + e (g = quad, e = 0..3), i.e. 16 of the 32 values; its partner lane (l ^ 32) the others. Returns the lane's 16 bytes
// of the block's 32 e4m3 bytes (to be stored at byte offset 16*lh of the block) and the block's E8M0 exponent byte (both lanes get it):
// 2^(ex) >= max|v| / 448, the smallest such power of two (frexp), so every value lands in e4m3's range with at most one binade unused.
__device__ __forceinline__ uint4 mx_quant_block(const float (&v)[4][4], int& e8m0) {
    float amax = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[g][e]));
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    int ex = -127;
    if (amax > 0.f) frexpf(amax * (1.f / 448.f), &ex);  // amax/448 = f * 2^ex, f in [0.5, 1)
    ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
    const float inv = ldexpf(1.f, -ex);
    uint32_t q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = pack_fp8x4(v[g][0] * inv, v[g][1] * inv, v[g][2] * inv, v[g][3] * inv);
    // lower lane: [own q0 | partner q0 | own q1 | partner q1] = columns 0-15; upper lane: [partner q2 | own q2 | partner q3 | own q3] = 16-31
    const auto r02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto r13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    e8m0 = ex + 127;
    return make_uint4(r02[0], r02[1], r13[0], r13[1]);
}

// Output pack of the LINEAR epilogues: the build's storage type, except -- fp16 build only -- the columns from VkGemmDesc.alt_cols_from on, which
// leave as bf16 (the V block of a fused q|k|v projection: common.h). `n` = any column of the 32-column fragment being packed (alt_cols_from % 32 == 0).
__device__ __forceinline__ uint32_t pack_out(float a, float b, const VkGemmDesc& p, int n) {
#if VK_F16
    if (p.alt_cols_from > 0 && n >= p.alt_cols_from) return pack_bf16x(a, b);
#endif
    return pack_bf16(a, b);
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
            if (m >= p.m_end) continue;  // both lanes of a row (l31, l31 + 32) leave together (m_end: the launch's row bound, = M unless a row range was asked for)
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
                    if (p.act == 1) { v[0] = gelu_erf_f(v[0]); v[1] = gelu_erf_f(v[1]); v[2] = gelu_erf_f(v[2]); v[3] = gelu_erf_f(v[3]); }
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
                        packed[g].x = pack_out(v[0], v[1], p, n);
                        packed[g].y = pack_out(v[2], v[3], p, n);
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
            if (m >= p.m_end) continue;
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

// ------------------------------------------------------------------------------------------------------------------------------------
// LDS-staged epilogue (round 3). The disassembly of the round-2 epilogue showed what its fixed ~7-15 us per tile were: every accumulator
// quad loaded its bias / LayerNorm column sums / row-vector float4s from global memory and WAITED for each (s_waitcnt vmcnt(0) after every
// load: 2-4 dependent L2 round trips per quad, 20 quads per 32x160 wave tile), and every fragment's residual load sat behind the previous
// fragment's stores (`out` may alias `res1` as far as the compiler can tell), i.e. five more dependent HBM round trips -- with all sixteen
// waves of the CU in the epilogue at once and nothing to overlap them with. Here
//   * the per-column vectors of the tile (bias, LayerNorm column sums) and the per-image row vectors of the <= EPI_NI images its rows span
//     are staged into LDS ONCE per tile, at tile start, under the first K-step's DMA (one float4 per thread), and read back with ds_read_b128
//     (lanes of a half-wave share the address: a broadcast read);
//   * the residual loads of a row block are issued up front (a ring of `D` fragments ahead of the stores);
//   * rows past M are clamped for the loads and predicated only at the stores, so the body is branch-free apart from kernel-uniform options.
// Arithmetic and operation order are those of gemm_epilogue: results are bitwise the same.
constexpr int EPI_NI = 4;  // images a tile's rows may span (256-row tiles: rows_per_vec >= 86)
__host__ __device__ constexpr int epi_vec_floats(int bn) { return (2 + 2 * EPI_NI) * bn; }  // bias | colsum | EPI_NI x rowvec | EPI_NI x rowvec2

struct EpiPlan {
    bool fast;  // block-uniform: this tile takes the LDS-staged epilogue
    int img0;   // first image (row-vector index) of the tile's rows
    int nimg;
};

template <int EPI, bool OUT_F32, int BM, int BN>
__device__ __forceinline__ EpiPlan epi_plan(const VkGemmDesc& p, int m0, int n0) {
    EpiPlan e{false, 0, 1};
    if (OUT_F32 || EPI == EPI_TRANS) return e;
    if (n0 + BN > p.N || (p.ldc % 8) != 0 || (((size_t)p.out) & 15) != 0) return e;
    // rows are addressed by 32-bit byte offsets from the (uniform) tensor bases
    const unsigned long long lim = 0xfffff000ull, rows = (unsigned long long)p.M * 2ull;
    if (rows * (unsigned)p.ldc >= lim || (p.res1 && rows * (unsigned)p.ld_res1 >= lim) || (p.res2 && rows * (unsigned)p.ld_res2 >= lim)) return e;
    if (p.bias && (((size_t)p.bias) & 15) != 0) return e;
    if (p.ln_stats && (((size_t)p.ln_colsum) & 15) != 0) return e;
    if (EPI == EPI_GEGLU) { e.fast = true; return e; }  // (N % 32 == 0 is validated by the launcher; BN % 32 == 0)
    if (p.res1 && ((p.ld_res1 % 8) != 0 || (((size_t)p.res1) & 15) != 0)) return e;
    if (p.res2 && ((p.ld_res2 % 8) != 0 || (((size_t)p.res2) & 15) != 0)) return e;
    if (p.rowvec || p.rowvec2) {
        if ((p.ldv % 4) != 0 || (p.rowvec && (((size_t)p.rowvec) & 15) != 0) || (p.rowvec2 && (((size_t)p.rowvec2) & 15) != 0)) return e;
        const int last = (m0 + BM < p.m_end ? m0 + BM : p.m_end) - 1;
        e.img0 = m0 / p.rows_per_vec;
        e.nimg = last / p.rows_per_vec - e.img0 + 1;
        if (e.nimg > EPI_NI) return e;
    }
    e.fast = true;
    return e;
}

// The tile-independent part of epi_plan for a launch whose N is a multiple of the tile width, as host + device code: true = EVERY tile of the
// launch takes the LDS-staged epilogue (kernels that carry no general epilogue -- gemm_pipe.hip -- take only such launches). Keep in step with epi_plan.
__host__ __device__ inline bool epi_fast_everywhere(const VkGemmDesc& p, int epi, int bm) {
    if (p.out_f32 || epi == EPI_TRANS) return false;
    if ((p.ldc % 8) != 0 || (((size_t)p.out) & 15) != 0) return false;
    const unsigned long long lim = 0xfffff000ull, rows = (unsigned long long)p.M * 2ull;
    if (rows * (unsigned)p.ldc >= lim || (p.res1 && rows * (unsigned)p.ld_res1 >= lim) || (p.res2 && rows * (unsigned)p.ld_res2 >= lim)) return false;
    if (p.bias && (((size_t)p.bias) & 15) != 0) return false;
    if (p.ln_stats && (((size_t)p.ln_colsum) & 15) != 0) return false;
    if (epi == EPI_GEGLU) return true;
    if (p.res1 && ((p.ld_res1 % 8) != 0 || (((size_t)p.res1) & 15) != 0)) return false;
    if (p.res2 && ((p.ld_res2 % 8) != 0 || (((size_t)p.res2) & 15) != 0)) return false;
    if (p.rowvec || p.rowvec2) {
        if ((p.ldv % 4) != 0 || (p.rowvec && (((size_t)p.rowvec) & 15) != 0) || (p.rowvec2 && (((size_t)p.rowvec2) & 15) != 0)) return false;
        if ((bm - 1) / p.rows_per_vec + 2 > EPI_NI) return false;   // bm consecutive rows span at most (bm - 1) / rows_per_vec + 2 images
    }
    return true;
}

// one float4 per thread: section s of the region = [bias | colsum | rowvec of images img0.. | rowvec2 of images img0..], absent ones zero
template <int BN, int NT>
__device__ __forceinline__ void epi_stage_vectors(const VkGemmDesc& p, float* ev, int n0, const EpiPlan& e, int tid) {
    constexpr int QN = BN / 4, NSEC = 2 + 2 * EPI_NI;
#pragma unroll
    for (int idx0 = 0; idx0 < NSEC * QN; idx0 += NT) {
        const int idx = idx0 + tid;
        if (idx < NSEC * QN) {
            const int sec = idx / QN, q = idx - sec * QN;
            const float* src = nullptr;
            if (sec == 0) src = p.bias;
            else if (sec == 1) src = p.ln_stats ? p.ln_colsum : nullptr;
            else if (sec < 2 + EPI_NI) { if (p.rowvec && sec - 2 < e.nimg) src = p.rowvec + (size_t)(e.img0 + sec - 2) * p.ldv; }
            else { if (p.rowvec2 && sec - 2 - EPI_NI < e.nimg) src = p.rowvec2 + (size_t)(e.img0 + sec - 2 - EPI_NI) * p.ldv; }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src) v = *(const float4*)(src + n0 + 4 * q);
            *(float4*)(ev + sec * BN + 4 * q) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// GroupNorm statistics of the OUTPUT, emitted by the LINEAR epilogue of the eight-wave 256x320 kernel (VkGemmDesc.gnstat_out, ABI v6): the
// stage-1 partial sums norm.hip's gn_stats_kernel would otherwise produce in a read pass of its own over the tensor this epilogue is writing
// (GroupNorm32 of openaimodel.py:195-199,227-234 / video_model.py:38-52 on the output of the preceding convolution). A wave tile is 64 rows x
// 160 columns = 64 tokens of ONE image (gn_rows % 64 == 0) x NG = 160 / CPG whole channel groups (CPG = N / 32 = 10, 20 or 40), so a wave owns
// the slot [row block][its NG groups] of the partials outright: no atomics, a fixed summation order, bitwise reproducible.
//   * accumulate (per lane, from the packed bf16 pairs it is about to store -- the values the apply pass will read): v_dot2_f32_bf16 with
//     (1, 1) for the sum and with itself for the sum of squares, one accumulator pair per group. A lane's columns are 4 lh + {0..3} of every
//     8: where a pair of the lower half-wave and the same pair of the upper half-wave fall into DIFFERENT groups (16 of a lane's 40 pairs per
//     row block at CPG = 10, 8 at CPG = 20, none at 40) the pair is masked by half-wave and added to both;
//   * reduce over the 64 lanes (32 rows x 2 half-waves) as a transposing butterfly: v_permlane32_swap pairs value i with value i + NG (sums
//     with sums of squares), then every xor step halves the register count by exchanging complementary halves, so the 2 NG values cost
//     ~3 NG instructions instead of 12 NG, and end up one per lane (pair): lane L holds value 32 * (L >> 5) + gb + ((L & 31) >> SH) of the slot.
template <int NG>
__device__ __forceinline__ void gn_reduce_store(float (&gs)[NG], float (&gq)[NG], float* __restrict__ slot, int gb, int lane, bool valid) {
    static_assert(NG == 16 || NG == 8 || NG == 4, "160-column wave tiles of 10 / 20 / 40-channel groups");
    float r[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {   // lower half-wave: the pair total of sum i; upper half-wave: of sum-of-squares i
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, gs[i]), __builtin_bit_cast(uint32_t, gq[i]), false, false);
        r[i] = __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);
    }
    int k = 16;
#pragma unroll
    for (int n = NG; n > 1; n >>= 1, k >>= 1) {   // lanes with bit k clear keep registers [0, n/2) and hand over [n/2, n); the others the reverse
        const bool up = (lane & k) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            float lo = r[i], hi = r[i + n / 2];
            asm volatile("" : "+v"(lo), "+v"(hi));   // two VALUES: left as array reads the selects become r[up ? i + n/2 : i], a 16-way v_cndmask chain each
            const float keep = up ? hi : lo;
            const float give = up ? lo : hi;
            r[i] = keep + __shfl_xor(give, k, 64);
        }
    }
#pragma unroll
    for (; k >= 1; k >>= 1) r[0] += __shfl_xor(r[0], k, 64);
    constexpr int SH = NG == 16 ? 1 : (NG == 8 ? 2 : 3);   // lanes per value inside a half-wave = 32 / NG
    if (valid && (lane & ((1 << SH) - 1)) == 0) slot[32 * (lane >> 5) + gb + ((lane & 31) >> SH)] = r[0];
}

// the two pairs (columns c .. c+3, c = 32 fi + 8 g + 4 lh of the wave tile) of one packed quad into the group accumulators
template <int CPG, int NG>
__device__ __forceinline__ void gn_accumulate_quad(float (&gs)[NG], float (&gq)[NG], const int fi, const int g, const uint2 packed, const uint32_t lo_mask) {
    constexpr uint32_t one = VK_ONE_PAIR;   // (1, 1) in the build's storage type; vk_dot2 = v_dot2c_f32_bf16 / v_dot2c_f32_f16 (common.h)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t w = h ? packed.y : packed.x;
        const int c0 = 32 * fi + 8 * g + 2 * h;                 // the pair's first column in a lane of the lower half-wave; + 4 in the upper one
        const int G0 = c0 / CPG, G1 = (c0 + 4) / CPG;
        if (G0 == G1) {
            gs[G0] = vk_dot2(w, one, gs[G0]);
            gq[G0] = vk_dot2(w, w, gq[G0]);
        } else {   // the half-waves' pairs belong to different groups: each lane adds its pair to its own group and zeros to the other
            const uint32_t v0 = w & lo_mask, v1 = w & ~lo_mask;
            gs[G0] = vk_dot2(v0, one, gs[G0]);
            gq[G0] = vk_dot2(v0, v0, gq[G0]);
            gs[G1] = vk_dot2(v1, one, gs[G1]);
            gq[G1] = vk_dot2(v1, v1, gq[G1]);
        }
    }
}

// LINEAR, unit by unit over the wave tile's U = FX*FY fragments (row block fj, column fragment fi): LayerNorm fold / bias / row vector from
// LDS, + res1, * alpha, + beta * (res2 + rowvec2), bf16 pack, row sums, two 16-byte stores. The residuals are the only global loads, issued
// as a RING that runs ahead of the stores: DA units' loads go out before the first unit is touched, and every finished unit -- its residual
// registers are dead then -- lets the next unit's loads go, so a load is in flight while the DA units before it are processed instead of
// starting only after their stores. DA is sized by NRES, the number of residual tensors (0, 1 or 2: template, chosen by a kernel-uniform
// branch), because the sixteen-wave kernels live at the 128-VGPR cap with 64-80 accumulator registers: a residual that had to be spilled
// would be WAITED for at the spill, which is exactly the serialisation this epilogue removes. Absent row vectors are staged as zeros and
// added unconditionally (x + 0 is exact); the LayerNorm fold is a template flag (two more ds_reads and eight fmas per quad).
template <int NRES, bool LN, bool MX, int FX, int FY, int FM, int FN, int BN, int CAP, int GNC = 0>
__device__ __forceinline__ void epilogue_linear_lds_body(const VkGemmDesc& p, f32x16_t (&acc)[FX][FY], int m0, int n0, int wm, int wn, int l31, int lh,
                                                         int stat_part, const float2* lnrow, const float* ev, int img0) {
    constexpr int MW = FM * 32, NW = FN * 32, U = FX * FY;
    // ring depth: what fits beside the accumulators (16 registers per unit), ~36 registers of addresses / per-quad temporaries and the
    // 128-VGPR cap, at 8 registers per unit and residual tensor
    constexpr int GN_NG = GNC ? NW / GNC : 1;       // GNC = channels per GroupNorm group of the output when the epilogue emits its statistics (gnstat_out), else 0
    // (the emitting bodies keep the same ring depth: their group accumulators come alive one fragment at a time while accumulator registers die
    //  sixteen at a time -- each such kernel compiles to 231-244 VGPRs without scratch, checked with -Rpass-analysis=kernel-resource-usage)
    constexpr int ROOM = (CAP - 16 * U - 36) / 8;  // CAP: the kernel's VGPR budget (128 for sixteen waves, 256 for eight)
    constexpr int DA = NRES == 0 ? 0 : ((ROOM / NRES < 1 ? 1 : ROOM / NRES) < U ? (ROOM / NRES < 1 ? 1 : ROOM / NRES) : U);
    constexpr bool mx_tile = MX;  // this whole column tile is MX-fp8 output (the dispatcher checked n0 < mx8_cols; mx8_cols % BN == 0)
    const bool a_is_r1 = p.res1 != nullptr;  // slot a = res1 if there is one, else res2; slot b = res2 when both exist
    const uint16_t* __restrict__ ra = (const uint16_t*)(a_is_r1 ? p.res1 : p.res2);
    const uint16_t* __restrict__ rb = (const uint16_t*)p.res2;
    const int lda_ = a_is_r1 ? p.ld_res1 : p.ld_res2;
    const int c0 = wn * NW + 4 * lh;             // tile-local column of this lane's quad 0 of fragment 0; quad g of fragment fi at + 32*fi + 8*g
    const int wide_off = n0 + wn * NW + 8 * lh;  // the lane's 16-byte chunk of a 16-column group (widen_pair / unwiden_pair)
    bool row_ok[FY];
    int mrow[FY];
    uint32_t rao[FY], rbo[FY], oo[FY];  // 32-bit BYTE offsets from the uniform bases (epi_plan checked the range): one register per row and tensor
    char* const outb = (char*)p.out;
#pragma unroll
    for (int fj = 0; fj < FY; ++fj) {
        const int m = m0 + wm * MW + fj * 32 + l31;
        row_ok[fj] = m < p.m_end;
        mrow[fj] = row_ok[fj] ? m : p.m_end - 1;  // loads of a row past M read the last row; only the stores are predicated
        if (NRES >= 1) rao[fj] = ((uint32_t)mrow[fj] * (uint32_t)lda_ + (uint32_t)wide_off) * 2u;
        if (NRES == 2) rbo[fj] = ((uint32_t)mrow[fj] * (uint32_t)p.ld_res2 + (uint32_t)wide_off) * 2u;
        oo[fj] = ((uint32_t)mrow[fj] * (uint32_t)p.ldc + (uint32_t)(wide_off - (p.mx8_out ? p.mx8_cols : 0))) * 2u;  // (MX tiles never use it)
    }
    static_assert(GNC == 0 || (MW % 64 == 0 && NW == 160 && NW % GNC == 0 && !MX), "gnstat_out: wave tiles of 64-row blocks x 160 columns of whole channel groups");
    float gs[GN_NG], gq[GN_NG];   // GNC: (sum, sum of squares) of the wave tile's GN_NG channel groups over its 64 rows, this lane's share
    if constexpr (GNC != 0) {
#pragma unroll
        for (int i = 0; i < GN_NG; ++i) { gs[i] = 0.f; gq[i] = 0.f; }
    }
    const uint32_t gn_lo_mask = lh ? 0u : 0xffffffffu;
    uint4 wa[U][2], wb[U][2];
    auto issue = [&](int fj, int fi, int u) {
        if (NRES >= 1) { wa[u][0] = *(const uint4*)((const char*)ra + rao[fj] + fi * 64); wa[u][1] = *(const uint4*)((const char*)ra + rao[fj] + fi * 64 + 32); }
        if (NRES == 2) { wb[u][0] = *(const uint4*)((const char*)rb + rbo[fj] + fi * 64); wb[u][1] = *(const uint4*)((const char*)rb + rbo[fj] + fi * 64 + 32); }
    };
#pragma unroll
    for (int u = 0; u < DA; ++u) issue(u / FX, u % FX, u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fj = 0; fj < FY; ++fj) {
        float nrm = 0.f, rs = 1.f;
        if (LN) { const float2 t = lnrow[mrow[fj] - m0]; rs = t.y; nrm = -t.x * t.y; }
        const int iv = (p.rowvec || p.rowvec2) ? mrow[fj] / p.rows_per_vec - img0 : 0;
        const float* evb = ev + c0;
        const float* evc = ev + BN + c0;
        const float* evr = ev + (2 + iv) * BN + c0;
        const float* evr2 = ev + (2 + EPI_NI + iv) * BN + c0;
        float ssum = 0.f, qsum = 0.f;
#pragma unroll
        for (int fi = 0; fi < FX; ++fi) {
            const int u = fj * FX + fi;
            uint2 packed[4], qa[4], qb[4];
            if (NRES >= 1) { unwiden_pair(wa[u][0], qa[0], qa[1]); unwiden_pair(wa[u][1], qa[2], qa[3]); }
            if (NRES == 2) { unwiden_pair(wb[u][0], qb[0], qb[1]); unwiden_pair(wb[u][1], qb[2], qb[3]); }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = fi * 32 + 8 * g;
                const float4 b = *(const float4*)(evb + co);
                const float4 rv = *(const float4*)(evr + co);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[fi][fj][4 * g + e];
                if (LN) {
                    const float4 c = *(const float4*)(evc + co);
                    v[0] = fmaf(rs, v[0], fmaf(nrm, c.x, b.x)); v[1] = fmaf(rs, v[1], fmaf(nrm, c.y, b.y));
                    v[2] = fmaf(rs, v[2], fmaf(nrm, c.z, b.z)); v[3] = fmaf(rs, v[3], fmaf(nrm, c.w, b.w));
                } else {
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                if (GNC == 0 && p.act == 1) { v[0] = gelu_erf_f(v[0]); v[1] = gelu_erf_f(v[1]); v[2] = gelu_erf_f(v[2]); v[3] = gelu_erf_f(v[3]); }   // (no activation / row sums in the emitting bodies: vk_gemm_pipe_gnstat_ok)
                if (NRES >= 1 && a_is_r1) {
                    const uint2 r = qa[g];
                    v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                if (NRES == 2 || (NRES == 1 && !a_is_r1)) {
                    const uint2 r = NRES == 2 ? qb[g] : qa[g];
                    const float4 b2 = *(const float4*)(evr2 + co);
                    const float t0 = bf16_lo(r.x) + b2.x, t1 = bf16_hi(r.x) + b2.y, t2 = bf16_lo(r.y) + b2.z, t3 = bf16_hi(r.y) + b2.w;
                    v[0] += p.beta * t0; v[1] += p.beta * t1; v[2] += p.beta * t2; v[3] += p.beta * t3;
                }
                packed[g].x = pack_out(v[0], v[1], p, n0 + wn * NW + fi * 32);
                packed[g].y = pack_out(v[2], v[3], p, n0 + wn * NW + fi * 32);
                if (GNC == 0 && p.rowstat_out) {
                    const float a0 = bf16_lo(packed[g].x), a1 = bf16_hi(packed[g].x), a2 = bf16_lo(packed[g].y), a3 = bf16_hi(packed[g].y);
                    ssum += (a0 + a1) + (a2 + a3);
                    qsum = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, qsum))));
                }
                if constexpr (GNC != 0) gn_accumulate_quad<GNC, GN_NG>(gs, gq, fi, g, packed[g], gn_lo_mask);
                // one quad's vectors in flight at a time: left alone, the scheduler hoists every ds_read of the row block and spills the
                // residuals it was told to keep in flight
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (mx_tile) {
                // BASELINE config 5: this column tile leaves as MX fp8 (the bf16 rounding above is what a bf16 consumer would have read; the
                // e4m3 values are taken from it so that both output forms of a tile agree to fp8 precision)
                float vq[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) { vq[g][0] = bf16_lo(packed[g].x); vq[g][1] = bf16_hi(packed[g].x); vq[g][2] = bf16_lo(packed[g].y); vq[g][3] = bf16_hi(packed[g].y); }
                int e8;
                const uint4 q16 = mx_quant_block(vq, e8);
                if (row_ok[fj]) {
                    const int col = n0 + wn * NW + fi * 32;
                    *(uint4*)((uint8_t*)p.mx8_out + (size_t)mrow[fj] * p.ld_mx8 + col + 16 * lh) = q16;
                    if (lh == 0) ((uint8_t*)p.mx8_scales)[(size_t)mrow[fj] * p.ld_mx8s + (col >> 5)] = (uint8_t)e8;
                }
            } else {
                const uint4 s0 = widen_pair(packed[0], packed[1]), s1 = widen_pair(packed[2], packed[3]);
                if (row_ok[fj]) {
                    epi_st16(outb + oo[fj] + fi * 64, s0);
                    epi_st16(outb + oo[fj] + fi * 64 + 32, s1);
                }
            }
            if (NRES >= 1 && u + DA < U) issue((u + DA) / FX, (u + DA) % FX, u + DA);  // this unit's residual registers are free again
            __builtin_amdgcn_sched_barrier(0);  // keep the order: this unit's stores, the next ring slot's loads, the next unit
        }
        if (GNC == 0 && p.rowstat_out) {
            const float so = __shfl_xor(ssum, 32, 64), qo = __shfl_xor(qsum, 32, 64);
            if (lh == 0 && row_ok[fj]) ((float2*)p.rowstat_out)[(size_t)stat_part * p.M + mrow[fj]] = make_float2(ssum + so, qsum + qo);
        }
        if constexpr (GNC != 0 && FY > 2) {   // 128-row wave tiles (the four-wave build): a slot per 64-row block = per pair of row fragments
            if ((fj & 1) == 1) {
                const int mb = m0 + wm * MW + (fj >> 1) * 64;
                gn_reduce_store<GN_NG>(gs, gq, (float*)p.gnstat_out + (size_t)(mb >> 6) * 64, (n0 + wn * NW) / GNC, l31 | (lh << 5), mb < p.m_end);
#pragma unroll
                for (int i = 0; i < GN_NG; ++i) { gs[i] = 0.f; gq[i] = 0.f; }
            }
        }
    }
    if constexpr (GNC != 0 && FY == 2) {   // slot = the wave tile's 64-row block (M % 64 == 0: all of its rows are inside the problem, or none)
        const int mb = m0 + wm * MW;
        gn_reduce_store<GN_NG>(gs, gq, (float*)p.gnstat_out + (size_t)(mb >> 6) * 64, (n0 + wn * NW) / GNC, l31 | (lh << 5), mb < p.m_end);
    }
}

// GN_CPG != 0: this kernel instantiation IS one of those that emit GroupNorm statistics of the output (VkGemmDesc.gnstat_out != NULL for every
// launch of it: the pipelined kernel's CONV3X3 / TEMPORAL3 loaders, one tile per workgroup, no split-K) for groups of GN_CPG channels and GN_NRES
// residual tensors, and carries that ONE body: two bodies behind a kernel-uniform branch already spill (40 bytes per lane), each alone takes 244 VGPRs
template <int FX, int FY, int FM, int FN, int BN, int CAP = 128, int GN_CPG = 0, int GN_NRES = 0>
__device__ __forceinline__ void gemm_epilogue_linear_lds(const VkGemmDesc& p, f32x16_t (&acc)[FX][FY], int m0, int n0, int wm, int wn, int l31, int lh,
                                                         int stat_part, const float2* lnrow, const float* ev, int img0) {
    if constexpr (GN_CPG != 0) {   // (the launcher checked: N = 32 GN_CPG, GN_NRES residuals, no LayerNorm fold / MX output / activation / row sums)
        epilogue_linear_lds_body<GN_NRES, false, false, FX, FY, FM, FN, BN, CAP, GN_CPG>(p, acc, m0, n0, wm, wn, l31, lh, stat_part, nullptr, ev, img0);
        return;
    }
    const int nres = (p.res1 != nullptr) + (p.res2 != nullptr);  // kernel-uniform
    if constexpr (BN == 320) {  // MX-fp8 output tiles (BASELINE config 5; validate(): 320-column tiles, no residuals): their own instantiation,
        if (p.mx8_out != nullptr && n0 < p.mx8_cols) {  // so that the bf16 bodies carry none of its registers
            if (lnrow != nullptr) epilogue_linear_lds_body<0, true, true, FX, FY, FM, FN, BN, CAP>(p, acc, m0, n0, wm, wn, l31, lh, stat_part, lnrow, ev, img0);
            else epilogue_linear_lds_body<0, false, true, FX, FY, FM, FN, BN, CAP>(p, acc, m0, n0, wm, wn, l31, lh, stat_part, lnrow, ev, img0);
            return;
        }
    }
#define VK_EPI_BODY(NR, LNF) epilogue_linear_lds_body<NR, LNF, false, FX, FY, FM, FN, BN, CAP>(p, acc, m0, n0, wm, wn, l31, lh, stat_part, lnrow, ev, img0)
    if (lnrow != nullptr) {
        if (nres == 0) VK_EPI_BODY(0, true);
        else if (nres == 1) VK_EPI_BODY(1, true);
        else VK_EPI_BODY(2, true);
    } else {
        if (nres == 0) VK_EPI_BODY(0, false);
        else if (nres == 1) VK_EPI_BODY(1, false);
        else VK_EPI_BODY(2, false);
    }
#undef VK_EPI_BODY
}

template <int FX, int FY, int FM, int FN, int BN>
__device__ __forceinline__ void gemm_epilogue_geglu_lds(const VkGemmDesc& p, f32x16_t (&acc)[FX][FY], int m0, int n0, int wm, int wn, int l31, int lh,
                                                        const float2* lnrow, const float* ev) {
    constexpr int MW = FM * 32, NW = FN * 32;
    const bool has_ln = lnrow != nullptr;
    const float* evb = ev + wn * NW + 4 * lh;       // packed row of the value quad g of fragment fi: + 32*fi + 8*g; its gate 16 rows further
    const float* evc = evb + BN;
#pragma unroll
    for (int fj = 0; fj < FY; ++fj) {
        const int m = m0 + wm * MW + fj * 32 + l31;
        const bool row_ok = m < p.m_end;
        const int mc = row_ok ? m : p.m_end - 1;
        float nrm = 0.f, rs = 1.f;
        if (has_ln) { const float2 t = lnrow[mc - m0]; rs = t.y; nrm = -t.x * t.y; }
        uint16_t* op = (uint16_t*)p.out + (size_t)mc * p.ldc + ((n0 + wn * NW) >> 1) + 8 * lh;
#pragma unroll
        for (int fi = 0; fi < FX; ++fi) {
            uint2 packed[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int co = fi * 32 + 8 * g;
                float a[4], gt[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = acc[fi][fj][4 * g + e]; gt[e] = acc[fi][fj][4 * (g + 2) + e]; }
                const float4 ba = *(const float4*)(evb + co), bg = *(const float4*)(evb + co + 16);
                if (has_ln) {
                    const float4 ca = *(const float4*)(evc + co), cg = *(const float4*)(evc + co + 16);
                    a[0] = fmaf(rs, a[0], fmaf(nrm, ca.x, ba.x)); a[1] = fmaf(rs, a[1], fmaf(nrm, ca.y, ba.y));
                    a[2] = fmaf(rs, a[2], fmaf(nrm, ca.z, ba.z)); a[3] = fmaf(rs, a[3], fmaf(nrm, ca.w, ba.w));
                    gt[0] = fmaf(rs, gt[0], fmaf(nrm, cg.x, bg.x)); gt[1] = fmaf(rs, gt[1], fmaf(nrm, cg.y, bg.y));
                    gt[2] = fmaf(rs, gt[2], fmaf(nrm, cg.z, bg.z)); gt[3] = fmaf(rs, gt[3], fmaf(nrm, cg.w, bg.w));
                } else {
                    a[0] += ba.x; a[1] += ba.y; a[2] += ba.z; a[3] += ba.w;
                    gt[0] += bg.x; gt[1] += bg.y; gt[2] += bg.z; gt[3] += bg.w;
                }
                packed[g].x = pack_bf16(a[0] * gelu_erf_f(gt[0]), a[1] * gelu_erf_f(gt[1]));
                packed[g].y = pack_bf16(a[2] * gelu_erf_f(gt[2]), a[3] * gelu_erf_f(gt[3]));
            }
            const uint4 w = widen_pair(packed[0], packed[1]);
            if (row_ok) epi_st16(op + fi * 16, w);
        }
    }
}

}  // namespace
