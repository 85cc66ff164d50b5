// GroupNorm(32)+SiLU and LayerNorm over token-major bf16 activations (gfx950). All HBM-bound: 16-B (8 x bf16)
// coalesced accesses, fp32 statistics (the reference's GroupNorm32 computes in fp32:
// vwm/modules/diffusionmodules/util.py:214-216), wave64 shuffle reductions.
//
// GroupNorm: x[n_img][S][C], 32 groups of cpg = C/32 channels; statistics per (image-group, channel-group) where an
// image-group is `frames_per_group` consecutive images -- 1 for the 2-D norms, T for the temporal ResBlock's 5-D norm
// whose statistics span (C/32, T, H, W) (openaimodel.py:196,228 on `b c t h w`). Two launches:
//   stats : every thread owns one fixed 16-B channel chunk and strides over tokens, keeping 8 per-channel fp32
//           partial (sum, sumsq) pairs; partials are combined in a fixed order (no atomics) through LDS and a
//           per-chunk workspace, then a finalize launch produces mean / rstd -> bitwise reproducible results.
//   apply : same ownership; per-channel scale/shift folded once into 16 registers, then a pure streaming pass
//           y = silu(x*a + b).
#include <stdlib.h>

#include "common.h"
#include "vista_hip.h"

namespace {

constexpr int GN_TOK = 128;  // tokens per workgroup

// stats pass 1: per (image, 128-token chunk) partial (sum, sumsq) of the 32 groups, reduced in a FIXED order
// (thread partials -> LDS [r][channel] -> per-channel over r -> per-group over channels): bitwise reproducible.
// Two-source form (x2 != NULL): the tensor is the channel concat [x | x2] of C1 + (C - C1) channels (UNet skip concat, never
// materialised); every thread owns one fixed 16-B channel chunk, so the source choice is a per-thread constant.
// AMAX (fp8 output mode of the apply pass): the workgroup also writes max|x| of its elements to amax_part[img][chunk] (plain store: no
// atomics, nothing to clear); the apply pass reduces the partials of its image group.
template <bool AMAX>
__global__ void gn_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int C1, float* __restrict__ partial, int S, int C,
                                int CG, int R, int tok_per_wg, float* __restrict__ amax_part) {
    __shared__ float wave_amax[16];
    extern __shared__ float lds[];  // [2][R][C]
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const int tok0 = blockIdx.x * tok_per_wg;
    const int tok1 = min(tok0 + tok_per_wg, S);
    const int chunk = tid % CG, r = tid / CG;
    const int cpg = C >> 5;
    float* lsum = lds;
    float* lsq = lds + R * C;
    float amx = 0.f;
    if (r < R) {
        float sm[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
        const bool second = x2 != nullptr && chunk * 8 >= C1;
        const int ld = x2 == nullptr ? C : (second ? C - C1 : C1);
        const uint16_t* p = (second ? x2 + (chunk * 8 - C1) : x + chunk * 8) + ((size_t)img * S) * ld;
        int t = tok0 + r;
        for (; t + 3 * R < tok1; t += 4 * R) {  // 4 independent 16-B loads in flight per thread
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(p + (size_t)(t + u * R) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sm[e] += f[e];
                    sq[e] = fmaf(f[e], f[e], sq[e]);
                    if (AMAX) amx = fmaxf(amx, fabsf(f[e]));
                }
            }
        }
        for (; t < tok1; t += R) {
            const uint4 v = *(const uint4*)(p + (size_t)t * ld);
            float f[8];
            unpack8(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sm[e] += f[e];
                sq[e] = fmaf(f[e], f[e], sq[e]);
                if (AMAX) amx = fmaxf(amx, fabsf(f[e]));
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lsum[r * C + chunk * 8 + e] = sm[e];
            lsq[r * C + chunk * 8 + e] = sq[e];
        }
    }
    if (AMAX) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
        if ((tid & 63) == 0) wave_amax[tid >> 6] = amx;
    }
    __syncthreads();
    if (AMAX && tid == 0) {
        float m = wave_amax[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wave_amax[w]);
        amax_part[(size_t)img * gridDim.x + blockIdx.x] = m;
    }
    for (int c = tid; c < C; c += blockDim.x) {  // per-channel totals over the R token lanes, fixed order
        float a = 0.f, b = 0.f;
        for (int rr = 0; rr < R; ++rr) { a += lsum[rr * C + c]; b += lsq[rr * C + c]; }
        lsum[c] = a;
        lsq[c] = b;
    }
    __syncthreads();
    if (tid < 32) {
        float a = 0.f, b = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += lsum[c]; b += lsq[c]; }
        float* dst = partial + ((size_t)img * gridDim.x + blockIdx.x) * 64;
        dst[tid] = a;
        dst[32 + tid] = b;
    }
}

// stats pass 2: one workgroup per image-group; sums the (frames_per_group x chunks) partials of each of the 64 values
// in a fixed order (4 strided lanes per value, then a fixed 4-way combine) and writes the raw [32 sums | 32 sums of squares]
// (kept raw so that a pixel-sharded multi-GPU run can all-reduce them before the apply pass).
//
// Large images (the VAE decoder's 576x1024 levels have up to 64512 partials per group) take it in two fixed-order levels:
// level 1 (gridDim.y = splits > 1) folds `per` consecutive partials into the slot of the first one, in place -- a workgroup
// reads only its own range before it writes -- and gn_finalize_level2_kernel sums those slots (`per` partials apart).
__global__ void gn_finalize_kernel(float* __restrict__ partial, float* __restrict__ sums, int nparts, int per, int in_place) {
    __shared__ float red[4][64];
    const int v = threadIdx.x & 63, j = threadIdx.x >> 6;  // 256 threads
    const int first = blockIdx.y * per;
    const int cnt = min(per, nparts - first);
    float* src = partial + ((size_t)blockIdx.x * nparts + first) * 64 + v;
    float a = 0.f;
    for (int i = j; i < cnt; i += 4) a += src[(size_t)i * 64];
    red[j][v] = a;
    __syncthreads();
    if (threadIdx.x < 64) {
        const float t = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
        if (in_place) src[0] = t;
        else sums[(size_t)blockIdx.x * 64 + v] = t;
    }
}

__global__ void gn_finalize_level2_kernel(const float* __restrict__ partial, float* __restrict__ sums, int nparts, int per, int splits) {
    __shared__ float red[4][64];
    const int v = threadIdx.x & 63, j = threadIdx.x >> 6;
    const float* src = partial + (size_t)blockIdx.x * nparts * 64 + v;
    float a = 0.f;
    for (int i = j; i < splits; i += 4) a += src[(size_t)i * per * 64];
    red[j][v] = a;
    __syncthreads();
    if (threadIdx.x < 64) sums[(size_t)blockIdx.x * 64 + v] = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
}

// 16-byte accesses with / without the non-temporal hint (A/B hook VISTA_GN_NT of the apply pass: bit 0 = loads, bit 1 = stores; same bytes either way)
typedef unsigned gn_u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 gn_ld16(const uint16_t* p) {
    if constexpr (NT) {
        const gn_u32x4_t v = __builtin_nontemporal_load((const gn_u32x4_t*)p);
        return make_uint4(v[0], v[1], v[2], v[3]);
    } else {
        return *(const uint4*)p;
    }
}
template <bool NT>
__device__ __forceinline__ void gn_st16(uint16_t* p, const uint4 v) {
    if constexpr (NT) {
        const gn_u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, (gn_u32x4_t*)p);
    } else {
        *(uint4*)p = v;
    }
}

// FOLD (round 6): `stats` holds the image groups' STAGE-1 slots ([group][nparts][64]) instead of their finished sums, and every workgroup sums its own
// group's slots first -- the arithmetic of gn_finalize_kernel (four strided lanes per value, then the fixed (0 + 1) + (2 + 3) combine), so the
// result is bitwise that of the finalize launch it replaces. nparts <= GN_FOLD_MAX slots = at most 64 KiB read from L2 per workgroup; the finalize
// launch it removes was 7.9 us of launch overhead per norm (105 norms per step, profiles/r05_step_kernels.txt).
constexpr int GN_FOLD_MAX = 256;
template <bool NTL, bool NTS, bool FOLD>
__global__ void gn_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int C1, uint16_t* __restrict__ y, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ stats, int S, int C, int CG, int R,
                                int frames_per_group, float inv_cnt, float eps, int do_silu, int tok_per_wg, int nparts) {
    __shared__ float fold_red[FOLD ? 4 : 1][64];
    __shared__ float fold_tot[64];
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const int tok0 = blockIdx.x * tok_per_wg;
    const int tok1 = min(tok0 + tok_per_wg, S);
    const int chunk = tid % CG, r = tid / CG;
    const float* st = stats + (size_t)(img / frames_per_group) * 64;  // [32 sums | 32 sums of squares]
    if constexpr (FOLD) {
        const float* src = stats + (size_t)(img / frames_per_group) * nparts * 64;
        for (int idx = tid; idx < 256; idx += blockDim.x) {   // (workgroups of 192 .. 320 threads: the 4 x 64 (lane, value) pairs of the finalize kernel)
            const int v = idx & 63, j = idx >> 6;
            float a = 0.f;
            for (int i = j; i < nparts; i += 4) a += src[(size_t)i * 64 + v];
            fold_red[j][v] = a;
        }
        __syncthreads();
        if (tid < 64) fold_tot[tid] = (fold_red[0][tid] + fold_red[1][tid]) + (fold_red[2][tid] + fold_red[3][tid]);
        __syncthreads();
        st = fold_tot;
    }
    if (r >= R) return;
    const int cpg = C >> 5;
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e;
        const int g = c / cpg;
        const float mean = st[g] * inv_cnt;
        const float var = fmaxf(st[32 + g] * inv_cnt - mean * mean, 0.f);
        a[e] = gamma[c] * rsqrtf(var + eps);
        b[e] = beta[c] - mean * a[e];
    }
    const size_t base = ((size_t)img * S) * C + chunk * 8;
    const bool second = x2 != nullptr && chunk * 8 >= C1;
    const int ld = x2 == nullptr ? C : (second ? C - C1 : C1);
    const uint16_t* xs = (second ? x2 + (chunk * 8 - C1) : x + chunk * 8) + ((size_t)img * S) * ld;
    int t = tok0 + r;
    for (; t + 3 * R < tok1; t += 4 * R) {  // 4 independent 16-B loads in flight per thread
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gn_ld16<NTL>(xs + (size_t)(t + u * R) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[8];
            unpack8(v[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = fmaf(f[e], a[e], b[e]);
                f[e] = do_silu ? silu_f(w) : w;
            }
            gn_st16<NTS>(y + base + (size_t)(t + u * R) * C, pack8(f));
        }
    }
    for (; t < tok1; t += R) {
        const uint4 v = gn_ld16<NTL>(xs + (size_t)t * ld);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = fmaf(f[e], a[e], b[e]);
            f[e] = do_silu ? silu_f(w) : w;
        }
        gn_st16<NTS>(y + base + (size_t)t * C, pack8(f));
    }
}

// GroupNorm(+SiLU) apply with fp8 (e4m3) output and ONE scale per image group, for the fp8 convolutions of BASELINE config 5: a conv
// output pixel sums taps of its own image only, so a per-image activation scale factors out of the implicit-GEMM K-sum (a per-token
// scale would not). The scale needs no extra pass: with max|x| of the image group (gn_stats_kernel<true>) every channel's output is
// bounded by |a_c| max|x| + |b_c| (y = a_c x + b_c, a_c = gamma_c rstd, b_c = beta_c - mean a_c), SiLU only shrinks magnitudes above
// 0.2785, and e4m3 is a floating format, so a bound that is loose by a small factor costs no precision. Every workgroup of the group
// derives the same scale (fixed-order max over the C channels); workgroup (0, first image of the group) publishes it.
__global__ void gn_apply_fp8_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int C1, uint8_t* __restrict__ y, float* __restrict__ scale_out,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ stats,
                                    const float* __restrict__ amax_part, int S, int C, int CG, int R, int frames_per_group, float inv_cnt,
                                    float eps, int do_silu, int tok_per_wg) {
    __shared__ unsigned bound_bits, xmax_bits;
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const int tok0 = blockIdx.x * tok_per_wg;
    const int tok1 = min(tok0 + tok_per_wg, S);
    const int chunk = tid % CG, r = tid / CG;
    const int cpg = C >> 5;
    const int grp = img / frames_per_group;
    const float* st = stats + (size_t)grp * 64;  // [32 sums | 32 sums of squares]
    if (tid == 0) { bound_bits = 0u; xmax_bits = 0u; }
    __syncthreads();
    {   // max|x| of the image group from the statistics pass' per-workgroup maxima (frames_per_group x chunks values)
        const int nparts = frames_per_group * gridDim.x;
        const float* pp = amax_part + (size_t)grp * nparts;
        float m = 0.f;
        for (int i = tid; i < nparts; i += blockDim.x) m = fmaxf(m, pp[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) atomicMax(&xmax_bits, __float_as_uint(m));  // LDS, one per wave; a maximum does not depend on the order
    }
    __syncthreads();
    const float xmax = __uint_as_float(xmax_bits);
    float a[8], b[8];
    if (r < R) {
        float bound = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = chunk * 8 + e;
            const int g = c / cpg;
            const float mean = st[g] * inv_cnt;
            const float var = fmaxf(st[32 + g] * inv_cnt - mean * mean, 0.f);
            a[e] = gamma[c] * rsqrtf(var + eps);
            b[e] = beta[c] - mean * a[e];
            bound = fmaxf(bound, fmaf(fabsf(a[e]), xmax, fabsf(b[e])));
        }
        if (r == 0) atomicMax(&bound_bits, __float_as_uint(bound));
    }
    __syncthreads();
    if (r >= R) return;
    float bound = __uint_as_float(bound_bits) * 1.0001f;  // fp32 rounding of a x + b against the bound
    if (do_silu) bound = fmaxf(bound, 0.2785f);
    bound = fmaxf(bound, 1e-20f);
    const float sc = bound * (1.f / 448.f);
    const float inv = 448.f / bound;
    if (blockIdx.x == 0 && tid == 0 && img % frames_per_group == 0) scale_out[grp] = sc;
    const size_t base = ((size_t)img * S) * C + chunk * 8;
    const bool second = x2 != nullptr && chunk * 8 >= C1;
    const int ld = x2 == nullptr ? C : (second ? C - C1 : C1);
    const uint16_t* xs = (second ? x2 + (chunk * 8 - C1) : x + chunk * 8) + ((size_t)img * S) * ld;
    auto emit = [&](const uint4& v, int t) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float w = fmaf(f[e], a[e], b[e]);
            if (do_silu) w = silu_f(w);
            f[e] = fminf(fmaxf(w * inv, -448.f), 448.f);  // the cvt yields NaN above 448, it does not saturate
        }
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *(int2*)(y + base + (size_t)t * C) = make_int2(lo, hi);
    };
    int t = tok0 + r;
    for (; t + 3 * R < tok1; t += 4 * R) {  // 4 independent 16-B loads in flight per thread
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(xs + (size_t)(t + u * R) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) emit(v[u], t + u * R);
    }
    for (; t < tok1; t += R) emit(*(const uint4*)(xs + (size_t)t * ld), t);
}

// LayerNorm: one wave per row, NCH 16-B chunks per lane (C <= 512*NCH). Two-pass (mean, centred variance) in
// registers. Optional per-image pre-add vector and write-back of the (bf16-rounded) sum.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                        uint16_t* __restrict__ sum_out, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ addvec, int rows,
                                                        int C, int rows_per_vec, int ldv, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int CG = C >> 3;
    const int nw = gridDim.x * 4;
    float gm[NCH][8], bt[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ch = lane + 64 * j;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gm[j][e] = (ch < CG) ? gamma[ch * 8 + e] : 0.f;
            bt[j][e] = (ch < CG) ? beta[ch * 8 + e] : 0.f;
        }
    }
    const float inv_c = 1.f / (float)C;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += nw) {
        float f[NCH][8];
        float s = 0.f;
        const float* av = addvec ? addvec + (size_t)(row / rows_per_vec) * ldv : nullptr;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            if (ch < CG) {
                const uint4 v = *(const uint4*)(x + (size_t)row * C + ch * 8);
                unpack8(v, f[j]);
                if (av) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[j][e] += av[ch * 8 + e];
                    const uint4 w = pack8(f[j]);
                    if (sum_out) *(uint4*)(sum_out + (size_t)row * C + ch * 8) = w;
                    unpack8(w, f[j]);  // normalise the bf16-rounded sum (what later residuals read)
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[j][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[j][e] = 0.f;
            }
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            if (ch < CG) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[j][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            if (ch < CG) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf((f[j][e] - mean) * rstd, gm[j][e], bt[j][e]);
                *(uint4*)(y + (size_t)row * C + ch * 8) = pack8(o);
            }
        }
    }
}

// LayerNorm fused with per-row fp8 (e4m3) quantisation: one wave per row, the normalised row stays in registers: mean / centred variance,
// y = (x - mean) * rstd * gamma + beta, scale = max|y| / 448, q = e4m3(y / scale). The bf16 normalised tensor is never written.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_quant_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ qo, float* __restrict__ scale,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int rows, int C,
                                                              float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int CG = C >> 3;
    const int nw = gridDim.x * 4;
    const float inv_c = 1.f / (float)C;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += nw) {
        float f[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            if (ch < CG) {
                unpack8(*(const uint4*)(x + (size_t)row * C + ch * 8), f[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[j][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[j][e] = 0.f;
            }
        }
        const float mean = wave_sum(s) * inv_c;
        float qv = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            if (lane + 64 * j < CG) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[j][e] - mean; qv += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(qv) * inv_c + eps);
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            if (ch < CG) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[j][e] = fmaf((f[j][e] - mean) * rstd, gamma[ch * 8 + e], beta[ch * 8 + e]);
                    amax = fmaxf(amax, fabsf(f[j][e]));
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
        const float inv = 1.f / sc;
        if (lane == 0) scale[row] = sc;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ch = lane + 64 * j;
            if (ch < CG) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(f[j][e] * inv, -448.f), 448.f);  // the cvt yields NaN above 448, it does not saturate
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
                *(int2*)(qo + (size_t)row * C + ch * 8) = make_int2(lo, hi);
            }
        }
    }
}

// Row sums for a LayerNorm folded into the consumer GEMM: one wave per row, stats[row] = (sum x, sum x^2), lane partials over the
// row's 16-B chunks then a fixed butterfly -- bitwise reproducible.
__global__ __launch_bounds__(256) void rowstats_kernel(const uint16_t* __restrict__ x, float2* __restrict__ stats, int rows, int C, long long ldx) {
    const int lane = threadIdx.x & 63;
    const int CG = C >> 3;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        float s = 0.f, q = 0.f;
        for (int ch = lane; ch < CG; ch += 64) {
            const uint4 v = *(const uint4*)(x + (size_t)row * ldx + ch * 8);
            float f[8];
            unpack8(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s += f[e]; q = fmaf(f[e], f[e], q); }
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) stats[row] = make_float2(s, q);
    }
}

}  // namespace

namespace {
struct GnGeom { int CG, R, threads, nchunks, ngroups, tok; };
inline bool gn_geom(int n_img, int S, int C, int fpg, GnGeom& g) {
    if (n_img <= 0 || S <= 0 || C <= 0 || (C % 32) != 0 || (C % 8) != 0 || C > 8192) return false;
    if (fpg <= 0 || (n_img % fpg) != 0) return false;
    g.CG = C / 8;
    if (g.CG > 1024) return false;
    g.R = 256 / g.CG;
    if (g.R < 1) g.R = 1;
    g.threads = ((g.CG * g.R + 63) / 64) * 64;
    // wide channels leave one token row per pass (R = 1): 32 tokens per workgroup instead of 128, so the deep, small levels
    // (C >= 1024, S = 144 / 576) launch enough workgroups to hide their serial token loop
    g.tok = (g.R >= 2) ? GN_TOK : 32;
    g.nchunks = (S + g.tok - 1) / g.tok;
    g.ngroups = n_img / fpg;
    return true;
}
}  // namespace

namespace {
// stage 2 of the statistics: sums[g][64] <- the nparts stage-1 slots of image group g, in a fixed order (one or two levels)
int gn_finalize(float* partial_ws, float* sums, int ngroups, int nparts, hipStream_t stream) {
    if (nparts <= 256) {
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(ngroups), dim3(256), 0, stream, partial_ws, sums, nparts, nparts, 0);
    } else {
        int per = 16;
        while ((long long)per * per < nparts) per += 4;  // ~sqrt(nparts), a multiple of the 4 accumulation lanes
        const int splits = (nparts + per - 1) / per;
        // level 1: partial[g][k*per] <- sum of partial[g][k*per .. k*per+per)
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(ngroups, splits), dim3(256), 0, stream, partial_ws, sums, nparts, per, 1);
        VK_CHECK_LAUNCH();
        // level 2: sums[g] <- sum over the `splits` slots, which sit `per` partials apart (group stride = nparts partials)
        hipLaunchKernelGGL(gn_finalize_level2_kernel, dim3(ngroups), dim3(256), 0, stream, (const float*)partial_ws, sums, nparts, per, splits);
    }
    VK_CHECK_LAUNCH();
    return VK_OK;
}

int gn_stats(const void* x, const void* x2, int C1, float* sums, float* partial_ws, int32_t n_img, int32_t S, int32_t C, int32_t frames_per_group,
             void* stream_, float* amax_part = nullptr, bool finalize = true) {
    hipStream_t stream = (hipStream_t)stream_;
    GnGeom g;
    if (!x || !sums || !partial_ws || !gn_geom(n_img, S, C, frames_per_group, g)) return VK_EINVAL;
    dim3 grid(g.nchunks, n_img);
    const size_t lds_bytes = (size_t)2 * g.R * C * sizeof(float);
    if (amax_part)
        hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(g.threads), lds_bytes, stream, (const uint16_t*)x, (const uint16_t*)x2, C1, partial_ws, S, C,
                           g.CG, g.R, g.tok, amax_part);
    else
        hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(g.threads), lds_bytes, stream, (const uint16_t*)x, (const uint16_t*)x2, C1, partial_ws, S, C,
                           g.CG, g.R, g.tok, (float*)nullptr);
    VK_CHECK_LAUNCH();
    if (!finalize) return VK_OK;   // (the apply pass folds the slots itself: gn_fold_parts)
    return gn_finalize(partial_ws, sums, g.ngroups, frames_per_group * g.nchunks, stream);
}

// Slots per image group of the statistics pass when the apply pass may fold them itself (0 = keep the finalize launch): at most GN_FOLD_MAX,
// VISTA_GN_FOLD=0 restores the three-launch GroupNorm everywhere (A/B hook; bitwise the same output either way).
int gn_fold_parts(int32_t n_img, int32_t S, int32_t C, int32_t frames_per_group) {
    static const bool on = [] { const char* e = getenv("VISTA_GN_FOLD"); return !e || atoi(e) != 0; }();
    GnGeom g;
    if (!on || !gn_geom(n_img, S, C, frames_per_group, g)) return 0;
    const long long np = (long long)frames_per_group * g.nchunks;
    return np <= GN_FOLD_MAX ? (int)np : 0;
}

// fold_nparts > 0: `sums` are the groups' stage-1 slots ([group][fold_nparts][64]) and the apply workgroups fold them themselves (no finalize launch)
int gn_apply(const void* x, const void* x2, int C1, void* y, const float* gamma, const float* beta, const float* sums, int32_t n_img, int32_t S,
             int32_t C, int32_t frames_per_group, float count, float eps, int32_t silu, void* stream_, int fold_nparts = 0) {
    hipStream_t stream = (hipStream_t)stream_;
    GnGeom g;
    if (!x || !y || !gamma || !beta || !sums || count <= 0.f || !gn_geom(n_img, S, C, frames_per_group, g)) return VK_EINVAL;
    const int tok_per_wg = g.tok;
    dim3 grid(g.nchunks, n_img);
    // Non-temporal hints for tensors that do not fit the 256 MiB Infinity Cache anyway (levels 0 and 1 at the BASELINE window: 295 / 147 MB): the
    // output is too large to survive until its consumer, and the input -- dead after this pass at most call sites; the ResBlock's first norm input
    // (re-read as the skip of conv2) and the transformer's opening norm input (the block residual) are read once more, much later, from HBM either
    // way at this size -- should not evict what the neighbouring kernels keep there. The stores carry most of the gain, the load hint the rest
    // (gn+conv 0.958 stores only / 0.940 both); the whole-step A/B (which contains the two re-reading call sites) is the one that decided. Same box, alternated processes (tools/gn_nt_ab.py, profiles/r05_gn_nontemporal_ab.txt): GroupNorm (statistics + fold +
    // apply) 0.190 -> 0.149 ms at level 0, 0.082 -> 0.078 at level 1, unchanged at level 2; GroupNorm + the 3x3 convolution that reads it 0.958 ->
    // 0.940 / 0.853 -> 0.843 / 0.783 -> 0.79 (level 2 fits the cache: no hint there); step 164.1 -> 163.6 ms. Bitwise the same output.
    // VISTA_GN_NT = 0..3 forces (bit 0 = loads, bit 1 = stores) for an A/B; unset = the size rule.
    static const int nt_env = [] { const char* e = getenv("VISTA_GN_NT"); return e ? atoi(e) & 3 : -1; }();
    const int nt_mode = nt_env >= 0 ? nt_env : ((long long)n_img * S * C * 2 >= (96LL << 20) ? 3 : 0);
    if (fold_nparts < 0 || fold_nparts > GN_FOLD_MAX) return VK_EINVAL;
#define VK_GN_APPLY(NL, NS, FO) hipLaunchKernelGGL((gn_apply_kernel<NL, NS, FO>), grid, dim3(g.threads), 0, stream, (const uint16_t*)x, (const uint16_t*)x2, C1, (uint16_t*)y, \
                                                   gamma, beta, sums, S, C, g.CG, g.R, frames_per_group, 1.f / count, eps, silu, tok_per_wg, fold_nparts)
    if (fold_nparts > 0) {
        if (nt_mode == 3) VK_GN_APPLY(true, true, true);
        else if (nt_mode == 2) VK_GN_APPLY(false, true, true);
        else if (nt_mode == 1) VK_GN_APPLY(true, false, true);
        else VK_GN_APPLY(false, false, true);
    } else {
        if (nt_mode == 3) VK_GN_APPLY(true, true, false);
        else if (nt_mode == 2) VK_GN_APPLY(false, true, false);
        else if (nt_mode == 1) VK_GN_APPLY(true, false, false);
        else VK_GN_APPLY(false, false, false);
    }
#undef VK_GN_APPLY
    VK_CHECK_LAUNCH();
    return VK_OK;
}
}  // namespace

extern "C" int vk_groupnorm_stats_bf16(const void* x, float* sums, float* partial_ws, int32_t n_img, int32_t S, int32_t C,
                                       int32_t frames_per_group, void* stream_) {
    return gn_stats(x, nullptr, 0, sums, partial_ws, n_img, S, C, frames_per_group, stream_);
}

extern "C" int vk_groupnorm_apply_bf16(const void* x, void* y, const float* gamma, const float* beta, const float* sums, int32_t n_img,
                                       int32_t S, int32_t C, int32_t frames_per_group, float count, float eps, int32_t silu, void* stream_) {
    return gn_apply(x, nullptr, 0, y, gamma, beta, sums, n_img, S, C, frames_per_group, count, eps, silu, stream_);
}

// ABI v7: the apply pass on STAGE-1 slots (a GEMM epilogue's VkGemmDesc.gnstat_out, one slot per 64 output rows: nchunks = S / 64 per image), folding
// them itself -- two launches per GroupNorm (producer, apply) instead of three. Needs frames_per_group * nchunks <= vk_groupnorm_fold_max() (else
// VK_EINVAL: run vk_groupnorm_finalize_partials + vk_groupnorm_apply_bf16). Bitwise the output of that pair.
extern "C" int vk_groupnorm_fold_max(void) {
    static const bool on = [] { const char* e = getenv("VISTA_GN_FOLD"); return !e || atoi(e) != 0; }();
    return on ? GN_FOLD_MAX : 0;
}

extern "C" int vk_groupnorm_apply_partials_bf16(const void* x, void* y, const float* gamma, const float* beta, const float* partial, int32_t n_img,
                                                int32_t S, int32_t C, int32_t nchunks, int32_t frames_per_group, float count, float eps, int32_t silu,
                                                void* stream_) {
    if (!partial || nchunks <= 0 || frames_per_group <= 0 || (long long)frames_per_group * nchunks > GN_FOLD_MAX) return VK_EINVAL;
    return gn_apply(x, nullptr, 0, y, gamma, beta, partial, n_img, S, C, frames_per_group, count, eps, silu, stream_, frames_per_group * nchunks);
}

// ABI v6: stage 2 alone, on the stage-1 slots a GEMM epilogue wrote (VkGemmDesc.gnstat_out: one slot per 64 output rows)
extern "C" int vk_groupnorm_finalize_partials(float* partial, float* sums, int32_t n_img, int32_t nchunks, int32_t frames_per_group, void* stream_) {
    if (!partial || !sums || n_img <= 0 || nchunks <= 0 || frames_per_group <= 0 || (n_img % frames_per_group) != 0) return VK_EINVAL;
    if ((long long)frames_per_group * nchunks > (1LL << 24)) return VK_EINVAL;
    return gn_finalize(partial, sums, n_img / frames_per_group, frames_per_group * nchunks, (hipStream_t)stream_);
}

extern "C" int vk_groupnorm_silu_cat_bf16(const void* x1, const void* x2, void* y, const float* gamma, const float* beta, float* stats_ws,
                                          int32_t n_img, int32_t S, int32_t C1, int32_t C2, int32_t frames_per_group, float eps, int32_t silu,
                                          void* stream_) {
    if (!x1 || !x2 || !stats_ws || frames_per_group <= 0 || n_img <= 0 || C1 <= 0 || C2 <= 0 || (C1 % 8) != 0 || (C2 % 8) != 0) return VK_EINVAL;
    const int C = C1 + C2;
    float* partial = stats_ws + (size_t)(n_img / frames_per_group) * 64;
    const int fold = gn_fold_parts(n_img, S, C, frames_per_group);
    int rc = gn_stats(x1, x2, C1, stats_ws, partial, n_img, S, C, frames_per_group, stream_, nullptr, fold == 0);
    if (rc != VK_OK) return rc;
    const float count = (float)(C / 32) * (float)S * (float)frames_per_group;
    return gn_apply(x1, x2, C1, y, gamma, beta, fold ? partial : stats_ws, n_img, S, C, frames_per_group, count, eps, silu, stream_, fold);
}

extern "C" int vk_groupnorm_silu_fp8(const void* x1, const void* x2, void* y8, float* scale_out, const float* gamma, const float* beta,
                                     float* stats_ws, int32_t n_img, int32_t S, int32_t C1, int32_t C2, int32_t frames_per_group, float eps,
                                     int32_t silu, void* stream_) {
    if (!x1 || !y8 || !scale_out || !gamma || !beta || !stats_ws || frames_per_group <= 0 || n_img <= 0 || C1 <= 0 || C2 < 0 || (C1 % 8) != 0 ||
        (C2 % 8) != 0 || (x2 == nullptr) != (C2 == 0))
        return VK_EINVAL;
    const int C = C1 + C2;
    GnGeom g;
    if (!gn_geom(n_img, S, C, frames_per_group, g)) return VK_EINVAL;
    // workspace: [groups][64] sums | [n_img][chunks] per-workgroup max|x| | [n_img][chunks][64] partial sums
    float* amax = stats_ws + (size_t)g.ngroups * 64;
    float* partial = amax + (size_t)n_img * g.nchunks;
    int rc = gn_stats(x1, x2, C1, stats_ws, partial, n_img, S, C, frames_per_group, stream_, amax);
    if (rc != VK_OK) return rc;
    const float count = (float)(C / 32) * (float)S * (float)frames_per_group;
    dim3 grid(g.nchunks, n_img);
    hipLaunchKernelGGL(gn_apply_fp8_kernel, grid, dim3(g.threads), 0, (hipStream_t)stream_, (const uint16_t*)x1, (const uint16_t*)x2, C1, (uint8_t*)y8,
                       scale_out, gamma, beta, (const float*)stats_ws, (const float*)amax, S, C, g.CG, g.R, frames_per_group, 1.f / count, eps, silu,
                       g.tok);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

extern "C" int vk_rowstats_bf16(const void* x, float* stats, int32_t rows, int32_t C, int64_t ldx, void* stream_) {
    if (!x || !stats || rows <= 0 || C <= 0 || (C % 8) != 0 || C > 1536 || ldx < C || (ldx % 8) != 0) return VK_EINVAL;
    long long want = ((long long)rows + 3) / 4;
    const long long cap = 256LL * 16;
    hipLaunchKernelGGL(rowstats_kernel, dim3((int)(want < cap ? want : cap)), dim3(256), 0, (hipStream_t)stream_, (const uint16_t*)x, (float2*)stats,
                       rows, C, (long long)ldx);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

extern "C" int vk_groupnorm_silu_bf16(const void* x, void* y, const float* gamma, const float* beta, float* stats_ws, int32_t n_img,
                                      int32_t S, int32_t C, int32_t frames_per_group, float eps, int32_t silu, void* stream_) {
    if (!stats_ws || frames_per_group <= 0 || n_img <= 0) return VK_EINVAL;
    // workspace: [n_img/fpg][64] sums, then [n_img][chunks][64] partial sums
    float* partial = stats_ws + (size_t)(n_img / frames_per_group) * 64;
    const int fold = gn_fold_parts(n_img, S, C, frames_per_group);
    int rc = gn_stats(x, nullptr, 0, stats_ws, partial, n_img, S, C, frames_per_group, stream_, nullptr, fold == 0);
    if (rc != VK_OK) return rc;
    const float count = (float)(C / 32) * (float)S * (float)frames_per_group;
    return gn_apply(x, nullptr, 0, y, gamma, beta, fold ? partial : stats_ws, n_img, S, C, frames_per_group, count, eps, silu, stream_, fold);
}

extern "C" int vk_layernorm_bf16(const void* x, void* y, void* sum_out, const float* gamma, const float* beta, const float* addvec,
                                 int32_t rows, int32_t C, int32_t rows_per_vec, int32_t ldv, float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || (C % 8) != 0 || C > 1536) return VK_EINVAL;
    if (addvec && rows_per_vec <= 0) return VK_EINVAL;
    const int CG = C / 8;
    const int nch = (CG + 63) / 64;
    long long want = ((long long)rows + 3) / 4;
    const long long cap = 256LL * 8;
    const int grid = (int)(want < cap ? want : cap);
#define LN_LAUNCH(N)                                                                                                              \
    hipLaunchKernelGGL(layernorm_kernel<N>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, (uint16_t*)sum_out, \
                       gamma, beta, addvec, rows, C, rows_per_vec, ldv, eps)
    if (nch == 1) LN_LAUNCH(1);
    else if (nch == 2) LN_LAUNCH(2);
    else LN_LAUNCH(3);
#undef LN_LAUNCH
    VK_CHECK_LAUNCH();
    return VK_OK;
}

extern "C" int vk_layernorm_quant_fp8(const void* x, void* q, float* scale, const float* gamma, const float* beta, int32_t rows, int32_t C,
                                      float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !q || !scale || !gamma || !beta || rows <= 0 || C <= 0 || (C % 8) != 0 || C > 1536) return VK_EINVAL;
    const int nch = (C / 8 + 63) / 64;
    long long want = ((long long)rows + 3) / 4;
    const long long cap = 256LL * 8;
    const int grid = (int)(want < cap ? want : cap);
#define LNQ_LAUNCH(N)                                                                                                                     \
    hipLaunchKernelGGL(layernorm_quant_kernel<N>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)x, (uint8_t*)q, scale, gamma, beta, rows, C, eps)
    if (nch == 1) LNQ_LAUNCH(1);
    else if (nch == 2) LNQ_LAUNCH(2);
    else LNQ_LAUNCH(3);
#undef LNQ_LAUNCH
    VK_CHECK_LAUNCH();
    return VK_OK;
}
