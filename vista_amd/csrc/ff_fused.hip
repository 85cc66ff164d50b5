// Fused FeedForward of the level-0 transformers (gfx950): GEGLU in-projection -> value * gelu(gate) -> out-projection in ONE kernel.
//   reference: vwm/modules/attention.py:85-128 (GEGLU, FeedForward.net), :524 (x = ff(norm3(x)) + x),
//              vwm/modules/video_attention.py:119-121,137-141 (ff_in / ff of the temporal block, AlphaBlender mix)
//   out[m][:] = epilogue2( sum_h  H[m][h] * W2[:][h] ),   H[m][h] = bf16( a * gelu(g) ),
//   (a, g) = LNfold( sum_k x[m][k] * W1[packed rows of h][k] ) + bias1            (the folded LayerNorm of VkGemmDesc.ln_*)
// The unfused pair writes H (460800 x 1280 bf16 = 1.18 GB per level-0 FeedForward) to HBM and reads it back: 15 FeedForwards x 2.36 GB
// per denoise step, and its GEGLU kernel runs the K-loop and the 32-gelu epilogue of a tile one after the other (matrix core 35 % busy,
// VALU 45 %, never together: profiles/r03_pmc_gemm_sq.txt). Here H never leaves the CU:
//   * a workgroup (8 waves, two per SIMD) owns 128 tokens; wave (tg, c) keeps the x fragments of its 32 tokens (all K = 320: 80 VGPRs)
//     and a 32-token x 160-column slice of the OUTPUT accumulator (80 VGPRs) in registers for the whole tile;
//   * the hidden dimension streams through in chunks of 32 (= 64 packed GEGLU rows = two 32-row MFMA fragments, one per wave of a
//     token pair): per chunk a wave runs 20 MFMAs of the in-projection, forms its 16 hidden values per token in registers (folded
//     LayerNorm, bias, gelu), hands them to its pair wave through 8 KB of LDS, and runs the 10 out-projection MFMAs of the PREVIOUS chunk
//     -- whose matrix work has no dependence on this chunk's gelu arithmetic, so VALU and MFMA overlap inside a wave;
//   * W1 / W2 chunks stream L2 -> LDS by LDS-DMA in a 2-deep ring (40 + 20 KB per chunk), one barrier per chunk;
//   * the hidden values are the out-projection's B operand in the order the lanes produced them: W2 is packed with its K axis
//     permuted inside every 16-group ([0-3, 8-11, 4-7, 12-15], ops.pack_linear(kperm16=True)), so no shuffle is needed;
//   * the out-projection's epilogue is the LINEAR family's LDS-staged epilogue (gemm_common.h): bias, residuals, AlphaBlender blend,
//     row vectors, LayerNorm row-sum emission -- same arithmetic, same operation order.
// Numerics: H is rounded to bf16 exactly where the unfused pair rounds it; the in-projection accumulates in the same order; the
// out-projection sums the same products with the 16-wide MFMA reduction walking a permuted slot order (fp32, not bitwise).
#include "common.h"
#include "vista_hip.h"

#include "gemm_common.h"

namespace {

constexpr int FF_C = 320;              // level-0 width: K of the in-projection, N of the out-projection
constexpr int FF_BM = 128;             // tokens per workgroup tile: four waves x 32 tokens, ONE wave per SIMD (up to 512 VGPRs each)
constexpr int FF_HC = 32;              // hidden units per chunk (= 64 packed GEGLU rows = two 32-row MFMA fragments)
constexpr int FF_MAXH = 1280;          // hidden width the per-column vector region is sized for
constexpr int FF_NT = 256;
constexpr int W1_SLOT = 2 * FF_HC * FF_C * 2;   // 64 packed rows x 320 k x 2 B = 40960: five [64 rows][64 k] slabs of 8 KB
constexpr int W2_SLOT = FF_C * FF_HC * 2;       // 320 rows x 32 hidden x 2 B = 20480
constexpr int OFF_W1 = 0, OFF_W2 = 2 * W1_SLOT, OFF_VEC = OFF_W2 + 2 * W2_SLOT;
constexpr int OFF_LN = OFF_VEC + 2 * (2 * FF_MAXH) * 4, FF_LDS = OFF_LN + FF_BM * 8;
static_assert(FF_LDS <= 163840, "LDS budget");
static_assert(epi_vec_floats(FF_C) * 4 <= W1_SLOT, "the out-projection epilogue's vectors overlay W1 slot 0");

// DBG (timing experiments only, results wrong): 1 = no LDS-DMA inside the chunk steps, 2 = gelu replaced by a plain product, 4 = no out-projection MFMAs
template <int DBG>
__global__ __launch_bounds__(FF_NT, 1) void ff_fused_kernel(const VkGemmDesc p1, const VkGemmDesc p2) {
    __shared__ __attribute__((aligned(16))) char smem[FF_LDS];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = token group: tokens 32 wave .. + 31 of the tile
    const int l31 = lane & 31, lh = lane >> 5;
    const int Hd = p2.K;             // hidden width (1280)
    const int nit = Hd / FF_HC;      // chunks
    const int ntiles = (p1.M + FF_BM - 1) / FF_BM;

    const uint16_t* __restrict__ W1g = (const uint16_t*)p1.Wt;
    const uint16_t* __restrict__ W2g = (const uint16_t*)p2.Wt;
    const uint16_t* __restrict__ Xg = (const uint16_t*)p1.A;
    float* const vecb = (float*)(smem + OFF_VEC);        // bias of the in-projection, packed row order (zeros if absent)
    float* const vecc = vecb + 2 * FF_MAXH;              // LayerNorm column sums (zeros without the fold)
    float2* const lnrow = (float2*)(smem + OFF_LN);

    // ---- per-lane LDS-DMA sources (element offsets; the XOR swizzle of the LDS image lives in the source address) ----
    // W1 chunk = five slabs (k = 64 i ..) of [64 rows][128 B]; a 1 KB piece = 8 rows of a slab; wave w stages row blocks w and w + 4 of every slab
    const int w1r = 8 * wave + (lane >> 3);                                    // packed row inside the chunk (first row block); (w1r + 32) >> 1 has the same low bits
    const int w1off = w1r * FF_C + 8 * ((lane & 7) ^ ((w1r >> 1) & 7));
    // W2 chunk = [320 rows][64 B]; a piece = 16 rows; wave w stages pieces w, w + 4, .., w + 16
    const int w2off = (16 * wave + (lane >> 2)) * Hd + 8 * ((lane & 3) ^ ((lane >> 4) & 3));   // ((row >> 2) & 3) == (lane >> 4) & 3

    // ---- per-lane fragment read offsets ----
    const int sw = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) frag_off[k4] = ((k4 * 2 + lh) ^ sw) << 4;
    const int w1row_off = l31 * 128;                     // fragment c of the chunk: + 32 c rows = 4096 c bytes
    const int sw2 = (l31 >> 2) & 3;
    int w2f_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w2f_off[s] = l31 * 64 + (((2 * s + lh) ^ sw2) << 4);   // + 32 f rows = 2048 f bytes; ((n >> 2) & 3) == sw2 for every f

    // ---- once per workgroup: per-column vectors of the in-projection ----
    for (int i = tid; i < (2 * Hd) / 4; i += FF_NT) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f), cs = b;
        if (p1.bias) b = *(const float4*)(p1.bias + 4 * i);
        if (p1.ln_stats) cs = *(const float4*)(p1.ln_colsum + 4 * i);
        *(float4*)(vecb + 4 * i) = b;
        *(float4*)(vecc + 4 * i) = cs;
    }

    auto dma_w1 = [&](int j, int slot) {
        const uint16_t* src = W1g + (size_t)j * (2 * FF_HC * FF_C) + w1off;
        char* dst = smem + OFF_W1 + slot * W1_SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 64 * i), (lptr_t)(dst + i * 8192), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 64 * i + 32 * FF_C), (lptr_t)(dst + i * 8192 + 4096), 16, 0, 0);
        }
    };
    auto dma_w2 = [&](int j, int slot) {
        const uint16_t* src = W2g + j * FF_HC + w2off;
        char* dst = smem + OFF_W2 + slot * W2_SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(64 * i) * Hd), (lptr_t)(dst + i * 4096), 16, 0, 0);
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * FF_BM;
        dma_w1(0, 0);
        // x fragments of this wave's 32 tokens: B operand of the in-projection, lane (l31, lh) holds k = 16 ks + 8 lh .. + 7 of token l31
        bf16x8_t xf[20];
        {
            int m = m0 + 32 * wave + l31;
            if (m >= p1.M) m = p1.M - 1;
            const uint16_t* xr = Xg + (size_t)m * p1.lda + 8 * lh;
#pragma unroll
            for (int ks = 0; ks < 20; ++ks) xf[ks] = *(const bf16x8_t*)(xr + 16 * ks);
        }
        if (p1.ln_stats != nullptr) {
            for (int r = tid; r < FF_BM; r += FF_NT) {
                const int m = m0 + r;
                lnrow[r] = ln_row_stats(p1, m < p1.M ? m : p1.M - 1);
            }
        }
        __syncthreads();
        float rs = 1.f, nrm = 0.f;
        if (p1.ln_stats != nullptr) { const float2 t = lnrow[32 * wave + l31]; rs = t.y; nrm = -t.x * t.y; }

        f32x16_t O[10][1];
#pragma unroll
        for (int f = 0; f < 10; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[f][0][r] = 0.f;

        // ---- one chunk step, hand-pipelined (one wave per SIMD: all overlap has to come from inside the wave) ----
        // Step(jc): in-projection MFMAs of chunk jc + 1 (40, two chains) woven 1 : ~7 with the GEGLU VALU work of chunk jc (its accumulators
        // Sc were finished by the previous step), then the 20 out-projection MFMAs of chunk jc on the hidden values the gelu left in
        // registers. Fragment reads run one group (5 MFMAs) ahead in a second register set; sched_barrier(0) pins the weave.
        // LDS at the top of the step: W1(jc + 1) in slot `s1`, W2(jc) in slot `s2`; the other two slots are free and receive W1(jc + 2) /
        // W2(jc + 1) (15 LDS-DMA pieces, issued two per MFMA group).
        auto step = [&](const int jc, f32x16_t (&Sc)[2], f32x16_t (&Sn)[2], const int s1, const int s2, const int jw1, const int jw2) {
            const char* w1s = smem + OFF_W1 + s1 * W1_SLOT + w1row_off;
            const char* w2s = smem + OFF_W2 + s2 * W2_SLOT;
            const uint16_t* d1 = W1g + (size_t)jw1 * (2 * FF_HC * FF_C) + w1off;
            char* l1 = smem + OFF_W1 + (s1 ^ 1) * W1_SLOT + wave * 1024;
            const uint16_t* d2 = W2g + jw2 * FF_HC + w2off;
            char* l2 = smem + OFF_W2 + (s2 ^ 1) * W2_SLOT + wave * 1024;
            bf16x8_t fr[2][5];
            float2 vv[2][4];
            uint32_t hq[2][4];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 16; ++r) Sn[cc][r] = 0.f;
            auto rd1 = [&](const int grp, bf16x8_t (&f)[5]) {   // in-projection MFMA i = 5 grp + n: k-substep i >> 1, fragment i & 1
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    const int i = 5 * grp + n, ks = i >> 1, cc = i & 1;
                    f[n] = *(const bf16x8_t*)(w1s + (ks >> 2) * 8192 + cc * 4096 + frag_off[ks & 3]);
                }
            };
            auto rdv = [&](const int u, float2 (&v)[4]) {       // unit u: fragment u >> 2, quad (u >> 1) & 1, element pair u & 1
                const int np = 2 * FF_HC * jc + 32 * (u >> 2) + 8 * ((u >> 1) & 1) + 4 * lh + 2 * (u & 1);
                v[0] = *(const float2*)(vecb + np); v[1] = *(const float2*)(vecb + np + 16);
                v[2] = *(const float2*)(vecc + np); v[3] = *(const float2*)(vecc + np + 16);
            };
            auto rd2 = [&](const int grp, bf16x8_t (&f)[5]) {   // out-projection MFMA i = 5 grp + n: k-substep i / 10, column fragment i % 10
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    const int i = 5 * grp + n;
                    f[n] = *(const bf16x8_t*)(w2s + w2f_off[i / 10] + (i % 10) * 2048);
                }
            };
            rd1(0, fr[0]);
            rdv(0, vv[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // reads of the next group, two DMA pieces, then MFMA n of this group next to stage n of this unit's two gates
                if (u < 7) { rd1(u + 1, fr[(u + 1) & 1]); rdv(u + 1, vv[(u + 1) & 1]); }
                else rd2(0, fr[0]);
                if (DBG & 1) {
                } else if (u < 5) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(d1 + 64 * u), (lptr_t)(l1 + u * 8192), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t)(d1 + 64 * u + 32 * FF_C), (lptr_t)(l1 + u * 8192 + 4096), 16, 0, 0);
                } else {
                    __builtin_amdgcn_global_load_lds((gptr_t)(d2 + (size_t)(64 * (2 * u - 10)) * Hd), (lptr_t)(l2 + (2 * u - 10) * 4096), 16, 0, 0);
                    if (u < 7) __builtin_amdgcn_global_load_lds((gptr_t)(d2 + (size_t)(64 * (2 * u - 9)) * Hd), (lptr_t)(l2 + (2 * u - 9) * 4096), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int cq = u >> 2, e0 = 4 * ((u >> 1) & 1) + 2 * (u & 1);
                const float2(&v)[4] = vv[u & 1];
                const bf16x8_t(&f)[5] = fr[u & 1];
                float a0, a1, g0, g1, t0, t1, q0, q1;
#define FF_MMA1(n) { const int i = 5 * u + (n); Sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[n], xf[i >> 1], Sn[i & 1], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                FF_MMA1(0)
                a0 = fmaf(rs, Sc[cq][e0], fmaf(nrm, v[2].x, v[0].x)); a1 = fmaf(rs, Sc[cq][e0 + 1], fmaf(nrm, v[2].y, v[0].y));
                g0 = fmaf(rs, Sc[cq][e0 + 8], fmaf(nrm, v[3].x, v[1].x)); g1 = fmaf(rs, Sc[cq][e0 + 9], fmaf(nrm, v[3].y, v[1].y));
                __builtin_amdgcn_sched_barrier(0);
                FF_MMA1(1)
                t0 = fminf(g0 * g0, 50.f); t1 = fminf(g1 * g1, 50.f);
                q0 = fmaf(t0, fmaf(t0, 0.0010142630f, -0.10677572f), -2.3011212f); q1 = fmaf(t1, fmaf(t1, 0.0010142630f, -0.10677572f), -2.3011212f);
                __builtin_amdgcn_sched_barrier(0);
                FF_MMA1(2)
                if (DBG & 2) { t0 = g0 * q0; t1 = g1 * q1; } else { t0 = __builtin_amdgcn_exp2f(g0 * q0); t1 = __builtin_amdgcn_exp2f(g1 * q1); }
                a0 *= g0; a1 *= g1;
                __builtin_amdgcn_sched_barrier(0);
                FF_MMA1(3)
                if (DBG & 2) { t0 = 1.f + t0; t1 = 1.f + t1; } else { t0 = __builtin_amdgcn_rcpf(1.f + t0); t1 = __builtin_amdgcn_rcpf(1.f + t1); }
                __builtin_amdgcn_sched_barrier(0);
                FF_MMA1(4)
                hq[cq][e0 >> 1] = pack_bf16(a0 * t0, a1 * t1);
                __builtin_amdgcn_sched_barrier(0);
#undef FF_MMA1
            }
            bf16x8_t hf[2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) hf[cc] = __builtin_bit_cast(bf16x8_t, make_uint4(hq[cc][0], hq[cc][1], hq[cc][2], hq[cc][3]));
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) {
                if (v2 < 3) rd2(v2 + 1, fr[(v2 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    const int i = 5 * v2 + n;
                    if (!(DBG & 4)) O[i % 10][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[v2 & 1][n], hf[i / 10], O[i % 10][0], 0, 0, 0);
                    else asm volatile("" :: "v"(fr[v2 & 1][n]), "v"(hf[i / 10]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // in-projection of chunk 0 on its own (prologue of the pipeline): two chains of 20 dependent MFMAs
        auto gemm1_first = [&](f32x16_t (&S)[2]) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[cc][r] = 0.f;
            const char* w1s = smem + OFF_W1 + w1row_off;
#pragma unroll
            for (int ks = 0; ks < 20; ++ks) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const bf16x8_t wf = *(const bf16x8_t*)(w1s + (ks >> 2) * 8192 + cc * 4096 + frag_off[ks & 3]);
                    S[cc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[ks], S[cc], 0, 0, 0);
                }
            }
        };

        f32x16_t Sa[2], Sb[2];
        dma_w1(1 < nit ? 1 : 0, 1);
        dma_w2(0, 0);
        gemm1_first(Sa);         // chunk 0 (W1 slot 0 was complete at the barrier above)
        __syncthreads();         // W1(1), W2(0) landed; slot W1[0] is free
        for (int j = 0; j < nit; j += 2) {   // nit is even (validated): two chunks per trip so that Sa / Sb keep static names
            // chunk j in Sa; chunk j + 1 -> Sb from W1 slot 1; W2(j) in slot 0; stream W1(j + 2) -> slot 0, W2(j + 1) -> slot 1
            step(j, Sa, Sb, 1, 0, j + 2 < nit ? j + 2 : nit - 1, j + 1);
            __syncthreads();
            // chunk j + 1 in Sb; chunk j + 2 -> Sa from slot 0 (past the end: a harmless recomputation of the last chunk)
            step(j + 1, Sb, Sa, 0, 1, j + 3 < nit ? j + 3 : nit - 1, j + 2 < nit ? j + 2 : nit - 1);
            __syncthreads();
        }
        // both W1 slots are dead now: stage the out-projection epilogue's vectors in slot 0
        float* const epi_vec = (float*)(smem + OFF_W1);
        const EpiPlan eplan = epi_plan<EPI_LINEAR, false, FF_BM, FF_C>(p2, m0, 0);
        if (eplan.fast) {
            epi_stage_vectors<FF_C, FF_NT>(p2, epi_vec, 0, eplan, tid);
            __syncthreads();
            gemm_epilogue_linear_lds<10, 1, 1, 10, FF_C, 512>(p2, O, m0, 0, wave, 0, l31, lh, 0, nullptr, epi_vec, eplan.img0);
        } else {
            gemm_epilogue<EPI_LINEAR, false, 10, 1, 1, 10>(p2, O, m0, 0, wave, 0, l31, lh, 0, nullptr);
        }
        if (tile + (int)gridDim.x < ntiles) __syncthreads();  // the next tile rewrites W1 slot 0 (epilogue vectors) and the row table
    }
}

inline int ff_validate(const VkGemmDesc* g, const VkGemmDesc* o) {
    if (!g || !o || !g->A || !g->Wt || !o->Wt || !o->out) return VK_EINVAL;
    if (g->amode != AMODE_DENSE || g->epi != EPI_GEGLU || g->out_f32 || g->A2) return VK_EINVAL;
    if (o->amode != AMODE_DENSE || o->epi != EPI_LINEAR || o->out_f32 || o->A2 || o->act || o->mx8_out || o->ln_stats) return VK_EINVAL;
    if (g->M <= 0 || g->M != o->M || g->K != FF_C || o->N != FF_C) return VK_EINVAL;
    if (o->K <= 0 || (o->K % (2 * FF_HC)) != 0 || o->K > FF_MAXH || g->N != 2 * o->K) return VK_EINVAL;
    if ((g->lda % 8) != 0 || (((size_t)g->A) & 15) != 0 || (o->ldc % 4) != 0) return VK_EINVAL;
    if (g->bias && (((size_t)g->bias) & 15) != 0) return VK_EINVAL;
    if (g->ln_stats && (!g->ln_colsum || (((size_t)g->ln_colsum) & 15) != 0 || g->ln_parts <= 0 || g->ln_parts > 64 || !(g->ln_eps > 0.f))) return VK_EINVAL;
    if ((o->rowvec || o->rowvec2) && o->rows_per_vec <= 0) return VK_EINVAL;
    if (o->rowvec2 && !o->res2) return VK_EINVAL;
    return VK_OK;
}

}  // namespace

// Row-sum slabs the fused kernel writes to out_proj->rowstat_out ([parts][M][2]): a wave owns whole 320-column rows.
extern "C" int vk_ff_fused_rowstat_parts(void) { return 1; }

extern "C" int vk_ff_fused_bf16(const VkGemmDesc* geglu, const VkGemmDesc* out_proj, void* stream_) {
    const int rc = ff_validate(geglu, out_proj);
    if (rc != VK_OK) return rc;
    const int ntiles = (geglu->M + FF_BM - 1) / FF_BM;
    const int grid = ntiles < 256 ? ntiles : 256;  // persistent: one workgroup per CU (141 KB of LDS) walks the tile list
    switch (geglu->tile_cfg) {  // (0 in the product; the timing experiments of tools/ff_fused_probe.py set 1 / 2 / 4)
        case 1: hipLaunchKernelGGL(ff_fused_kernel<1>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        case 2: hipLaunchKernelGGL(ff_fused_kernel<2>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        case 4: hipLaunchKernelGGL(ff_fused_kernel<4>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        default: hipLaunchKernelGGL(ff_fused_kernel<0>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj);
    }
    VK_CHECK_LAUNCH();
    return VK_OK;
}
