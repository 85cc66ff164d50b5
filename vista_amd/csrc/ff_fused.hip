// Fused FeedForward of the level-0 transformers (gfx950): GEGLU in-projection -> value * gelu(gate) -> out-projection in ONE kernel.
//   reference: vwm/modules/attention.py:85-128 (GEGLU, FeedForward.net), :524 (x = ff(norm3(x)) + x),
//              vwm/modules/video_attention.py:119-121,137-141 (ff_in / ff of the temporal block, AlphaBlender mix)
//   out[m][:] = epilogue2( sum_h  H[m][h] * W2[:][h] ),   H[m][h] = bf16( a * gelu(g) ),
//   (a, g) = LNfold( sum_k x[m][k] * W1[packed rows of h][k] ) + bias1            (the folded LayerNorm of VkGemmDesc.ln_*)
// The unfused pair writes H (460800 x 1280 bf16 = 1.18 GB per level-0 FeedForward) to HBM and reads it back: 15 FeedForwards x 2.36 GB
// per denoise step, and its GEGLU kernel runs the K-loop and the 32-gelu epilogue of a tile one after the other (matrix core 35 % busy,
// VALU 45 %, never together: profiles/r03_pmc_gemm_sq.txt). Here H never leaves the CU:
//   * a workgroup owns 128 tokens and runs EIGHT waves in two roles, one wave of each role on every SIMD:
//       in-projection waves 0-3 keep the x fragments of their 32 tokens (all K = 320: 80 VGPRs) in registers; per 32-wide hidden chunk they
//         run 40 MFMAs (two accumulator chains) woven 1 : ~7 with the GEGLU arithmetic of the PREVIOUS chunk (folded LayerNorm, bias, gelu:
//         16 gates per lane) and hand the bf16 hidden values to their partner through 8 KB of LDS;
//       out-projection waves 4-7 keep the 32-token x 320-column OUTPUT accumulator (160 registers) for the whole tile; per chunk they run
//         20 bare MFMAs and issue ALL of the workgroup's LDS-DMA (60 KB of W1 / W2 per chunk = 15 pieces per wave) between them -- a DMA
//         issue blocks its wave for 60-190 cycles, which a wave that also has to feed 40 MFMAs and 260 VALU instructions cannot afford
//         (measured on the one-wave-per-SIMD predecessor of this kernel: 2100 of 6400 cycles per chunk, profiles/r04_ff_fused_notes.txt);
//     so matrix work of one role runs under the VALU / DMA-issue work of the other on every SIMD, and neither role exceeds 256 registers;
//   * W1 / W2 chunks stream L2 -> LDS by LDS-DMA in a 2-deep ring (40 + 20 KB per chunk), one barrier per chunk;
//   * the hidden values are the out-projection's B operand in the order the lanes produced them: W2 is packed with its K axis
//     permuted inside every 16-group ([0-3, 8-11, 4-7, 12-15]) and chunk-major (ops.pack_ff_out), so no shuffle is needed;
//   * the out-projection's epilogue is the LINEAR family's LDS-staged epilogue (gemm_common.h): bias, residuals, AlphaBlender blend,
//     row vectors, LayerNorm row-sum emission -- same arithmetic, same operation order.
// Numerics: H is rounded to bf16 exactly where the unfused pair rounds it; the in-projection accumulates in the same order; the
// out-projection sums the same products with the 16-wide MFMA reduction walking a permuted slot order (fp32, not bitwise).
#include "common.h"
#include "vista_hip.h"

#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int FF_C = 320;              // level-0 width: K of the in-projection, N of the out-projection
constexpr int FF_BM = 128;             // tokens per workgroup tile: four token groups of 32
constexpr int FF_HC = 32;              // hidden units per chunk (= 64 packed GEGLU rows = two 32-row MFMA fragments)
constexpr int FF_MAXH = 1280;          // hidden width the per-column vector region is sized for
constexpr int FF_NT = 512;             // waves 0-3: in-projection + GEGLU; waves 4-7: out-projection + LDS-DMA
constexpr int W1_SLOT = 2 * FF_HC * FF_C * 2;   // 64 packed rows x 320 k x 2 B = 40960: five [64 rows][64 k] slabs of 8 KB
constexpr int W2_SLOT = FF_C * FF_HC * 2;       // 320 rows x 32 hidden x 2 B = 20480
constexpr int H_SLOT = FF_BM * FF_HC * 2;       // 128 tokens x 32 hidden x 2 B = 8192
constexpr int OFF_W1 = 0, OFF_W2 = 2 * W1_SLOT, OFF_H = OFF_W2 + 2 * W2_SLOT, OFF_VEC = OFF_H + 2 * H_SLOT;
constexpr int OFF_LN = OFF_VEC + 2 * (2 * FF_MAXH) * 4, FF_LDS = OFF_LN + FF_BM * 8;
static_assert(FF_LDS <= 163840, "LDS budget");
static_assert(epi_vec_floats(FF_C) * 4 <= 2 * W2_SLOT && 4 * 20480 <= 2 * W1_SLOT, "epilogue vectors overlay the W2 ring, the accumulator hand-over the W1 ring");

// DBG (timing experiments only, results wrong): 1 = no LDS-DMA inside the chunk steps, 2 = gelu replaced by a plain product, 4 = no out-projection MFMAs
template <int DBG>
__global__ __launch_bounds__(FF_NT, 2) void ff_fused_kernel(const VkGemmDesc p1, const VkGemmDesc p2) {
    __shared__ __attribute__((aligned(16))) char smem[FF_LDS];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_in = wave < 4;      // in-projection wave (else out-projection wave); waves w and w + 4 share a SIMD
    const int tgw = wave & 3;           // token group: tokens 32 tgw .. + 31 of the tile
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
    // W1 chunk = five slabs (k = 64 i ..) of [64 rows][128 B]; a 1 KB piece = 8 rows of a slab; DMA wave d (= tgw of an out-projection wave)
    // stages row blocks d and d + 4 of every slab
    const int w1r = 8 * tgw + (lane >> 3);                                     // packed row inside the chunk (first row block); (w1r + 32) >> 1 has the same low bits
    const int w1off = w1r * FF_C + 8 * ((lane & 7) ^ ((w1r >> 1) & 7));
    // W2 chunk (chunk-major [chunk][320 rows][32]) = [320 rows][64 B]; a piece = 16 rows; DMA wave d stages pieces d, d + 4, .., d + 16
    const int w2off = (16 * tgw + (lane >> 2)) * FF_HC + 8 * ((lane & 3) ^ ((lane >> 4) & 3));   // ((row >> 2) & 3) == (lane >> 4) & 3

    // ---- per-lane fragment read offsets ----
    const int sw = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) frag_off[k4] = ((k4 * 2 + lh) ^ sw) << 4;
    const int w1row_off = l31 * 128;                     // fragment c of the chunk: + 32 c rows = 4096 c bytes
    const int sw2 = (l31 >> 2) & 3;
    int w2f_off[2], h_off[2];                            // k-substep s of a chunk: W2 fragment rows / this lane's token row of the H slot
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        w2f_off[s] = l31 * 64 + (((2 * s + lh) ^ sw2) << 4);   // + 32 f rows = 2048 f bytes; ((n >> 2) & 3) == sw2 for every f
        h_off[s] = (32 * tgw + l31) * 64 + (((2 * s + lh) ^ sw2) << 4);   // written by the in-projection lane (l31, lh) of fragment s, read by the same lane index
    }

    // ---- once per workgroup: per-column vectors of the in-projection ----
    for (int i = tid; i < (2 * Hd) / 4; i += FF_NT) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f), cs = b;
        if (p1.bias) b = *(const float4*)(p1.bias + 4 * i);
        if (p1.ln_stats) cs = *(const float4*)(p1.ln_colsum + 4 * i);
        *(float4*)(vecb + 4 * i) = b;
        *(float4*)(vecc + 4 * i) = cs;
    }

    // one of a DMA wave's 15 pieces of a step: q = 0..9: W1 chunk jw1 -> W1 slot `sl` (slab q >> 1, row block tgw + 4 (q & 1));
    // q = 10..14: W2 chunk jw2 -> W2 slot `sl` (piece tgw + 4 (q - 10)). Address = uniform base (SGPR pair: tensor + chunk + piece) + ONE
    // 32-bit per-lane byte offset per tensor: fifteen 64-bit per-lane pointers would not fit beside the 160 accumulator registers.
    const uint32_t w1voff_ = (uint32_t)w1off * 2u, w2voff_ = (uint32_t)w2off * 2u;
    auto dma_piece = [&](const int q, const int sl, const int jw1, const int jw2) {
        // (the offsets pass through an empty asm at every use: otherwise the compiler hoists all fifteen 64-bit per-lane addresses out of the
        //  chunk loop -- 30 registers the 160-register accumulator leaves no room for; they were spilled and reloaded around every tile)
        uint32_t w1voff = w1voff_, w2voff = w2voff_;
        asm volatile("" : "+v"(w1voff), "+v"(w2voff));
        if (q < 10) {
            const char* ub = (const char*)W1g + (size_t)jw1 * (2 * FF_HC * FF_C * 2) + (64 * (q >> 1) + (q & 1) * 32 * FF_C) * 2;
            __builtin_amdgcn_global_load_lds((gptr_t)(ub + w1voff), (lptr_t)(smem + OFF_W1 + sl * W1_SLOT + tgw * 1024 + (q >> 1) * 8192 + (q & 1) * 4096), 16, 0, 0);
        } else {
            const char* ub = (const char*)W2g + (size_t)jw2 * (FF_C * FF_HC * 2) + (q - 10) * 64 * FF_HC * 2;
            __builtin_amdgcn_global_load_lds((gptr_t)(ub + w2voff), (lptr_t)(smem + OFF_W2 + sl * W2_SLOT + tgw * 1024 + (q - 10) * 4096), 16, 0, 0);
        }
    };

    // DBG & 8: per-phase cycle counts (s_memtime) summed over the tiles of this workgroup, written by lane 0 of every wave of workgroup 0 to
    // p1.splitk_ws as 8 x int64 per wave: in-projection waves {A, B, C, barrier wait, steps}, out-projection waves {MFMA + DMA, barrier wait, steps, epilogue, MFMA + DMA again}
    long long tacc[5] = {0, 0, 0, 0, 0};
    auto now = [&]() -> long long { return (DBG & 8) ? (long long)__builtin_amdgcn_s_memtime() : 0ll; };
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * FF_BM;
        // ---------------- prologue: W1(0) -> slot 0 (DMA waves), x fragments (in-projection waves), row statistics (everyone) ----------------
        if (!role_in) {
#pragma unroll
            for (int q = 0; q < 10; ++q) dma_piece(q, 0, 0, 0);
        }
        if (p1.ln_stats != nullptr) {
            for (int r = tid; r < FF_BM; r += FF_NT) {
                const int m = m0 + r;
                lnrow[r] = ln_row_stats(p1, m < p1.M ? m : p1.M - 1);
            }
        }
        __syncthreads();

        // the out-projection epilogue's per-column / per-image vectors overlay the W2 ring (dead by then); block-uniform plan
        float* const epi_vec = (float*)(smem + OFF_W2);
        const EpiPlan eplan = epi_plan<EPI_LINEAR, false, FF_BM, FF_C>(p2, m0, 0);
        f32x16_t E[5][1];   // the wave's share of the output tile in the epilogue: 32 tokens x 160 columns
        if (role_in) {
            // =============================== in-projection waves ===============================
            float rs = 1.f, nrm = 0.f;
            if (p1.ln_stats != nullptr) { const float2 t = lnrow[32 * tgw + l31]; rs = t.y; nrm = -t.x * t.y; }
            // x fragments of this wave's 32 tokens: B operand of the in-projection, lane (l31, lh) holds k = 16 ks + 8 lh .. + 7 of token l31.
            // Loaded HERE, not next to the prologue's DMA: issued before the prologue barrier they were live across its bookkeeping, and the
            // allocator (one register budget for both roles) spilled 12 of the 20 fragments to scratch and reloaded them -- 300 bytes per lane
            // and tile = 560 MB of scratch writes per launch (rocprofv3 WRITE_SIZE 657 MB for 295 MB of output).
            bf16x8_t xf[20];
            {
                int m = m0 + 32 * tgw + l31;
                if (m >= p1.M) m = p1.M - 1;
                const uint16_t* xr = Xg + (size_t)m * p1.lda + 8 * lh;
#pragma unroll
                for (int ks = 0; ks < 20; ++ks) xf[ks] = *(const bf16x8_t*)(xr + 16 * ks);
            }
            // Step j (W1(j) in W1 slot j & 1): (1) the 40 in-projection MFMAs, bare, the two fragments' accumulator chains alternating (a VALU
            // instruction between two MFMAs of ONE chain costs the chain its back-to-back forwarding: +43 cycles per MFMA, measured as 1750
            // cycles for 20 woven MFMAs against 920 bare); (2) the GEGLU of all 16 gates of the lane in one unfenced block (eight independent
            // dependency chains for the scheduler to interleave; fenced 2-gate units ran at 195 cycles each). Meanwhile the SIMD's
            // out-projection wave issues the step's DMA during (1) and runs its 20 MFMAs during (2). Hidden values -> H slot j & 1.
            for (int j = 0; j < nit; ++j) {
                const int sl = j & 1;
                const char* w1s = smem + OFF_W1 + sl * W1_SLOT + w1row_off;
                char* hs = smem + OFF_H + sl * H_SLOT;
                bf16x8_t fr[2][5];
                f32x16_t S[2];
                float4 vb[2][2][2], vc[2][2][2];   // [fragment][quad][value | gate]: bias / LayerNorm column sums of this lane's 16 gates
                auto rd1 = [&](const int grp, bf16x8_t (&f)[5]) {   // in-projection MFMA i = 5 grp + n: k-substep i >> 1, fragment i & 1
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        const int i = 5 * grp + n, ks = i >> 1, cc = i & 1;
                        f[n] = *(const bf16x8_t*)(w1s + (ks >> 2) * 8192 + cc * 4096 + frag_off[ks & 3]);
                    }
                };
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[cc][r] = 0.f;
                const long long tA = now();
                rd1(0, fr[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g < 7) rd1(g + 1, fr[(g + 1) & 1]);
                    else {   // the gelu phase's per-column vectors, under the last five MFMAs (read at its start they cost four exposed LDS round trips)
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int np = 2 * FF_HC * j + 32 * cc + 4 * lh + 8 * q;
                                vb[cc][q][0] = *(const float4*)(vecb + np); vb[cc][q][1] = *(const float4*)(vecb + np + 16);
                                vc[cc][q][0] = *(const float4*)(vecc + np); vc[cc][q][1] = *(const float4*)(vecc + np + 16);
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        const int i = 5 * g + n;
                        S[i & 1] = vk_mfma(fr[g & 1][n], xf[i >> 1], S[i & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const long long tB = now();
                // GEGLU: per fragment the value quads g = 0, 1 and their gate quads g + 2 (packed rows 8 g + 4 lh + e, gates 16 rows further)
                // -> 8 bf16 per lane and fragment, in the k-slot order of the out-projection's B operand
                // Written stage by stage over all 16 gates (sixteen independent instructions per stage, fenced): left to itself the scheduler
                // emits each gate's chain fma -> exp -> add -> rcp -> mul back to back, and with one or two waves per SIMD every dependent
                // transcendental then stalls for its full latency (measured: 130 cycles per gate).
                uint32_t hq[2][4];
                float av[16], gv[16], uv[16];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const float4 ba = vb[cc][g][0], bg = vb[cc][g][1], ca = vc[cc][g][0], cg = vc[cc][g][1];
                        const int o = 8 * cc + 4 * g;
                        av[o + 0] = fmaf(rs, S[cc][4 * g + 0], fmaf(nrm, ca.x, ba.x)); av[o + 1] = fmaf(rs, S[cc][4 * g + 1], fmaf(nrm, ca.y, ba.y));
                        av[o + 2] = fmaf(rs, S[cc][4 * g + 2], fmaf(nrm, ca.z, ba.z)); av[o + 3] = fmaf(rs, S[cc][4 * g + 3], fmaf(nrm, ca.w, ba.w));
                        gv[o + 0] = fmaf(rs, S[cc][4 * g + 8], fmaf(nrm, cg.x, bg.x)); gv[o + 1] = fmaf(rs, S[cc][4 * g + 9], fmaf(nrm, cg.y, bg.y));
                        gv[o + 2] = fmaf(rs, S[cc][4 * g + 10], fmaf(nrm, cg.z, bg.z)); gv[o + 3] = fmaf(rs, S[cc][4 * g + 11], fmaf(nrm, cg.w, bg.w));
                    }
                }
                // (an empty asm with a "+v" operand pins a value's position in the instruction stream: IR-level passes move pure arithmetic
                //  across __builtin_amdgcn_sched_barrier, which only binds the machine scheduler)
#define FF_PIN16(arr) _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(arr[i]))
                FF_PIN16(gv);
#pragma unroll
                for (int i = 0; i < 16; ++i) uv[i] = fminf(gv[i] * gv[i], 50.f);
#pragma unroll
                for (int i = 0; i < 16; ++i) uv[i] = gv[i] * fmaf(uv[i], fmaf(uv[i], 0.0010142630f, -0.10677572f), -2.3011212f);   // gelu_erf_f, common.h
                FF_PIN16(uv);
#pragma unroll
                for (int i = 0; i < 16; ++i) uv[i] = (DBG & 2) ? uv[i] : __builtin_amdgcn_exp2f(uv[i]);
                FF_PIN16(uv);
#pragma unroll
                for (int i = 0; i < 16; ++i) uv[i] = 1.f + uv[i];
                FF_PIN16(uv);
#pragma unroll
                for (int i = 0; i < 16; ++i) uv[i] = gv[i] * ((DBG & 2) ? uv[i] : __builtin_amdgcn_rcpf(uv[i]));   // = gelu_erf_f(gv[i]), operation for operation
                FF_PIN16(uv);
#undef FF_PIN16
#pragma unroll
                for (int i = 0; i < 8; ++i) hq[i >> 2][i & 3] = pack_bf16(av[2 * i] * uv[2 * i], av[2 * i + 1] * uv[2 * i + 1]);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) *(uint4*)(hs + h_off[cc]) = make_uint4(hq[cc][0], hq[cc][1], hq[cc][2], hq[cc][3]);
                const long long tD = now();
                __syncthreads();
                if (DBG & 8) { const long long tE = now(); tacc[0] += tB - tA; tacc[2] += tD - tB; tacc[3] += tE - tD; tacc[4] += 1; }
            }
            __syncthreads();                         // (the out-projection waves' drain step)
            // epilogue hand-over: the partner out-projection wave parks columns 160..319 of these 32 tokens in the (dead) W1 ring, lane for lane
            if (eplan.fast) epi_stage_vectors<FF_C, FF_NT / 2>(p2, epi_vec, 0, eplan, tid);
            __syncthreads();
            const char* xs = smem + OFF_W1 + tgw * 20480 + lane * 16;
#pragma unroll
            for (int f = 0; f < 5; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *(const float4*)(xs + (f * 4 + q) * 1024);
                    E[f][0][4 * q] = v.x; E[f][0][4 * q + 1] = v.y; E[f][0][4 * q + 2] = v.z; E[f][0][4 * q + 3] = v.w;
                }
        } else {
            // =============================== out-projection waves (also the workgroup's DMA engine) ===============================
            f32x16_t O[10][1];   // out-projection accumulator: 32 tokens x 320 columns
#pragma unroll
            for (int f = 0; f < 10; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[f][0][r] = 0.f;
            // Step 0: nothing to multiply yet; DMA W1(1) -> W1 slot 1, W2(0) -> W2 slot 0.
            // Step j = 1..nit: out-projection of chunk j - 1 (hidden values in H slot (j - 1) & 1, W2(j - 1) in W2 slot (j - 1) & 1): 20 MFMAs, with
            // this step's DMA -- W1(j + 1) -> W1 slot (j + 1) & 1, W2(j) -> W2 slot j & 1, both last read in step j - 1 -- one piece after each of
            // the first 15 MFMAs. Past the end the chunk index is clamped: the last chunk is re-staged into a dead slot (branch-free steps).
            if (!(DBG & 1)) {
#pragma unroll
                for (int q = 0; q < 15; ++q) dma_piece(q, q < 10 ? 1 : 0, 1, 0);
            }
            __syncthreads();
            for (int j = 1; j <= nit; ++j) {
                const int sr = (j - 1) & 1;
                const char* w2s = smem + OFF_W2 + sr * W2_SLOT;
                const char* hs = smem + OFF_H + sr * H_SLOT;
                const int jw1 = j + 1 < nit ? j + 1 : nit - 1, jw2 = j < nit ? j : nit - 1;
                bf16x8_t fr[2][5], hf[2];
                const long long tA = now();
                auto rd2 = [&](const int grp, bf16x8_t (&f)[5]) {   // out-projection MFMA i = 5 grp + n: k-substep i / 10, column fragment i % 10
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        const int i = 5 * grp + n;
                        f[n] = *(const bf16x8_t*)(w2s + w2f_off[i / 10] + (i % 10) * 2048);
                    }
                };
                const long long tM = now();
                hf[0] = *(const bf16x8_t*)(hs + h_off[0]);
                hf[1] = *(const bf16x8_t*)(hs + h_off[1]);
                rd2(0, fr[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v2 = 0; v2 < 4; ++v2) {
                    if (v2 < 3) rd2(v2 + 1, fr[(v2 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        const int i = 5 * v2 + n;
                        if (!(DBG & 4)) O[i % 10][0] = vk_mfma(fr[v2 & 1][n], hf[i / 10], O[i % 10][0]);
                        else asm volatile("" :: "v"(fr[v2 & 1][n]), "v"(hf[i / 10]));
                        // One DMA piece after each of the first 15 MFMAs. Measured alternatives: all 15 pieces first, then the MFMAs (1.43 ms per
                        // launch against 1.42: the late MFMAs stretch the partner's gelu phase -- MFMA and VALU issue of the two waves of a SIMD
                        // add up rather than overlap); MFMAs first, then the pieces (1.59: the DMA lands ~1000 cycles after the step's work is
                        // done); half of the pieces issued by the in-projection waves (1.54: a piece costs its issuing wave ~120 cycles
                        // wherever it sits, and those waves are the longer role).
                        if (!(DBG & 1) && i < 15) {
                            __builtin_amdgcn_sched_barrier(0);
                            dma_piece(i, i < 10 ? sr : (sr ^ 1), jw1, jw2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (DBG & 8) tacc[4] += now() - tM;
                const long long tB = now();
                __syncthreads();
                if (DBG & 8) { const long long tC = now(); tacc[0] += tB - tA; tacc[1] += tC - tB; tacc[2] += 1; }
            }
            const long long tE0 = now();
            // ---- epilogue hand-over: all eight waves run the out-projection's epilogue, 32 tokens x 160 columns each. Four waves alone took
            // 32 k cycles per tile for it (19 % of the tile: 160 accumulator registers leave no room for the residual ring, and half the CU's
            // load / store issue capacity idles). Columns 160..319 go to the partner in-projection wave through the W1 ring (dead since the
            // barrier before the drain step; 4 x 20 KB, lane-linear 16-byte pieces: conflict-free both ways).
            {
                char* xs = smem + OFF_W1 + tgw * 20480 + lane * 16;
#pragma unroll
                for (int f = 0; f < 5; ++f)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *(float4*)(xs + (f * 4 + q) * 1024) = make_float4(O[5 + f][0][4 * q], O[5 + f][0][4 * q + 1], O[5 + f][0][4 * q + 2], O[5 + f][0][4 * q + 3]);
            }
            __syncthreads();
#pragma unroll
            for (int f = 0; f < 5; ++f) E[f][0] = O[f][0];
            if (DBG & 8) tacc[3] += now() - tE0;
        }
        // ---- the LINEAR family's epilogue (gemm_common.h) on every wave: wave column 0 = the out-projection waves (columns 0..159), 1 = the
        // in-projection waves; its per-column / per-image vectors were staged into the (dead) W2 ring before the hand-over barrier; row-sum
        // slab = wave column.
        {
            const long long tE1 = now();
            const int wn = role_in ? 1 : 0;
            if (eplan.fast) gemm_epilogue_linear_lds<5, 1, 1, 5, FF_C, 256>(p2, E, m0, 0, tgw, wn, l31, lh, wn, nullptr, epi_vec, eplan.img0);
            else gemm_epilogue<EPI_LINEAR, false, 5, 1, 1, 5>(p2, E, m0, 0, tgw, wn, l31, lh, wn, nullptr);
            if ((DBG & 8) && !role_in) tacc[3] += now() - tE1;
        }
        if (tile + (int)gridDim.x < ntiles) __syncthreads();  // the next tile rewrites the rings (epilogue vectors, hand-over) and the row table
    }
    if ((DBG & 8) && blockIdx.x == 0 && lane == 0 && p1.splitk_ws) {
        long long* o = (long long*)p1.splitk_ws + wave * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = tacc[i];
    }
}

inline int ff_validate(const VkGemmDesc* g, const VkGemmDesc* o) {
    if (!g || !o || !g->A || !g->Wt || !o->Wt || !o->out) return VK_EINVAL;
    if (g->amode != AMODE_DENSE || g->epi != EPI_GEGLU || g->out_f32 || g->A2) return VK_EINVAL;
    if (o->amode != AMODE_DENSE || o->epi != EPI_LINEAR || o->out_f32 || o->A2 || o->act || o->mx8_out || o->ln_stats) return VK_EINVAL;
    if (g->M <= 0 || g->M != o->M || g->K != FF_C || o->N != FF_C) return VK_EINVAL;
    if (o->K <= 0 || (o->K % (2 * FF_HC)) != 0 || o->K < 4 * FF_HC || o->K > FF_MAXH || g->N != 2 * o->K) return VK_EINVAL;
    if ((g->lda % 8) != 0 || (((size_t)g->A) & 15) != 0 || (o->ldc % 4) != 0) return VK_EINVAL;
    if (g->bias && (((size_t)g->bias) & 15) != 0) return VK_EINVAL;
    if (g->ln_stats && (!g->ln_colsum || (((size_t)g->ln_colsum) & 15) != 0 || g->ln_parts <= 0 || g->ln_parts > 64 || !(g->ln_eps > 0.f))) return VK_EINVAL;
    if ((o->rowvec || o->rowvec2) && o->rows_per_vec <= 0) return VK_EINVAL;
    if (o->rowvec2 && !o->res2) return VK_EINVAL;
    return VK_OK;
}

}  // namespace

// Row-sum slabs the fused kernel writes to out_proj->rowstat_out ([parts][M][2]): one per wave column of the epilogue (2 x 160 columns).
extern "C" int vk_ff_fused_rowstat_parts(void) { return 2; }

extern "C" int vk_ff_fused_bf16(const VkGemmDesc* geglu, const VkGemmDesc* out_proj, void* stream_) {
    const int rc = ff_validate(geglu, out_proj);
    if (rc != VK_OK) return rc;
    const int ntiles = (geglu->M + FF_BM - 1) / FF_BM;
    // persistent: one workgroup per CU (141 KB of LDS) walks the tile list. VISTA_FF_WALK=0: one workgroup per tile (A/B hook, as VISTA_GEGLU_WALK)
    static const bool walk = [] { const char* e = getenv("VISTA_FF_WALK"); return !e || atoi(e) != 0; }();
    const int grid = (ntiles < 256 || !walk) ? ntiles : 256;
    VkGemmDesc g1 = *geglu, o1 = *out_proj;       // the shared epilogues bound their rows by m_end (ABI v5 row ranges: not offered here, all rows)
    g1.m_begin = o1.m_begin = 0;
    g1.m_end = o1.m_end = geglu->M;
    geglu = &g1;
    out_proj = &o1;
#ifdef FF_TIMING   // tuning builds only (tools/ff_fused_probe.py): geglu->tile_cfg selects a timing variant; 1 / 2 / 4 give WRONG results by design
    switch (geglu->tile_cfg) {
        case 1: hipLaunchKernelGGL(ff_fused_kernel<1>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        case 2: hipLaunchKernelGGL(ff_fused_kernel<2>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        case 8: hipLaunchKernelGGL(ff_fused_kernel<8>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        case 4: hipLaunchKernelGGL(ff_fused_kernel<4>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj); break;
        default: hipLaunchKernelGGL(ff_fused_kernel<0>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj);
    }
#else   // the product library carries ONE instantiation; tile_cfg (vk_gemm_bf16's tile-variant selector) has no meaning here and is ignored
    hipLaunchKernelGGL(ff_fused_kernel<0>, dim3(grid), dim3(FF_NT), 0, (hipStream_t)stream_, *geglu, *out_proj);
#endif
    VK_CHECK_LAUNCH();
    return VK_OK;
}
