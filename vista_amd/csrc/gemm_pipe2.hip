// Four-wave pipelined 128x320 variant of the tiled GEMM, TWO workgroups per CU (tile_cfg bit 4): DENSE loader x LINEAR / GEGLU
// epilogue, 16-bit out. Same math, fragment layout and epilogues (gemm_common.h) as the eight-wave 256x320 kernel of gemm_pipe.hip -- the same MFMA
// sequence per output element, so results are bitwise the same -- but the CU holds two independent half-height tiles instead of one:
//   * one workgroup = four waves, wave tile 64 x 160 as in gemm_pipe.hip (2 activation x 5 weight fragments, 160 accumulator registers, 7
//     ds_read_b128 per 10 MFMAs), 256-VGPR budget; two workgroups = eight waves per CU = two per SIMD, out of phase with each other: one's
//     epilogue (stores, residual reads, GEGLU arithmetic -- 30-50 % of a short-K tile, profiles/r04_gemm_pipe.txt section 3) runs under the other's
//     K-loop, which one 162 KB workgroup per CU cannot do;
//   * what pays for it is LDS: 80 KB per workgroup. K-steps are 32 deep (64-byte LDS rows, chunk index XOR (row >> 1) & 3), the weight tile
//     (320 rows: L2-resident, short latency) is double-buffered and the activation tile (128 rows: may come from HBM) triple-buffered -- 2 x 20 KB
//     + 3 x 8 KB = 64 KB + the epilogue vectors / LayerNorm row statistics (14 KB);
//   * per K-step and wave: five weight pieces of K-step kt + 1 and two activation pieces of K-step kt + 2 (buffer_load ... lds, 1 KB each),
//     issued between the ten MFMAs that follow the K-step barrier; the wait before the next barrier is a COUNTED one (vmcnt(2): the two
//     youngest pieces -- the activation rows two K-steps ahead -- stay in flight across it);
//   * fragments double-buffered across the barrier exactly as in gemm_pipe.hip: the second k-substep's MFMAs of K-step kt run after the barrier
//     and cover the first fragment reads of K-step kt + 1.
// The price list this design was measured against (profiles/r06_gemm_pipe2.txt): 1.56x the LDS-fill bytes per FLOP of the 256x320 tile
// (2 x (128 + 320) rows against 256 + 320) and 14 instead of 9 LDS-DMA issues per 40 MFMAs and wave.
#include <stdlib.h>

#include "common.h"
#include "vista_hip.h"

#include "gemm_common.h"

namespace {

constexpr int QBM = 128, QBN = 320, QNT = 256, QBK = 32;
constexpr int QROW = QBK * 2;                    // bytes per LDS row
constexpr int QA_ST = QBM * QROW, QW_ST = QBN * QROW;   // 8192 / 20480 bytes per stage
constexpr int QNA = 3, QNW = 2;                  // ring depths
constexpr int QAP = QBM / 64, QWP = QBN / 64;    // pieces (64 rows x 64 B = one 16-byte chunk per thread) per K-step: 2 + 5
constexpr int QFX = 5, QFY = 2;
constexpr unsigned long long Q_LIMIT = 0xfffff000ull;
constexpr unsigned Q_OOB = 0xffffff00u;         // an offset no resource below reaches (host: every operand < Q_LIMIT bytes)

typedef __attribute__((address_space(3))) void* qlptr_t;

#define Q_SB() __builtin_amdgcn_sched_barrier(0)

template <int N> __device__ __forceinline__ void q_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, bool NT_A>
__global__ __launch_bounds__(QNT, 2) void gemm_pipe2_kernel(const VkGemmDesc p) {
    constexpr int OFF_W = QNA * QA_ST, OFF_LN = OFF_W + QNW * QW_ST, OFF_EV = OFF_LN + QBM * 8;
    __shared__ __attribute__((aligned(16))) char smem[OFF_EV + epi_vec_floats(QBN) * 4];

    const int tilesN = p.N / QBN;
    const int tilesM = (p.m_end - p.m_begin + QBM - 1) / QBM;
    const int ntiles = tilesM * tilesN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // tile order as gemm_pipe.hip: column tile fastest unless the weights overflow the L2, then panels of 8 row tiles
    int tm, tn;
    {
        const int logical = xcd_remap(blockIdx.x, ntiles);
        if (tilesN < 8 || (long long)p.N * p.K * 2 <= (3LL << 20)) {
            tn = logical % tilesN;
            tm = logical / tilesN;
        } else {
            constexpr int GM = 16;
            const int panel = logical / (GM * tilesN), r = logical - panel * (GM * tilesN);
            const int gm = (tilesM - panel * GM < GM) ? tilesM - panel * GM : GM;
            tm = panel * GM + r % gm;
            tn = r / gm;
        }
    }
    const int m0 = p.m_begin + tm * QBM, n0 = tn * QBN;

    // ---- staging assignment: thread t moves the 16-byte chunk lc of piece row lr (64 rows x 4 chunks per piece); the swizzle lives in the SOURCE chunk ----
    const int lc = tid & 3, lr = tid >> 2;
    const int lsrc = lc ^ ((lr >> 1) & 3);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff_w = ((unsigned)lr * (unsigned)p.K + (unsigned)lsrc * 8u) * 2u;
    const unsigned voff_a = ((unsigned)lr * (unsigned)p.lda + (unsigned)lsrc * 8u) * 2u;
    const unsigned wpass = 64u * (unsigned)p.K * 2u, apass = 64u * (unsigned)p.lda * 2u;
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const uint16_t*)p.Wt + (size_t)n0 * p.K), 0, (int)((unsigned)QBN * (unsigned)p.K * 2u), 0x00020000);
    const int rows = (p.m_end - m0 < QBM) ? p.m_end - m0 : QBM;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint16_t*)p.A + (size_t)m0 * p.lda), 0,
                                                                        (int)(((unsigned)(rows - 1) * (unsigned)p.lda + (unsigned)p.K) * 2u), 0x00020000);
    // the pieces of K-step `k` (rows past M: out-of-range offsets, which the hardware returns as zeros). `live` = false turns the piece into an
    // out-of-range read as well (zeros into a stage nobody reads any more): the K-loop issues the same seven pieces in every iteration, so it is
    // straight-line code with ONE counted wait -- a wave-uniform branch around a piece splits the MFMA sequence into basic blocks, and the fragment
    // vectors crossing them came back as v_perm_b32 shuffles between the MFMAs
    auto dma_w = [&](const int i, const int k, const bool live) __attribute__((always_inline)) {
        char* const dst = smem + OFF_W + (k % QNW) * QW_ST + i * 64 * QROW + wave_u * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (qlptr_t)dst, 16, live ? voff_w : Q_OOB, (unsigned)i * wpass + (unsigned)k * QROW, 0, 0);
    };
    auto dma_a = [&](const int j, const int k, const bool live) __attribute__((always_inline)) {
        char* const dst = smem + (k % QNA) * QA_ST + j * 64 * QROW + wave_u * 1024;
        const unsigned v = live ? voff_a + (unsigned)j * apass : Q_OOB;
        if (NT_A) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (qlptr_t)dst, 16, v, (unsigned)k * QROW, 0, 2);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (qlptr_t)dst, 16, v, (unsigned)k * QROW, 0, 0);
    };

    f32x16_t acc[QFX][QFY];
    // fragment addresses: weight rows wn*160 + 32 f + l31, activation rows wm*64 + 32 f + l31; k-substep ks = chunk (2 ks + lh) ^ ((row >> 1) & 3)
    const int sw = (l31 >> 1) & 3;
    const int xrow = OFF_W + (wn * 160 + l31) * QROW, yrow = (wm * 64 + l31) * QROW;
    auto load_frags = [&](const int k, const int ks, bf16x8_t* xf, bf16x8_t* yf) __attribute__((always_inline)) {
        const int co = ((ks * 2 + lh) ^ sw) << 4;
        const char* sa = smem + (k % QNA) * QA_ST + yrow + co;
        const char* sx = smem + (k % QNW) * QW_ST + xrow + co;
#pragma unroll
        for (int f = 0; f < QFY; ++f) yf[f] = *(const bf16x8_t*)(sa + f * 32 * QROW);
#pragma unroll
        for (int f = 0; f < QFX; ++f) xf[f] = *(const bf16x8_t*)(sx + f * 32 * QROW);
    };
    auto mma = [&](const bf16x8_t* xf, const bf16x8_t* yf) __attribute__((always_inline)) {
#pragma unroll
        for (int fi = 0; fi < QFX; ++fi)
#pragma unroll
            for (int fj = 0; fj < QFY; ++fj) acc[fi][fj] = vk_mfma(xf[fi], yf[fj], acc[fi][fj]);
    };
    // ten MFMAs with the seven pieces of later K-steps between them: W(kw) after the first five, A(ka) after the next two
    const int nk = p.K / QBK;
    auto mma_dma = [&](const bf16x8_t* xf, const bf16x8_t* yf, const int kw, const int ka) __attribute__((always_inline)) {
        const bool do_w = kw < nk, do_a = ka < nk;
#pragma unroll
        for (int fi = 0; fi < QFX; ++fi)
#pragma unroll
            for (int fj = 0; fj < QFY; ++fj) {
                acc[fi][fj] = vk_mfma(xf[fi], yf[fj], acc[fi][fj]);
                const int q = fi * 2 + fj;
                if (q < QWP + QAP) {
                    Q_SB();
                    if (q < QWP) dma_w(q, kw, do_w);
                    else dma_a(q - QWP, ka, do_a);
                    Q_SB();
                }
            }
    };

    float2* const lnrow = (float2*)(smem + OFF_LN);
    float* const epi_vec = (float*)(smem + OFF_EV);
    EpiPlan eplan{true, 0, 1};
    if (EPI == EPI_LINEAR && (p.rowvec || p.rowvec2)) {
        const int last = (m0 + QBM < p.m_end ? m0 + QBM : p.m_end) - 1;
        eplan.img0 = m0 / p.rows_per_vec;
        eplan.nimg = last / p.rows_per_vec - eplan.img0 + 1;
    }

    // ---- prologue: A(0), W(0), then the tile's epilogue vectors / row statistics (plain loads), then A(1), W(1), A(2) ----
#pragma unroll
    for (int j = 0; j < QAP; ++j) dma_a(j, 0, true);
#pragma unroll
    for (int i = 0; i < QWP; ++i) dma_w(i, 0, true);
    {
        int tv = tid;
        asm volatile("" : "+v"(tv));   // (nothing lane-derived of this pass is to live across the K-loop)
        epi_stage_vectors<QBN, QNT>(p, epi_vec, n0, eplan, tv);
        if (p.ln_stats != nullptr) {
            for (int r = tv; r < QBM; r += QNT) {
                const int m = m0 + r;
                lnrow[r] = ln_row_stats(p, m < p.m_end ? m : p.m_end - 1);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < QAP; ++j) dma_a(j, 1, nk > 1);
#pragma unroll
    for (int i = 0; i < QWP; ++i) dma_w(i, 1, nk > 1);
#pragma unroll
    for (int j = 0; j < QAP; ++j) dma_a(j, 2, nk > 2);
    q_wait_vmcnt<QAP + QWP + QAP>();   // A(0), W(0) have landed when at most the nine pieces issued after them are outstanding
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the vector / row-statistics ds_writes)
    __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int i = 0; i < QFX; ++i)
#pragma unroll
        for (int j = 0; j < QFY; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifdef PIPE_TIMING   // s_memtime stamps of one wave per SIMD of workgroup 8 (tools/build_variant.sh -DPIPE_TIMING; tools/gemm_pipe2_probe.py prints them)
    unsigned long long tq_k0, tq_wait = 0, tq_t0, tq_t1, tq_t2;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq_t0) :: "memory");
    tq_k0 = tq_t0;
#endif
    bf16x8_t xa[QFX], ya[QFY], xb[QFX], yb[QFY];
    load_frags(0, 0, xa, ya);
    load_frags(0, 1, xb, yb);
    // Invariant at the top of iteration kt: stages of K-step kt hold its data for every wave's pieces (barrier passed), xa / xb = its two
    // k-substeps; in flight: W(kt + 1), A(kt + 1) (issued one iteration ago -- or in the prologue) and, youngest, A(kt + 2).
    for (int kt = 0; kt + 1 < nk; ++kt) {
        Q_SB();
        mma(xa, ya);
        Q_SB();
        // K-step kt + 1 landed for this wave: everything but the two youngest pieces (A(kt + 2), live or not)
#ifdef PIPE_TIMING
        unsigned long long tq_a, tq_b;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq_a) :: "memory");
#endif
        q_wait_vmcnt<QAP>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of K-step kt are done
        __builtin_amdgcn_s_barrier();   // ... every wave's: its stages are free, K-step kt + 1 is complete (a raw barrier: __syncthreads() would add vmcnt(0))
#ifdef PIPE_TIMING
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq_b) :: "memory");
        tq_wait += tq_b - tq_a;
#endif
        load_frags(kt + 1, 0, xa, ya);
        Q_SB();
        mma_dma(xb, yb, kt + 2, kt + 3);   // W(kt + 2) over W(kt), A(kt + 3) over A(kt)
        Q_SB();
        load_frags(kt + 1, 1, xb, yb);
    }
    Q_SB();
    mma(xa, ya);
    Q_SB();
    mma(xb, yb);
    Q_SB();

#ifdef PIPE_TIMING
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq_t1) :: "memory");
#endif
    int te = tid;
    asm volatile("" : "+v"(te));
    const int e_wave = te >> 6, e_l31 = te & 31, e_lh = (te >> 5) & 1;
    const int e_wm = e_wave >> 1, e_wn = e_wave & 1;
    const float2* const lnp = p.ln_stats != nullptr ? lnrow : nullptr;
    if constexpr (EPI == EPI_GEGLU) gemm_epilogue_geglu_lds<QFX, QFY, 2, 5, QBN>(p, acc, m0, n0, e_wm, e_wn, e_l31, e_lh, lnp, epi_vec);
    else gemm_epilogue_linear_lds<QFX, QFY, 2, 5, QBN, 256>(p, acc, m0, n0, e_wm, e_wn, e_l31, e_lh, tn * 2 + e_wn, lnp, epi_vec, eplan.img0);
#ifdef PIPE_TIMING
    if (blockIdx.x == 8 && lane == 0 && p.splitk_ws) {   // per wave: [vmcnt + barrier wait, K-steps, K-loop, epilogue, kernel] in s_memtime ticks
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tq_t2) :: "memory");
        float* o = (float*)p.splitk_ws + wave * 8;
        o[0] = (float)tq_wait; o[1] = (float)nk; o[2] = (float)(tq_t1 - tq_t0); o[3] = (float)(tq_t2 - tq_t1); o[4] = (float)(tq_t2 - tq_k0); o[5] = 0.f;
    }
#endif
}

template <int EPI>
int pipe2_launch(const VkGemmDesc* d, hipStream_t stream) {
    const int tilesN = d->N / QBN, tilesM = (d->m_end - d->m_begin + QBM - 1) / QBM;
    const int ntiles = tilesM * tilesN;
    const bool nt_a = tilesN <= 4;   // as gemm_pipe.hip: activation rows that few column tiles re-read are streamed non-temporally
    if (nt_a) hipLaunchKernelGGL((gemm_pipe2_kernel<EPI, true>), dim3(ntiles), dim3(QNT), 0, stream, *d);
    else hipLaunchKernelGGL((gemm_pipe2_kernel<EPI, false>), dim3(ntiles), dim3(QNT), 0, stream, *d);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

}  // namespace

// 1 = the two-per-CU pipelined variant takes this (already validated, row range normalised) problem
extern "C" int vk_gemm_pipe2_fit(const VkGemmDesc* d) {
    if (d->amode != AMODE_DENSE || (d->epi != EPI_LINEAR && d->epi != EPI_GEGLU) || d->out_f32 || (d->N % QBN) != 0 || d->mx8_out || d->A2 || d->gnstat_out) return 0;
    if ((d->K % QBK) != 0 || d->K < 2 * QBK) return 0;
    if (!epi_fast_everywhere(*d, d->epi, QBM)) return 0;   // the kernel carries the LDS-staged epilogues only
    if ((unsigned long long)QBN * d->K * 2ull >= Q_LIMIT || (unsigned long long)QBM * d->lda * 2ull >= Q_LIMIT) return 0;
    return 1;
}

extern "C" int vk_gemm_pipe2_launch(const VkGemmDesc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!vk_gemm_pipe2_fit(d)) return VK_EINVAL;
    return d->epi == EPI_GEGLU ? pipe2_launch<EPI_GEGLU>(d, stream) : pipe2_launch<EPI_LINEAR>(d, stream);
}
