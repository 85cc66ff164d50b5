// Eight-wave pipelined 256x320 variant of the tiled bf16 GEMM (tile_cfg 7): DENSE / CONV3X3 (stride 1 | 2, asymmetric pad, fused nearest x2
// upsample) / TEMPORAL3 (no halo frames) loaders x LINEAR epilogue, DENSE x GEGLU, and the split-K form of the LINEAR ones; bf16 out.
// Same math, LDS image, fragment layout and epilogues (gemm_common.h) as gemm.hip's sixteen-wave 256x320 kernel -- results are bitwise the
// same -- but the K-step is written for the matrix pipe instead of for occupancy (measurements: profiles/r04_gemm_pipe.txt):
//   * eight waves, wave tile 64 x 160 (2 activation x 5 weight fragments, 160 accumulator registers, 256-VGPR budget, two waves per SIMD):
//     7 ds_read_b128 per 10 MFMAs instead of 6 per 5;
//   * fragments double-buffered ACROSS the K-step barrier: the last k-substep's ten MFMAs are issued after the barrier and cover the first
//     fragment reads of the next stage, so no LDS latency is exposed anywhere in the loop (the sixteen-wave kernel waits on lgkmcnt(0)
//     before almost every MFMA pair and relies on its four waves per SIMD to fill the holes);
//   * the nine LDS-DMA pieces of the next K-step (activation pieces first: they are the ones that may come from HBM) are issued BETWEEN the
//     MFMAs of the first k-substep, one after each MFMA. An LDS-DMA issue stalls its wave for ~120 cycles (profiles/r04_ff_fused_notes.txt);
//     interleaved, the partner wave's MFMAs run in the gaps (worth 1-2 % over issuing all nine ahead of the MFMAs);
//   * the pieces are BUFFER loads (buffer_load_dwordx4 ... lds): a wave-uniform resource per operand in SGPRs, ONE 32-bit per-lane offset for
//     the weights and one per activation row group, everything that changes per piece / K-step / tap in scalar registers. Out-of-image conv
//     taps, frames outside the window and rows past M are out-of-range offsets, which the hardware returns as zeros: no clamping, no zero
//     word, no 64-bit per-lane pointers (the sixteen-wave kernel keeps nine of them live across the loop);
//   * LINEAR launches run one tile per workgroup (a workgroup that ends does not wait for its stores); GEGLU launches walk the tile list with
//     256 resident workgroups and stage the next tile's first K-step before the epilogue.
// Internal entry points, called by gemm.hip's launcher.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "vista_hip.h"

#include "gemm_common.h"

#ifdef PIPE_W4   // the second compile of this file has entry points (and a kernel name in the profiles) of its own
#define vk_gemm_pipe_fit vk_gemm_pipe4_fit
#define vk_gemm_pipe_gnstat_ok vk_gemm_pipe4_gnstat_ok
#define vk_gemm_pipe_launch vk_gemm_pipe4_launch
#define gemm_pipe_kernel gemm_pipe4_kernel
#endif

namespace {

// PIPE_W4 (gemm_pipe4.hip compiles this file a second time with it): the SAME kernel as FOUR waves of 128 x 160 -- one wave per SIMD, a 512-register
// budget (320 accumulator registers, most of them AGPRs), 9 ds_read_b128 per 20 MFMAs instead of 7 per 10 and half the waves at every barrier; the
// pieces are 32 rows (4 waves x 8 rows), 10 weight + 8 activation pieces per wave and K-step. Round-6 experiment: profiles/r06_gemm_pipe4.txt.
#ifdef PIPE_W4
constexpr int PBM = 256, PBN = 320, PNT = 256;
constexpr int PFX = 5, PFY = 4;
#else
constexpr int PBM = 256, PBN = 320, PNT = 512;
constexpr int PFX = 5, PFY = 2;                // weight (MFMA row operand) / activation (column operand) fragments per wave
#endif
constexpr int PRPP = PNT / 8;                  // tile rows per piece (every wave moves 8 rows of a piece)
constexpr int PAP = PBM / PRPP, PWP = PBN / PRPP;
constexpr int PA_BYTES = PBM * 128, PSTAGE = (PBM + PBN) * 128;
constexpr int PCAP = PNT == 512 ? 256 : 432;   // what the epilogue sizes its residual ring by: 16 registers per accumulator + the ring must fit the ARCH VGPRs (four-wave build: 256 of the 320 accumulator registers are AGPRs, so 432 - 320 - 36 = 76 registers of ring)
constexpr int APAR0 = 6 * PAP;                 // first bit of the CONV3X3 upsample parities in the validity word
typedef std::conditional<(8 * PAP > 32), unsigned long long, unsigned>::type amask_t;
constexpr unsigned P_OOB = 0xffffff00u;        // an offset no resource below reaches (host: every operand < P_LIMIT bytes)
constexpr unsigned long long P_LIMIT = 0xfffff000ull;
static_assert(PAP + PWP <= PFX * PFY && 8 * PAP <= 64, "the piece schedule issues one piece after each MFMA of the first k-substep");

typedef __attribute__((address_space(3))) void* lptr_t;

// The MFMA of the K-loop. Eight-wave build: the builtin. Four-wave build: 320 accumulator registers do not fit the 256 AGPRs, and left to itself the register
// allocator shuttles accumulators between the two halves of the file inside the loop (1060 v_accvgpr moves and 109 scratch accesses per K-step in the first
// build) -- so the accumulators of weight fragments 0..3 are PINNED to AGPRs and those of fragment 4 to VGPRs by the operand constraints of an inline-asm MFMA.
#ifdef PIPE_W4
#if VK_F16
#define PIPE_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define PIPE_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif
template <bool AG>
__device__ __forceinline__ void pipe_mfma(const bf16x8_t& x, const bf16x8_t& y, f32x16_t& c) {
    if constexpr (AG) asm volatile(PIPE_MFMA_OP " %0, %1, %2, %0" : "+a"(c) : "v"(x), "v"(y));
    else asm volatile(PIPE_MFMA_OP " %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y));
}
#define PIPE_MMA(FI, X, Y, C) do { if ((FI) < 4) pipe_mfma<true>(X, Y, C); else pipe_mfma<false>(X, Y, C); } while (0)
#else
#define PIPE_MMA(FI, X, Y, C) C = vk_mfma(X, Y, C)
#endif

#define PIPE_SB() __builtin_amdgcn_sched_barrier(0)

// GN_CPG != 0: an instantiation whose LINEAR epilogue also emits the GroupNorm statistics of the output (VkGemmDesc.gnstat_out, gemm_common.h) for
// groups of GN_CPG channels with GN_NRES residual tensors; kernels of their own so that the others keep their register allocation, and one body
// each (244 VGPRs, no scratch; any two bodies in one kernel spill)
// PF (round 6, DENSE loader): L2 prefetch of the activation rows TWO K-steps ahead. The activation pieces of a dense K-step come from HBM / the Infinity
// Cache (~2 us under load) but are issued only one K-step (~1.3 us of MFMAs) before the barrier that needs them: the phase timers show 450-800 of a
// 3700-4600-tick K-step period waiting on the wave's own pieces (profiles/r04_gemm_pipe.txt section 3, r06_gemm_pipe2.txt), which the convolution loaders
// (taps re-read from L2) do not have. One plain 4-byte buffer_load per lane and K-step -- thread t touches the 64-byte half (t & 1) of row (t >> 1)'s
// 128-byte K-step segment, destination a dummy register -- issued right AFTER the nine pieces of K-step kt + 1 pulls K-step kt + 2's lines into the
// XCD's L2, so that the LDS-DMA of the next iteration finds them there. It is the youngest vector-memory operation at the K-step barrier, whose wait
// therefore becomes vmcnt(1) + a raw s_barrier (__syncthreads() would wait for it: vmcnt(0)). No arithmetic changes: results are bitwise the same.
template <int AMODE, int EPI, bool NT_A, bool SPLIT, int GN_CPG = 0, int GN_NRES = 0, bool PF = false>
__global__ __launch_bounds__(PNT, PNT == 512 ? 2 : 1) void gemm_pipe_kernel(const VkGemmDesc p, const int ksplit) {
    static_assert(!PF || AMODE == AMODE_DENSE, "the L2 prefetch is written for the dense loader");
    constexpr int LN_OFF = 2 * PSTAGE, EV_OFF = LN_OFF + PBM * 8;
    constexpr int NTAPS = (AMODE == AMODE_CONV3X3) ? 9 : (AMODE == AMODE_TEMPORAL3) ? 3 : 1;
    constexpr bool WALK = (EPI == EPI_GEGLU);   // persistent tile walk (pipe_launch); the LINEAR instantiations are compiled as one tile per workgroup
    __shared__ __attribute__((aligned(16))) char smem[2 * PSTAGE + PBM * 8 + epi_vec_floats(PBN) * 4];
#ifdef PIPE_TIMING
    unsigned long long tm_k0, tm_dma = 0, tm_bar = 0, tm_loop = 0, tm_epi = 0, tm_ksteps = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tm_k0) :: "memory");
#endif

    const int tilesN = p.N / PBN;
    const int tilesM = (p.m_end - p.m_begin + PBM - 1) / PBM;   // row tiles of this launch's row range [m_begin, m_end)
    const int ntiles = tilesM * tilesN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    // tile order: as gemm.hip (column tile fastest unless the weights overflow the L2, then panels of 8 row tiles). The launch is
    // PERSISTENT: workgroup b walks tiles b, b + gridDim.x, ... (gridDim.x is a multiple of 8 whenever it is smaller than the tile count, so a
    // workgroup's tiles stay in its XCD's range of the remap)
    // SPLIT (small-M, deep-K LINEAR problems, as gemm.hip's split-K): workgroup = (K slice, tile), the K slice as the slow index; a slice's
    // fp32 partial tile goes to the workspace [slice][M][N] and gemm.hip's finishing pass applies the epilogue
    int kslice = 0;
    auto tile_of = [&](const int bid, int& tm, int& tn) __attribute__((always_inline)) {
        int logical = xcd_remap(bid, SPLIT ? ntiles * ksplit : ntiles);
        if (SPLIT) { kslice = logical / ntiles; logical -= kslice * ntiles; }
        if (tilesN < 8 || (long long)p.N * p.K * 2 <= (3LL << 20)) {
            tn = logical % tilesN;
            tm = logical / tilesN;
        } else {
            constexpr int GM = 8;
            const int panel = logical / (GM * tilesN), r = logical - panel * (GM * tilesN);
            const int gm = (tilesM - panel * GM < GM) ? tilesM - panel * GM : GM;
            tm = panel * GM + r % gm;
            tn = r / gm;
        }
        // TEMPORAL3 (round 6): walk the row tiles FRAME-fastest. Rows are (clip, frame, pixel); a tile of frame t reads the same pixels of frames t - 1, t, t + 1,
        // so every 128-byte piece of the input is read by three tiles. In row order those three are a whole frame (S / 256 tiles) apart -- about one tile time on
        // an XCD, during which it streams ~17 MB through its 4 MiB L2 -- and two of the three reads come back from the Infinity Cache / HBM (650-750 ticks of own-DMA
        // wait per K-step, profiles/r04_gemm_pipe.txt section 3). With the frame as the fast index the tiles of ONE pixel block of consecutive frames run side by
        // side on one XCD (xcd_remap keeps consecutive logical tiles together) and the neighbours' reads are L2 hits. Only where a frame's rows overflow the L2
        // (S * Cin * 2 > 3 MB: level 0 of the BASELINE window, 5.9 MB per frame: 0.313 -> 0.302 ms per launch; at level 1 -- 2.9 MB per frame, the row order's
        // re-reads already hit -- the frame-fastest order measured 4 % SLOWER: profiles/r06_temporal_tile_order.txt), whole pixel blocks (S % 256 == 0), whole
        // row range; the arithmetic per tile is untouched: bitwise the same output.
        if constexpr (AMODE == AMODE_TEMPORAL3) {
            if (p.tile_cfg & 8) return;   // (A/B: VISTA_T3_ORDER=0 keeps the row order)
            const int nblk = p.S / PBM;
            if (nblk * PBM == p.S && p.m_begin == 0 && tilesM == (p.M / p.S) * nblk && (long long)p.S * p.Cin * 2 > (3LL << 20)) {
                const int per_clip = p.T * nblk, clip = tm / per_clip, r = tm - clip * per_clip;
                const int j = r / p.T, t = r - j * p.T;
                tm = clip * per_clip + t * nblk + j;
            }
        }
    };

    // ---- staging assignment: 16-byte chunk lc of tile rows lr + 64 * i; the XOR swizzle lives in the SOURCE chunk index (gemm.hip) ----
    const int lc = tid & 7, lr = tid >> 3;
    const int lsrc = lc ^ ((lr >> 1) & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff_w = ((unsigned)lr * (unsigned)p.K + (unsigned)lsrc * 8u) * 2u;
    const unsigned wpass = (unsigned)PRPP * (unsigned)p.K * 2u;   // bytes between the row groups of two weight pieces
    const unsigned apass = (unsigned)PRPP * (unsigned)(AMODE == AMODE_DENSE ? p.lda : p.Cin) * 2u;   // DENSE / TEMPORAL3: ... of two activation pieces

    // address state of the tile being staged: a resource per operand, per-lane offsets of the activation row groups, validity bits
    __amdgpu_buffer_rsrc_t rw, ra;
    unsigned voff_a[PAP];   // DENSE / TEMPORAL3: [0] only (the row groups are the uniform stride `apass` apart)
    amask_t amask = 0;      // CONV3X3: per piece 3 row-valid + 3 column-valid bits (tap (ky, kx) valid = row bit ky & column bit kx); TEMPORAL3: 3 frame bits
    auto setup = [&](const int m0, const int n0) __attribute__((always_inline)) {
        rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint16_t*)p.Wt + (size_t)n0 * p.K), 0, (int)((unsigned)PBN * (unsigned)p.K * 2u), 0x00020000);
        amask = 0;
        if (AMODE == AMODE_DENSE) {
            const int rows = (p.m_end - m0 < PBM) ? p.m_end - m0 : PBM;
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint16_t*)p.A + (size_t)m0 * p.lda), 0,
                                                   (int)(((unsigned)(rows - 1) * (unsigned)p.lda + (unsigned)p.K) * 2u), 0x00020000);
            voff_a[0] = ((unsigned)lr * (unsigned)p.lda + (unsigned)lsrc * 8u) * 2u;
        } else if (AMODE == AMODE_CONV3X3) {
            const int hw = p.Hout * p.Wout;
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned)(p.M / hw) * (unsigned)(p.H * p.Wd) * (unsigned)p.Cin * 2u), 0x00020000);
#pragma unroll
            for (int i = 0; i < PAP; ++i) {
                const int m = m0 + lr + PRPP * i;
                const int img = m / hw;
                const int rem = m - img * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                const int y0 = oy * p.stride - (p.asym_pad ? 0 : 1), x0 = ox * p.stride - (p.asym_pad ? 0 : 1);
                // Fused nearest x2 upsample of the source (ups = 2, stride 1; openaimodel.py:100-102): tap (ky, kx) of output pixel (oy, ox) reads
                // source pixel ((oy - 1 + ky) >> 1, (ox - 1 + kx) >> 1) = (oy >> 1) - 1 + ((ky + 1 + (oy & 1)) >> 1), likewise in x: the row /
                // column parities ride in amask's top byte and the lane offset is that of source pixel ((oy >> 1) - 1, (ox >> 1) - 1)
                const int sh = p.ups - 1;
                const int ry = sh ? ((y0 + 1) >> 1) - 1 : y0, rx = sh ? ((x0 + 1) >> 1) - 1 : x0;
                if (sh) amask |= (amask_t)(((y0 + 1) & 1) | (((x0 + 1) & 1) << 1)) << (APAR0 + 2 * i);
                // tap (0, 0); wraps for ry / rx = -1, where it is only ever used with a valid tap's offset added
                voff_a[i] = ((unsigned)((img * p.H + ry) * p.Wd + rx) * (unsigned)p.Cin + (unsigned)lsrc * 8u) * 2u;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (m < p.m_end && y0 + t >= 0 && y0 + t < (p.H << sh)) mk |= 1u << t;
                    if (x0 + t >= 0 && x0 + t < (p.Wd << sh)) mk |= 8u << t;
                }
                amask |= (amask_t)mk << (6 * i);
            }
        } else {  // TEMPORAL3: m = (b*T + t)*S + s over [clips*T][S][Cin]; frames outside the window are the conv's zero padding
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned)p.M * (unsigned)p.Cin * 2u), 0x00020000);
            voff_a[0] = ((unsigned)(m0 + lr) * (unsigned)p.Cin + (unsigned)lsrc * 8u) * 2u;
#pragma unroll
            for (int i = 0; i < PAP; ++i) {
                const int m = m0 + lr + PRPP * i;
                const int t = (m / p.S) % p.T;
                if (m < p.m_end) amask |= (amask_t)((t > 0 ? 1u : 0u) | 2u | (t + 1 < p.T ? 4u : 0u)) << (3 * i);
            }
        }
    };

#ifdef PIPE_W4
#ifndef PIPE_W4_SPREAD
#define PIPE_W4_SPREAD 0
#endif
    constexpr bool w4_spread = PIPE_W4_SPREAD != 0;   // A/B at COMPILE time (-DPIPE_W4_SPREAD=1): two K-loop bodies behind a run-time flag spill 1.2-1.7 KB per lane
#endif
    const int nk_all = p.K / BK;
    int nk = nk_all;        // K-steps of this workgroup (SPLIT: of its slice, set below)
    int tap = 0, c0b = 0;   // (tap, channel-slab byte offset) of the NEXT K-step to stage (conv loaders: K-step = (slab kt / NTAPS, tap kt % NTAPS))
    int kb = 0;             // its byte offset inside a weight row (dense: also inside an activation row)

    // piece i (i < 5: 64 weight rows, else 64 activation rows) of the next K-step into `stage`; dma_next() after a K-step's last piece
    auto dma_piece = [&](const int i, const int stage) __attribute__((always_inline)) {
        char* const sA = smem + stage * PSTAGE + wave_u * 1024;
        if (i < PWP) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(sA + PA_BYTES + i * PRPP * 128), 16, voff_w, (unsigned)i * wpass + (unsigned)kb, 0, 0);
            return;
        }
        const int j = i - PWP;
        lptr_t dst = (lptr_t)(sA + j * PRPP * 128);
        if (AMODE == AMODE_DENSE) {
            const unsigned v = voff_a[0] + (unsigned)j * apass;   // (the row term stays in the bounds-checked part of the address: rows past M read zeros)
            if (NT_A) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, v, (unsigned)kb, 0, 2);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, v, (unsigned)kb, 0, 0);
        } else if (AMODE == AMODE_CONV3X3) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool ok = ((amask >> (6 * j + ky)) & (amask >> (6 * j + 3 + kx)) & 1u) != 0;
            unsigned v;
            if (p.ups == 2) {   // (kernel-uniform) per-lane tap offset: the source step of a tap depends on the output pixel's parity
                const unsigned dy = ((unsigned)((amask >> (APAR0 + 2 * j)) & 1u) + (unsigned)(ky + 1)) >> 1, dx = ((unsigned)((amask >> (APAR0 + 1 + 2 * j)) & 1u) + (unsigned)(kx + 1)) >> 1;
                v = voff_a[j] + (dy * (unsigned)p.Wd + dx) * ((unsigned)p.Cin * 2u);
            } else {
                v = voff_a[j] + (unsigned)((ky * p.Wd + kx) * p.Cin) * 2u;
            }
            if (!ok) v = P_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, v, (unsigned)c0b, 0, 0);
        } else {
            const unsigned tapoff = (unsigned)((tap - 1) * p.S * p.Cin) * 2u;
            const bool ok = ((amask >> (3 * j + tap)) & 1u) != 0;
            const unsigned v = ok ? voff_a[0] + (unsigned)j * apass + tapoff : P_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, v, (unsigned)c0b, 0, 0);
        }
    };
    auto dma_next = [&]() __attribute__((always_inline)) {
        kb += BK * 2;
        if (AMODE != AMODE_DENSE) {
            if (++tap == NTAPS) { tap = 0; c0b += BK * 2; }
        }
    };
    // q-th piece of a K-step in issue order: the activation pieces first (they are the ones that may come from HBM)
    auto dma_q = [&](const int q, const int stage) __attribute__((always_inline)) { dma_piece(q < PAP ? PWP + q : q - PAP, stage); };

    // PF: the lane's offset into the tile's activation rows; the dummy destination stays a live register until the kernel ends (a load still in
    // flight must not land in a register the allocator has handed to something else)
    unsigned pf_dummy = 0;
    const unsigned pf_voff = ((unsigned)(tid >> 1) * (unsigned)p.lda + (unsigned)(tid & 1) * 32u) * 2u;
    auto prefetch = [&](const bool live) __attribute__((always_inline)) {
        if constexpr (PF) {
            const unsigned v = live ? pf_voff : P_OOB;   // (rows past M and K-steps past the tile: out of range = no memory access)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(pf_dummy) : "v"(v), "s"(ra), "s"((unsigned)kb) : "memory");
        }
    };

    f32x16_t acc[PFX][PFY];

    // fragment addresses: weights rows wn*160 + 32*f + l31 of sW, activation rows wm*64 + 32*f + l31 of sA, k-substep ks = chunk (2*ks + lh) ^ sw
    const int sw = (l31 >> 1) & 7;
    const int xrow = PA_BYTES + (wn * 160 + l31) * 128, yrow = (wm * (PFY * 32) + l31) * 128;
    auto load_frags = [&](const int stage, const int ks, bf16x8_t* xf, bf16x8_t* yf) __attribute__((always_inline)) {
        const char* sb = smem + stage * PSTAGE + ((((ks * 2 + lh) ^ sw)) << 4);
#pragma unroll
        for (int f = 0; f < PFY; ++f) yf[f] = *(const bf16x8_t*)(sb + yrow + f * 32 * 128);
#pragma unroll
        for (int f = 0; f < PFX; ++f) xf[f] = *(const bf16x8_t*)(sb + xrow + f * 32 * 128);
    };
    auto mma = [&](const bf16x8_t* xf, const bf16x8_t* yf) __attribute__((always_inline)) {
#pragma unroll
        for (int fi = 0; fi < PFX; ++fi)
#pragma unroll
            for (int fj = 0; fj < PFY; ++fj) PIPE_MMA(fi, xf[fi], yf[fj], acc[fi][fj]);
    };
    // the ten MFMAs of one k-substep with the nine pieces of the next K-step, one after each of the first nine MFMAs
    auto mma_dma = [&](const bf16x8_t* xf, const bf16x8_t* yf, const int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int fi = 0; fi < PFX; ++fi)
#pragma unroll
            for (int fj = 0; fj < PFY; ++fj) {
                PIPE_MMA(fi, xf[fi], yf[fj], acc[fi][fj]);
                if (fi * PFY + fj < PWP + PAP) {
                    PIPE_SB();
                    dma_q(fi * PFY + fj, stage);
                    PIPE_SB();
                }
            }
    };

#ifdef PIPE_W4
    // four-wave build: with one wave per SIMD nothing runs in the shadow of a piece's issue stall, and eighteen pieces in a row cost more than eighteen
    // spread out (MI355X_MICROARCH: 60 cycles among bare MFMAs, 100-185 inside a phase already carrying pieces) -> one piece after every SECOND MFMA of the
    // first two k-substeps (q0 = 0 / PWP + PAP - 9 ...): pieces [q0, q0 + n) of the next K-step
    auto mma_dma_spread = [&](const bf16x8_t* xf, const bf16x8_t* yf, const int stage, const int q0, const int n) __attribute__((always_inline)) {
#pragma unroll
        for (int fi = 0; fi < PFX; ++fi)
#pragma unroll
            for (int fj = 0; fj < PFY; ++fj) {
                PIPE_MMA(fi, xf[fi], yf[fj], acc[fi][fj]);
                const int u = fi * PFY + fj;
                if ((u & 1) == 0 && (u >> 1) < n) {
                    PIPE_SB();
                    dma_q(q0 + (u >> 1), stage);
                    PIPE_SB();
                }
            }
    };
#endif

    float2* const lnrow = (float2*)(smem + LN_OFF);
    float* const epi_vec = (float*)(smem + EV_OFF);
    // every tile takes the LDS-staged epilogue (the host checked epi_fast_everywhere; this kernel carries no general one): only the images
    // a tile's rows span remain to be found
    auto plan_of = [&](const int m0) __attribute__((always_inline)) {
        EpiPlan e{true, 0, 1};
        if (EPI == EPI_LINEAR && (p.rowvec || p.rowvec2)) {
            const int last = (m0 + PBM < p.m_end ? m0 + PBM : p.m_end) - 1;
            e.img0 = m0 / p.rows_per_vec;
            e.nimg = last / p.rows_per_vec - e.img0 + 1;
        }
        return e;
    };
    // a tile's epilogue vectors and folded-LayerNorm row statistics -> LDS (issued under the tile's first pieces)
    auto stage_tile_vectors = [&](const int m0, const int n0, const EpiPlan& e) __attribute__((always_inline)) {
        int tv = tid;
        asm volatile("" : "+v"(tv));   // (as the epilogue: nothing lane-derived of this pass is to live across the K-loop)
        if (SPLIT) return;   // (the finishing pass applies the epilogue)
        epi_stage_vectors<PBN, PNT>(p, epi_vec, n0, e, tv);
        if (p.ln_stats != nullptr) {
            for (int r = tv; r < PBM; r += PNT) {
                const int m = m0 + r;
                lnrow[r] = ln_row_stats(p, m < p.m_end ? m : p.m_end - 1);
            }
        }
    };

    // ---- first tile: its first K-step's nine pieces, then (under them) the vectors ----
    int bid = blockIdx.x;
    int tm, tn;
    tile_of(bid, tm, tn);
    int m0 = p.m_begin + tm * PBM, n0 = tn * PBN;
    setup(m0, n0);
    if (SPLIT) {
        const int kt0 = (int)((long long)kslice * nk_all / ksplit), kt1 = (int)((long long)(kslice + 1) * nk_all / ksplit);
        nk = kt1 - kt0;
        kb = kt0 * BK * 2;
        if (AMODE != AMODE_DENSE) { tap = kt0 % NTAPS; c0b = (kt0 / NTAPS) * BK * 2; }
    }
    int st = 0;   // stage holding the current K-step
#pragma unroll
    for (int q = 0; q < PWP + PAP; ++q) dma_q(q, 0);
    dma_next();
    EpiPlan eplan = plan_of(m0);
    stage_tile_vectors(m0, n0, eplan);
    __syncthreads();

    while (true) {
#ifdef PIPE_TIMING
        unsigned long long tm_t0;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tm_t0) :: "memory");
#endif
#pragma unroll
        for (int i = 0; i < PFX; ++i)
#pragma unroll
            for (int j = 0; j < PFY; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bf16x8_t xa[PFX], ya[PFY], xb[PFX], yb[PFY];
        load_frags(st, 0, xa, ya);
        load_frags(st, 1, xb, yb);
        for (int kt = 0; kt + 1 < nk; ++kt) {
            PIPE_SB();
#ifdef PIPE_W4
            if constexpr (w4_spread) {
                mma_dma_spread(xa, ya, st ^ 1, 0, 9);
                PIPE_SB();
                load_frags(st, 2, xa, ya);
                PIPE_SB();
                mma_dma_spread(xb, yb, st ^ 1, 9, PWP + PAP - 9);
                dma_next();
            } else {
                mma_dma(xa, ya, st ^ 1);
                dma_next();
                PIPE_SB();
                load_frags(st, 2, xa, ya);
                PIPE_SB();
                mma(xb, yb);
            }
            PIPE_SB();
#else
            mma_dma(xa, ya, st ^ 1);
            dma_next();
            prefetch(kt + 2 < nk);   // (kb now names K-step kt + 2)
            PIPE_SB();
            load_frags(st, 2, xa, ya);
            PIPE_SB();
            mma(xb, yb);
            PIPE_SB();
#endif
            load_frags(st, 3, xb, yb);
            PIPE_SB();
            mma(xa, ya);
            PIPE_SB();
#ifdef PIPE_TIMING   // s_memtime stamps around the barrier: own-DMA wait and barrier wait per K-step
            unsigned long long t1, t2, t3;
            asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
            if (PF) asm volatile("s_waitcnt vmcnt(1)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
#endif
            if constexpr (PF) {   // everything but the youngest operation (the prefetch) has landed; a raw barrier: __syncthreads() would add vmcnt(0)
                asm volatile("s_waitcnt vmcnt(1)\n s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else
            __syncthreads();   // vmcnt(0): this wave's pieces of K-step kt + 1 have landed; lgkmcnt(0): its reads of stage st are done
#ifdef PIPE_TIMING
            asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t3) :: "memory");
            tm_dma += t2 - t1; tm_bar += t3 - t2; ++tm_ksteps;
#endif
            st ^= 1;
            load_frags(st, 0, xa, ya);
            PIPE_SB();
            mma(xb, yb);   // (the last k-substep of K-step kt runs after the barrier and covers the first reads of the new stage)
            PIPE_SB();
            load_frags(st, 1, xb, yb);
        }
        // last K-step: nothing of this tile left to stage
        PIPE_SB();
        mma(xa, ya);
        PIPE_SB();
        load_frags(st, 2, xa, ya);
        PIPE_SB();
        mma(xb, yb);
        PIPE_SB();
        load_frags(st, 3, xb, yb);
        PIPE_SB();
        mma(xa, ya);
        PIPE_SB();
        mma(xb, yb);
        PIPE_SB();
#ifdef PIPE_W4   // the hazard recogniser does not see inside the inline-asm MFMAs: their results are read (v_accvgpr_read / VALU) only after the pipe has drained
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
#endif
#ifdef PIPE_TIMING
        unsigned long long tm_t1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tm_t1) :: "memory");
        tm_loop += tm_t1 - tm_t0;
#endif

        // the NEXT tile's first K-step goes out before this tile's epilogue: stage st ^ 1 has been free since the last barrier, and the address
        // state of this tile is dead (the epilogue works from m0 / n0 / eplan)
        const int nbid = bid + (int)gridDim.x;
        const bool more = WALK && nbid < ntiles;
        int ntm = 0, ntn = 0;
        if (more) {
            tile_of(nbid, ntm, ntn);
            setup(p.m_begin + ntm * PBM, ntn * PBN);
            kb = 0; tap = 0; c0b = 0;
#pragma unroll
            for (int q = 0; q < PWP + PAP; ++q) dma_q(q, st ^ 1);
            dma_next();
        }
        PIPE_SB();

        // the epilogue's lane-derived state is formed from an OPAQUE copy of the thread index: derived from `tid` it is hoisted out of the
        // tile loop, lives across the K-loop next to 216 accumulator / fragment registers and is spilled (its reloads then queue behind the
        // pieces just issued)
        int te = tid;
        asm volatile("" : "+v"(te));
        const int e_wave = te >> 6, e_l31 = te & 31, e_lh = (te >> 5) & 1;
        const int e_wm = e_wave >> 1, e_wn = e_wave & 1;
        const float2* const lnp = p.ln_stats != nullptr ? lnrow : nullptr;
        if constexpr (SPLIT) {   // fp32 partial tile of this K slice (gemm.hip: gemm_kernel's split-K branch)
            VkGemmDesc q = p;
            q.out = (float*)p.splitk_ws + (size_t)kslice * p.M * p.N;
            q.ldc = p.N;
            q.bias = nullptr; q.rowvec = nullptr; q.res1 = nullptr; q.res2 = nullptr;
            q.alpha = 1.f; q.beta = 0.f;
            q.rowvec2 = nullptr; q.ln_stats = nullptr; q.rowstat_out = nullptr; q.act = 0;
            gemm_epilogue<EPI_LINEAR, true, PFX, PFY, PFY, 5>(q, acc, m0, n0, e_wm, e_wn, e_l31, e_lh);
        } else if constexpr (EPI == EPI_GEGLU) gemm_epilogue_geglu_lds<PFX, PFY, PFY, 5, PBN>(p, acc, m0, n0, e_wm, e_wn, e_l31, e_lh, lnp, epi_vec);
        else gemm_epilogue_linear_lds<PFX, PFY, PFY, 5, PBN, PCAP, GN_CPG, GN_NRES>(p, acc, m0, n0, e_wm, e_wn, e_l31, e_lh, tn * 2 + e_wn, lnp, epi_vec, eplan.img0);
#ifdef PIPE_TIMING
        unsigned long long tm_t2;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tm_t2) :: "memory");
        tm_epi += tm_t2 - tm_t1;
#endif
        if (!more) break;
        __syncthreads();   // every wave is done with this tile's vectors / row statistics
        bid = nbid; tm = ntm; tn = ntn;
        m0 = p.m_begin + tm * PBM; n0 = tn * PBN;
        eplan = plan_of(m0);
        stage_tile_vectors(m0, n0, eplan);
        __syncthreads();   // + vmcnt(0): the next tile's first K-step has landed
        st ^= 1;
    }
    if constexpr (PF) asm volatile("" :: "v"(pf_dummy));   // (live until here: see its declaration)
#ifdef PIPE_TIMING
    if (blockIdx.x == 8 && lane == 0 && p.splitk_ws) {   // per wave: [own-DMA wait, barrier wait, K-steps timed, K-loops, epilogues (+ next tile's setup), kernel] in s_memtime ticks
        unsigned long long tm_end;
        asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tm_end) :: "memory");
        float* o = (float*)p.splitk_ws + wave * 8;
        o[0] = (float)tm_dma; o[1] = (float)tm_bar; o[2] = (float)tm_ksteps; o[3] = (float)tm_loop; o[4] = (float)tm_epi; o[5] = (float)(tm_end - tm_k0);
    }
#endif
}

template <int AMODE, int EPI>
int pipe_launch(const VkGemmDesc* d, hipStream_t stream, int ksplit) {
    const int tilesN = d->N / PBN, tilesM = (d->m_end - d->m_begin + PBM - 1) / PBM;   // (the caller normalised the row range)
    const int ntiles = tilesM * tilesN;
    VkGemmDesc desc = *d;
    if constexpr (AMODE == AMODE_TEMPORAL3) {   // VISTA_T3_ORDER=0: the row tile order of rounds 4 / 5 instead of the frame-fastest one (A/B hook; kernel: tile_of)
        static const bool rows = [] { const char* e = getenv("VISTA_T3_ORDER"); return e && atoi(e) == 0; }();
        desc.tile_cfg = rows ? (desc.tile_cfg | 8) : (desc.tile_cfg & ~8);
    }
    const bool nt_a = (AMODE == AMODE_DENSE && tilesN <= 4);   // as gemm.hip's launch_cfg: activation rows that few column tiles re-read are streamed non-temporally
    if constexpr (EPI == EPI_LINEAR) {
        if (ksplit > 1) {   // K slices x tiles; the caller (gemm.hip) runs the finishing pass
            if (nt_a) hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, AMODE == AMODE_DENSE, true>), dim3(ntiles * ksplit), dim3(PNT), 0, stream, desc, ksplit);
            else hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, false, true>), dim3(ntiles * ksplit), dim3(PNT), 0, stream, desc, ksplit);
            VK_CHECK_LAUNCH();
            return VK_OK;
        }
    }
    if (ksplit > 1) return VK_EINVAL;
    // GEGLU: 256 resident workgroups (one per CU: 162 KB of LDS) walk the tile list, +1-9 % over one workgroup per tile. LINEAR: one workgroup
    // per tile -- a workgroup that ends does not wait for its stores, so the next one's first pieces fly while they drain, whereas the tile
    // walk's first barrier (vmcnt(0)) of the next tile waits for every store of the last: measured -7..-10 % on the short-K level-0 / level-1
    // shapes, +-1 % on the deep-K ones (same-box sweep, profiles/r04_gemm_pipe.txt)
    // VISTA_GEGLU_WALK=0: one workgroup per GEGLU tile as well (A/B hook: do two concurrent half-batch launches interleave better than two tile walks?)
    static const bool geglu_walk = [] { const char* e = getenv("VISTA_GEGLU_WALK"); return !e || atoi(e) != 0; }();
    const int grid = (EPI == EPI_GEGLU && ntiles > 256 && geglu_walk) ? 256 : ntiles;
    if constexpr (EPI == EPI_LINEAR && AMODE != AMODE_DENSE) {
        if (desc.gnstat_out) {   // (vk_gemm_pipe_launch checked vk_gemm_pipe_gnstat_ok: N = 320 / 640 / 1280, at most one residual tensor)
            const int nres = (desc.res1 != nullptr) + (desc.res2 != nullptr);
#define VK_PIPE_GN(CPG, NR) hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, false, false, CPG, NR>), dim3(grid), dim3(PNT), 0, stream, desc, 1)
            if (desc.N == 320) { if (nres == 0) VK_PIPE_GN(10, 0); else VK_PIPE_GN(10, 1); }
            else if (desc.N == 640) { if (nres == 0) VK_PIPE_GN(20, 0); else VK_PIPE_GN(20, 1); }
            else { if (nres == 0) VK_PIPE_GN(40, 0); else VK_PIPE_GN(40, 1); }
#undef VK_PIPE_GN
            VK_CHECK_LAUNCH();
            return VK_OK;
        }
    }
    if constexpr (AMODE == AMODE_DENSE) {
        // L2 prefetch of the activation rows two K-steps ahead (template flag PF): where the rows are whole 128-byte lines (lda and the base pointer), the
        // K-loop is long enough to have something to prefetch AND the launch is at most one round of tiles. Measured (profiles/r06_gemm_prefetch.txt): a
        // launch that fills the chip for several rounds is bound by request throughput, not latency, and the second request per line costs 2-5 % (step +1 ms
        // with the prefetch everywhere); an under-filled launch (the deep levels, every level of a frame-sharded rank) waits out the full memory latency in
        // each K-step and gains 3-33 %. VISTA_GEMM_PF: unset = this rule, 1 = every dense launch, 0 = never (A/B hooks; bitwise the same results).
        static const int pf_env = [] { const char* e = getenv("VISTA_GEMM_PF"); return e ? atoi(e) : -1; }();
        const bool pf_want = PNT == 512 && (pf_env < 0 ? ntiles <= 256 : pf_env != 0);   // (the prefetch's lane -> row map is written for 512 threads)
        if (pf_want && d->K >= 3 * BK && (d->lda % 64) == 0 && (((size_t)d->A) & 127) == 0) {
            if (nt_a) hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, true, false, 0, 0, true>), dim3(grid), dim3(PNT), 0, stream, desc, 1);
            else hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, false, false, 0, 0, true>), dim3(grid), dim3(PNT), 0, stream, desc, 1);
            VK_CHECK_LAUNCH();
            return VK_OK;
        }
    }
    if (nt_a) hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, AMODE == AMODE_DENSE, false>), dim3(grid), dim3(PNT), 0, stream, desc, 1);
    else hipLaunchKernelGGL((gemm_pipe_kernel<AMODE, EPI, false, false>), dim3(grid), dim3(PNT), 0, stream, desc, 1);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

}  // namespace

// 1 = the pipelined variant takes this (already validated) problem
extern "C" int vk_gemm_pipe_fit(const VkGemmDesc* d) {
    if ((d->epi != EPI_LINEAR && d->epi != EPI_GEGLU) || d->out_f32 || (d->N % PBN) != 0 || d->mx8_out || d->A2) return 0;
    if (d->epi == EPI_GEGLU && d->amode != AMODE_DENSE) return 0;
    if (!epi_fast_everywhere(*d, d->epi, PBM)) return 0;   // the kernel carries the LDS-staged epilogues only
    if ((unsigned long long)PBN * d->K * 2ull >= P_LIMIT) return 0;
    if (d->amode == AMODE_DENSE) return ((unsigned long long)PBM * d->lda * 2ull < P_LIMIT) ? 1 : 0;
    if (d->amode == AMODE_CONV3X3) {
        if (d->ups == 2 && (d->stride != 1 || d->asym_pad)) return 0;   // (the fused nearest x2 upsample comes with stride 1, pad 1 only)
        const long long hw = (long long)d->Hout * d->Wout;
        if (hw <= 0 || (d->M % hw) != 0) return 0;
        return ((unsigned long long)(d->M / hw) * d->H * d->Wd * d->Cin * 2ull < P_LIMIT) ? 1 : 0;
    }
    if (d->amode == AMODE_TEMPORAL3) {
        if (d->halo_prev || d->halo_next) return 0;   // neighbour ranks' halo frames are other tensors: the sixteen-wave kernel's per-lane pointers
        return ((unsigned long long)d->M * d->Cin * 2ull < P_LIMIT) ? 1 : 0;
    }
    return 0;
}

// 1 = a launch of `d` on this kernel with `ksplit` K slices can emit the GroupNorm statistics of its output (VkGemmDesc.gnstat_out; d->gnstat_out
// itself is not looked at: vk_gemm_gnstat_fit asks before the caller has a buffer). The epilogue's conditions: a wave tile = 64 rows of ONE
// image x whole channel groups, every 64-row block entirely inside the problem, the finished accumulator in registers.
extern "C" int vk_gemm_pipe_gnstat_ok(const VkGemmDesc* d, int ksplit) {
    if (!vk_gemm_pipe_fit(d) || ksplit != 1 || d->epi != EPI_LINEAR || d->out_f32) return 0;
    if (d->amode != AMODE_CONV3X3 && d->amode != AMODE_TEMPORAL3) return 0;   // (the instantiations built with the emitting bodies)
    if (d->N != 320 && d->N != 640 && d->N != 1280) return 0;                 // 32 groups of 10 / 20 / 40 channels: 160-column wave tiles hold whole groups
    if (d->ln_stats || d->mx8_out || d->act || d->rowstat_out || (d->res1 && d->res2)) return 0;   // (the emitting bodies: plain epilogue, at most one residual)
    if (d->gn_rows <= 0 || (d->gn_rows % 64) != 0 || (d->M % d->gn_rows) != 0) return 0;
    if (d->m_begin != 0 || (d->m_end != 0 && d->m_end != d->M)) return 0;     // (row ranges: the tail split would hand rows to another kernel)
    return 1;
}

extern "C" int vk_gemm_pipe_launch(const VkGemmDesc* d, void* stream_, int ksplit) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!vk_gemm_pipe_fit(d)) return VK_EINVAL;
    if (ksplit > 1 && (d->epi != EPI_LINEAR || !d->splitk_ws || d->ln_stats || d->rowstat_out || d->act)) return VK_EINVAL;
    if (d->gnstat_out && !vk_gemm_pipe_gnstat_ok(d, ksplit)) return VK_EINVAL;
#ifndef VK_PIPE_ONLY   // (-DVK_PIPE_ONLY=<amode>: a one-loader build for tuning sessions)
#define VK_PIPE_ONLY -1
#endif
    constexpr int only = VK_PIPE_ONLY;
    if constexpr (only < 0 || only == AMODE_DENSE) if (d->amode == AMODE_DENSE)
        return d->epi == EPI_GEGLU ? pipe_launch<AMODE_DENSE, EPI_GEGLU>(d, stream, ksplit) : pipe_launch<AMODE_DENSE, EPI_LINEAR>(d, stream, ksplit);
    if constexpr (only < 0 || only == AMODE_CONV3X3) if (d->amode == AMODE_CONV3X3) return pipe_launch<AMODE_CONV3X3, EPI_LINEAR>(d, stream, ksplit);
    if constexpr (only < 0 || only == AMODE_TEMPORAL3) if (d->amode == AMODE_TEMPORAL3) return pipe_launch<AMODE_TEMPORAL3, EPI_LINEAR>(d, stream, ksplit);
    return VK_EINVAL;
}
