// bf16 MFMA GEMM family for the VideoUNet hot path (gfx950):
//   out[m][n] = epilogue( sum_k A(m,k) * W[n][k] )
// A(m,k) is produced by one of three loaders, all over token-major (NHWC) bf16 activations:
//   DENSE      A[m][k]                       nn.Linear / 1x1 conv     (vwm/modules/attention.py:344-346,421; openaimodel.py:241)
//   CONV3X3    implicit-GEMM 3x3 conv, pad 1, stride 1|2, optional fused nearest x2 upsample of the source
//                                            (openaimodel.py:198,232 ResBlock convs; :136 Downsample; :100-102 Upsample)
//   TEMPORAL3  3x1x1 conv over frames, pad (1,0,0)   (video_model.py:38-52 time_stack)
//   CONV3D     3x3x3 conv over (frames, H, W), pad 1 (temporal VAE decoder: autoencoding/temporal_ae.py:24-37,82-87)
// W is [Npad][K] with K contiguous (= nn.Linear.weight layout; conv weights are packed [Cout][Cin/64][tap][64]: the K-loop walks all
// taps of one 64-channel slab before the next slab, so a tile re-reads its input rows on CONSECUTIVE K-steps and they hit the XCD's L2.
// With the tap-major order of round 1 ([Cout][tap][Cin]) the re-read came 5..20 K-steps x 32 workgroups later, missed the 4 MiB L2 and
// went through the fabric: 2.65 GB fetched per level-0 conv launch for 0.3 GB of input (rocprofv3 FETCH_SIZE, profiles/r02_pmc_traffic.txt).
//
// Tiling (template): block tile BM x BN x 64 with WM x WN waves, each wave FM x FN MFMA 32x32x16 bf16 tiles, fp32 accumulate.
// Every layout of THIS file keeps a wave at <= 128 VGPRs, i.e. FOUR waves per SIMD: co-resident waves cover LDS-read latency, LDS-DMA
// issue and the per-K-step barrier (profiles/r01_vendor_blas_yardstick.txt). Round 4 added the opposite design for the 256x320 tile --
// eight waves, 256 VGPRs, a hand-ordered K-step (gemm_pipe.hip: +14-17 % on the convolutions) -- which the launcher below now prefers
// wherever it takes the problem (tile variant 7), and two special-purpose kernels (gemm_stream.hip: K = 320 projections; ff_fused.hip:
// the level-0 FeedForward); the kernels here remain for the other tiles, CONV3D, fp32 output, two-source A, halo frames and TRANS.
//   256x320 (16 waves 8x2, wave tile 32x160, single-buffered fragments)  N a multiple of 320 (every channel count of the
//                                            shipped UNet): no padded columns, the conv gather of A runs once per 320 channels
//   256x256 (16 waves 4x4, wave tile 64x64)  GEGLU and N % 320 != 0
//   256x128 (16 waves 4x4, wave tile 64x32)  other narrow N
//   128x128 ( 8 waves 4x2, wave tile 32x64)  small problems, two workgroups per CU
// Small-M, deep-K LINEAR problems run the largest tile over 2..8 K slices (split-K; fp32 partials in the caller's workspace,
// fixed-order finishing pass) so that tiles x slices fills the chip.
// LDS: two stages of (A tile + W tile); rows are 128 B (64 bf16) and the 16-B chunk index is XOR-swizzled
// with (row>>1)&7 so the ds_read_b128 fragment reads are bank-conflict free. Tiles stream HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4; the swizzle, the conv gathers and the zero padding live in the per-lane SOURCE address): tile t+1
// lands while the MFMAs consume tile t, one barrier per K-step, fragment reads double-buffered in registers where they fit.
// The MFMA is issued "swapped" (weights are the row/A operand, activations the column/B operand) so that a lane
// owns one output row m and 4 consecutive output columns per accumulator quad -> 8-byte bf16x4 stores and
// per-lane-contiguous fused epilogues (bias, per-image row vector, residuals, GEGLU).
#include <stdlib.h>

#include "common.h"
#include "vista_hip.h"

#include "gemm_common.h"

// gemm_pipe.hip: the eight-wave pipelined 256x320 variant (tile_cfg 7); fit = 1 when it takes the problem
extern "C" int vk_gemm_pipe_fit(const VkGemmDesc* d);
extern "C" int vk_gemm_pipe_launch(const VkGemmDesc* d, void* stream, int ksplit);
// gemm_pipe4.hip: the same kernel as four waves of 128 x 160 (round-6 experiment, measured 3-20 % behind: profiles/r06_gemm_pipe4.txt). NOT part of the
// default build (VISTA_BUILD_PIPE4=1 python -m vista_amd.build adds it): a weak reference, resolved only in an A/B library
extern "C" int vk_gemm_pipe4_launch(const VkGemmDesc* d, void* stream, int ksplit) __attribute__((weak));
extern "C" int vk_gemm_pipe_gnstat_ok(const VkGemmDesc* d, int ksplit);
// gemm_pipe2.hip: the four-wave pipelined 128x320 variant, two workgroups per CU (tile_cfg bit 4 / VISTA_GEMM_PIPE2: an A/B option, off by default)
extern "C" int vk_gemm_pipe2_fit(const VkGemmDesc* d);
extern "C" int vk_gemm_pipe2_launch(const VkGemmDesc* d, void* stream);

namespace {

// 128x160 two-per-CU tiles for DENSE LINEAR GEMMs with N <= K <= this. Same-box sweep (profiles/r03_tile5_sweep.txt): K = N = 320 projections
// +20 % (0.305 -> 0.255 ms, 3.5 TB/s algorithmic), K = N = 640 +14 %, K = N = 1280 -7 %; N = 3K (q|k|v) -15 % and K = 4N (FF out) -3..-18 %: the
// extra column tiles re-read the activations / the deep K-loop is MFMA-bound and wants the big tile.
// Round 4: at K = N = 640 the pipelined 256x320 kernel (gemm_pipe.hip) beats the 128x160 tile (level 1, 115200 rows, + residual + row sums:
// 0.166 vs 0.175 ms, profiles/r04_gemm_pipe.txt), so the default bound is 320 now; the 128x160 tile keeps K = N = 320 (0.240 vs 0.248 ms) and the
// under-filled small problems of the rule further down.
constexpr int TILE5_MAX_K_DEFAULT = 320;

__device__ uint4 g_zero16;  // source of the zero fill for out-of-image conv taps on the LDS-DMA path (zero-initialised)

template <int AMODE, int EPI, bool OUT_F32, int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 16) ? 4 : 2) void gemm_kernel(const VkGemmDesc p, const int ksplit) {
    constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    constexpr int NT = WM * WN * 64;              // threads
    constexpr int RPP = NT / 8;                   // tile rows staged per pass (8 x 16-B chunks per 128-B row)
    constexpr int AP = BM / RPP, WP = (BN + RPP - 1) / RPP;   // staging passes for the A / W tiles (the last W pass may be partial)
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    constexpr bool TR = (EPI == EPI_TRANS);
    constexpr bool PERSIST = (EPI == EPI_GEGLU);
    constexpr int FX = TR ? FM : FN, FY = TR ? FN : FM;  // MFMA row-operand / column-operand fragments per wave
    static_assert(BM % RPP == 0 && BN % 8 == 0, "A rows must be a multiple of the staging pass, W rows of a wave's 8-row slice");
    constexpr bool FRAG_DB = (NT <= 512) || (FX * FY <= 4);  // 16 waves x 32x160 wave tiles: no registers for double-buffered fragments
    // ONE LDS object (a second __shared__ array makes hipcc drain vmcnt before every K-step's first ds_read): two tile stages, then the
    // (mean, rstd) table of the tile's BM activation rows for the folded LayerNorm
    // ... then the epilogue's per-column / per-image vectors (gemm_common.h, "LDS-staged epilogue")
    constexpr int LN_OFF = 2 * STAGE_BYTES, EV_OFF = LN_OFF + BM * 8;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + BM * 8 + epi_vec_floats(BN) * 4];

    const int tilesN = (p.N + BN - 1) / BN;
    const int tilesM = (p.m_end - p.m_begin + BM - 1) / BM;   // row tiles of this launch's row range [m_begin, m_end) (launchers normalise m_end)
    // split-K (small-M problems): workgroup = (K slice, tile), K slice as the slow index so neighbours still share weight tiles
    const int ntiles = tilesM * tilesN;
    // PERSIST (GEGLU kernels): a launch of 256 workgroups walks the tile list with stride gridDim.x instead of one workgroup per tile: no
    // workgroup drain / dispatch between tiles, and the next tile's first DMA overlaps this tile's store drain: +3-6 % on the GEGLU shapes
    // (longest epilogue of the family), neutral to -3 % on the other epilogues, which keep one workgroup per tile
    // (profiles/r02_gemm_sweep_persistent_experiment.jsonl). Also staging the next tile's first K-step BEFORE the gelu epilogue (two
    // statistics tables, addresses formed on the fly) was correct but measured -1..-2.5 % in a same-box A/B (profiles/r02_ab_notes.txt):
    // the per-tile overhead is the epilogue itself, not the prologue's memory round trip. No state is carried across tiles except the loop counter; gridDim.x is a
    // multiple of 8, so a workgroup's tiles stay in its XCD's range of the remap.
    const int total_wg = ntiles * ksplit;
    int bid = blockIdx.x;
  do {
    int tid = threadIdx.x;
    if constexpr (PERSIST) asm volatile("" : "+v"(tid));  // opaque per iteration: keeps the lane-derived address state from being hoisted
                                                          // out of the tile loop (hoisted, it stays live across the body and spills)
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lin = xcd_remap(bid, total_wg);
    const int kslice = lin / ntiles, logical = lin - kslice * ntiles;
    // Tile order. Narrow outputs (tilesN < 8): column tile fastest -- the few column tiles of a row tile run together, share its
    // activation rows in L2, and the whole weight matrix is L2-resident anyway. Wide outputs (GEGLU: 10-40 column tiles, weights of
    // 6-26 MB against a 4 MiB L2): panels of GM = 8 row tiles, row tile fastest inside a panel, so the 32 workgroups an XCD runs at a
    // time form an 8 x 4 block that streams 8 activation + 4 weight K-slices per step instead of 1 + 32 (rocprofv3 FETCH_SIZE of the
    // level-2 GEGLU: 3.06 GB per launch with the column-fastest order = every tile re-streaming its weight tile through the fabric;
    // 1.24 GB with the panels; same-box A/B +4-8 % at level 2, +0-2 % at level 1, -2-3 % at level 0 where it is therefore not used).
    int tn, tm;
    if (tilesN < 8 || (long long)p.N * p.K * 2 <= (3LL << 20)) {  // (weights that fit the L2 are re-read from it whatever the order: level-0 GEGLU)
        tn = logical % tilesN;
        tm = logical / tilesN;
    } else {
        constexpr int GM = 8;
        const int panel = logical / (GM * tilesN), r = logical - panel * (GM * tilesN);
        const int gm = (tilesM - panel * GM < GM) ? tilesM - panel * GM : GM;
        tm = panel * GM + r % gm;
        tn = r / gm;
    }
    const int m0 = p.m_begin + tm * BM, n0 = tn * BN;

    const uint16_t* __restrict__ Ag = (const uint16_t*)p.A;
    const uint16_t* __restrict__ Wg = (const uint16_t*)p.Wt;
    // address of the zero word that out-of-image taps read: formed ONCE and made opaque -- left to itself the compiler re-derives it in
    // every K-step (s_getpc + s_load_dwordx2 of the GOT entry + s_waitcnt lgkmcnt(0), which also drains the wave's LDS reads)
    const uint16_t* zsrc = (const uint16_t*)&g_zero16;
    asm volatile("" : "+s"(zsrc));

    // ---- per-thread global->LDS staging assignment: chunk lc of rows lr + 32*i ----
    const int lc = tid & 7;
    const int lr = tid >> 3;
    // LDS-DMA (global_load_lds) writes lane l's 16 B at wave_base + 16*l, i.e. row lr, physical slot lc. To land the
    // swizzled image (logical chunk c at slot c ^ ((row>>1)&7)) the lane must FETCH logical chunk lc ^ sw instead.
    const int lsrc = lc ^ ((lr >> 1) & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    const uint16_t* wptr[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) wptr[i] = Wg + (size_t)(n0 + lr + RPP * i) * p.K + lsrc * 8;

    // A-row state
    const uint16_t* aptr[AP];
    int a_y0[AP], a_x0[AP];
    bool a_ok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int m = m0 + lr + RPP * i;
        a_ok[i] = m < p.m_end;
        if (m >= p.m_end) m = p.m_end - 1;
        if (AMODE == AMODE_DENSE) {
            aptr[i] = Ag + (size_t)m * p.lda + lsrc * 8;
            a_y0[i] = m;  // row index, for the second source of a channel-concatenated A
            a_x0[i] = 0;
        } else if (AMODE == AMODE_CONV3X3) {
            const int hw = p.Hout * p.Wout;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_y0[i] = oy * p.stride - (p.asym_pad ? 0 : 1);  // asym_pad: padding only below / right of the image
            a_x0[i] = ox * p.stride - (p.asym_pad ? 0 : 1);
            aptr[i] = Ag + (size_t)img * p.H * p.Wd * p.Cin + lsrc * 8;
        } else if (AMODE == AMODE_CONV3D) {  // 3x3x3 conv over (frames, H, W): m = frame*H*W + oy*W + ox, frame = b*T + t
            const int hw = p.H * p.Wd;
            const int fr = m / hw;
            const int rem = m - fr * hw;
            const int oy = rem / p.Wd, ox = rem - oy * p.Wd;
            a_y0[i] = (fr << 12) | oy;              // oy, ox < 4096; frame index and in-clip index packed above them
            a_x0[i] = ((fr % p.T) << 12) | ox;
            aptr[i] = Ag + lsrc * 8;
        } else {  // TEMPORAL3: m = (b*T + t)*S + s
            const int fr = m / p.S;
            a_y0[i] = fr % p.T;  // frame index t
            a_x0[i] = ((fr / p.T) * p.S + (m - fr * p.S)) * p.Cin + lsrc * 8;  // element offset of (clip b, pixel s) in a halo frame
            aptr[i] = Ag + (size_t)m * p.Cin + lsrc * 8;
        }
    }

    const int nk_all = p.K / BK;
    const int kt0 = (int)((long long)kslice * nk_all / ksplit), kt1 = (int)((long long)(kslice + 1) * nk_all / ksplit);
    // (tap, channel offset) of the next K-step to stage, for the conv loaders: K-step kt = (channel slab kt / NTAPS, tap kt % NTAPS);
    // starts at this workgroup's first K-step
    constexpr int NTAPS = (AMODE == AMODE_CONV3X3) ? 9 : (AMODE == AMODE_TEMPORAL3) ? 3 : (AMODE == AMODE_CONV3D) ? 27 : 1;
    int tap = (AMODE == AMODE_DENSE) ? 0 : kt0 % NTAPS, c0 = (AMODE == AMODE_DENSE) ? 0 : (kt0 / NTAPS) * BK;

    // direct global -> LDS staging of tile kt into `stage` (one 1-KiB global_load_lds_dwordx4 per wave and 8-row group)
    auto dma_tile = [&](int kt, int stage) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int k0 = kt * BK;
        char* sA = smem + stage * STAGE_BYTES + wave_u * 1024;
        char* sW = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            if (RPP * (i + 1) <= BN || RPP * i + 8 * wave_u < BN)  // wave-uniform: waves past the tile's last row skip the partial pass
                __builtin_amdgcn_global_load_lds((gptr_t)(wptr[i] + k0), (lptr_t)(sW + i * RPP * 128), 16, 0, 0);
        if (AMODE == AMODE_DENSE) {
            // channel concat [A | A2] folded into the loader: K-steps past k_split read the second tensor (block-uniform choice)
            const uint16_t* asrc[AP];
            if (p.A2 != nullptr && k0 >= p.k_split) {
                const uint16_t* A2g = (const uint16_t*)p.A2 + (k0 - p.k_split) + lsrc * 8;
#pragma unroll
                for (int i = 0; i < AP; ++i) asrc[i] = A2g + (size_t)a_y0[i] * p.lda2;
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i) asrc[i] = aptr[i] + k0;
            }
            if (p.tile_cfg & 32) {  // set by launch_cfg: non-temporal policy for an activation stream with little reuse
#pragma unroll
                for (int i = 0; i < AP; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)asrc[i], (lptr_t)(sA + i * RPP * 128), 16, 0, 2);
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)asrc[i], (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        } else if (AMODE == AMODE_CONV3X3) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int sh = p.ups - 1;
            const int He = p.H << sh, We = p.Wd << sh;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int iy = a_y0[i] + ky, ix = a_x0[i] + kx;
                const bool ok = a_ok[i] && iy >= 0 && iy < He && ix >= 0 && ix < We;
                const int sy = iy >> sh, sx = ix >> sh;
                const uint16_t* src = ok ? aptr[i] + ((size_t)(sy * p.Wd + sx) * p.Cin + c0) : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        } else if (AMODE == AMODE_CONV3D) {
            // tap = kt3*9 + ky*3 + kx; zero padding in time (clip ends) and space (video_model-style Conv3d(k=3, pad=1):
            // vwm/modules/autoencoding/temporal_ae.py:24-37,82-87)
            const int kt3 = tap / 9, r9 = tap - kt3 * 9;
            const int ky = r9 / 3, kx = r9 - ky * 3;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int fr = a_y0[i] >> 12, oy = a_y0[i] & 4095;
                const int t = a_x0[i] >> 12, ox = a_x0[i] & 4095;
                const int tt = t + kt3 - 1, iy = oy + ky - 1, ix = ox + kx - 1;
                const bool ok = a_ok[i] && tt >= 0 && tt < p.T && iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
                const uint16_t* src = ok ? aptr[i] + (((size_t)(fr + kt3 - 1) * p.H + iy) * p.Wd + ix) * p.Cin + c0 : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        } else {
            // frames before / after the local range come from the neighbour ranks' halo frames ([clips][S][Cin], frame-sharded
            // multi-GPU runs) or are the conv's zero padding at the ends of the window (halo pointer NULL)
            const int dt = tap - 1;
            const uint16_t* hprev = (const uint16_t*)p.halo_prev;
            const uint16_t* hnext = (const uint16_t*)p.halo_next;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int t = a_y0[i] + dt;
                const uint16_t* src = aptr[i] + ((ptrdiff_t)dt * p.S * p.Cin + c0);
                if (t < 0) src = hprev ? hprev + (a_x0[i] + c0) : zsrc;
                if (t >= p.T) src = hnext ? hnext + (a_x0[i] + c0) : zsrc;
                if (!a_ok[i]) src = zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
            }
        }
        if (++tap == NTAPS) { tap = 0; c0 += BK; }
    };

    f32x16_t acc[FX][FY];
#pragma unroll
    for (int i = 0; i < FX; ++i)
#pragma unroll
        for (int j = 0; j < FY; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // X = MFMA row operand, Y = MFMA column operand. Normal: X = weights, Y = activations. TRANS: swapped.
    constexpr int MW = FM * 32, NW = FN * 32;  // wave tile extents along m / n
    const int xoff = TR ? wm * MW : wn * NW;
    const int yoff = TR ? wn * NW : wm * MW;
    const int xbase = TR ? 0 : A_BYTES;   // X tile lives in sA (TRANS) or sW
    const int ybase = TR ? A_BYTES : 0;
    const int sw = (l31 >> 1) & 7;
    int frag_off[4];  // byte offset of k-substep ks inside a row
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) frag_off[ks] = ((ks * 2 + lh) ^ sw) << 4;
    const int xrow_off = (xoff + l31) * 128, yrow_off = (yoff + l31) * 128;

    // K-step compute with double-buffered fragments: the ds_read_b128s of k-substep ks+1 are issued BEFORE the MFMAs of
    // substep ks, so the LDS latency is covered by FX*FY MFMAs instead of being exposed at the head of every substep.
    auto load_frags = [&](const char* sb, int ks, bf16x8_t* xf, bf16x8_t* yf) {
#pragma unroll
        for (int f = 0; f < FX; ++f) xf[f] = *(const bf16x8_t*)(sb + xbase + xrow_off + f * 32 * 128 + frag_off[ks]);
#pragma unroll
        for (int f = 0; f < FY; ++f) yf[f] = *(const bf16x8_t*)(sb + ybase + yrow_off + f * 32 * 128 + frag_off[ks]);
    };
    auto mma = [&](const bf16x8_t* xf, const bf16x8_t* yf) {
#pragma unroll
        for (int fi = 0; fi < FX; ++fi)
#pragma unroll
            for (int fj = 0; fj < FY; ++fj)
                acc[fi][fj] = vk_mfma(xf[fi], yf[fj], acc[fi][fj]);
    };
    auto compute = [&](int stage) {
        const char* sb = smem + stage * STAGE_BYTES;
        if constexpr (!FRAG_DB) {  // four waves per SIMD cover the LDS latency instead of a second fragment buffer
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8_t xs[FX], ys[FY];
                load_frags(sb, ks, xs, ys);
                mma(xs, ys);
            }
            return;
        }
        bf16x8_t xa[FX], ya[FY], xb[FX], yb[FY];
        // sched_barrier(0) pins the phase order; without it hipcc sinks every read next to its first use again
        load_frags(sb, 0, xa, ya);
        load_frags(sb, 1, xb, yb);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, ya);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(sb, 2, xa, ya);
        __builtin_amdgcn_sched_barrier(0);
        mma(xb, yb);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(sb, 3, xb, yb);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, ya);
        __builtin_amdgcn_sched_barrier(0);
        mma(xb, yb);
    };
    const int nk = kt1;
    // LDS-DMA pipeline: tile kt+1 streams straight into the free LDS stage while the MFMAs consume tile kt; the
    // barrier's vmcnt(0) retires the DMA. No staging VGPRs, no ds_write pass.
    float2* const lnrow_cur = (float2*)(smem + LN_OFF);
    float* const epi_vec = (float*)(smem + EV_OFF);
    const EpiPlan eplan = (ksplit > 1) ? EpiPlan{false, 0, 1} : epi_plan<EPI, OUT_F32, BM, BN>(p, m0, n0);
    dma_tile(kt0, kt0 & 1);
    if (eplan.fast) epi_stage_vectors<BN, NT>(p, epi_vec, n0, eplan, tid);  // like the row statistics: under the first tile's DMA
    if (p.ln_stats != nullptr) {
        // row statistics of this tile's activation rows -> LDS. Issued AFTER the first tile's DMA so that both HBM round trips are in
        // flight together; visible to every wave after the barrier below.
        for (int r = tid; r < BM; r += NT) {
            const int m = m0 + r;
            lnrow_cur[r] = ln_row_stats(p, m < p.m_end ? m : p.m_end - 1);
        }
    }
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) dma_tile(kt + 1, stage ^ 1);
        compute(stage);
        __syncthreads();
    }

    if constexpr (EPI == EPI_LINEAR) {
        if (ksplit > 1) {  // fp32 partial tile of this K slice -> workspace [kslice][M][N]; splitk_finish_kernel applies the epilogue
            VkGemmDesc q = p;
            q.out = (float*)p.splitk_ws + (size_t)kslice * p.M * p.N;
            q.ldc = p.N;
            q.bias = nullptr; q.rowvec = nullptr; q.res1 = nullptr; q.res2 = nullptr;
            q.alpha = 1.f; q.beta = 0.f;
            q.rowvec2 = nullptr; q.ln_stats = nullptr; q.rowstat_out = nullptr; q.act = 0;
            gemm_epilogue<EPI_LINEAR, true, FX, FY, FM, FN>(q, acc, m0, n0, wm, wn, l31, lh);
            continue;  // (split-K launches are never persistent: one slice-tile per workgroup)
        }
    }
    if constexpr (!OUT_F32 && EPI == EPI_LINEAR) {
        if (eplan.fast) gemm_epilogue_linear_lds<FX, FY, FM, FN, BN, (WM * WN == 16) ? 128 : 256>(p, acc, m0, n0, wm, wn, l31, lh, tn * WN + wn, p.ln_stats != nullptr ? lnrow_cur : nullptr, epi_vec, eplan.img0);
        else gemm_epilogue<EPI, OUT_F32, FX, FY, FM, FN>(p, acc, m0, n0, wm, wn, l31, lh, tn * WN + wn, p.ln_stats != nullptr ? lnrow_cur : nullptr);
    } else if constexpr (EPI == EPI_GEGLU) {
        if (eplan.fast) gemm_epilogue_geglu_lds<FX, FY, FM, FN, BN>(p, acc, m0, n0, wm, wn, l31, lh, p.ln_stats != nullptr ? lnrow_cur : nullptr, epi_vec);
        else gemm_epilogue<EPI, OUT_F32, FX, FY, FM, FN>(p, acc, m0, n0, wm, wn, l31, lh, tn * WN + wn, p.ln_stats != nullptr ? lnrow_cur : nullptr);
    } else {
        gemm_epilogue<EPI, OUT_F32, FX, FY, FM, FN>(p, acc, m0, n0, wm, wn, l31, lh, tn * WN + wn, p.ln_stats != nullptr ? lnrow_cur : nullptr);
    }
    if (PERSIST && bid + (int)gridDim.x < total_wg) __syncthreads();  // the next tile rewrites the LDS row-statistics table and stage 0
  } while (PERSIST && (bid += gridDim.x) < total_wg);
}

// Second pass of a split-K GEMM: out = alpha*(sum_s partial[s] + bias + rowvec + res1) + beta*res2, partials summed in slice order
// (bitwise reproducible). One thread per 4 consecutive columns of a row.
template <bool OUT_F32>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const VkGemmDesc p, const int ksplit) {
    const long long quads = (long long)p.M * (p.N >> 2);
    const uint16_t* __restrict__ res1 = (const uint16_t*)p.res1;
    const uint16_t* __restrict__ res2 = (const uint16_t*)p.res2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / (p.N >> 2));
        const int n = (int)(i - (long long)m * (p.N >> 2)) * 4;
        const float* ws = (const float*)p.splitk_ws + (size_t)m * p.N + n;
        float4 a = *(const float4*)ws;
        for (int s = 1; s < ksplit; ++s) {
            const float4 b = *(const float4*)(ws + (size_t)s * p.M * p.N);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
        if (p.bias) {
            const float4 b = *(const float4*)(p.bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.rowvec) {
            const float4 b = *(const float4*)(p.rowvec + (size_t)(m / p.rows_per_vec) * p.ldv + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (res1) {
            const uint2 r = *(const uint2*)(res1 + (size_t)m * p.ld_res1 + n);
            v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        if (res2) {
            const uint2 r = *(const uint2*)(res2 + (size_t)m * p.ld_res2 + n);
            float t[4] = {bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y)};
            if (p.rowvec2) {
                const float4 b = *(const float4*)(p.rowvec2 + (size_t)(m / p.rows_per_vec) * p.ldv + n);
                t[0] += b.x; t[1] += b.y; t[2] += b.z; t[3] += b.w;
            }
            v[0] += p.beta * t[0]; v[1] += p.beta * t[1]; v[2] += p.beta * t[2]; v[3] += p.beta * t[3];
        }
        if (OUT_F32) *(float4*)((float*)p.out + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        else *(uint2*)((uint16_t*)p.out + (size_t)m * p.ldc + n) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    }
}

template <int AMODE, int EPI, bool OUT_F32, int WM, int WN, int FM, int FN>
int launch_cfg(const VkGemmDesc* d, hipStream_t stream, int ksplit = 1) {
    constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    const int tilesN = (d->N + BN - 1) / BN;
    const int tilesM = (d->m_end - d->m_begin + BM - 1) / BM;   // (vk_gemm_bf16 normalised the row range)
    VkGemmDesc desc = *d;
    // Activation rows that at most 4 column-tiles ever read are streamed with the non-temporal policy: they would only evict the
    // weight tile every workgroup shares (measured +6-8 % on the K=320 level-0 projections, -5-11 % when 10+ column tiles re-read A).
    desc.tile_cfg = (desc.tile_cfg & ~32) | ((AMODE == AMODE_DENSE && tilesN <= 4) ? 32 : 0);
    int grid = tilesM * tilesN * ksplit;
    if (EPI == EPI_GEGLU) {  // persistent tile walk (see the kernel): one (two for the 8-wave 128x128 variant) resident workgroup per CU
        const int resident = 256 * ((WM * WN == 16) ? 1 : 2);
        if (grid > resident) grid = resident;
    }
    hipLaunchKernelGGL((gemm_kernel<AMODE, EPI, OUT_F32, WM, WN, FM, FN>), dim3(grid), dim3(WM * WN * 64), 0, stream, desc, ksplit);
    VK_CHECK_LAUNCH();
    if (ksplit > 1) {
        const long long quads = (long long)d->M * (d->N >> 2);
        const int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
        hipLaunchKernelGGL((splitk_finish_kernel<OUT_F32>), dim3(grid), dim3(256), 0, stream, desc, ksplit);
        VK_CHECK_LAUNCH();
    }
    return VK_OK;
}

// Tile-shape choice (host side, per problem). Preference order: 256x320 (N a multiple of 320, not GEGLU), 256x256 (unless
// rounding N up to 256 wastes > 10% of the MFMA work), 256x128, 128x128 -- but a variant is only taken if its grid covers the
// chip (>= 192 workgroups, i.e. at least 3/4 of the CUs with one workgroup each; the 128x128 variant runs two per CU). Small-M problems (deep UNet levels, and every
// level of a frame-sharded multi-GPU run) therefore fall back to smaller tiles instead of leaving CUs idle.
struct TileChoice { int cfg, ksplit; };  // cfg: 1 = 128x128, 2 = 256x128, 3 = 256x256, 4 = 256x320, 5 = 128x160 (two workgroups per CU), 7 = 256x320 pipelined (eight waves)

// cfg 5 exists for DENSE / LINEAR / bf16-out GEMMs only (the HBM-bound K = C projections it is meant for): everything else maps it to 4
inline bool cfg5_ok(const VkGemmDesc* d) { return d->amode == AMODE_DENSE && d->epi == EPI_LINEAR && !d->out_f32 && (d->N % 160) == 0; }
inline TileChoice choose_tile(const VkGemmDesc* d) {
    static const int tile5_max_k = [] { const char* e = getenv("VISTA_TILE5_MAXK"); return e ? atoi(e) : TILE5_MAX_K_DEFAULT; }();
    const int amode = d->amode, epi = d->epi;
    const int force = d->tile_cfg & 7;  // 0 = auto (tests / tuning force a variant)
    // tile_cfg bit 4 (alone): the two-per-CU pipelined kernel is asked for. Where it takes the problem the answer is the 320-column, two-wave-column
    // geometry of variant 7 without K slices (row-sum slabs, vk_gemm_tile_choice), whatever the size rules below would pick; launch() then routes it.
    if ((d->tile_cfg & 16) && force == 0 && vk_gemm_pipe2_fit(d)) return {7, 1};
    int cfg = force;
    if (d->mx8_out) cfg = 4;  // MX-fp8 output lives in the LDS-staged epilogue of whole 320-column tiles (validate() checked N and mx8_cols)
    if (cfg == 5 && !cfg5_ok(d)) cfg = 4;
    if (cfg == 7 && !vk_gemm_pipe_fit(d)) cfg = 4;
    if (cfg == 4 && amode == AMODE_CONV3D) cfg = 3;  // the 27-tap loader's extra address state does not fit the 256x320 register budget
    if (cfg == 0) {
        const int rows = d->m_end - d->m_begin;   // (the row range of this call: all M rows unless a caller asked for a part)
        auto wgs = [&](int bm, int bn) { return (long long)((rows + bm - 1) / bm) * ((d->N + bn - 1) / bn); };
        const int n256 = (d->N + 255) / 256 * 256;
        // GEGLU could run on the 256x320 tile since the fragment-local packing, but measured 2-9 % slower there than on 256x256
        const bool ok320 = (epi != EPI_GEGLU) && (amode != AMODE_CONV3D) && (d->N % 320 == 0);
        const bool ok256 = n256 * 10 <= d->N * 11;
        const long long need = 192;  // >= 75 % of the 256 CUs in a single round still beats the smaller, less efficient tiles
        // HBM-bound projections (K = C = N or thereabouts: under ~250 FLOP per byte of activations moved, against a ridge of 312): the
        // 256x320 tile leaves ONE workgroup per CU whose DMA / MFMA / residual-read / store phases never overlap (2.9 TB/s algorithmic at
        // K = N = 320, round 2). 128x160 tiles (81 KB of LDS) put TWO four-wave workgroups on a CU, out of phase with each other; the second
        // column tile re-reads the activation rows from L2. Set VISTA_TILE5_MAXK (ops.py -> tile_cfg) to tune / disable.
        if (cfg5_ok(d) && tile5_max_k > 0 && d->K <= tile5_max_k && d->N <= d->K && wgs(128, 160) >= 1024) cfg = 5;
        else if (ok320 && wgs(256, 320) >= need) cfg = 4;
        else if (ok256 && wgs(256, 256) >= need) cfg = 3;
        else if (wgs(256, 128) >= need) cfg = 2;
        else cfg = 1;
        // Round-aware corrections for SMALL problems (one rank of a frame-sharded run: 7-8 images instead of 50; the deepest level of the
        // full problem). Same-box sweep at 7 images, all five tiles forced in turn (profiles/r03_gemm_sweep_rank7.jsonl):
        //  * GEGLU walks its tile list with 256 persistent workgroups, so its time is ceil(tiles / 256) tile-times: 4032 x 10240 x 1280 is
        //    640 tiles of 256x256 = 3 rounds but 512 tiles of 256x320 = exactly 2 (126 -> 108 us); the wider tile is ~5 % slower per
        //    FLOP (the 2-9 % measured at full size, where the rounds are 18 / 36 / 71 and the rule keeps 256x256).
        //    Since the LDS-staged epilogue the wider tile is no longer slower per FLOP once K >= 640 (same-box sweep, 50 images: level 1
        //    0.875 -> 0.842 ms, level 2 0.740 -> 0.738; level 0 / K = 320 1.037 -> 1.079, where it stays 5 % behind): weight it +3 % there.
        if (epi == EPI_GEGLU && amode == AMODE_DENSE && (d->N % 320) == 0 && cfg == 3) {
            const long long r3 = (wgs(256, 256) + 255) / 256, r4 = (wgs(256, 320) + 255) / 256;
            if (r4 * 320 * 100 < r3 * 256 * (d->K >= 640 ? 103 : 95)) cfg = 4;
        }
        //  * DENSE LINEAR: when the tile picked above fills < 75 % of its last round (e.g. 16128 x 640: 126 tiles of 256x320 do not reach
        //    `need`, 315 of 256x128 are 1.23 rounds), 128x160 tiles -- two co-resident workgroups per CU, which degrade gracefully in a
        //    partial round -- were 16-34 % faster on every K (640 .. 5120), including the K = 4N FeedForward out-projections that prefer
        //    the big tile at full size. Needs >= 192 of them; smaller problems are left to the split-K rule below.
        if (cfg5_ok(d) && cfg != 5 && tile5_max_k > 0 && wgs(128, 160) >= need) {
            int bm = 256, bn = 320, slots = 256;
            if (cfg == 3) bn = 256;
            else if (cfg == 2) bn = 128;
            else if (cfg == 1) { bm = 128; bn = 128; slots = 512; }
            const long long w = wgs(bm, bn), rounds = (w + slots - 1) / slots;
            if (w * 100 < rounds * slots * 75) cfg = 5;
        }
    }
    // Split-K for small-M, deep-K problems (deep UNet levels; every level of a frame-sharded multi-GPU rank): when even the
    // 128-wide tiles would leave the chip under-filled or ragged, run the LARGEST tile over 2..8 K slices instead, so that
    // tiles x slices ~ one full round of CUs; fp32 partials go to the caller's workspace and a finishing pass applies the epilogue.
    // Not combined with the LayerNorm fold / row-sum emission (their epilogues need the finished accumulator in registers).
    int ksplit = 1;
    if (epi == EPI_LINEAR && force == 0 && d->splitk_ws && cfg != 4 && cfg != 3 && cfg != 5 && !d->ln_stats && !d->rowstat_out && !d->act && !d->alt_cols_from &&
        d->m_begin == 0 && d->m_end == d->M) {   // (the finishing pass walks all M rows: no split-K on a row range)
        const bool ok320s = (amode != AMODE_CONV3D) && (d->N % 320 == 0);
        const int bn = ok320s ? 320 : 256;
        const long long tiles = (long long)((d->m_end - d->m_begin + 255) / 256) * ((d->N + bn - 1) / bn);
        const int nk = d->K / BK;
        int s = (int)(256 / tiles);
        if (s > 8) s = 8;
        if (s > nk / 8) s = nk / 8;  // at least 8 K-steps per slice
        if (s >= 2 && tiles * s >= 128 && (long long)s * d->M * d->N * 4 <= d->splitk_ws_bytes) {
            ksplit = s;
            cfg = ok320s ? 4 : 3;
        }
    }
    // 256x320 chosen by the rules above (not forced): the eight-wave pipelined kernel where it takes the problem (gemm_pipe.hip; same-box
    // sweep +14-17 % on the 3x3 convs, +5-15 % on the dense / temporal shapes, bitwise the same results). VISTA_GEMM_PIPE=0: A/B hook.
    static const bool pipe_on = [] { const char* e = getenv("VISTA_GEMM_PIPE"); return !e || atoi(e) != 0; }();
    if (cfg == 4 && force == 0 && pipe_on && !d->out_f32 && vk_gemm_pipe_fit(d)) cfg = 7;   // (also the split-K launches of the rule above)
    // GEGLU: +10-11 % at level 1 and +16-17 % at level 2 over the better of 256x256 / 256x320 (sixteen waves), also on one rank's small
    // problems; K = 320 (level 0 without the fused FeedForward) stays on the persistent 256x256 kernel, where the pipelined one is 5 % behind
    if (epi == EPI_GEGLU && (cfg == 3 || cfg == 4) && force == 0 && pipe_on && d->K >= 640 && vk_gemm_pipe_fit(d)) cfg = 7;
    if (epi == EPI_GEGLU && cfg == 7 && force == 0 && d->K < 640) cfg = 3;
    return {cfg, ksplit};
}

// (block-tile width, wave columns) of a variant: the row-sum slabs of rowstat_out are one per (column tile, wave column)
inline void tile_geometry(int cfg, int& bn, int& wn) {
    if (cfg == 5) { bn = 160; wn = 1; }
    else if (cfg == 4 || cfg == 7) { bn = 320; wn = 2; }
    else if (cfg == 3) { bn = 256; wn = 4; }
    else if (cfg == 2) { bn = 128; wn = 4; }
    else { bn = 128; wn = 2; }
}

// Tail split of a one-tile-per-workgroup launch of the 256x320 pipelined kernel (LINEAR epilogue). With one 162 KB workgroup per CU the launch
// takes ceil(tiles / 256) tile times, and the BASELINE shapes leave the last round almost empty: level 0 is 460800 rows = 1800 row tiles =
// 7.03 rounds at N = 320 (eight tiles keep the chip for an eighth round: 12 % of the launch) and 21.09 at N = 960. When the last round would
// be filled to at most VISTA_GEMM_TAIL percent, the whole rounds run as rows [m_begin, m_split) on the pipelined kernel and the remaining
// rows [m_split, m_end) as a second launch of 128x160 tiles (four waves, two workgroups per CU: a quarter of the work per workgroup, all of
// them resident at once) -- the same MFMA sequence per output element, the same row-sum slabs (160 columns each): bitwise the same result
// as the single launch (tests/test_kernels_gpu.py::test_gemm_tail_split_and_row_ranges_are_bitwise). Returns the split row, 0 = no split.
// MEASURED (round 5, profiles/r05_negative_results.txt section 1) and NOT adopted: the split changes nothing (level-0 conv3x3 0.7239 vs 0.7245 ms,
// q|k|v 0.4602 vs 0.4626, temporal conv 0.318 vs 0.306; step 169.4 vs 169.5 ms). The eight tiles of the "eighth round" do not cost a round: alone
// on the chip they run ~2.4x faster than a tile among 255 others (no HBM / L2 contention, and the clock is no longer held down by the power
// of 256 busy CUs), which is about what the extra launch costs. The rule stays available as VISTA_GEMM_TAIL=<percent> (A/B hook), default off.
inline int tail_split_row(const VkGemmDesc* d, const TileChoice& t) {
    static const int env_pct = [] { const char* e = getenv("VISTA_GEMM_TAIL"); return e ? atoi(e) : 0; }();   // OFF by default: see the note above
    const int max_pct = (d->tile_cfg & 64) ? 40 : env_pct;   // tile_cfg bit 6: the caller asks for the split rule (tests)
    if (max_pct <= 0 || t.cfg != 7 || t.ksplit != 1 || (d->tile_cfg & 7) != 0 || d->epi != EPI_LINEAR || d->out_f32 || (d->N % 320) != 0) return 0;
    const int rows = d->m_end - d->m_begin;
    const long long tilesN = d->N / 320, tilesM = (rows + 255) / 256, ntiles = tilesM * tilesN;
    const long long full = ntiles / 256, rem = ntiles % 256;
    if (full == 0 || rem == 0 || rem * 100 > (long long)max_pct * 256) return 0;
    const long long tm_main = full * 256 / tilesN;   // whole row tiles the full rounds cover
    if (tm_main <= 0 || tm_main >= tilesM) return 0;
    const int m_split = d->m_begin + (int)tm_main * 256;
    const long long small = (long long)((d->m_end - m_split + 127) / 128) * (d->N / 160);
    if (small > 512) return 0;   // more than one round of the two-per-CU tiles: the tail would no longer be a single short round
    return m_split;
}

// The two-per-CU pipelined kernel (gemm_pipe2.hip) instead of the launcher's choice `t`? Only where it leaves the row-sum slab geometry alone (the
// 320-column variants 4 / 5 / 7 all write one slab per 160 columns, as it does; GEGLU launches emit none) and never for a split-K launch.
// tile_cfg bit 4 forces it (tests / probes); VISTA_GEMM_PIPE2 = 1 (LINEAR) | 2 (GEGLU) | 3 (both) turns it on for whole-round problems with
// K <= VISTA_PIPE2_MAXK (default 1280). Off by default: profiles/r06_gemm_pipe2.txt.
inline bool pipe2_wanted(const VkGemmDesc* d, const TileChoice& t) {
    static const int mode = [] { const char* e = getenv("VISTA_GEMM_PIPE2"); return e ? atoi(e) : 0; }();
    static const int maxk = [] { const char* e = getenv("VISTA_PIPE2_MAXK"); return e ? atoi(e) : 1280; }();
    const bool forced = (d->tile_cfg & 16) != 0;
    if (!forced && !(mode & (d->epi == EPI_GEGLU ? 2 : 1))) return false;
    if (t.ksplit != 1 || d->amode != AMODE_DENSE || d->out_f32) return false;
    if (d->epi == EPI_LINEAR && t.cfg != 4 && t.cfg != 5 && t.cfg != 7) return false;
    if (!forced) {
        if ((d->tile_cfg & 7) != 0 || d->K > maxk) return false;
        const long long tiles = (long long)((d->m_end - d->m_begin + 127) / 128) * (d->N / 320);
        if (tiles < 512) return false;   // less than one round of two workgroups per CU: the small-problem rules stay
    }
    return vk_gemm_pipe2_fit(d) != 0;
}

template <int AMODE, int EPI, bool OUT_F32>
int launch(const VkGemmDesc* d, hipStream_t stream) {
    const TileChoice t = choose_tile(d);
    if constexpr (AMODE == AMODE_DENSE && (EPI == EPI_LINEAR || EPI == EPI_GEGLU) && !OUT_F32) {
        if (pipe2_wanted(d, t)) return vk_gemm_pipe2_launch(d, stream);
    }
    if constexpr (EPI == EPI_LINEAR && !OUT_F32 && (AMODE == AMODE_DENSE || AMODE == AMODE_CONV3X3 || AMODE == AMODE_TEMPORAL3)) {
        if (const int m_split = tail_split_row(d, t)) {
            VkGemmDesc head = *d, tail = *d;
            head.m_end = m_split;
            tail.m_begin = m_split;
            const int rc = vk_gemm_pipe_launch(&head, stream, 1);
            if (rc != VK_OK) return rc;
            return launch_cfg<AMODE, EPI, OUT_F32, 4, 1, 1, 5>(&tail, stream);
        }
    }
    if constexpr (AMODE == AMODE_DENSE && EPI == EPI_LINEAR && !OUT_F32) {
        if (t.cfg == 5) return launch_cfg<AMODE, EPI, OUT_F32, 4, 1, 1, 5>(d, stream);  // 128x160: four 32x160 wave tiles, two workgroups per CU
    }
    if constexpr (AMODE != AMODE_CONV3D) {
        // 256x320 as sixteen 32x160 wave tiles (<= 128 VGPRs with single-buffered fragments): +4-11 % over eight 64x160 tiles
        // on the projections and the implicit-GEMM convs (tools/gemm_sweep.py), for the same reason as the 256x256 case below
        if (t.cfg == 7) {   // eight 64x160 wave tiles, pipelined K-step (gemm_pipe.hip); VISTA_GEMM_PIPE4: its four-wave build (gemm_pipe4.hip, A/B hook:
            // 1 = every launch of this variant, 2 = the convolution loaders only, 3 = the dense loader only)
            static const int pipe4 = [] { const char* e = getenv("VISTA_GEMM_PIPE4"); return e ? atoi(e) : 0; }();
            const bool w4 = vk_gemm_pipe4_launch != nullptr && (pipe4 == 1 || (pipe4 == 2 && AMODE != AMODE_DENSE) || (pipe4 == 3 && AMODE == AMODE_DENSE));
            const int rc = w4 ? vk_gemm_pipe4_launch(d, stream, t.ksplit) : vk_gemm_pipe_launch(d, stream, t.ksplit);
            if (rc != VK_OK || t.ksplit == 1) return rc;
            const long long quads = (long long)d->M * (d->N >> 2);   // the split-K finishing pass, as launch_cfg's
            const int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
            hipLaunchKernelGGL((splitk_finish_kernel<OUT_F32>), dim3(grid), dim3(256), 0, stream, *d, t.ksplit);
            VK_CHECK_LAUNCH();
            return VK_OK;
        }
        if (t.cfg == 4) return launch_cfg<AMODE, EPI, OUT_F32, 8, 2, 1, 5>(d, stream, t.ksplit);
    }
    // 256x256 runs as SIXTEEN waves (4 per SIMD, 64x64 wave tiles, <= 128 VGPRs): same bytes per FLOP as the 8-wave layout, but twice
    // the waves to cover LDS-read latency, DMA issue and the per-K-step barrier (+5-10 % on GEGLU and the N % 320 != 0 projections).
    if (t.cfg == 3) return launch_cfg<AMODE, EPI, OUT_F32, 4, 4, 2, 2>(d, stream, t.ksplit);
    if (t.cfg == 2) return launch_cfg<AMODE, EPI, OUT_F32, 4, 4, 2, 1>(d, stream);  // 256x128, sixteen 64x32 wave tiles
    return launch_cfg<AMODE, EPI, OUT_F32, 4, 2, 1, 2>(d, stream);  // 128x128 as eight 32x64 wave tiles, two workgroups per CU
}

inline int validate(const VkGemmDesc* d) {
    if (!d || !d->A || !d->Wt || !d->out) return VK_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % BK) != 0 || (d->N % 4) != 0 || d->tile_cfg < 0 || (d->tile_cfg & ~(64 | 16)) > 7) return VK_EINVAL;
    if (d->m_begin < 0 || d->m_end < 0 || d->m_end > d->M || (d->m_end != 0 && d->m_begin >= d->m_end) || (d->m_end == 0 && d->m_begin >= d->M)) return VK_EINVAL;
    if ((d->m_begin != 0 || (d->m_end != 0 && d->m_end != d->M)) && d->epi == EPI_TRANS) return VK_EINVAL;   // (row ranges: LINEAR / GEGLU)
    if (d->amode != AMODE_DENSE && (d->Cin <= 0 || (d->Cin % BK) != 0)) return VK_EINVAL;
    if (d->amode == AMODE_DENSE && (d->lda % 8) != 0) return VK_EINVAL;
    if (d->amode == AMODE_CONV3X3 && (d->K != 9 * d->Cin || d->stride < 1 || d->stride > 2 || d->ups < 1 || d->ups > 2)) return VK_EINVAL;
    if (d->amode == AMODE_TEMPORAL3 && (d->K != 3 * d->Cin || d->T <= 0 || d->S <= 0)) return VK_EINVAL;
    if (d->amode == AMODE_CONV3D && (d->K != 27 * d->Cin || d->T <= 0 || d->H <= 0 || d->Wd <= 0 || d->H >= 4096 || d->Wd >= 4096 ||
                                     (long long)d->M != (long long)(d->M / (d->H * d->Wd)) * d->H * d->Wd)) return VK_EINVAL;
    if ((d->rowvec || d->rowvec2) && d->rows_per_vec <= 0) return VK_EINVAL;
    if (d->rowvec2 && (!d->res2 || d->epi != EPI_LINEAR)) return VK_EINVAL;
    if (d->A2 && (d->amode != AMODE_DENSE || d->k_split <= 0 || d->k_split >= d->K || (d->k_split % BK) != 0 || (d->lda2 % 8) != 0)) return VK_EINVAL;
    if (d->ln_stats && (d->amode != AMODE_DENSE || d->A2 || !d->ln_colsum || d->ln_parts <= 0 || d->ln_parts > 64 || !(d->ln_eps > 0.f))) return VK_EINVAL;
    if (d->rowstat_out && (d->epi != EPI_LINEAR || d->out_f32)) return VK_EINVAL;
    if (d->act != 0 && (d->act != 1 || d->epi != EPI_LINEAR)) return VK_EINVAL;
    if (d->mx8_out && (!d->mx8_scales || d->epi != EPI_LINEAR || d->out_f32 || (d->N % 320) != 0 || d->mx8_cols <= 0 || (d->mx8_cols % 320) != 0 ||
                       d->mx8_cols > d->N || (d->ld_mx8 % 16) != 0 || d->ld_mx8 < d->mx8_cols || d->ld_mx8s * 32 < d->mx8_cols ||
                       (((size_t)d->mx8_out) & 15) != 0 || d->rowstat_out || (d->ldc % 8) != 0 || (((size_t)d->out) & 15) != 0 ||
                       (d->amode != AMODE_DENSE) || d->res1 || d->res2 || d->rowvec || d->rowvec2 ||
                       (unsigned long long)d->M * 2ull * (unsigned)d->ldc >= 0xfffff000ull))  // (what epi_plan needs to take the LDS-staged epilogue)
        return VK_EINVAL;
    // (ABI v7, read by the fp16 build only -- validated in both) columns from alt_cols_from on leave as bf16: LINEAR, 16-bit out, whole fragments, no statistics of the output
    if (d->alt_cols_from != 0 && (d->alt_cols_from < 0 || (d->alt_cols_from % 32) != 0 || d->alt_cols_from >= d->N || d->epi != EPI_LINEAR || d->out_f32 ||
                                  d->rowstat_out || d->gnstat_out || d->mx8_out))
        return VK_EINVAL;
    return VK_OK;
}

}  // namespace

// gemm_stream.hip: the weight-stationary streaming kernel for the level-0 K = 320 projections (0 = does not take this problem)
extern "C" int vk_gemm_stream_fit(const VkGemmDesc* d);
extern "C" int vk_gemm_stream_launch(const VkGemmDesc* d, void* stream);
static int stream_fit(const VkGemmDesc* d) {
    static const bool on = [] { const char* e = getenv("VISTA_GEMM_STREAM"); return !e || atoi(e) != 0; }();   // A/B hook: 0 = tiled kernels only
    if (d->tile_cfg & 16) return 0;   // (the two-per-CU pipelined kernel was asked for)
    return (on || (d->tile_cfg & 7) == 6) ? vk_gemm_stream_fit(d) : 0;
}

extern "C" int vk_gemm_rowstat_parts(const VkGemmDesc* d) {
    const int rc = validate(d);
    if (rc != VK_OK) return rc;
    if (d->epi != EPI_LINEAR || d->out_f32) return VK_EINVAL;
    VkGemmDesc q = *d;
    norm_row_range(q);
    q.rowstat_out = (float*)1;  // what the launcher will see (callers size the buffer from this answer BEFORE they can set the pointer): the
                                // streaming kernel is not chosen on its own for a row-sum emitting launch, and such a launch is never split-K
    if (const int fit = (q.m_begin == 0 && q.m_end == q.M) ? stream_fit(&q) : 0) return d->N / (32 * fit);  // one slab per column tile: the workgroup combines its waves' row sums
    if ((q.tile_cfg & 7) == 6) q.tile_cfg &= ~7;
    const TileChoice t = choose_tile(&q);
    int bn, wn;
    tile_geometry(t.cfg, bn, wn);
    return ((d->N + bn - 1) / bn) * wn;
}

// The launcher's decision for `d`, without launching: (block-tile variant 1..5) * 16 + K slices. Pure host arithmetic (the CPU test-suite pins
// the launch rules with it); negative = the error vk_gemm_bf16 would return.
extern "C" int vk_gemm_tile_choice(const VkGemmDesc* d) {
    const int rc = validate(d);
    if (rc != VK_OK) return rc;
    VkGemmDesc q = *d;
    norm_row_range(q);
    if (q.m_begin == 0 && q.m_end == q.M && stream_fit(d)) return 6 * 16 + 1;
    if ((q.tile_cfg & 7) == 6) q.tile_cfg &= ~7;
    const TileChoice t = choose_tile(&q);
    return t.cfg * 16 + t.ksplit;
}

// The row at which vk_gemm_bf16 would split `d` into a pipelined launch of whole rounds + a 128x160 tail launch (tail_split_row), 0 = one launch;
// negative = the error vk_gemm_bf16 would return. Pure host arithmetic.
extern "C" int vk_gemm_tail_split(const VkGemmDesc* d) {
    const int rc = validate(d);
    if (rc != VK_OK) return rc;
    VkGemmDesc q = *d;
    norm_row_range(q);
    if (q.m_begin == 0 && q.m_end == q.M && stream_fit(d)) return 0;
    if ((q.tile_cfg & 7) == 6) q.tile_cfg &= ~7;
    if (q.epi != EPI_LINEAR || q.out_f32 || q.amode == AMODE_CONV3D) return 0;
    return tail_split_row(&q, choose_tile(&q));
}

// ABI v6: can the launch vk_gemm_bf16 would make for `d` emit the GroupNorm statistics of its output (VkGemmDesc.gnstat_out)? The answer is formed
// from the launcher's own decision chain (streaming kernel, tile choice, K slices, tail split) -- none of which reads gnstat_out, so the launch
// that follows with the pointer set takes the same kernel; vk_gemm_bf16 re-checks and refuses rather than dropping the statistics.
extern "C" int vk_gemm_gnstat_fit(const VkGemmDesc* d) {
    const int rc = validate(d);
    if (rc != VK_OK) return rc;
    VkGemmDesc q = *d;
    norm_row_range(q);
    if (q.epi != EPI_LINEAR || q.out_f32 || (q.amode != AMODE_CONV3X3 && q.amode != AMODE_TEMPORAL3)) return 0;
    if (q.m_begin == 0 && q.m_end == q.M && stream_fit(d)) return 0;
    if ((q.tile_cfg & 7) == 6) q.tile_cfg &= ~7;
    const TileChoice t = choose_tile(&q);
    if (t.cfg != 7 || t.ksplit != 1 || tail_split_row(&q, t) != 0) return 0;
    return vk_gemm_pipe_gnstat_ok(&q, 1) ? q.M / 64 : 0;
}

extern "C" int vk_gemm_bf16(const VkGemmDesc* d_in, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int rc = validate(d_in);
    if (rc != VK_OK) return rc;
    if (d_in->gnstat_out && vk_gemm_gnstat_fit(d_in) <= 0) return VK_EINVAL;   // (never a silent launch without the statistics the caller will read)
    VkGemmDesc dq = *d_in;
    norm_row_range(dq);   // m_end = M unless the caller asked for a row range (validate() checked it)
    if (dq.m_begin == 0 && dq.m_end == dq.M && stream_fit(d_in)) return vk_gemm_stream_launch(&dq, stream_);
    if ((dq.tile_cfg & 7) == 6) dq.tile_cfg &= ~7;   // streaming variant requested for a problem it does not take: the launcher's own choice
    const VkGemmDesc* d = &dq;
    const bool f32 = d->out_f32 != 0;
    switch (d->epi) {
        case EPI_LINEAR:
            if ((d->ldc % 4) != 0) return VK_EINVAL;
            if (d->amode == AMODE_DENSE) return f32 ? launch<AMODE_DENSE, EPI_LINEAR, true>(d, stream) : launch<AMODE_DENSE, EPI_LINEAR, false>(d, stream);
            if (d->amode == AMODE_CONV3X3) return f32 ? launch<AMODE_CONV3X3, EPI_LINEAR, true>(d, stream) : launch<AMODE_CONV3X3, EPI_LINEAR, false>(d, stream);
            if (d->amode == AMODE_TEMPORAL3) return f32 ? launch<AMODE_TEMPORAL3, EPI_LINEAR, true>(d, stream) : launch<AMODE_TEMPORAL3, EPI_LINEAR, false>(d, stream);
            if (d->amode == AMODE_CONV3D) return f32 ? launch<AMODE_CONV3D, EPI_LINEAR, true>(d, stream) : launch<AMODE_CONV3D, EPI_LINEAR, false>(d, stream);
            return VK_EINVAL;
        case EPI_GEGLU:
            if (d->amode != AMODE_DENSE || f32 || (d->N % 32) != 0) return VK_EINVAL;  // whole [16 value | 16 gate] fragments
            return launch<AMODE_DENSE, EPI_GEGLU, false>(d, stream);
        case EPI_TRANS:
            if (d->amode != AMODE_DENSE || f32 || d->S <= 0 || (d->S % 4) != 0) return VK_EINVAL;
            return launch<AMODE_DENSE, EPI_TRANS, false>(d, stream);
    }
    return VK_EINVAL;
}
