// Weight-stationary streaming GEMM for the level-0 K = 320 projections (gfx950): out[m][n] = epilogue( sum_k A[m][k] * W[n][k] ), K = 320,
// N = 320 (proj_in / proj_out / attention out-projections: vwm/modules/attention.py:579,602,421) or 960 (the fused q|k|v projection,
// attention.py:344-346: three column tiles of 320), M in the hundreds of thousands.
// These GEMMs are HBM-bound by arithmetic (160 FLOP per byte of activations against a ridge of 312) but ran at 0.45 of the HBM peak on the
// tiled kernel (gemm.hip, 128x160 tiles): its K-loop re-streams the weight tile from L2 for every 128 rows -- 20 of the 36 LDS-DMA pieces of a
// K-step -- and one LDS-DMA issue costs the issuing wave ~120 cycles (profiles/r04_ff_fused_notes.txt): 9 pieces per wave and K-step are 1080
// cycles of issue against 640 cycles of MFMAs. Here the WEIGHTS never move:
//   * wave w of a workgroup keeps W[32 w .. 32 w + 31][0 .. 319] as twenty MFMA A-operand fragments in registers (80 VGPRs) for the whole launch;
//   * the activations stream HBM -> LDS by LDS-DMA in 32-row tiles (20 KB: two pieces per wave and tile), three tiles ahead in a ring of four;
//     every wave of the workgroup multiplies the same tile against its own 32 output columns (20 MFMAs, B operand from LDS), one raw
//     s_barrier per tile;
//   * residual tile (N = 320) and LayerNorm row statistics ride DMA rings of their own, so the loop contains no ordinary load: the only wait
//     is one counted s_waitcnt vmcnt per tile (the rule is derived at `nwait` below) and nothing ever drains the queue;
//   * epilogue per lane: 16 outputs of one row (bias / folded LayerNorm / per-image row vector from LDS, residual from the LDS tile, two
//     16-byte stores); the row sums for the NEXT LayerNorm are combined across the workgroup's waves through LDS into ONE slab.
// The workgroup walks 32-row tiles persistently (one workgroup per CU); for N = 960 the three workgroups that share a row tile sit on one XCD
// and walk in step, so the tile comes from HBM once and from that XCD's L2 twice.
#include "common.h"
#include "vista_hip.h"

#include "gemm_common.h"

namespace {

constexpr int GS_K = 320, GS_ROWS = 32, GS_NS = 4, GS_RNS = 3, GS_MAXPARTS = 4;
constexpr int GS_TILE = GS_ROWS * GS_K * 2;   // 20480 bytes: [32 rows][640 B]

template <int N> __device__ __forceinline__ void gs_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void gs_wait(int n) {  // wave-uniform, 0..8
    switch (n) {
        case 0: gs_wait_vmcnt<0>(); break;
        case 1: gs_wait_vmcnt<1>(); break;
        case 2: gs_wait_vmcnt<2>(); break;
        case 3: gs_wait_vmcnt<3>(); break;
        case 4: gs_wait_vmcnt<4>(); break;
        case 5: gs_wait_vmcnt<5>(); break;
        case 6: gs_wait_vmcnt<6>(); break;
        case 7: gs_wait_vmcnt<7>(); break;
        default: gs_wait_vmcnt<8>(); break;
    }
}

template <int NW, bool RES>
struct GsLds {
    static constexpr int NTILE = NW * 32;
    static constexpr int RT = GS_ROWS * NTILE * 2;
    static constexpr int OFF_A = 0, OFF_R = GS_NS * GS_TILE, OFF_VEC = OFF_R + (RES ? GS_RNS * RT : 0);
    static constexpr int OFF_LN = OFF_VEC + 3 * NTILE * 4, OFF_ST = OFF_LN + GS_NS * GS_MAXPARTS * 256;
    static constexpr int BYTES = OFF_ST + 2 * NW * 256;
};

template <int NW, bool RES>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void gemm_stream_kernel(const VkGemmDesc p, const int tiles_n) {
    using L = GsLds<NW, RES>;
    constexpr int NTILE = L::NTILE, RT = L::RT;
    static_assert(L::BYTES <= 163840, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[L::BYTES];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    // workgroup -> (column tile, row walker): the column tiles of one walker sit on one XCD (block b runs on XCD b % 8) and walk the same rows
    const int b = blockIdx.x;
    int nt = 0, walker = b, nwalk = gridDim.x;
    if (tiles_n > 1) { const int slot = b >> 3; nt = slot % tiles_n; walker = (b & 7) + 8 * (slot / tiles_n); nwalk = gridDim.x / tiles_n; }   // grid = 8 tiles_n k
    const int tilesM = (p.M + GS_ROWS - 1) / GS_ROWS;
    const int ncol0 = nt * NTILE;            // first output column of this workgroup
    const int nb = ncol0 + 32 * w;           // ... of this wave
    const bool has_ln = p.ln_stats != nullptr, has_res = RES && p.res1 != nullptr, has_rv = p.rowvec != nullptr, has_stat = p.rowstat_out != nullptr;
    const int parts = has_ln ? p.ln_parts : 0;

    float* const vbias = (float*)(smem + L::OFF_VEC);
    float* const vcs = vbias + NTILE;
    float* const vrv = vcs + NTILE;
    float* const lnring = (float*)(smem + L::OFF_LN);     // [slot][part][32 rows][2]
    float2* const stpart = (float2*)(smem + L::OFF_ST);   // [parity][wave][32 rows]

    // ---- weights: this wave's 32 output columns x K as 20 A-operand fragments (lane (l31, lh): row nb + l31, k = 16 ks + 8 lh ..) ----
    bf16x8_t wf[20];
    {
        const uint16_t* wr = (const uint16_t*)p.Wt + (size_t)(nb + l31) * GS_K + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) wf[ks] = *(const bf16x8_t*)(wr + 16 * ks);
    }
    for (int i = tid; i < NTILE; i += NW * 64) {
        vbias[i] = p.bias ? p.bias[ncol0 + i] : 0.f;
        vcs[i] = has_ln ? p.ln_colsum[ncol0 + i] : 0.f;
        vrv[i] = 0.f;
    }
    int cur_img = -1;

    // ---- per-lane DMA geometry: the [32][640 B] tile image is lane-linear; this wave stages pieces w and w + 10 (waves 0..9) ----
    // LDS byte 1024 q + 16 lane -> row r, physical chunk c' of the row; it must hold logical chunk (c' & ~7) | ((c' & 7) ^ ((r >> 1) & 7))
    int prow[2], pchunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int byte = 1024 * (w + 10 * i) + 16 * lane;
        prow[i] = byte / 640;
        const int cp = (byte - prow[i] * 640) >> 4;
        pchunk[i] = (cp & ~7) | ((cp & 7) ^ ((prow[i] >> 1) & 7));
    }
    const int pa = w < 10 ? 2 : 0, pr = (has_res && w < 10) ? 2 : 0, ps = (w < parts) ? 1 : 0;
    const int pw = pa + pr + ps;   // this wave's LDS-DMA pieces per tile

    // Stage row tile tm as ring entry seq: `what` & 1 = the residual tile, & 2 = activations + row statistics. Per iteration the wave issues
    // R(seq + 2) first, then A / S(seq + 3) (the residual ring has three entries, the others four).
    auto issue = [&](int tm, int seq, int what) {
        const int m0 = tm * GS_ROWS;
        if (RES && pr && (what & 1)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = m0 + prow[i];
                if (m >= p.M) m = p.M - 1;
                const uint16_t* src = (const uint16_t*)p.res1 + (size_t)m * p.ld_res1 + ncol0 + 8 * pchunk[i];
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + L::OFF_R + (seq % GS_RNS) * RT + 1024 * (w + 10 * i)), 16, 0, 0);
            }
        }
        if (pa && (what & 2)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = m0 + prow[i];
                if (m >= p.M) m = p.M - 1;
                const uint16_t* src = (const uint16_t*)p.A + (size_t)m * p.lda + 8 * pchunk[i];
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + L::OFF_A + (seq % GS_NS) * GS_TILE + 1024 * (w + 10 * i)), 16, 0, 0);
            }
        }
        if (ps && (what & 2)) {   // wave w < parts: the tile's 32 (sum, sum of squares) pairs of slab w, 4 bytes per lane
            int m = m0 + (lane >> 1);
            if (m >= p.M) m = p.M - 1;
            const float* src = p.ln_stats + ((size_t)w * p.M + m) * 2 + (lane & 1);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)((char*)lnring + ((seq % GS_NS) * GS_MAXPARTS + w) * 256), 4, 0, 0);
        }
    };

    // fragment read offsets inside a tile: row l31, logical chunk 2 ks + lh
    const int sw = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) frag_off[k4] = l31 * 640 + (((k4 * 2 + lh) ^ sw) << 4);
    // residual chunks of this lane: logical chunk 4 w + 2 gp + lh of row l31 (8 bf16 each)
    int res_off[2];
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) { const int c = 4 * w + 2 * gp + lh; res_off[gp] = l31 * (NTILE * 2) + (((c & ~7) | ((c & 7) ^ sw)) << 4); }

    const int ntiles_mine = walker < tilesM ? (tilesM - walker + nwalk - 1) / nwalk : 0;
    if (ntiles_mine == 0) return;
    auto tile_of = [&](int seq) { const int s = seq < ntiles_mine ? seq : ntiles_mine - 1; return walker + s * nwalk; };   // (past the end: re-stage the last tile)
    // prologue, in the steady-state order (R(i + 2), then A / S(i + 3) per iteration i = -3, -2, -1)
    issue(tile_of(0), 0, 2);
    issue(tile_of(0), 0, 1); issue(tile_of(1), 1, 2);
    issue(tile_of(1), 1, 1); issue(tile_of(2), 2, 2);
    // Wait rule at the top of iteration seq (before its own issues): R(seq), A(seq), S(seq) must have landed. VMEM loads retire in order among
    // themselves, stores likewise, but not with respect to each other; the loads younger than R(seq) are A / S(seq + 1), R(seq + 1), A / S(seq + 2)
    // = pw + pa + ps. "At most that many operations outstanding" therefore implies R(seq) and everything older have landed whatever share of the
    // outstanding operations are stores -- and it leaves room for the previous tile's two output stores, so that no iteration waits for its
    // predecessor's stores to be acknowledged (with only two tiles in flight and vmcnt(pw) every iteration did: 3.3 us per tile, latency-bound).
    const int nwait = pw + pa + ps;
    const float inv_k = 1.f / (float)GS_K;

#ifdef GS_TIMING
    long long tt[4] = {0, 0, 0, 0};
#define GS_T(i) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); tt[i] += t_ - t0_; t0_ = t_; }
    long long t0_ = (long long)__builtin_amdgcn_s_memtime();
#else
#define GS_T(i)
#endif
    for (int seq = 0; seq < ntiles_mine; ++seq) {
        const int tm = walker + seq * nwalk, m0 = tm * GS_ROWS;
        gs_wait(nwait);                                // entry `seq` has landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // ... every wave's pieces; all reads of entry seq - 1's slots are done as well
        GS_T(0)
        issue(tile_of(seq + 2), seq + 2, 1);
        issue(tile_of(seq + 3), seq + 3, 2);
        GS_T(1)
        // the previous tile's row sums: lane r < 32 of wave (r % NW) adds row r's NW partials in wave order and writes the single slab
        if (has_stat && seq > 0 && lane < GS_ROWS && (lane % NW) == w) {
            const int mp = (tm - nwalk) * GS_ROWS + lane;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) { const float2 v = stpart[(((seq - 1) & 1) * NW + i) * GS_ROWS + lane]; s += v.x; q += v.y; }
            if (mp < p.M) ((float2*)p.rowstat_out)[(size_t)(ncol0 / NTILE) * p.M + mp] = make_float2(s, q);
        }
        // per-image row vector: re-staged when the walk enters another image (tiles never straddle: rows_per_vec % 32 == 0)
        if (has_rv) {
            const int img = m0 / p.rows_per_vec;
            if (img != cur_img) {
                cur_img = img;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();          // (everyone is done with the previous image's vector)
                for (int i = tid; i < NTILE; i += NW * 64) vrv[i] = p.rowvec[(size_t)img * p.ldv + ncol0 + i];
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        // ---- epilogue operands first (their LDS latency then hides under the MFMAs; read at the point of use they were four exposed LDS
        // round trips per tile): LayerNorm statistics of row l31, the lane's two residual chunks ----
        const int m = m0 + l31;
        float ls = 0.f, lq = 0.f;
        if (has_ln) {
            const float* lt = lnring + ((seq % GS_NS) * GS_MAXPARTS) * 64 + 2 * l31;
            for (int i = 0; i < parts; ++i) { ls += lt[i * 64]; lq += lt[i * 64 + 1]; }
        }
        uint4 rw0 = make_uint4(0, 0, 0, 0), rw1 = rw0;
        if (RES && has_res) {
            const char* rt = smem + L::OFF_R + (seq % GS_RNS) * RT;
            rw0 = *(const uint4*)(rt + res_off[0]);
            rw1 = *(const uint4*)(rt + res_off[1]);
        }
        // ---- 32 rows x 32 columns: 20 MFMAs, B operand = the tile's rows from LDS. (Two accumulator chains by k-substep parity: no faster --
        // the three-wave SIMDs are bound by their 60 MFMAs per tile -- and their 16 extra registers spilled: every reload in this loop is a
        // vmcnt(0) that drains the DMA ring.) ----
        const char* at = smem + L::OFF_A + (seq % GS_NS) * GS_TILE;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) {
            const bf16x8_t xf = *(const bf16x8_t*)(at + (ks >> 2) * 128 + frag_off[ks & 3]);
            acc = vk_mfma(wf[ks], xf, acc);
        }
        GS_T(2)
        // ---- epilogue: lane (l31, lh) owns row m0 + l31, columns nb + 8 g + 4 lh + e ----
        float rs = 1.f, nrm = 0.f;
        if (has_ln) {
            const float mu = ls * inv_k;
            rs = rsqrtf(fmaxf(lq * inv_k - mu * mu, 0.f) + p.ln_eps);
            nrm = -mu * rs;
        }
        uint2 rq[4];
        unwiden_pair(rw0, rq[0], rq[1]);
        unwiden_pair(rw1, rq[2], rq[3]);
        uint2 packed[4];
        float ssum = 0.f, qsum = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * w + 8 * g + 4 * lh;
            const float4 bb = *(const float4*)(vbias + c), cc = *(const float4*)(vcs + c), rv = *(const float4*)(vrv + c);
            float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            v[0] = fmaf(rs, v[0], fmaf(nrm, cc.x, bb.x)); v[1] = fmaf(rs, v[1], fmaf(nrm, cc.y, bb.y));   // (no fold: rs = 1, nrm = 0, colsum = 0)
            v[2] = fmaf(rs, v[2], fmaf(nrm, cc.z, bb.z)); v[3] = fmaf(rs, v[3], fmaf(nrm, cc.w, bb.w));
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            v[0] += bf16_lo(rq[g].x); v[1] += bf16_hi(rq[g].x); v[2] += bf16_lo(rq[g].y); v[3] += bf16_hi(rq[g].y);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
            packed[g].x = pack_bf16(v[0], v[1]);
            packed[g].y = pack_bf16(v[2], v[3]);
            const float a0 = bf16_lo(packed[g].x), a1 = bf16_hi(packed[g].x), a2 = bf16_lo(packed[g].y), a3 = bf16_hi(packed[g].y);
            ssum += (a0 + a1) + (a2 + a3);
            qsum = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, qsum))));
        }
        const uint4 s0 = widen_pair(packed[0], packed[1]), s1 = widen_pair(packed[2], packed[3]);
        if (m < p.M) {
            uint16_t* o = (uint16_t*)p.out + (size_t)m * p.ldc + nb + 8 * lh;
            *(uint4*)o = s0;
            *(uint4*)(o + 16) = s1;
        }
        if (has_stat) {
            const float so = __shfl_xor(ssum, 32, 64), qo = __shfl_xor(qsum, 32, 64);
            if (lh == 0) stpart[((seq & 1) * NW + w) * GS_ROWS + l31] = make_float2(ssum + so, qsum + qo);
        }
        GS_T(3)
    }
#ifdef GS_TIMING
    if (blockIdx.x == 0 && lane == 0 && p.splitk_ws) {
        long long* o = (long long*)p.splitk_ws + w * 8;
        for (int i = 0; i < 4; ++i) o[i] = tt[i];
        o[4] = ntiles_mine;
    }
#endif
    if (has_stat) {   // the last tile's row sums
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (lane < GS_ROWS && (lane % NW) == w) {
            const int seq = ntiles_mine - 1;
            const int mp = (walker + seq * nwalk) * GS_ROWS + lane;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) { const float2 v = stpart[((seq & 1) * NW + i) * GS_ROWS + lane]; s += v.x; q += v.y; }
            if (mp < p.M) ((float2*)p.rowstat_out)[(size_t)(ncol0 / NTILE) * p.M + mp] = make_float2(s, q);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-staged tiles past the end)
}

}  // namespace

// Does the weight-stationary streaming kernel take this problem? (host arithmetic; vk_gemm_bf16 / vk_gemm_rowstat_parts / vk_gemm_tile_choice ask)
// Returns 0 = no, 10 = yes (ten-wave workgroups, 320 output columns each; N = 320, 640 or 960).
extern "C" int vk_gemm_stream_fit(const VkGemmDesc* d) {
    if (!d || d->amode != 0 || d->epi != 0 || d->out_f32 || d->K != GS_K || d->A2 || d->act || d->mx8_out || d->res2 || d->rowvec2 || d->alt_cols_from) return 0;
    if ((d->tile_cfg & 7) != 0 && (d->tile_cfg & 7) != 6) return 0;                 // a forced tiled variant
    // auto: only when every CU gets >= 8 row tiles, and only the form that measured faster than the tiled kernels at the BASELINE shape
    // (profiles/r04_gemm_stream.txt): N = 320 with a residual and neither row-sum emission, row vector nor folded LayerNorm (proj_out: 0.194 vs
    // 0.250 ms = 4.5 vs 3.5 TB/s algorithmic); the other epilogues / N = 960 run level with or 5 % behind the tiled kernels and stay there
    if ((d->tile_cfg & 7) == 0 && (d->M < 32 * 256 * 8 || d->N != 320 || !d->res1 || d->rowstat_out || d->rowvec || d->ln_stats)) return 0;
    if ((d->lda % 8) != 0 || (d->ldc % 8) != 0 || (((size_t)d->out) & 15) != 0 || (((size_t)d->A) & 15) != 0 || d->beta != 0.f) return 0;
    if (d->rowvec && (d->rows_per_vec <= 0 || (d->rows_per_vec % GS_ROWS) != 0)) return 0;
    if (d->ln_stats && (d->ln_parts <= 0 || d->ln_parts > GS_MAXPARTS || !d->ln_colsum)) return 0;
    if (d->N != 320 && d->N != 640 && d->N != 960) return 0;      // one to three column tiles of 320 (ten waves x 32 columns)
    if (d->res1 && (d->N != 320 || (d->ld_res1 % 8) != 0 || (((size_t)d->res1) & 15) != 0)) return 0;
    return 10;
}

extern "C" int vk_gemm_stream_launch(const VkGemmDesc* d, void* stream_) {
    const int fit = vk_gemm_stream_fit(d);
    if (!fit) return VK_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles_n = d->N / 320;
    const int grid = 8 * tiles_n * (32 / tiles_n);   // whole groups of tiles_n workgroups per XCD (256 / 256 / 240)
    hipLaunchKernelGGL((gemm_stream_kernel<10, true>), dim3(grid), dim3(640), 0, stream, *d, tiles_n);
    VK_CHECK_LAUNCH();
    return VK_OK;
}
