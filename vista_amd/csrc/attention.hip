// Attention cores of the VideoUNet on gfx950 (wave64, MFMA 32x32x16 bf16, fp32 softmax/accumulate).
//
// vk_attn_spatial_bf16 -- flash-style spatial self-attention (BasicTransformerBlock.attn1;
//   vwm/modules/attention.py:370-407,514-518): per (image, head), N = H*W tokens, d = 64, no mask.
//   Work decomposition: one workgroup = 512 (S >= 4096: 8 waves x 64 rows, two 32-row query blocks per wave so that every K / V^T fragment
//   read feeds two MFMAs), 256 (S >= 2048: 8 x 32) or 128 (4 x 32) query rows of one (image, head); KV is streamed in 64-key tiles
//   through a 2-stage LDS ring (K tile [64 keys][64 d], V^T tile [64 d][64 keys], 8 KiB each).
//   The score MFMA is issued swapped, S^T = K . Q^T, so a lane owns ONE query column and 32 of the tile's keys:
//   the row max / row sum are lane-local plus a single lane^32 exchange, and the bf16 probabilities are already
//   in MFMA B-operand position for O^T = V^T . P^T. The K tile's rows are stored PERMUTED (row bits [g1 g0 h e1 e0] hold
//   key [g1 h g0 e1 e0]) so that the 8 accumulator registers feeding one PV k-step hold 8 contiguous keys: the matching
//   V^T fragment is a single aligned ds_read_b128. No cross-lane permutes, no P round trip through LDS.
//   V^T ([img][head][64][S]) is produced directly by the value projection GEMM (EPI_TRANS). Both tiles are filled by
//   LDS-DMA (global_load_lds_dwordx4; the row permutation and the bank swizzle live in the per-lane SOURCE address,
//   keys past the sequence end read a zero word), 2-stage ring, one barrier per 64-key tile.
//   LDS swizzle (both tiles, 128-B rows): 16-B chunk ^= (row>>1)&7 -> ds_read_b128 conflict-free.
//   Occupancy: 117 VGPRs, 32 KiB LDS -> TWO workgroups (16 waves) per CU for the 32-row-per-wave forms; the 64-row form uses 200 VGPRs
//   (one 8-wave workgroup per CU, 2 waves per SIMD) and is 4 % faster at S = 9216. What the loop costs is close to the SUM of its MFMA issue
//   time (16 x 32 cycles per wave and tile) and its VALU issue time (~150 instructions, 33 of them v_exp at ~2.7x a plain VALU op):
//   ablations with the softmax or 3/4 of the MFMAs removed, the per-instruction issue rates and three software-pipelined variants
//   (32-key sub-steps with the next scores / this tile's exponentials / the previous PV interleaved in one wave, 3-stage ring) are in
//   profiles/r02_attn_ablation.txt -- those pipelined forms (7 VALU per MFMA, one accumulator chain) were correct and no faster; round 5's
//   attn_spatial_pipe_kernel below (5 fillers per MFMA, alternating accumulators, two workgroups per CU) is, and serves the long sequences
//   made of whole 256-row blocks (levels 0 / 1). For this kernel -- every other length, and the pipelined kernel's fallback -- the levers are
//   instruction counts:
//     * no row-maximum tree on the fast path: exponentials are taken against the EXISTING base and the tile is validated afterwards
//       through its row sums; a tile that fails is redone with a re-base (see "MAX-FREE FAST PATH" in the kernel)
//     * row sums from the bf16-rounded probabilities, two per v_dot2c_f32_bf16
//     * the lane-pair maximum of the slow path through v_permlane32_swap (VALU) instead of ds_bpermute
//     * no s_setprio flips (measured +1.9 % without them)
//   O and l are re-based only when a row maximum grows by more than ~2^6..2^11 (RESCALE_THR: see the fast path's validation).
//
// vk_attn_temporal_bf16 -- per-pixel attention over the T (<=32) frames (VideoTransformerBlock.attn1;
//   vwm/modules/video_attention.py:116-127, attention.py:384-399): one wave per (batch, pixel, head); Q/K
//   fragments are loaded straight from HBM (rows are frames, stride S*ld), V goes through a 4 KiB wave-private
//   LDS tile and is gathered in the accumulator's key order. HBM-bound by construction (12.5 FLOP/B).
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "vista_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 6.0f;  // spatial attention: re-base the online softmax only for max growth > 2^6

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ bf16x8_t make_frag(uint2 a, uint2 b) {
    uint4 v = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ uint4 g_attn_zero16;

// MX-fp8 quantisation of one 32-element block held as four accumulator quads across the lane pair (l, l ^ 32): see mx_quant_block in gemm_common.h
__device__ __forceinline__ uint4 mx_quant_block_attn(const float (&v)[4][4], int& e8m0) {
    float amax = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[g][e]));
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    int ex = -127;
    if (amax > 0.f) frexpf(amax * (1.f / 448.f), &ex);
    ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
    const float inv = ldexpf(1.f, -ex);
    uint32_t q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][0] * inv, v[g][1] * inv, 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[g][2] * inv, v[g][3] * inv, w, true);
        q[g] = (uint32_t)w;
    }
    const auto r02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto r13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    e8m0 = ex + 127;
    return make_uint4(r02[0], r02[1], r13[0], r13[1]);
}  // zero source for keys past the end of the sequence (LDS-DMA cannot predicate its write)

// actual key (within a 32-key subtile) held by K-tile LDS row rho: bits [g1 g0 h e1 e0] -> [g1 h g0 e1 e0].
// With this row permutation the accumulator registers 8*(J&1)..+7 of lane-half h hold the 8 CONTIGUOUS keys 16J + 8h + 0..7,
// so the matching V^T fragment is one aligned 16-byte LDS read.
__device__ __forceinline__ int key_of_row(int rho32) {
    const int g = rho32 >> 3, h = (rho32 >> 2) & 1, e = rho32 & 3;
    return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + e;
}

// max over the two lanes (l, l^32) that share a query column, without the LDS round trip of ds_bpermute (v_permlane32_swap is VALU)
__device__ __forceinline__ float pair_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, v), __builtin_bit_cast(uint32_t, v), false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}

typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_ones(uint32_t packed, float acc) {  // acc + lo(packed) + hi(packed), both bf16
    const bf16x2v_t one = {(__bf16)1.0f, (__bf16)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, packed), one, acc, false);
}

// NW waves per workgroup, QW 32-row query blocks per wave: NW*QW*32 query rows share each 64-key K / V^T tile. QW = 2 (long sequences)
// feeds every K / V^T fragment read to TWO MFMAs: the K-loop of these kernels is LDS-throughput-bound (profiles/r02_ab_notes.txt A/B 2),
// and the price -- twice the accumulators, 2 waves per SIMD instead of 4 -- is paid in latency hiding the loop does not depend on.
// VROW: V is the token-major [S][ldv] column block of the fused q|k|v projection (no V^T tensor, no TRANS GEMM): its 64-key tile lands in LDS
// as [key][64 d] rows and the PV MFMA's A operand (V^T fragment: row d, 8 consecutive keys) is read with ds_read_b64_tr_b16 -- per 16-lane
// group a 4x16 -> 16x4 transpose of the 64 addressed bf16 (tools/probes/tr16_probe.hip), two reads per fragment. The lanes of a group
// address 4 consecutive keys x 16 d; the 16-byte chunks of a row are swizzled by key bit 1 (swap of the 64-byte halves) so that the 4 rows
// x 64 bytes a half-wave touches cover all 64 banks (tools/probes/tr16_layout_probe.hip: same rate as the b128 reads of the V^T image).
// PRE: q arrives multiplied by (softmax scale x log2 e) -- folded into the query projection's weights at pack time, where it costs no extra
// rounding -- so a score IS the base-2 exponent, and rows whose running maximum lies within +-60 octaves use the base ZERO: the probability
// is v_exp_f32 of the accumulator itself, no scale/base fma (64 of the ~224 VALU instructions per 64-key tile and wave of a kernel that
// is bound by its VALU, not by the matrix core: 64 quarter-rate exponentials = 1024 cycles already equal the tile's 32 MFMAs). Any base is
// exact in exact arithmetic; base 0 is safe in fp32 / bf16 because the first tile's maximum >= -60 bounds the row sum away from zero and an
// exponent above 127 overflows to +inf, fails the row-sum test and re-bases the row on its true maximum (general path from then on).
// The whole kernel as a device function over the workgroup's LDS (2 x 16 KiB) and its logical block index, so that the pipelined kernel below
// (attn_spatial_pipe_kernel) can fall back to it for a workgroup whose rows leave the zero-base range.
template <int NW, int QW, bool VROW, bool PRE>
__device__ __forceinline__ void attn_spatial_body(char* const smem, const int logical, const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                  const uint16_t* __restrict__ vt, uint16_t* __restrict__ o, int n_img, int heads, int S, int ldq,
                                                  int ldk, int ldo, float scale_log2, float rescale_thr, int ldv) {
    constexpr int QB = NW * QW * 32;  // query rows per workgroup
    constexpr int GPW = 8 / NW;       // 8-row DMA groups of each tile handled per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    const int nqb = (S + QB - 1) / QB;
    const int bh = logical / nqb, qb = logical - bh * nqb;
    const int img = bh / heads, head = bh - img * heads;

    // ---- Q fragments (B operand: column = query row, 8 consecutive d at 16*ks + 8*lh) ----
    int qrow[QW];
    bool q_ok[QW];
    bf16x8_t qf[QW][4];
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        qrow[b] = qb * QB + (wave * QW + b) * 32 + l31;
        q_ok[b] = qrow[b] < S;
        if (!q_ok[b]) qrow[b] = S - 1;
        const uint16_t* qptr = q + ((size_t)img * S + qrow[b]) * ldq + head * 64 + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const bf16x8_t*)(qptr + 16 * ks);
    }

    // ---- K / V^T staging by LDS-DMA: wave w fills the 8-row groups w, w+NW, .. of both tiles; lane -> row (lane>>3) of the
    //      group, physical 16-B slot (lane&7); the swizzle (slot = chunk ^ ((row>>1)&7)) is applied on the SOURCE chunk ----
    const int dj = lane >> 3, dslot = lane & 7;
    int k_key[GPW], d_row[GPW], src_chunk[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int rho = 8 * (wave + NW * i) + dj;                // LDS row of both tiles handled by this lane
        src_chunk[i] = dslot ^ ((rho >> 1) & 7);
        k_key[i] = (rho & 32) + key_of_row(rho & 31);            // key (within the 64-key tile) stored in K row rho
        d_row[i] = rho;                                           // V^T row = head-dim index d
    }
    const uint16_t* kbase = k + (size_t)img * S * ldk + head * 64;
    const uint16_t* vbase = VROW ? vt + (size_t)img * S * ldv + head * 64 : vt + ((size_t)(img * heads + head) * 64) * S;
    int v_chunk[GPW];  // VROW: logical 16-byte chunk of V row (= key) rho fetched into physical slot dslot
#pragma unroll
    for (int i = 0; i < GPW; ++i) v_chunk[i] = dslot ^ (((d_row[i] >> 1) & 1) << 2);

    // Per-lane byte offsets inside a tile are loop-invariant; the tile itself moves the UNIFORM base (scalar adds), so a full tile costs
    // no per-lane address arithmetic. Only a ragged LAST tile needs the per-key bounds test (its missing keys read the zero word).
    uint32_t k_off[GPW], v_off[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        k_off[i] = (uint32_t)(k_key[i] * ldk + src_chunk[i] * 8) * 2u;
        v_off[i] = VROW ? (uint32_t)(d_row[i] * ldv + v_chunk[i] * 8) * 2u : (uint32_t)(d_row[i] * S + src_chunk[i] * 8) * 2u;
    }
    const bool ragged = (S & 63) != 0;
    const int n_tiles = (S + 63) >> 6;
    auto dma_tile = [&](int t, int stage) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int key0 = t * 64;
        char* sK = smem + stage * 16384 + wave_u * 1024;
        char* sV = sK + 8192;
        if (NW == 8 && !(ragged && t == n_tiles - 1)) {  // (the 4-wave kernel of the short sequences has no registers to spare for it)
            const char* kt = (const char*)kbase + (size_t)key0 * ldk * 2;  // uniform
            const char* vtile = (const char*)vbase + (VROW ? (size_t)key0 * ldv * 2 : (size_t)key0 * 2);
#pragma unroll
            for (int i = 0; i < GPW; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)(kt + k_off[i]), (lptr_t)(sK + i * NW * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(vtile + v_off[i]), (lptr_t)(sV + i * NW * 1024), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int key = key0 + k_key[i];
            const uint16_t* ksrc = (key < S) ? kbase + (size_t)key * ldk + src_chunk[i] * 8 : (const uint16_t*)&g_attn_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)ksrc, (lptr_t)(sK + i * NW * 1024), 16, 0, 0);
            const int kk = VROW ? key0 + d_row[i] : key0 + src_chunk[i] * 8;  // V row = key | first key of this 8-key chunk (S % 8 == 0)
            const uint16_t* vsrc = (kk >= S) ? (const uint16_t*)&g_attn_zero16
                                             : (VROW ? vbase + (size_t)kk * ldv + v_chunk[i] * 8 : vbase + (size_t)d_row[i] * S + kk);
            __builtin_amdgcn_global_load_lds((gptr_t)vsrc, (lptr_t)(sV + i * NW * 1024), 16, 0, 0);
        }
    };

    // fragment read offsets: row l31 (+32 per subtile), logical chunk 2*ks + lh, same swizzle for K and V^T tiles
    const int fsw = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) frag_off[ks] = l31 * 128 + (((ks * 2 + lh) ^ fsw) << 4);

    // VROW fragment reads: lane (l31, lh) of 16-lane group g1 = (l31 >> 4) & 1 addresses key 16J + 8lh + 4hf + kq (kq = (l31 & 15) >> 2), the 4 d at
    // 32dd + 16 g1 + 4c (c = l31 & 3); key bit 1 = kq bit 1 swaps the row's 64-byte halves, i.e. flips dd. One base per dd, immediates for (J, hf).
    int vtr_base[2];
    {
        const int kq = (l31 & 15) >> 2, g1 = (l31 >> 4) & 1, c = l31 & 3, sw1 = (kq >> 1) & 1;
        const int b0 = (8 * lh + kq) * 128 + (2 * g1 + (c >> 1)) * 16 + (c & 1) * 8;
        vtr_base[0] = b0 + (0 ^ sw1) * 64;
        vtr_base[1] = b0 + (1 ^ sw1) * 64;
    }
    auto v_frag = [&](const char* sV, int d, int J) -> bf16x8_t {
        if constexpr (VROW) {
            typedef short s4_t __attribute__((ext_vector_type(4)));
            typedef s4_t __attribute__((address_space(3))) * lds_s4_t;
            const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sV + vtr_base[d] + J * 2048));
            const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sV + vtr_base[d] + J * 2048 + 512));
            bf16x8_t f;
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
            f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            return f;
        } else {
            return *(const bf16x8_t*)(sV + d * 32 * 128 + frag_off[J]);
        }
    };

    f32x16_t oacc[QW][2];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        m_run[b] = NEG_BIG;
        l_run[b] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[b][d][r] = 0.f;
    }

    const int nt = n_tiles;
    const float psum_limit = PRE ? 0x1p100f : fast_exp2(fminf(rescale_thr, 100.f) + 5.f);  // finite whatever the tuning value: an overflowed row sum (+inf) must fail the test
    bool zero_base = false;  // PRE: every row of this wave uses base 0 (wave-uniform)
    dma_tile(0, 0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int stage = t & 1;
        if (t + 1 < nt) dma_tile(t + 1, stage ^ 1);
        const char* sK = smem + stage * 16384;
        const char* sV = sK + 8192;

        f32x16_t sacc[QW][2];
        bf16x8_t pf[QW][4];
        float psum[QW];
        // ---- S^T[key][q] = K . Q^T : every K fragment feeds the QW query blocks of the wave ----
        auto scores = [&]() {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int b = 0; b < QW; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[b][c][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(sK + c * 32 * 128 + frag_off[ks]);
#pragma unroll
                    for (int b = 0; b < QW; ++b) sacc[b][c] = vk_mfma(kf, qf[b][ks], sacc[b][c]);
                }
            }
            if (t == nt - 1 && (S & 63)) {  // mask keys past the end of the sequence (last tile only)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // accumulator row rho32 = (r&3) + 8*(r>>2) + 4*lh holds key key_of_row(rho32) of subtile c
                        const int key = t * 64 + c * 32 + key_of_row((r & 3) + 8 * (r >> 2) + 4 * lh);
                        if (key >= S) {
#pragma unroll
                            for (int b = 0; b < QW; ++b) sacc[b][c][r] = NEG_BIG;
                        }
                    }
            }
        };
        // ---- probabilities under the CURRENT base m_run: the lane's 32 scores all belong to query column l31. P^T fragments: k-step
        //      J = accumulator regs 8*(J&1)..+7 of key subtile J>>1 = keys 16J + 8*lh + 0..7. The row sum is taken from the ROUNDED
        //      probabilities, two per v_dot2c_f32_bf16 (half the adds, and l matches what the PV MFMAs accumulate) ----
        auto probabilities = [&](auto zb_tag) __attribute__((always_inline)) {
            constexpr bool ZB = decltype(zb_tag)::value;
#pragma unroll
            for (int b = 0; b < QW; ++b) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[b][c][r] = fast_exp2(ZB ? sacc[b][c][r] : (PRE ? sacc[b][c][r] - m_run[b] : fmaf(sacc[b][c][r], scale_log2, -m_run[b])));
                psum[b] = 0.f;
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    const int c = J >> 1, r0 = 8 * (J & 1);
                    uint4 v;
                    v.x = pack_bf16x(sacc[b][c][r0 + 0], sacc[b][c][r0 + 1]);
                    v.y = pack_bf16x(sacc[b][c][r0 + 2], sacc[b][c][r0 + 3]);
                    v.z = pack_bf16x(sacc[b][c][r0 + 4], sacc[b][c][r0 + 5]);
                    v.w = pack_bf16x(sacc[b][c][r0 + 6], sacc[b][c][r0 + 7]);
                    psum[b] = dot2_ones(v.x, psum[b]);
                    psum[b] = dot2_ones(v.y, psum[b]);
                    psum[b] = dot2_ones(v.z, psum[b]);
                    psum[b] = dot2_ones(v.w, psum[b]);
                    pf[b][J] = __builtin_bit_cast(bf16x8_t, v);
                }
            }
        };
        // ---- re-base of the online softmax on this tile's row maxima: m_run <- max(m_run, max_keys s), O and l scaled to the new base ----
        auto rebase = [&]() {
#pragma unroll
            for (int b = 0; b < QW; ++b) {
                float mx = sacc[b][0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[b][0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][1][r]);
                mx = fmaxf(PRE ? pair_max(mx) : pair_max(mx) * scale_log2, NEG_BIG);
                float m_new = fmaxf(m_run[b], mx);
                if (PRE && fabsf(m_new) <= 60.f) m_new = 0.f;
                const float alpha = fast_exp2(m_run[b] - m_new);
                m_run[b] = m_new;
                l_run[b] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][d][r] *= alpha;
            }
            if constexpr (PRE) {
                bool z = m_run[0] == 0.f;
#pragma unroll
                for (int b = 1; b < QW; ++b) z = z && (m_run[b] == 0.f);
                zero_base = __all(z);
            }
        };
        auto probs = [&]() __attribute__((always_inline)) {
            if constexpr (PRE) {
                if (zero_base) probabilities(std::true_type{});
                else probabilities(std::false_type{});
            } else {
                probabilities(std::false_type{});
            }
        };

        // MAX-FREE FAST PATH. The loop costs close to the SUM of its VALU and MFMA issue time (profiles/r02_attn_ablation.txt), and the
        // row-maximum tree is ~30 of its ~150 VALU instructions per tile -- for a result that changes nothing on almost every tile:
        // after the first tiles a row's maximum hardly ever grows by more than a few octaves. So the exponentials are taken against
        // the EXISTING base without looking at the maximum, and the tile is validated AFTERWARDS through the row sum it needs anyway:
        // if some lane's 32 probabilities sum to more than 2^(thr+5) (one of them exceeded the old maximum by > 2^thr..2^(thr+5);
        // +inf / NaN from an overflowing exp2 fail the <= test too), the tile is redone the slow way -- scores recomputed from the
        // K tile still in LDS, re-base on their maxima, probabilities again. fp32 exp2 cannot overflow before the score exceeds the
        // base by 128 octaves, bf16 P and the fp32 accumulators keep their RELATIVE precision whatever the common factor, and the
        // final division by l removes it. Tile 0 always takes the slow path (it defines the first base).
        scores();
        if (__builtin_expect(t == 0, 0)) rebase();
        probs();
        bool bad = !(psum[0] <= psum_limit);
#pragma unroll
        for (int b = 1; b < QW; ++b) bad = bad || !(psum[b] <= psum_limit);
        if (__builtin_expect(__any(bad), 0)) {  // wave-uniform, rare
            asm volatile("" ::: "memory");  // re-read the fragments: keeping the fast path's copies alive for this branch costs spills
            scores();
            rebase();
            probabilities(std::false_type{});  // (the general form is right for any base, zero included)
        }
#pragma unroll
        for (int b = 0; b < QW; ++b) l_run[b] += psum[b];

        // ---- O^T[d][q] += V^T . P^T : V^T fragment (row d = 32*dd + l31, keys 16J + 8*lh ..+7) is one ds_read_b128, shared by the
        //      wave's query blocks ----
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const bf16x8_t vf = v_frag(sV, d, J);
#pragma unroll
                for (int b = 0; b < QW; ++b) oacc[b][d] = vk_mfma_bf16x(vf, pf[b][J], oacc[b][d]);
            }
        }
        __syncthreads();  // retires the DMA of tile t+1 (vmcnt(0)) and frees this stage
    }

#pragma unroll
    for (int b = 0; b < QW; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv = 1.f / l_tot;
        if (q_ok[b]) {
            uint16_t* optr = o + ((size_t)img * S + qrow[b]) * ldo + head * 64 + 4 * lh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w;
                    w.x = pack_bf16(oacc[b][d][4 * g + 0] * inv, oacc[b][d][4 * g + 1] * inv);
                    w.y = pack_bf16(oacc[b][d][4 * g + 2] * inv, oacc[b][d][4 * g + 3] * inv);
                    *(uint2*)(optr + 32 * d + 8 * g) = w;
                }
        }
    }
}

template <int NW, int QW, bool VROW, bool PRE>
__global__ __launch_bounds__(NW * 64, QW == 2 ? 2 : 4) void attn_spatial_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                                const uint16_t* __restrict__ vt, uint16_t* __restrict__ o,
                                                                                int n_img, int heads, int S, int ldq, int ldk, int ldo,
                                                                                float scale_log2, float rescale_thr, int ldv) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];  // per stage: K tile 8 KiB | V^T tile 8 KiB
    const int nqb = (S + NW * QW * 32 - 1) / (NW * QW * 32);
    attn_spatial_body<NW, QW, VROW, PRE>(smem, xcd_remap(blockIdx.x, nqb * n_img * heads), q, k, vt, o, n_img, heads, S, ldq, ldk, ldo, scale_log2,
                                         rescale_thr, ldv);
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: software-pipelined form of the zero-base kernel (PRE, VROW) for long sequences, S a multiple of the workgroup's NW x 64 query rows.
// The un-pipelined loop above costs the SUM of its matrix and its VALU issue time (scores -> barrier-free but serial: 16 MFMAs, then ~160 VALU,
// then 16 MFMAs per wave and tile). Here the work is cut into UNITS of (32 keys x 32 queries) -- 4 score MFMAs chained on one accumulator,
// 16 v_exp + 8 v_cvt_pk + 16 v_add, 4 PV MFMAs -- and unit n's VALU work is issued in the gaps between the score MFMAs of unit n + 1 and the
// PV MFMAs of unit n - 1: one MFMA, then 2 exp + 1 cvt + 2 add (the five single-issue fillers that fit one 32-cycle MFMA of a wave that owns
// its SIMD, MI355X_MICROARCH.md "Per-instruction cycle constants"), then the next MFMA on ANOTHER accumulator (score k-steps alternate with
// the two PV accumulators: an MFMA never waits on its predecessor). tools/probes/attn_issue_probe.hip measures exactly this stream.
//   * four units per 64-key tile: (key block c, query block b) = (0,0) (0,1) (1,0) (1,1); two score accumulators and two packed-P buffers
//     alternate with the unit parity. The K fragments of a key block serve two consecutive units (b = 1 of one key block pair ... b = 0 of the
//     next) and so do the V^T fragments one unit later: each fragment register is RELOADED with the next key block's fragment right after the
//     MFMA that used it last (ds_read_b128 after the even MFMAs, two ds_read_b64_tr_b16 after the odd ones, in the b = 0 units), eight gaps
//     before its next use -- single-buffered fragments, 32 registers instead of 64, at most two LDS reads per gap;
//   * no row maxima and no per-tile validation at all: every row runs on base 0 (a score is the base-2 exponent, the query carries
//     scale x log2 e) and the kernel checks ONCE, at the end, that each row's sum is finite and within 2^+-100; a workgroup that fails the
//     test (scores beyond +-100 octaves: never on real activations) recomputes its rows with attn_spatial_body, the general online softmax;
//   * row sums are plain f32 adds of the un-rounded probabilities (v_dot2c on the packed pair costs ~10 cycles beside MFMAs, same guide table);
//   * LDS rings, ONE barrier per tile (B_t, after the first MFMA of tile t's third unit): K(t)'s last fragment read is in the first unit of
//     tile t and K(t + 1)'s first one right after B_t -> K ring of two 8 KiB stages, the DMA of K(t + 2) goes out at B_t; V(t) is read in the
//     first and third unit of tile t, i.e. still after B_t -> V ring of THREE stages, the DMA of V(t + 2) goes out at B_t into V(t - 1)'s;
//     every DMA has a whole tile to land and is retired by the vmcnt(0) of the next barrier. 40 KiB of LDS;
//   * 256-register budget (launch bound 2 waves per SIMD): the score accumulators must be arch VGPRs for v_exp to read them -- with the
//     512-register budget of a one-wave-per-SIMD declaration the compiler parks MFMA results in AGPRs and pays a v_accvgpr_read per score.
//     NW = 4: four waves, one per SIMD when one workgroup runs per CU, or two unsynchronised workgroups per CU; NW = 8: two waves per SIMD.
template <int NW, bool ONE_PER_CU>
__global__ __launch_bounds__(NW * 64, 2) void attn_spatial_pipe_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                        const uint16_t* __restrict__ v, uint16_t* __restrict__ o, int n_img,
                                                                        int heads, int S, int ldq, int ldk, int ldo, int ldv) {
    constexpr int QB = NW * 64;   // query rows per workgroup: two 32-row blocks per wave
    constexpr int GPW = 8 / NW;   // 8-row DMA groups of each tile handled per wave
    // K ring: 2 x 8 KiB at 0; V ring ([key][64 d] rows): 3 x 8 KiB at 16 KiB. ONE_PER_CU: the declaration is padded past half of the CU's
    // 160 KiB so that a four-wave workgroup has its SIMDs to itself (the padding is never touched)
    constexpr int V_RING = 16384;
    __shared__ __attribute__((aligned(16))) char smem[40960 + (ONE_PER_CU ? 48 * 1024 : 0)];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef short s4_t __attribute__((ext_vector_type(4)));
    typedef s4_t __attribute__((address_space(3))) * lds_s4_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nqb = S / QB;
    const int logical = xcd_remap(blockIdx.x, nqb * n_img * heads);
    const int bh = logical / nqb, qb = logical - bh * nqb;
    const int img = bh / heads, head = bh - img * heads;

    // ---- Q fragments (B operand of the score MFMA: column = query row, 8 consecutive d at 16 ks + 8 lh) ----
    bf16x8_t qf[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qrow = qb * QB + (wave * 2 + b) * 32 + l31;
        const uint16_t* qptr = q + ((size_t)img * S + qrow) * ldq + head * 64 + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const bf16x8_t*)(qptr + 16 * ks);
    }

    // ---- K / V staging by LDS-DMA, images exactly as attn_spatial_body<.., VROW = true> lays them out: two per-lane source offsets live
    //      across the loop, everything else of a piece's address is uniform ----
    const char* const kbase = (const char*)(k + (size_t)img * S * ldk + head * 64);
    const char* const vbase = (const char*)(v + (size_t)img * S * ldv + head * 64);
    uint32_t k_off0, v_off0;   // group 0's per-lane source offsets; group i (NW = 4: i = 0, 1) is 32 keys further: a uniform + 32 rows
    {
        const int rho = 8 * wave + (lane >> 3), dslot = lane & 7;                   // LDS row of both tiles handled by this lane (group 0)
        const int key = (rho & 32) + key_of_row(rho & 31);                          // key stored in K row rho (permuted: see key_of_row)
        k_off0 = (uint32_t)(key * ldk + (dslot ^ ((rho >> 1) & 7)) * 8) * 2u;        // swizzle on the SOURCE chunk
        v_off0 = (uint32_t)(rho * ldv + (dslot ^ (((rho >> 1) & 1) << 2)) * 8) * 2u; // V row rho = key rho: 64-byte halves swapped by key bit 1
    }
    // piece q of tile t's 2 * GPW pieces of this wave: q even = K group q / 2, q odd = V group q / 2
    auto dma_piece = [&](const int t, const int kst, const int vst, const int q) __attribute__((always_inline)) {
        const int i = q >> 1;   // (NW * i = 4 i: rows rho + 32 i -> keys + 32 i under both images' layouts)
        if ((q & 1) == 0) {
            const char* kt = kbase + (size_t)t * 64 * ldk * 2;   // uniform
            __builtin_amdgcn_global_load_lds((gptr_t)(kt + (size_t)(32 * i) * ldk * 2 + k_off0), (lptr_t)(smem + kst * 8192 + wave_u * 1024 + i * NW * 1024), 16, 0, 0);
        } else {
            const char* vt = vbase + (size_t)t * 64 * ldv * 2;
            __builtin_amdgcn_global_load_lds((gptr_t)(vt + (size_t)(32 * i) * ldv * 2 + v_off0), (lptr_t)(smem + V_RING + vst * 8192 + wave_u * 1024 + i * NW * 1024), 16, 0, 0);
        }
    };
    auto dma_tile = [&](const int t, const int kst, const int vst) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2 * GPW; ++q) dma_piece(t, kst, vst, q);
    };

    // fragment read offsets (as attn_spatial_body): K row l31 (+ 32 per key block), logical chunk (2 ks + lh) ^ fsw under the row swizzle = the
    // ks = 0 offset with bits 5-6 XORed by ks: ONE address register + a v_xor per read instead of four (registers are what this loop is short
    // of); V through the transposing read, whose two d-half bases differ by the 64-byte half swap = XOR 64 likewise
    const int fsw = (l31 >> 1) & 7;
    const int kfrag0 = l31 * 128 + ((lh ^ fsw) << 4);
    int vtr0;
    {
        const int kq = (l31 & 15) >> 2, g1 = (l31 >> 4) & 1, c = l31 & 3, sw1 = (kq >> 1) & 1;
        vtr0 = (8 * lh + kq) * 128 + (2 * g1 + (c >> 1)) * 16 + (c & 1) * 8 + sw1 * 64;
    }

    // ---- pipeline state ----
    f32x16_t oacc[2][2];          // [query block b][d half]
    f32x16_t sc[2];               // score accumulators: unit parity
    uint32_t pw[2][8];            // packed probabilities of a unit (two k-slot fragments): unit parity
    bf16x8_t kf[4];               // K fragments of the key block the coming score MFMAs use, [k-step]
    bf16x8_t vf[4];               // V^T fragments of the key block the coming PV MFMAs use, [2 * d half + k-slot within the block]
    float ps[2][2];               // row-sum partials [query block][chain]
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        ps[b][0] = 0.f; ps[b][1] = 0.f;
        oacc[b][0] = zero16; oacc[b][1] = zero16;
    }
    auto read_k = [&](const char* sK, const int c, const int ks) __attribute__((always_inline)) {
        kf[ks] = *(const bf16x8_t*)(sK + c * 32 * 128 + (kfrag0 ^ (ks << 5)));
    };
    // V fragment f = 2 * d + j of key block c: row d half d, keys 32 c + 16 j + 8 lh ... -- two transposing 8-byte reads
    auto read_v = [&](const char* sV, const int c, const int f) __attribute__((always_inline)) {
        const int d = f >> 1, J = 2 * c + (f & 1);
        const int vb = vtr0 ^ (d << 6);
        const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sV + vb + J * 2048));
        const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sV + vb + J * 2048 + 512));
        bf16x8_t r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        vf[f] = r;
    };
    auto p_frag = [&](const uint32_t* w4) __attribute__((always_inline)) -> bf16x8_t {
        const uint4 u = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        return __builtin_bit_cast(bf16x8_t, u);
    };

    const int nt = S >> 6;
    // ---- prologue: tiles 0 and 1 in flight, K(0, block 0) fragments, the first unit's scores. The loop body is uniform: the first unit's
    //      "previous unit" is a zero probability block (pw[1] = 0 against finite V fragments adds nothing), and the last tile's reads of a
    //      tile t + 1 hit a stale (finite or not: never consumed) stage ----
    dma_tile(0, 0, 0);
    if (nt > 1) dma_tile(1, 1, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) pw[1][i] = 0u;
#pragma unroll
    for (int f = 0; f < 4; ++f) vf[f] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) read_k(smem, 0, ks);
    sc[0] = vk_mfma(kf[0], qf[0][0], zero16);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) sc[0] = vk_mfma(kf[ks], qf[0][ks], sc[0]);
    __builtin_amdgcn_sched_barrier(0);

    // One unit slot. SL = 0..3 = (key block c = SL >> 1, query block b = SL & 1) of tile t; kst / vst = ring stages of tile t (K: t & 1, V: t % 3).
    //   VALU : softmax of sc[SL & 1] -> pw[SL & 1], ps[b]
    //   MFMA : even gaps -- scores of the NEXT unit into sc[(SL & 1) ^ 1] from kf; odd gaps -- PV of the PREVIOUS unit from pw[(SL & 1) ^ 1] and vf
    //   LDS  : b = 0 units reload each fragment right after its last use: SL 0: kf <- K(t, block 1), vf <- V(t, block 0);
    //                                                                  SL 2: kf <- K(t + 1, block 0), vf <- V(t, block 1)
    //   SL 2 : the tile's barrier B_t after the first MFMA (K(t + 1), V(t + 1) landed; K(t), V(t - 1) dead), then the DMA of tile t + 2
    auto slot = [&](auto sl_tag, const int t, const int kst, const int vst) __attribute__((always_inline)) {
        constexpr int SL = decltype(sl_tag)::value;
        constexpr int b = SL & 1, par = SL & 1;
        constexpr int nb = b ^ 1, pb = b ^ 1;                   // query block of the next / previous unit
        const char* const sK_cur = smem + kst * 8192;
        const char* const sK_nxt = smem + (kst ^ 1) * 8192;
        const char* const sV_cur = smem + V_RING + vst * 8192;
        f32x16_t& cur = sc[par];
        f32x16_t& nxt = sc[par ^ 1];
        uint32_t* const pcur = pw[par];
        const uint32_t* const pprev = pw[par ^ 1];
        float ea = 0.f, eb = 0.f;   // the exponentials of score pair g, taken ONE gap ahead of their conversion and row-sum adds
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if ((g & 1) == 0) {
                const int ks = g >> 1;
                nxt = vk_mfma(kf[ks], qf[nb][ks], ks == 0 ? zero16 : nxt);
                if (SL == 2 && g == 0) __syncthreads();   // B_t. vmcnt(0): this wave's pieces of tile t + 1 have landed; lgkmcnt(0): no fragment read is pending
                // the DMA of tile t + 2 behind B_t, all pieces at once: K(t + 2) over K(t); V(t + 2) over V(t - 1): stage (t + 2) % 3 = (t - 1) % 3.
                // (One piece per even gap of this unit instead measured -25 %: 6.28 vs 4.71 ms at level 0, profiles/r05_attn_pipe.txt.)
                if (SL == 2 && g == 0 && t + 2 < nt) dma_tile(t + 2, kst, vst == 0 ? 2 : vst - 1);
                if (SL == 0) read_k(sK_cur, 1, ks);
                if (SL == 2) read_k(sK_nxt, 0, ks);
            } else {
                const int d = (g >> 1) & 1, j = g >> 2, f = 2 * d + j;
                oacc[pb][d] = vk_mfma_bf16x(vf[f], p_frag(pprev + 4 * j), oacc[pb][d]);
                if (SL == 0) read_v(sV_cur, 0, f);
                if (SL == 2) read_v(sV_cur, 1, f);
            }
            {   // the fillers of this gap: exponentials of pair g + 1 (gap 0: of pair 0 too), conversion and row-sum adds of pair g. A
                // transcendental's result may not be read by the next instruction; one gap of distance keeps every use clear of that
                // hazard without the s_nop the compiler otherwise puts between each v_exp pair and its v_cvt_pk (20 per tile).
                if (g == 0) { ea = fast_exp2(cur[0]); eb = fast_exp2(cur[1]); }
                float na = 0.f, nb2 = 0.f;
                if (g < 7) { na = fast_exp2(cur[2 * g + 2]); nb2 = fast_exp2(cur[2 * g + 3]); }
                uint32_t w = pack_bf16x(ea, eb);
                asm volatile("" : "+v"(w));   // pinned HERE: the optimiser otherwise sinks the conversions to their use, a whole unit later
                pcur[g] = w;
                ps[b][0] += ea;   // (single v_add_f32: the file is built with -fno-slp-vectorize -- packed into v_pk_add_f32 the sixteen adds of
                ps[b][1] += eb;   //  a unit are gathered at its end, keep sixteen exponentials live, and cost more beside MFMAs, same guide table)
                asm volatile("" : "+v"(ps[b][0]), "+v"(ps[b][1]));   // pinned like the conversions: left alone the adds sink into the next basic block in a clump
                ea = na; eb = nb2;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        int vst = 0;
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            slot(std::integral_constant<int, 0>{}, t, t & 1, vst);
            slot(std::integral_constant<int, 1>{}, t, t & 1, vst);
            slot(std::integral_constant<int, 2>{}, t, t & 1, vst);
            slot(std::integral_constant<int, 3>{}, t, t & 1, vst);
            vst = vst == 2 ? 0 : vst + 1;
        }
    }
    // PV of the very last unit (key block 1, query block 1)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[1][d] = vk_mfma_bf16x(vf[2 * d + j], p_frag(pw[1] + 4 * j), oacc[1][d]);

    // ---- the one validity test of the zero-base run: finite row sums inside 2^+-100 ----
    float inv[2];
    bool bad = false;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float l = ps[b][0] + ps[b][1];
        const float l_tot = l + __shfl_xor(l, 32, 64);
        bad = bad || !(l_tot >= 0x1p-100f && l_tot <= 0x1p100f);
        inv[b] = 1.f / l_tot;
    }
    if (__syncthreads_or(bad)) {   // (also a barrier: every wave is done with the LDS ring) -- workgroup-uniform, never taken on real activations
#ifndef VK_ATTN_NO_FALLBACK   // (-DVK_ATTN_NO_FALLBACK: an A/B build without the inlined general kernel -- the only code of this kernel that uses scratch;
                              //  timing only, tools/build_variant.sh; profiles/r06_attn_scratch_ab.txt)
        attn_spatial_body<NW, 2, true, true>(smem, logical, q, k, v, o, n_img, heads, S, ldq, ldk, ldo, 1.0f, RESCALE_THR, ldv);
#endif
        return;
    }
    // Output: lane (l31, lh) holds d = 32 dd + 8 g + 4 lh + e of row l31. Swapping the upper half-wave's quad g with the lower half-wave's quad
    // g + 1 (v_permlane32_swap) leaves the lower lane with the 8 contiguous d of quad g and the upper lane with those of quad g + 1: 16-byte
    // stores, half as many (the store tail of an attention workgroup is issue-bound: MI355X_MICROARCH.md, cycle constants). ldo % 8 == 0 and a
    // 16-byte aligned `o` are the launcher's conditions for this kernel.
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qrow = qb * QB + (wave * 2 + b) * 32 + l31;
        uint16_t* optr = o + ((size_t)img * S + qrow) * ldo + head * 64 + 8 * lh;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint2 qa, qb2;
                qa.x = pack_bf16(oacc[b][d][8 * gp + 0] * inv[b], oacc[b][d][8 * gp + 1] * inv[b]);
                qa.y = pack_bf16(oacc[b][d][8 * gp + 2] * inv[b], oacc[b][d][8 * gp + 3] * inv[b]);
                qb2.x = pack_bf16(oacc[b][d][8 * gp + 4] * inv[b], oacc[b][d][8 * gp + 5] * inv[b]);
                qb2.y = pack_bf16(oacc[b][d][8 * gp + 6] * inv[b], oacc[b][d][8 * gp + 7] * inv[b]);
                const auto rx = __builtin_amdgcn_permlane32_swap(qa.x, qb2.x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(qa.y, qb2.y, false, false);
                *(uint4*)(optr + 32 * d + 16 * gp) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
    }
}

// WIDE (default): 16-byte output stores (quad g of the upper half-wave swapped with quad g + 1 of the lower one, as the spatial kernels); needs
// ldo % 8 == 0 and a 16-byte aligned `o`. WSYNC (measured, NOT the default): the V tile is wave-private, so the two workgroup barriers of an
// iteration can be replaced by wave-local ordering (LDS executes a wave's instructions in order; the fences only pin the compiler).
// VISTA_ATTN_T (launcher) selects: bit 0 = WSYNC, bit 1 = WIDE; bitwise the same results. Same box, alternated processes, ms per launch at
// B = 2, T = 25 (tools/attn_t_ab.py, profiles/r05_attn_temporal_ab.txt):        level 0 (S 9216, 5 heads)   level 1 (2304, 10)   level 2 (576, 20)
//   0  barriers, 8-byte stores (rounds 1-4 of this repository)                  0.3152  3.74 TB/s            0.1586               0.0856
//   1  wave-local sync, 8-byte stores                                            0.3387-0.3799               0.1707               0.0936   (slower: the barriers keep a
//   2  barriers, 16-byte stores  <- default                                      0.2830  4.17 TB/s            0.1441               0.0796    workgroup's four problems -- four
//   3  wave-local sync, 16-byte stores                                           0.2858                       0.1456               0.0811    heads of one pixel -- in step)
template <bool WSYNC, bool WIDE>
__global__ __launch_bounds__(256) void attn_temporal_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o, int B, int T,
                                                            int S, int heads, int ld, int k_off, int v_off, int ldo,
                                                            float scale_log2, long long nprob, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 4096];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    char* sV = smem + wave * 4096;
    const bool t_ok = l31 < T;

    for (int it = 0; it < iters; ++it) {
        long long prob = ((long long)it * gridDim.x + blockIdx.x) * 4 + wave;
        const bool active = prob < nprob;
        if (!active) prob = nprob - 1;
        const int head = (int)(prob % heads);
        const long long bs = prob / heads;
        const int s = (int)(bs % S);
        const int b = (int)(bs / S);
        const size_t row0 = (size_t)b * T * S + s;  // row of frame t = row0 + t*S

        // Q (B operand) and K (A operand) fragments straight from HBM: row = frame l31, 8 d at 16ks + 8lh
        bf16x8_t qf[4], kf[4];
        {
            const uint16_t* rp = qkv + (row0 + (size_t)(t_ok ? l31 : 0) * S) * ld + head * 64 + 8 * lh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 a = *(const uint4*)(rp + 16 * ks);
                uint4 c = *(const uint4*)(rp + k_off + 16 * ks);
                if (!t_ok) { a = make_uint4(0, 0, 0, 0); c = make_uint4(0, 0, 0, 0); }
                qf[ks] = __builtin_bit_cast(bf16x8_t, a);
                kf[ks] = __builtin_bit_cast(bf16x8_t, c);
            }
        }
        // V tile [32 frames][64 d] -> wave-private LDS (rows >= T zero)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cidx = lane + 64 * i;
            const int vr = cidx >> 3, vc = cidx & 7;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (vr < T) v = *(const uint4*)(qkv + (row0 + (size_t)vr * S) * ld + v_off + head * 64 + vc * 8);
            *(uint4*)(sV + vr * 128 + vc * 16) = v;
        }

        f32x16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sacc = vk_mfma(kf[ks], qf[ks], sacc);

        float mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= T) sacc[r] = NEG_BIG;
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mb = mx * scale_log2;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float pv = fast_exp2(fmaf(sacc[r], scale_log2, -mb));
            if (key >= T) pv = 0.f;
            sacc[r] = pv;
            psum += pv;
        }
        psum += __shfl_xor(psum, 32, 64);
        const float inv = 1.f / psum;

        bf16x8_t pf[2];
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            uint4 v;
            v.x = pack_bf16x(sacc[8 * J + 0], sacc[8 * J + 1]);
            v.y = pack_bf16x(sacc[8 * J + 2], sacc[8 * J + 3]);
            v.z = pack_bf16x(sacc[8 * J + 4], sacc[8 * J + 5]);
            v.w = pack_bf16x(sacc[8 * J + 6], sacc[8 * J + 7]);
            pf[J] = __builtin_bit_cast(bf16x8_t, v);
        }

        if (WSYNC) {   // V tile visible to the other lanes of THIS wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();  // V tile visible (uniform trip count: every wave reaches this)
        }

        f32x16_t oacc[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                // V^T fragment in accumulator key order: element i <-> frame 16J + 8*(i>>2) + 4*lh + (i&3)
                uint32_t w[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const int ka = 16 * J + 8 * ((2 * i2) >> 2) + 4 * lh + ((2 * i2) & 3);
                    const uint16_t lo = *(const uint16_t*)(sV + ka * 128 + (32 * d + l31) * 2);
                    const uint16_t hi = *(const uint16_t*)(sV + (ka + 1) * 128 + (32 * d + l31) * 2);
                    w[i2] = (uint32_t)lo | ((uint32_t)hi << 16);
                }
                const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
                oacc[d] = vk_mfma_bf16x(__builtin_bit_cast(bf16x8_t, v), pf[J], oacc[d]);
            }
        }
        if (WIDE) {   // (the swaps run in every lane; only the stores are predicated)
            uint16_t* optr = o + (row0 + (size_t)(t_ok ? l31 : 0) * S) * ldo + head * 64 + 8 * lh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint2 qa, qb;
                    qa.x = pack_bf16(oacc[d][8 * gp + 0] * inv, oacc[d][8 * gp + 1] * inv);
                    qa.y = pack_bf16(oacc[d][8 * gp + 2] * inv, oacc[d][8 * gp + 3] * inv);
                    qb.x = pack_bf16(oacc[d][8 * gp + 4] * inv, oacc[d][8 * gp + 5] * inv);
                    qb.y = pack_bf16(oacc[d][8 * gp + 6] * inv, oacc[d][8 * gp + 7] * inv);
                    const auto rx = __builtin_amdgcn_permlane32_swap(qa.x, qb.x, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(qa.y, qb.y, false, false);
                    if (active && t_ok) *(uint4*)(optr + 32 * d + 16 * gp) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                }
        } else if (active && t_ok) {
            uint16_t* optr = o + (row0 + (size_t)l31 * S) * ldo + head * 64 + 4 * lh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w2;
                    w2.x = pack_bf16(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
                    w2.y = pack_bf16(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
                    *(uint2*)(optr + 32 * d + 8 * g) = w2;
                }
        }
        if (WSYNC) {   // this wave's reads of its tile are done before its next writes (in-order LDS; compiler pinned)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();  // LDS tile free for the next problem
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Row softmax for the VAE decoder's single-head 512-wide AttnBlock (vwm/modules/diffusionmodules/model.py:135-182):
// scores arrive fp32 (already scaled by the GEMM epilogue), probabilities leave as bf16 rows for the P.V GEMM.
// One 256-thread workgroup per row; a row of up to 16384 columns stays in registers between the max / sum / write passes.
namespace {
constexpr int SM_MAXV = 16;  // float4 chunks per thread
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int cols, long long ldx,
                                                           long long ldy) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float4* xr = (const float4*)(x + (size_t)row * ldx);
    const int nv = cols >> 2;
    float4 v[SM_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = tid + i * 256;
        if (c < nv) {
            v[i] = xr[c];
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float L2E = 1.4426950408889634f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = tid + i * 256;
        if (c < nv) {
            v[i].x = __builtin_amdgcn_exp2f((v[i].x - mx) * L2E);
            v[i].y = __builtin_amdgcn_exp2f((v[i].y - mx) * L2E);
            v[i].z = __builtin_amdgcn_exp2f((v[i].z - mx) * L2E);
            v[i].w = __builtin_amdgcn_exp2f((v[i].w - mx) * L2E);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
    uint2* yr = (uint2*)(y + (size_t)row * ldy);
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = tid + i * 256;
        if (c < nv) {
            uint2 o;
            o.x = pack_bf16(v[i].x * inv, v[i].y * inv);
            o.y = pack_bf16(v[i].z * inv, v[i].w * inv);
            yr[c] = o;
        }
    }
}
}  // namespace

extern "C" int vk_softmax_rows_f32_bf16(const float* x, void* y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy, void* stream_) {
    if (!x || !y || rows <= 0 || rows > 0x7fffffffLL || cols <= 0 || (cols & 3) || cols > SM_MAXV * 256 * 4 || (ldx & 3) || (ldy & 3) ||
        ldx < cols || ldy < cols)
        return VK_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream_, x, (uint16_t*)y, cols, (long long)ldx,
                       (long long)ldy);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// BASELINE config 5: spatial self-attention with the score MFMA in fp8 (vk_attn_spatial_fp8qk). Q and K arrive as e4m3 bytes with one E8M0
// scale per row and 32 head-dim elements -- written by the fused q|k|v projection's epilogue (VkGemmDesc.mx8_out: no quantisation pass) --
// and S^T = K . Q^T of a 32-key x 32-query block is ONE v_mfma_scale_f32_32x32x64_f8f6f4 (d = 64 = its K depth) with both operands' block
// scales applied inside the instruction, instead of four bf16 32x32x16 MFMAs: a quarter of the score instructions at half their total time,
// and a K tile of 4 KiB instead of 8. Softmax, bf16 P, and the PV product on the bf16 V rows are the bf16 kernel's (same fragments, same
// key permutation); the output may leave as MX fp8 as well (o8 / os: the attention-out projection then runs as an fp8 GEMM with a_mx).
// Operand layout of the scaled MFMA (probed in round 2, tools/probes/mx_scale_probe.hip): lane l holds row l & 31; VGPRs 0-3 = bytes
// [16 lh, 16 lh + 16) of the row's first 32-byte K-block, VGPRs 4-7 = the same half of the second block; the E8M0 scale of (row r, block b)
// is byte op_sel of the scale VGPR of lane r + 32 b, i.e. lane half lh supplies block lh.
namespace {
typedef __attribute__((ext_vector_type(8))) int i32x8a_t;

// PRE: as in attn_spatial_kernel -- the query carries scale x log2 e, zero-base rows take v_exp_f32 of the score itself (512-row form only).
template <int NW, int QW, bool PRE>
__global__ __launch_bounds__(NW * 64, QW == 2 ? 2 : 4) void attn_spatial_fp8qk_kernel(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                                                      const uint8_t* __restrict__ qs, const uint8_t* __restrict__ ks,
                                                                                      const uint16_t* __restrict__ v, uint16_t* __restrict__ o,
                                                                                      uint8_t* __restrict__ o8, uint8_t* __restrict__ os, int n_img, int heads,
                                                                                      int S, int ldq8, int ldk8, int ldqs, int ldks, int ldv, int ldo, int ldo8,
                                                                                      int ldos, float scale_log2, float rescale_thr) {
    constexpr int QB = NW * QW * 32;
    constexpr int GPW = 8 / NW;  // 8-row DMA groups of the V tile handled per wave
    __shared__ __attribute__((aligned(16))) char smem[2 * 12288];  // per stage: K8 tile [64 keys][64 B] 4 KiB | V tile [64 keys][64 d] bf16 8 KiB
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nqb = (S + QB - 1) / QB;
    const int logical = xcd_remap(blockIdx.x, nqb * n_img * heads);
    const int bh = logical / nqb, qb = logical - bh * nqb;
    const int img = bh / heads, head = bh - img * heads;

    // ---- Q fragments (B operand) and their block scales ----
    int qrow[QW];
    bool q_ok[QW];
    i32x8a_t qf[QW];
    int qsc[QW];
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        qrow[b] = qb * QB + (wave * QW + b) * 32 + l31;
        q_ok[b] = qrow[b] < S;
        if (!q_ok[b]) qrow[b] = S - 1;
        const size_t row = (size_t)img * S + qrow[b];
        const uint8_t* qp = q8 + row * ldq8 + head * 64 + 16 * lh;
        const uint4 lo = *(const uint4*)qp, hi = *(const uint4*)(qp + 32);
        qf[b][0] = lo.x; qf[b][1] = lo.y; qf[b][2] = lo.z; qf[b][3] = lo.w;
        qf[b][4] = hi.x; qf[b][5] = hi.y; qf[b][6] = hi.z; qf[b][7] = hi.w;
        qsc[b] = qs[row * ldqs + 2 * head + lh];
    }

    // ---- staging: K8 tile by waves 0-3 (16 rows x 64 B per LDS-DMA instruction), V tile as in the bf16 kernel (VROW form) ----
    const int krho = 16 * (wave & 3) + (lane >> 2);                       // K8 LDS row filled by this lane
    const int kchunk = (lane & 3) ^ ((krho >> 2) & 3);                    // logical 16-byte chunk it fetches (slot = chunk ^ ((row >> 2) & 3))
    const int kkey = (krho & 32) + key_of_row(krho & 31);                 // key (within the tile) stored in that row
    const uint32_t k_off = (uint32_t)(kkey * ldk8 + kchunk * 16);
    const uint8_t* kbase = k8 + (size_t)img * S * ldk8 + head * 64;
    const int dj = lane >> 3, dslot = lane & 7;
    int v_row[GPW], v_chunk[GPW];
    uint32_t v_off[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        v_row[i] = 8 * (wave + NW * i) + dj;
        v_chunk[i] = dslot ^ (((v_row[i] >> 1) & 1) << 2);
        v_off[i] = (uint32_t)(v_row[i] * ldv + v_chunk[i] * 8) * 2u;
    }
    const uint16_t* vbase = v + (size_t)img * S * ldv + head * 64;
    const uint8_t* ksb = ks + (size_t)img * S * ldks + 2 * head + lh;     // + key * ldks: this lane half's block scale of a key
    const bool ragged = (S & 63) != 0;
    const int nt = (S + 63) >> 6;
    auto dma_tile = [&](int t, int stage) __attribute__((always_inline)) {
        const int key0 = t * 64;
        char* sK = smem + stage * 12288;
        char* sV = sK + 4096 + wave_u * 1024;
        const bool last_ragged = ragged && t == nt - 1;
        if (NW == 4 || wave_u < 4) {
            const uint8_t* src = (!last_ragged || key0 + kkey < S) ? kbase + (size_t)key0 * ldk8 + k_off : (const uint8_t*)&g_attn_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sK + (wave_u & 3) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const uint16_t* src = (!last_ragged || key0 + v_row[i] < S) ? (const uint16_t*)((const char*)vbase + (size_t)key0 * ldv * 2 + v_off[i])
                                                                         : (const uint16_t*)&g_attn_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sV + i * NW * 1024), 16, 0, 0);
        }
    };
    // K block scales of tile t for this lane: key of accumulator-row l31 in each 32-key subtile (clamped past the end: those scores are masked)
    auto k_scales = [&](int t, int (&out)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            int key = t * 64 + c * 32 + key_of_row(l31);
            key = key < S ? key : S - 1;
            out[c] = ksb[(size_t)key * ldks];
        }
    };

    // fragment read offsets: K8 row l31 (+32 per subtile), chunks lh and 2 + lh under the 64-byte-row swizzle
    const int ksw = (l31 >> 2) & 3;
    const int kf_lo = l31 * 64 + ((lh ^ ksw) << 4), kf_hi = l31 * 64 + (((2 + lh) ^ ksw) << 4);
    int vtr_base[2];
    {
        const int kq = (l31 & 15) >> 2, g1 = (l31 >> 4) & 1, c = l31 & 3, sw1 = (kq >> 1) & 1;
        const int b0 = (8 * lh + kq) * 128 + (2 * g1 + (c >> 1)) * 16 + (c & 1) * 8;
        vtr_base[0] = b0 + (0 ^ sw1) * 64;
        vtr_base[1] = b0 + (1 ^ sw1) * 64;
    }

    f32x16_t oacc[QW][2];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        m_run[b] = NEG_BIG;
        l_run[b] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[b][d][r] = 0.f;
    }
    const float psum_limit = PRE ? 0x1p100f : fast_exp2(fminf(rescale_thr, 100.f) + 5.f);
    bool zero_base = false;
    int ksc[2], ksc_next[2];
    dma_tile(0, 0);
    k_scales(0, ksc);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int stage = t & 1;
        if (t + 1 < nt) { dma_tile(t + 1, stage ^ 1); k_scales(t + 1, ksc_next); }
        const char* sK = smem + stage * 12288;
        const char* sV = sK + 4096;

        f32x16_t sacc[QW][2];
        bf16x8_t pf[QW][4];
        float psum[QW];
        auto scores = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint4 lo = *(const uint4*)(sK + c * 32 * 64 + kf_lo), hi = *(const uint4*)(sK + c * 32 * 64 + kf_hi);
                i32x8a_t kf;
                kf[0] = lo.x; kf[1] = lo.y; kf[2] = lo.z; kf[3] = lo.w; kf[4] = hi.x; kf[5] = hi.y; kf[6] = hi.z; kf[7] = hi.w;
#pragma unroll
                for (int b = 0; b < QW; ++b) {
                    f32x16_t z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    sacc[b][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[b], z, 0, 0, 0, ksc[c], 0, qsc[b]);
                }
            }
            if (t == nt - 1 && (S & 63)) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + c * 32 + key_of_row((r & 3) + 8 * (r >> 2) + 4 * lh);
                        if (key >= S) {
#pragma unroll
                            for (int b = 0; b < QW; ++b) sacc[b][c][r] = NEG_BIG;
                        }
                    }
            }
        };
        auto probabilities = [&](auto zb_tag) __attribute__((always_inline)) {
            constexpr bool ZB = decltype(zb_tag)::value;
#pragma unroll
            for (int b = 0; b < QW; ++b) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[b][c][r] = fast_exp2(ZB ? sacc[b][c][r] : (PRE ? sacc[b][c][r] - m_run[b] : fmaf(sacc[b][c][r], scale_log2, -m_run[b])));
                psum[b] = 0.f;
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    const int c = J >> 1, r0 = 8 * (J & 1);
                    uint4 w;
                    w.x = pack_bf16x(sacc[b][c][r0 + 0], sacc[b][c][r0 + 1]);
                    w.y = pack_bf16x(sacc[b][c][r0 + 2], sacc[b][c][r0 + 3]);
                    w.z = pack_bf16x(sacc[b][c][r0 + 4], sacc[b][c][r0 + 5]);
                    w.w = pack_bf16x(sacc[b][c][r0 + 6], sacc[b][c][r0 + 7]);
                    psum[b] = dot2_ones(w.x, psum[b]);
                    psum[b] = dot2_ones(w.y, psum[b]);
                    psum[b] = dot2_ones(w.z, psum[b]);
                    psum[b] = dot2_ones(w.w, psum[b]);
                    pf[b][J] = __builtin_bit_cast(bf16x8_t, w);
                }
            }
        };
        auto rebase = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < QW; ++b) {
                float mx = sacc[b][0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[b][0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][1][r]);
                mx = fmaxf(PRE ? pair_max(mx) : pair_max(mx) * scale_log2, NEG_BIG);
                float m_new = fmaxf(m_run[b], mx);
                if (PRE && fabsf(m_new) <= 60.f) m_new = 0.f;
                const float alpha = fast_exp2(m_run[b] - m_new);
                m_run[b] = m_new;
                l_run[b] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[b][d][r] *= alpha;
            }
            if constexpr (PRE) {
                bool z = m_run[0] == 0.f;
#pragma unroll
                for (int b = 1; b < QW; ++b) z = z && (m_run[b] == 0.f);
                zero_base = __all(z);
            }
        };
        // the bf16 kernel's max-free fast path (see attn_spatial_kernel): exponentials against the existing base, validated by the row sums
        scores();
        if (__builtin_expect(t == 0, 0)) rebase();
        if constexpr (PRE) {
            if (zero_base) probabilities(std::true_type{});
            else probabilities(std::false_type{});
        } else {
            probabilities(std::false_type{});
        }
        bool bad = !(psum[0] <= psum_limit);
#pragma unroll
        for (int b = 1; b < QW; ++b) bad = bad || !(psum[b] <= psum_limit);
        if (__builtin_expect(__any(bad), 0)) {
            asm volatile("" ::: "memory");
            scores();
            rebase();
            probabilities(std::false_type{});
        }
#pragma unroll
        for (int b = 0; b < QW; ++b) l_run[b] += psum[b];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                typedef short s4_t __attribute__((ext_vector_type(4)));
                typedef s4_t __attribute__((address_space(3))) * lds_s4_t;
                const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sV + vtr_base[d] + J * 2048));
                const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sV + vtr_base[d] + J * 2048 + 512));
                bf16x8_t vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3]; vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
                for (int b = 0; b < QW; ++b) oacc[b][d] = vk_mfma_bf16x(vf, pf[b][J], oacc[b][d]);
            }
        }
        __syncthreads();
        ksc[0] = ksc_next[0];
        ksc[1] = ksc_next[1];
    }

#pragma unroll
    for (int b = 0; b < QW; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv = 1.f / l_tot;
        const size_t row = (size_t)img * S + qrow[b];
        if (o8 != nullptr) {  // MX fp8 output: one E8M0 scale per row and 32 head-dim elements (= one accumulator set d)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                float vq[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) vq[g][e] = oacc[b][d][4 * g + e] * inv;
                int e8;
                const uint4 q16 = mx_quant_block_attn(vq, e8);
                if (q_ok[b]) {
                    *(uint4*)(o8 + row * ldo8 + head * 64 + 32 * d + 16 * lh) = q16;
                    if (lh == 0) {
                        os[row * ldos + 2 * head + d] = (uint8_t)e8;
                        if (d == 1 && head == heads - 1)  // pad bytes of the scale row (ldos is a multiple of 4 for the consumer GEMM's dword reads):
                            for (int j = 2 * heads; j < ldos; ++j) os[row * ldos + j] = 127;  // 2^0 -- an uninitialised 0xFF would be an E8M0 NaN
                    }
                }
            }
        } else if (q_ok[b]) {
            uint16_t* optr = o + row * ldo + head * 64 + 4 * lh;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w;
                    w.x = pack_bf16(oacc[b][d][4 * g + 0] * inv, oacc[b][d][4 * g + 1] * inv);
                    w.y = pack_bf16(oacc[b][d][4 * g + 2] * inv, oacc[b][d][4 * g + 3] * inv);
                    *(uint2*)(optr + 32 * d + 8 * g) = w;
                }
        }
    }
}
}  // namespace

extern "C" int vk_attn_spatial_fp8qk(const void* q8, const void* k8, const void* q_scales, const void* k_scales, const void* v, void* o, void* o8,
                                     void* o_scales, int32_t n_img, int32_t heads, int32_t S, int32_t ldq8, int32_t ldk8, int32_t ldqs, int32_t ldks,
                                     int32_t ldv, int32_t ldo, int32_t ldo8, int32_t ldos, float scale, void* stream_) {
#if VK_F16
    return VK_EINVAL;   // BASELINE config 5 (fp8) exists in the bf16 build only (include/vista_hip.h: vk_act_dtype)
#endif
    if (!q8 || !k8 || !q_scales || !k_scales || !v || (!o && !o8) || (o8 && !o_scales) || n_img <= 0 || heads <= 0 || S <= 0) return VK_EINVAL;
    if ((S % 8) != 0 || (ldq8 % 16) != 0 || (ldk8 % 16) != 0 || (ldv % 8) != 0 || (o && (ldo % 4) != 0) || (o8 && (ldo8 % 16) != 0)) return VK_EINVAL;
    if ((((size_t)q8) & 15) || (((size_t)k8) & 15) || (o8 && (((size_t)o8) & 15))) return VK_EINVAL;
    static const float thr = [] { const char* e = getenv("VISTA_ATTN_RESCALE_THR"); return e ? (float)atof(e) : RESCALE_THR; }();
    const int cls = S >= 4096 ? 2 : (S >= 2048 ? 1 : 0);
    const int qb_rows = cls == 2 ? 512 : (cls == 1 ? 256 : 128);
    const long long nblk = (long long)((S + qb_rows - 1) / qb_rows) * n_img * heads;
    if (nblk > 0x7fffffffLL) return VK_EINVAL;
    const bool pre = scale == 0.f;  // scale == 0: the query already carries softmax_scale * log2(e) (see vk_attn_spatial_qkv_log2_bf16)
    if (pre) scale = 1.f / LOG2E;
#define FP8QK_LAUNCH(NW, QW, PR)                                                                                                                \
    hipLaunchKernelGGL((attn_spatial_fp8qk_kernel<NW, QW, PR>), dim3((unsigned)nblk), dim3(NW * 64), 0, (hipStream_t)stream_, (const uint8_t*)q8,   \
                       (const uint8_t*)k8, (const uint8_t*)q_scales, (const uint8_t*)k_scales, (const uint16_t*)v, (uint16_t*)o, (uint8_t*)o8,     \
                       (uint8_t*)o_scales, n_img, heads, S, ldq8, ldk8, ldqs, ldks, ldv, ldo, ldo8, ldos, scale * LOG2E, thr)
    if (cls == 2 && pre) FP8QK_LAUNCH(8, 2, true);
    else if (cls == 2) FP8QK_LAUNCH(8, 2, false);
    else if (cls == 1) FP8QK_LAUNCH(8, 1, false);
    else FP8QK_LAUNCH(4, 1, false);
#undef FP8QK_LAUNCH
    VK_CHECK_LAUNCH();
    return VK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Small dense attention for the conditioner's OpenCLIP ViT-H/14 image tower (vwm/modules/encoders/modules.py:251-399 -> open_clip
// VisionTransformer resblocks: nn.MultiheadAttention over 257 tokens, 16 heads of dim 80): softmax(q k^T * scale) v per (image, head),
// any head dim D <= 128 that is a multiple of 8, S up to a few hundred tokens. It runs once per sampling window on a handful of images
// (0.1 % of a window's FLOPs), so it is a plain fp32 VALU kernel: one thread per query row, the head's K and V slices staged once per
// workgroup in LDS as bf16 (broadcast reads), online softmax in registers.
namespace {
template <int D>
__global__ __launch_bounds__(64) void attn_small_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o, int heads, int S, int ld,
                                                        int k_off, int v_off, int ldo, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];  // K [S][D] bf16 | V [S][D] bf16
    uint16_t* sK = (uint16_t*)dsm;
    uint16_t* sV = sK + (size_t)S * D;
    const int bh = blockIdx.x, img = bh / heads, head = bh - img * heads;
    const uint16_t* base = qkv + (size_t)img * S * ld + head * D;
    constexpr int CH = D / 8;  // 16-byte chunks per row
    for (int i = threadIdx.x; i < S * CH; i += 64) {
        const int r = i / CH, c = i - r * CH;
        *(uint4*)(sK + (size_t)r * D + c * 8) = *(const uint4*)(base + (size_t)r * ld + k_off + c * 8);
        *(uint4*)(sV + (size_t)r * D + c * 8) = *(const uint4*)(base + (size_t)r * ld + v_off + c * 8);
    }
    __syncthreads();
    const int s = blockIdx.y * 64 + threadIdx.x;
    if (s >= S) return;
    float q[D], acc[D];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float f[8];
        unpack8(*(const uint4*)(base + (size_t)s * ld + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { q[8 * c + e] = f[e] * scale_log2; acc[8 * c + e] = 0.f; }
    }
    float m = NEG_BIG, l = 0.f;
    for (int j = 0; j < S; ++j) {
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float f[8];
            unpack8(*(const uint4*)(sK + (size_t)j * D + c * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sc = fmaf(q[8 * c + e], f[e], sc);
        }
        const float m_new = fmaxf(m, sc);
        const float alpha = fast_exp2(m - m_new), pj = fast_exp2(sc - m_new);
        m = m_new;
        l = fmaf(l, alpha, pj);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float f[8];
            unpack8(*(const uint4*)(sV + (size_t)j * D + c * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[8 * c + e] = fmaf(acc[8 * c + e], alpha, pj * f[e]);
        }
    }
    const float inv = 1.f / l;
    uint16_t* op = o + ((size_t)img * S + s) * ldo + head * D;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = acc[8 * c + e] * inv;
        *(uint4*)(op + c * 8) = pack8(f);
    }
}
}  // namespace

extern "C" int vk_attn_small_bf16(const void* qkv, void* o, int32_t n_img, int32_t heads, int32_t S, int32_t D, int32_t ld, int32_t k_off,
                                  int32_t v_off, int32_t ldo, float scale, void* stream_) {
    if (!qkv || !o || n_img <= 0 || heads <= 0 || S <= 0 || (ld % 8) != 0 || (k_off % 8) != 0 || (v_off % 8) != 0 || (ldo % 8) != 0) return VK_EINVAL;
    const size_t lds = (size_t)S * D * 2 * 2;
    if (lds > 160 * 1024) return VK_EINVAL;
    const dim3 grid((unsigned)(n_img * heads), (unsigned)((S + 63) / 64));
#define SMALL_LAUNCH(DD)                                                                                                                      \
    do {                                                                                                                                       \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attn_small_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return VK_ELAUNCH;                                                                                                                 \
        hipLaunchKernelGGL((attn_small_kernel<DD>), grid, dim3(64), lds, (hipStream_t)stream_, (const uint16_t*)qkv, (uint16_t*)o, heads, S, ld,  \
                           k_off, v_off, ldo, scale * LOG2E);                                                                                  \
    } while (0)
    if (D == 80) SMALL_LAUNCH(80);
    else if (D == 64) SMALL_LAUNCH(64);
    else if (D == 128) SMALL_LAUNCH(128);
    else return VK_EINVAL;
#undef SMALL_LAUNCH
    VK_CHECK_LAUNCH();
    return VK_OK;
}

static int attn_spatial_launch(const void* q, const void* k, const void* vt, void* o, int32_t n_img, int32_t heads,
                               int32_t S, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream_, bool pre = false) {
    if (!q || !k || !vt || !o || n_img <= 0 || heads <= 0 || S <= 0) return VK_EINVAL;
    if ((S % 8) != 0 || (ldq % 8) != 0 || (ldk % 8) != 0 || (ldo % 4) != 0) return VK_EINVAL;
    // long sequences: 256 query rows (8 waves) per workgroup halve the K/V^T stream per FLOP; short ones keep 128 rows so the
    // ragged last q-block wastes less (S = 144, 576 at the deep levels)
    static const float thr = [] { const char* e = getenv("VISTA_ATTN_RESCALE_THR"); return e ? (float)atof(e) : RESCALE_THR; }();  // tuning / A-B
    // long sequences: 8 waves x 64 query rows (512 per workgroup): every K / V^T fragment read feeds two MFMAs and the K/V^T stream
    // per FLOP halves again; medium ones 8 x 32; short ones 4 x 32 so the ragged last q-block wastes less (S = 144, 576 at the deep levels)
    static const int qw_env = [] { const char* e = getenv("VISTA_ATTN_QW"); return e ? atoi(e) : 0; }();  // tuning / A-B: 1 or 2
    int cls = S >= 4096 ? (qw_env == 1 ? 1 : 2) : (S >= 2048 ? (qw_env == 2 ? 2 : 1) : 0);
    if (cls == 2 && qw_env == 0) {
        // few images (one rank of a frame-sharded run): the 512-row workgroups run ONE per CU, so e.g. 7 images x 5 heads x 18 blocks = 630 of
        // them are 2.46 rounds = 3; 256-row workgroups (two per CU) finish the same work 8.5 % sooner (0.88 -> 0.805 ms), while at 8 / 13 / 50
        // images (2.8 / 4.6 / 17.6 rounds) the 512-row form stays 1-4 % ahead (profiles/r03_rank_proxy.txt)
        const long long n2 = (long long)((S + 511) / 512) * n_img * heads;
        const long long rounds = (n2 + 255) / 256;
        if (n2 * 100 < rounds * 256 * 85) cls = 1;
    }
    const int qb_rows = cls == 2 ? 512 : (cls == 1 ? 256 : 128);
    const int nqb = (S + qb_rows - 1) / qb_rows;
    const long long nblk = (long long)nqb * n_img * heads;
    if (nblk > 0x7fffffffLL) return VK_EINVAL;
#define ATTN_LAUNCH(NW, QW, VR, PR)                                                                                                             \
    hipLaunchKernelGGL((attn_spatial_kernel<NW, QW, VR, PR>), dim3((unsigned)nblk), dim3(NW * 64), 0, (hipStream_t)stream_, (const uint16_t*)q,   \
                       (const uint16_t*)k, (const uint16_t*)vt, (uint16_t*)o, n_img, heads, S, ldq, ldk, ldo, scale * LOG2E, thr, ldv)
    if (pre && ldv <= 0) return VK_EINVAL;  // the pre-scaled form exists for the q | k | v column-block layout only
    if (pre) scale = 1.f / LOG2E;           // a score is already the base-2 exponent: scale_log2 = 1 for the general kernels below
    // Round 5: the software-pipelined zero-base kernel for long sequences made of whole workgroups of query rows (level 0: S = 9216).
    // VISTA_ATTN_PIPE: 0 = off (A/B hook), 1 = four waves x 64 rows, two workgroups per CU, 2 = four waves, one workgroup per CU (one wave per
    // SIMD), 3 = eight waves (two per SIMD in one workgroup).
    static const int pipe_mode = [] { const char* e = getenv("VISTA_ATTN_PIPE"); return e ? atoi(e) : 1; }();
    if (pre && pipe_mode > 0 && S >= 2048 && (ldo % 8) == 0 && (((size_t)o) & 15) == 0) {
        const int rows = pipe_mode == 3 ? 512 : 256;
        if (S % rows == 0) {
            const long long nb = (long long)(S / rows) * n_img * heads;
            if (nb > 0x7fffffffLL) return VK_EINVAL;
#define ATTN_PIPE_LAUNCH(NW, ONE)                                                                                                             \
    hipLaunchKernelGGL((attn_spatial_pipe_kernel<NW, ONE>), dim3((unsigned)nb), dim3(NW * 64), 0, (hipStream_t)stream_, (const uint16_t*)q,    \
                       (const uint16_t*)k, (const uint16_t*)vt, (uint16_t*)o, n_img, heads, S, ldq, ldk, ldo, ldv)
            if (pipe_mode == 3) ATTN_PIPE_LAUNCH(8, false);
            else if (pipe_mode == 2) ATTN_PIPE_LAUNCH(4, true);
            else ATTN_PIPE_LAUNCH(4, false);
#undef ATTN_PIPE_LAUNCH
            VK_CHECK_LAUNCH();
            return VK_OK;
        }
    }
    if (pre && cls == 2) {
        // zero-base kernel for the 512-row form only (241 VGPRs of its 256): 5.12 -> 4.95 ms per level-0 launch (1.06 -> 1.10 PFLOP/s). The
        // 256- / 128-row forms live at 128 VGPRs for four waves per SIMD; the second probability path spills there (0.69 -> 0.81 ms at
        // S = 2304), so they run the general kernel on the pre-scaled query (profiles/r03_attn_zero_base.txt).
        ATTN_LAUNCH(8, 2, true, true);
    } else if (ldv > 0) {
        if (cls == 2) ATTN_LAUNCH(8, 2, true, false);
        else if (cls == 1) ATTN_LAUNCH(8, 1, true, false);
        else ATTN_LAUNCH(4, 1, true, false);
    } else {
        if (cls == 2) ATTN_LAUNCH(8, 2, false, false);
        else if (cls == 1) ATTN_LAUNCH(8, 1, false, false);
        else ATTN_LAUNCH(4, 1, false, false);
    }
#undef ATTN_LAUNCH
    VK_CHECK_LAUNCH();
    return VK_OK;
}

extern "C" int vk_attn_spatial_bf16(const void* q, const void* k, const void* vt, void* o, int32_t n_img, int32_t heads,
                                    int32_t S, int32_t ldq, int32_t ldk, int32_t ldo, float scale, void* stream_) {
    return attn_spatial_launch(q, k, vt, o, n_img, heads, S, ldq, ldk, 0, ldo, scale, stream_);
}

extern "C" int vk_attn_spatial_qkv_bf16(const void* q, const void* k, const void* v, void* o, int32_t n_img, int32_t heads,
                                        int32_t S, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream_) {
    if (ldv <= 0 || (ldv % 8) != 0) return VK_EINVAL;
    return attn_spatial_launch(q, k, v, o, n_img, heads, S, ldq, ldk, ldv, ldo, scale, stream_);
}

extern "C" int vk_attn_spatial_qkv_log2_bf16(const void* q, const void* k, const void* v, void* o, int32_t n_img, int32_t heads,
                                             int32_t S, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, void* stream_) {
    if (ldv <= 0 || (ldv % 8) != 0) return VK_EINVAL;
    return attn_spatial_launch(q, k, v, o, n_img, heads, S, ldq, ldk, ldv, ldo, 1.0f, stream_, true);
}

extern "C" int vk_attn_temporal_bf16(const void* qkv, void* o, int32_t B, int32_t T, int32_t S, int32_t heads, int32_t ld,
                                     int32_t k_off, int32_t v_off, int32_t ldo, float scale, void* stream_) {
    if (!qkv || !o || B <= 0 || T <= 0 || T > 32 || S <= 0 || heads <= 0) return VK_EINVAL;
    if ((ld % 8) != 0 || (k_off % 8) != 0 || (v_off % 8) != 0 || (ldo % 4) != 0) return VK_EINVAL;
    const long long nprob = (long long)B * S * heads;
    const long long want = (nprob + 3) / 4;
    const long long cap = 256LL * 16;  // 16 workgroups per CU keeps plenty of loads in flight
    const int grid = (int)(want < cap ? want : cap);
    const int iters = (int)((want + grid - 1) / grid);
    static const int mode_env = [] { const char* e = getenv("VISTA_ATTN_T"); return e ? atoi(e) : 2; }();   // A/B hook: bit 0 = wave-local sync, bit 1 = 16-byte stores (default: measured below)
    int mode = mode_env & 3;
    if ((ldo % 8) != 0 || (((size_t)o) & 15) != 0) mode &= 1;   // 16-byte stores need 16-byte aligned rows
#define VK_ATTN_T(WS, WD) hipLaunchKernelGGL((attn_temporal_kernel<WS, WD>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const uint16_t*)qkv, \
                                             (uint16_t*)o, B, T, S, heads, ld, k_off, v_off, ldo, scale * LOG2E, nprob, iters)
    if (mode == 3) VK_ATTN_T(true, true);
    else if (mode == 2) VK_ATTN_T(false, true);
    else if (mode == 1) VK_ATTN_T(true, false);
    else VK_ATTN_T(false, false);
#undef VK_ATTN_T
    VK_CHECK_LAUNCH();
    return VK_OK;
}
