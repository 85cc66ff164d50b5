// fp8 (OCP e4m3) dense GEMM for the UNet's 1x1 / Linear projections: BASELINE config 5 ("fp8 1x1 conv-as-GEMM path").
//
//   out[m][n] = epilogue( a_scale[m] * w_scale[n] * sum_k Aq[m][k] * Wq[n][k] )
//
// Aq: activations quantised per ROW (token) by vk_quantize_rows_fp8, Wq: weights quantised per OUTPUT CHANNEL at pack time.
// Same structure as gemm.hip (LDS-DMA 2-stage pipeline, 128-byte swizzled LDS rows, swapped MFMA orientation, shared fused
// epilogue) with 128 fp8 K-elements per row and v_mfma_scale_f32_32x32x64_f8f6f4 (scale operands 0 = unscaled fp8 x fp8):
// per K-step a wave issues the same number of MFMA cycles as the bf16 kernel for twice the FLOPs, and stages half the bytes
// per FLOP through the LDS-DMA path that bounds the bf16 kernel.
// Operand layout of the 32x32x64 instruction (probed, tools/probes/fp8_mfma_probe.hip): lane l holds row l&31 and the 32
// consecutive k-bytes [32*(l>>5), +32) in 8 VGPRs; C/D layout is the bf16 32x32 one.
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ uint4 g_zero16_f8;

struct F8Args {
    const float* a_scale;  // [M] per-row activation scale
    const float* w_scale;  // [N padded] per-output-channel weight scale (GEGLU: in packed row order)
    int K;                 // real K (multiple of 16); p.K = weight row stride in bytes (multiple of 128, zero-filled past K)
};

template <int EPI, bool OUT_F32, int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 16) ? 4 : 2) void gemm_fp8_kernel(const VkGemmDesc p, const F8Args q) {
    constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;
    constexpr int AP = BM / RPP, WP = BN / RPP;
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    constexpr int FX = FN, FY = FM;  // X = weights (MFMA row operand), Y = activations
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tilesN = (p.N + BN - 1) / BN;
    const int tilesM = (p.M + BM - 1) / BM;
    const int logical = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int tn = logical % tilesN, tm = logical / tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const uint8_t* __restrict__ Ag = (const uint8_t*)p.A;
    const uint8_t* __restrict__ Wg = (const uint8_t*)p.Wt;

    const int lc = tid & 7, lr = tid >> 3;
    const int lsrc = lc ^ ((lr >> 1) & 7);  // logical 16-B chunk this lane fetches so that the swizzled image lands (gemm.hip)
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    const uint8_t* wptr[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) wptr[i] = Wg + (size_t)(n0 + lr + RPP * i) * p.K + lsrc * 16;
    const uint8_t* aptr[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int m = m0 + lr + RPP * i;
        if (m >= p.M) m = p.M - 1;
        aptr[i] = Ag + (size_t)m * p.lda + lsrc * 16;
    }

    auto dma_tile = [&](int kt, int stage) {
        const int k0 = kt * 128;
        char* sA = smem + stage * STAGE_BYTES + wave_u * 1024;
        char* sW = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(wptr[i] + k0), (lptr_t)(sW + i * RPP * 128), 16, 0, 0);
        const bool inside = k0 + lsrc * 16 < q.K;  // K tail (K = 320: 2.5 K-steps): chunks past the row end are zero-filled
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const uint8_t* src = inside ? aptr[i] + k0 : (const uint8_t*)&g_zero16_f8;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + i * RPP * 128), 16, 0, 0);
        }
    };

    f32x16_t acc[FX][FY];
#pragma unroll
    for (int i = 0; i < FX; ++i)
#pragma unroll
        for (int j = 0; j < FY; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int MW = FM * 32, NW = FN * 32;
    const int xoff = wn * NW, yoff = wm * MW;
    const int sw = (l31 >> 1) & 7;
    const int xrow_off = A_BYTES + (xoff + l31) * 128, yrow_off = (yoff + l31) * 128;

    auto load_frag = [&](const char* rowp, int ks) {
        const int c = ks * 4 + lh * 2;  // the lane's 32 k-bytes = logical chunks c, c+1 of the 128-byte row
        const i32x4_t lo = *(const i32x4_t*)(rowp + ((c ^ sw) << 4));
        const i32x4_t hi = *(const i32x4_t*)(rowp + (((c + 1) ^ sw) << 4));
        i32x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    auto compute = [&](int stage) {
        const char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x8_t xf[FX], yf[FY];
#pragma unroll
            for (int f = 0; f < FX; ++f) xf[f] = load_frag(sb + xrow_off + f * 32 * 128, ks);
#pragma unroll
            for (int f = 0; f < FY; ++f) yf[f] = load_frag(sb + yrow_off + f * 32 * 128, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fi = 0; fi < FX; ++fi)
#pragma unroll
                for (int fj = 0; fj < FY; ++fj)
                    acc[fi][fj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xf[fi], yf[fj], acc[fi][fj], 0, 0, 0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nk = p.K / 128;
    dma_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) dma_tile(kt + 1, stage ^ 1);
        compute(stage);
        __syncthreads();
    }

    // dequantise in registers (row scale of the lane's output row x channel scales of its column quads), then the shared epilogue
#pragma unroll
    for (int fj = 0; fj < FY; ++fj) {
        const int m = m0 + wm * MW + fj * 32 + l31;
        const float sa = q.a_scale[m < p.M ? m : p.M - 1];
#pragma unroll
        for (int fi = 0; fi < FX; ++fi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 s4 = *(const float4*)(q.w_scale + n0 + wn * NW + fi * 32 + 8 * g + 4 * lh);  // padded to the tile
                acc[fi][fj][4 * g + 0] *= sa * s4.x;
                acc[fi][fj][4 * g + 1] *= sa * s4.y;
                acc[fi][fj][4 * g + 2] *= sa * s4.z;
                acc[fi][fj][4 * g + 3] *= sa * s4.w;
            }
    }
    gemm_epilogue<EPI, OUT_F32, FX, FY, FM, FN>(p, acc, m0, n0, wm, wn, l31, lh);
}

template <int EPI, bool OUT_F32, int WM, int WN, int FM, int FN>
int launch_cfg(const VkGemmDesc* d, const F8Args& q, hipStream_t stream) {
    constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
    const int tiles = ((d->N + BN - 1) / BN) * ((d->M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI, OUT_F32, WM, WN, FM, FN>), dim3(tiles), dim3(WM * WN * 64), 0, stream, *d, q);
    VK_CHECK_LAUNCH();
    return VK_OK;
}

template <int EPI, bool OUT_F32>
int launch(const VkGemmDesc* d, const F8Args& q, hipStream_t stream) {
    int cfg = d->tile_cfg & 7;
    if (cfg == 0) {
        auto wgs = [&](int bm, int bn) { return (long long)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn); };
        const int n256 = (d->N + 255) / 256 * 256;
        if (EPI != EPI_GEGLU && d->N % 320 == 0 && wgs(256, 320) >= 192) cfg = 4;  // GEGLU: 256x256 measured faster
        else if (n256 * 10 <= d->N * 11 && wgs(256, 256) >= 192) cfg = 3;
        else cfg = 1;
    }
    if (cfg == 4) return launch_cfg<EPI, OUT_F32, 4, 2, 2, 5>(d, q, stream);
    if (cfg == 3) return launch_cfg<EPI, OUT_F32, 4, 4, 2, 2>(d, q, stream);  // sixteen 64x64 wave tiles, 4 waves per SIMD (as in gemm.hip)
    return launch_cfg<EPI, OUT_F32, 2, 2, 2, 2>(d, q, stream);
}

// one wave per row: amax -> scale = amax / 448, q = e4m3(x / scale)
__global__ __launch_bounds__(256) void quantize_rows_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ qo, float* __restrict__ scale,
                                                            int M, int K, long long ldx, long long ldq) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    constexpr int MAXC = 10;  // 8-element chunks per lane: K <= 5120
    const int nch = K >> 3;
    uint4 v[MAXC];
    float amax = 0.f;
    const uint16_t* xr = x + (size_t)row * ldx;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            v[i] = *(const uint4*)(xr + c * 8);
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) scale[row] = sc;
    uint8_t* qr = qo + (size_t)row * ldq;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * inv, -448.f), 448.f);  // the cvt yields NaN above 448, it does not saturate
            int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            *(int2*)(qr + c * 8) = make_int2(lo, hi);
        }
    }
}

}  // namespace

extern "C" int vk_gemm_fp8(const VkGemmDesc* d, const float* a_scale, const float* w_scale, int32_t k_real, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !d->A || !d->Wt || !d->out || !a_scale || !w_scale) return VK_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % 128) != 0 || (d->N % 4) != 0 || k_real <= 0 || k_real > d->K || (k_real % 16) != 0 ||
        d->lda < k_real || (d->lda % 16) != 0 || d->amode != AMODE_DENSE || d->tile_cfg < 0 || (d->tile_cfg & 7) > 4 || d->tile_cfg > 7)
        return VK_EINVAL;
    if ((d->rowvec || d->rowvec2) && d->rows_per_vec <= 0) return VK_EINVAL;
    if (d->ln_stats || d->rowstat_out || d->A2 || (d->rowvec2 && !d->res2)) return VK_EINVAL;  // bf16-GEMM-only features
    const F8Args q{a_scale, w_scale, k_real};
    const bool f32 = d->out_f32 != 0;
    if (d->epi == EPI_LINEAR) return f32 ? launch<EPI_LINEAR, true>(d, q, stream) : launch<EPI_LINEAR, false>(d, q, stream);
    if (d->epi == EPI_GEGLU && !f32 && (d->N % 32) == 0) return launch<EPI_GEGLU, false>(d, q, stream);
    return VK_EINVAL;
}

extern "C" int vk_quantize_rows_fp8(const void* x, void* q, float* scale, int32_t M, int32_t K, int64_t ldx, int64_t ldq, void* stream_) {
    if (!x || !q || !scale || M <= 0 || K <= 0 || (K % 8) != 0 || K > 5120 || ldx < K || ldq < K || (ldx % 8) != 0 || (ldq % 8) != 0) return VK_EINVAL;
    hipLaunchKernelGGL(quantize_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream_, (const uint16_t*)x, (uint8_t*)q, scale, M, K,
                       (long long)ldx, (long long)ldq);
    VK_CHECK_LAUNCH();
    return VK_OK;
}
