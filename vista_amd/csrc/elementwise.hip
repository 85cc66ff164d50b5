// Small HBM-bound elementwise / layout kernels around the UNet and the EulerEDM sampler step (gfx950).
// Reference call sites are listed per entry point in include/vista_hip.h.
#include "common.h"
#include "vista_hip.h"

namespace {

constexpr int EW_THREADS = 256;
__host__ inline int ew_grid(long long n) {
    long long g = (n + EW_THREADS - 1) / EW_THREADS;
    const long long cap = 256LL * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ---- channel concat: 16-B chunks ----
__global__ void concat_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, long long rows, int g1,
                              int g2) {
    const int g = g1 + g2;
    const long long n = rows * g;
    GRID_STRIDE(i, n) {
        const long long r = i / g;
        const int c = (int)(i - r * g);
        out[i] = (c < g1) ? a[r * g1 + c] : b[r * g2 + (c - g1)];
    }
}

// ---- NCHW f32 -> token-major bf16 padded to Cpad channels; one thread per (img, pixel, 8-channel chunk) ----
__global__ void nchw_to_tokens_kernel(const float* __restrict__ x, uint4* __restrict__ out, int n_img, int C, int HW, int Cpad) {
    const int cg = Cpad >> 3;
    const long long n = (long long)n_img * HW * cg;
    GRID_STRIDE(i, n) {
        const int ch = (int)(i % cg);
        const long long ip = i / cg;
        const int pix = (int)(ip % HW);
        const int img = (int)(ip / HW);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            f[e] = (c < C) ? x[((size_t)img * C + c) * HW + pix] : 0.f;
        }
        out[i] = pack8(f);
    }
}

__global__ void tokens_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int n_img, int C, int HW, int ldx) {
    const long long n = (long long)n_img * C * HW;
    GRID_STRIDE(i, n) {
        const int pix = (int)(i % HW);
        const long long ic = i / HW;
        const int c = (int)(ic % C);
        const int img = (int)(ic / C);
        out[i] = x[((size_t)img * HW + pix) * ldx + c];
    }
}

template <bool F32>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, void* __restrict__ out_, int n, int dim, float neg_log_period) {
    const int half = dim >> 1;
    const long long tot = (long long)n * dim;
    GRID_STRIDE(i, tot) {
        const int j = (int)(i % dim);
        const int r = (int)(i / dim);
        float v = 0.f;
        if (j < 2 * half) {
            const int jj = (j < half) ? j : j - half;
            const float freq = expf(neg_log_period * (float)jj / (float)half);
            const float arg = t[r] * freq;
            v = (j < half) ? cosf(arg) : sinf(arg);
        }
        if (F32) ((float*)out_)[i] = v;
        else ((uint16_t*)out_)[i] = f32_to_bf16(v);
    }
}

__global__ void emb_combine_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                   const float* __restrict__ mask, float* __restrict__ emb, uint16_t* __restrict__ silu_out, int n, int dim) {
    const long long tot = (long long)n * dim;
    GRID_STRIDE(i, tot) {
        const int r = (int)(i / dim);
        float v;
        if (a) {
            const float m = mask[r];
            v = a[i] * m + b[i] * (1.f - m);
        } else {
            v = b[i];
        }
        if (c) v += c[i];
        emb[i] = v;
        if (silu_out) silu_out[i] = f32_to_bf16(silu_f(v));
    }
}

__global__ void silu_cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, long long n, int do_silu) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        y[i] = f32_to_bf16(do_silu ? silu_f(v) : v);
    }
}

// ---- sampler ----
__global__ void sampler_prepare_kernel(float* __restrict__ x, const float* __restrict__ cond_frame, const float* __restrict__ mask,
                                       const float* __restrict__ concat_uc, const float* __restrict__ concat_c, uint4* __restrict__ net_in,
                                       int T, int HW, int Cpad, float c_in, int replace) {
    const int cg = Cpad >> 3;
    const long long n = (long long)T * HW * cg;
    GRID_STRIDE(i, n) {
        const int ch = (int)(i % cg);
        const long long ip = i / cg;
        const int pix = (int)(ip % HW);
        const int t = (int)(ip / HW);
        uint4 vu = make_uint4(0, 0, 0, 0), vc = make_uint4(0, 0, 0, 0);
        if (ch == 0) {
            float fu[8], fc[8];
            const float m = replace ? mask[t] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const size_t idx = ((size_t)t * 4 + c) * HW + pix;
                float xv = x[idx];
                if (replace) {
                    xv = xv * (1.f - m) + cond_frame[idx] * m;
                    x[idx] = xv;
                }
                fu[c] = fc[c] = xv * c_in;
                fu[4 + c] = concat_uc ? concat_uc[idx] : 0.f;
                fc[4 + c] = concat_c ? concat_c[idx] : 0.f;
            }
            vu = pack8(fu);
            vc = pack8(fc);
        }
        net_in[((size_t)t * HW + pix) * cg + ch] = vu;
        net_in[((size_t)(T + t) * HW + pix) * cg + ch] = vc;
    }
}

__global__ void sampler_update_kernel(float* __restrict__ x, const float* __restrict__ net, const float* __restrict__ scale, int T, int HW,
                                      int ld, float c_out, float c_skip, float sigma, float sigma_next) {
    const long long n = (long long)T * 4 * HW;
    GRID_STRIDE(i, n) {
        const int pix = (int)(i % HW);
        const long long ic = i / HW;
        const int c = (int)(ic % 4);
        const int t = (int)(ic / 4);
        const float xv = x[i];
        const float du = net[((size_t)t * HW + pix) * ld + c] * c_out + xv * c_skip;
        const float dc = net[((size_t)(T + t) * HW + pix) * ld + c] * c_out + xv * c_skip;
        const float g = du + scale[t] * (dc - du);
        const float d = (xv - g) / sigma;
        x[i] = xv + d * (sigma_next - sigma);
    }
}

__global__ void denoiser_combine_kernel(const float* __restrict__ net, const float* __restrict__ x, const float* __restrict__ c_out,
                                        const float* __restrict__ c_skip, float* __restrict__ out, int n_img, int chw) {
    const long long n = (long long)n_img * chw;
    GRID_STRIDE(i, n) {
        const int r = (int)(i / chw);
        out[i] = net[i] * c_out[r] + x[i] * c_skip[r];
    }
}
__global__ void cfg_combine_kernel(const float* __restrict__ x2, const float* __restrict__ scale, float* __restrict__ out, int T, int chw) {
    const long long n = (long long)T * chw;
    GRID_STRIDE(i, n) {
        const int t = (int)(i / chw);
        const float u = x2[i], c = x2[n + i];
        out[i] = u + scale[t] * (c - u);
    }
}
__global__ void euler_step_kernel(const float* __restrict__ x, const float* __restrict__ den, const float* __restrict__ sigma,
                                  const float* __restrict__ sigma_next, float* __restrict__ out, int n_img, int chw) {
    const long long n = (long long)n_img * chw;
    GRID_STRIDE(i, n) {
        const int r = (int)(i / chw);
        const float d = (x[i] - den[i]) / sigma[r];
        out[i] = x[i] + d * (sigma_next[r] - sigma[r]);
    }
}
__global__ void mask_replace_kernel(const float* __restrict__ x, const float* __restrict__ cond, const float* __restrict__ mask,
                                    float* __restrict__ out, int n_img, int chw) {
    const long long n = (long long)n_img * chw;
    GRID_STRIDE(i, n) {
        const float m = mask[(int)(i / chw)];
        out[i] = x[i] * (1.f - m) + cond[i] * m;
    }
}
__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out, int n_img, int chw) {
    const long long n = (long long)n_img * chw;
    GRID_STRIDE(i, n) { out[i] = x[i] * s[(int)(i / chw)]; }
}

__global__ void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ out,
                                       int n_img, int chw, float scale) {
    const long long n = (long long)n_img * chw;
    GRID_STRIDE(i, n) {
        const long long img = i / chw, r = i - img * chw;
        const float mean = mom[img * 2 * chw + r];
        float v = mean;
        if (noise) {
            const float lv = fminf(fmaxf(mom[img * 2 * chw + chw + r], -30.f), 20.f);
            v = fmaf(__expf(0.5f * lv), noise[i], mean);
        }
        out[i] = v * scale;
    }
}

// sum over i of  sum_e (x[e][i] - mean_e x[.][i])^2 / (E - 1): per-block partials in a fixed order, then one block adds them.
__global__ __launch_bounds__(256) void ens_var_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, int E, long long n) {
    __shared__ double red[256];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float mean = 0.f;
        for (int e = 0; e < E; ++e) mean += x[(size_t)e * n + i];
        mean /= (float)E;
        float d2 = 0.f;
        for (int e = 0; e < E; ++e) {
            const float d = x[(size_t)e * n + i] - mean;
            d2 = fmaf(d, d, d2);
        }
        acc += (double)(d2 / (float)(E - 1));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void ens_var_final_kernel(const double* __restrict__ partial, double* __restrict__ out, int nblk) {
    double a = 0.0;
    for (int i = 0; i < nblk; ++i) a += partial[i];
    out[0] = a;
}


// ---------------------------------------------------------------------------------------------------------------
constexpr int CLIP_MAX_KS = 63;  // largest odd Gaussian kernel of the antialias blur (a side of ~7000 px resized to 224)
// OpenCLIP image preprocessing + patchify in one pass (FrozenOpenCLIPImageEmbedder.preprocess, vwm/modules/encoders/modules.py:304-315,
// then the 14x14 / stride-14 patch convolution's im2col): kornia 0.6.9 `resize(x, (224, 224), "bicubic", align_corners=True, antialias=True)`
// = separable Gaussian blur (reflect border; sigma = (factor - 1) / 2 per axis, kernel size int(max(4 sigma, 3)) made odd -- only when
// downscaling) followed by F.interpolate(mode="bicubic", align_corners=True) (A = -0.75, clamped taps); then (x + 1) / 2 and the CLIP
// mean / std. One thread per output pixel; its value is written straight into the patch-embedding GEMM's A operand:
//   out[(img * (1 + gp*gp) + 1 + py*gp + px)][c*ps*ps + ky*ps + kx]   bf16, row stride ldo (columns 3*ps*ps .. ldo-1 zero), row 0 of each
//   image (the class-token slot) zero.
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ int reflect_idx(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * (n - 1) - i : i; }

__global__ __launch_bounds__(EW_THREADS) void clip_preprocess_kernel(const float* __restrict__ img, uint16_t* __restrict__ out, int n_img, int H, int W,
                                                                     int out_hw, int ps, int ldo, float sig_y, float sig_x, int ks_y, int ks_x,
                                                                     float m0, float m1, float m2, float s0, float s1, float s2) {
    const int gp = out_hw / ps;
    const long long total = (long long)n_img * 3 * out_hw * out_hw;
    // Gaussian taps (the same for every output value): once per workgroup, in LDS, in the order and arithmetic of the per-thread form they
    // replace (a runtime-indexed register array lives in scratch: 2 x 63 floats per lane); ks == 1 = no blur along that axis
    __shared__ float gy[CLIP_MAX_KS], gx[CLIP_MAX_KS];
    if ((int)threadIdx.x < ks_y) { const float d = (float)((int)threadIdx.x - ks_y / 2); gy[threadIdx.x] = ks_y > 1 ? expf(-d * d / (2.f * sig_y * sig_y)) : 1.f; }
    if ((int)threadIdx.x < ks_x) { const float d = (float)((int)threadIdx.x - ks_x / 2); gx[threadIdx.x] = ks_x > 1 ? expf(-d * d / (2.f * sig_x * sig_x)) : 1.f; }
    __syncthreads();
    float ny = 0.f, nx = 0.f;
    for (int k = 0; k < ks_y; ++k) ny += gy[k];
    for (int k = 0; k < ks_x; ++k) nx += gx[k];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % out_hw);
        const int oy = (int)((i / out_hw) % out_hw);
        const int c = (int)((i / ((long long)out_hw * out_hw)) % 3);
        const int n = (int)(i / ((long long)3 * out_hw * out_hw));
        const float* src = img + ((size_t)n * 3 + c) * H * W;
        const float ry = out_hw > 1 ? (float)oy * ((float)(H - 1) / (float)(out_hw - 1)) : 0.f;
        const float rx = out_hw > 1 ? (float)ox * ((float)(W - 1) / (float)(out_hw - 1)) : 0.f;
        const int iy = (int)floorf(ry), ix = (int)floorf(rx);
        const float ty = ry - (float)iy, tx = rx - (float)ix;
        const float A = -0.75f;
        const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
        float acc = 0.f;
        for (int a = 0; a < 4; ++a) {
            int yy = iy - 1 + a;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);  // bicubic taps are clamped to the (blurred) image
            float rowacc = 0.f;
            for (int b = 0; b < 4; ++b) {
                int xx = ix - 1 + b;
                xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                float v = 0.f;  // blurred(yy, xx): reflect-padded separable Gaussian
                for (int ky = 0; ky < ks_y; ++ky) {
                    const float* row = src + (size_t)reflect_idx(yy + ky - ks_y / 2, H) * W;
                    float r = 0.f;
                    for (int kx = 0; kx < ks_x; ++kx) r = fmaf(gx[kx], row[reflect_idx(xx + kx - ks_x / 2, W)], r);
                    v = fmaf(gy[ky], r, v);
                }
                rowacc = fmaf(wx[b], v / (ny * nx), rowacc);
            }
            acc = fmaf(wy[a], rowacc, acc);
        }
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        const float val = ((acc + 1.f) * 0.5f - mean) / sd;
        const int py = oy / ps, ky = oy - py * ps, px = ox / ps, kx = ox - px * ps;
        out[((size_t)n * (1 + gp * gp) + 1 + py * gp + px) * ldo + (c * ps + ky) * ps + kx] = f32_to_bf16(val);
    }
}

}  // namespace

#define EW_LAUNCH(kernel, count, ...)                                                                          \
    hipLaunchKernelGGL(kernel, dim3(ew_grid(count)), dim3(EW_THREADS), 0, (hipStream_t)stream, __VA_ARGS__); \
    VK_CHECK_LAUNCH();                                                                                         \
    return VK_OK

extern "C" int vk_concat_channels_bf16(const void* a, const void* b, void* out, int64_t rows, int32_t C1, int32_t C2, void* stream) {
    if (!a || !b || !out || rows <= 0 || C1 <= 0 || C2 <= 0 || (C1 % 8) || (C2 % 8)) return VK_EINVAL;
    EW_LAUNCH(concat_kernel, rows * ((C1 + C2) / 8), (const uint4*)a, (const uint4*)b, (uint4*)out, (long long)rows, C1 / 8, C2 / 8);
}
extern "C" int vk_nchw_to_tokens_bf16(const float* x, void* out, int32_t n_img, int32_t C, int32_t HW, int32_t Cpad, void* stream) {
    if (!x || !out || n_img <= 0 || C <= 0 || HW <= 0 || Cpad < C || (Cpad % 8)) return VK_EINVAL;
    EW_LAUNCH(nchw_to_tokens_kernel, (long long)n_img * HW * (Cpad / 8), x, (uint4*)out, n_img, C, HW, Cpad);
}
extern "C" int vk_tokens_to_nchw_f32(const float* x, float* out, int32_t n_img, int32_t C, int32_t HW, int32_t ldx, void* stream) {
    if (!x || !out || n_img <= 0 || C <= 0 || HW <= 0 || ldx < C) return VK_EINVAL;
    EW_LAUNCH(tokens_to_nchw_kernel, (long long)n_img * C * HW, x, out, n_img, C, HW, ldx);
}
extern "C" int vk_timestep_embedding_bf16(const float* t, void* out, int32_t n, int32_t dim, float max_period, void* stream) {
    if (!t || !out || n <= 0 || dim < 2 || max_period <= 0.f) return VK_EINVAL;
    EW_LAUNCH(timestep_embedding_kernel<false>, (long long)n * dim, t, out, n, dim, -logf(max_period));
}
extern "C" int vk_timestep_embedding_f32(const float* t, float* out, int32_t n, int32_t dim, float max_period, void* stream) {
    if (!t || !out || n <= 0 || dim < 2 || max_period <= 0.f) return VK_EINVAL;
    EW_LAUNCH(timestep_embedding_kernel<true>, (long long)n * dim, t, (void*)out, n, dim, -logf(max_period));
}
extern "C" int vk_emb_combine(const float* a, const float* b, const float* c, const float* mask, float* emb, void* silu_out, int32_t n,
                              int32_t dim, void* stream) {
    if (!b || !emb || n <= 0 || dim <= 0 || (a && !mask)) return VK_EINVAL;
    EW_LAUNCH(emb_combine_kernel, (long long)n * dim, a, b, c, mask, emb, (uint16_t*)silu_out, n, dim);
}
extern "C" int vk_silu_f32_to_bf16(const float* x, void* y, int64_t count, void* stream) {
    if (!x || !y || count <= 0) return VK_EINVAL;
    EW_LAUNCH(silu_cast_kernel, count, x, (uint16_t*)y, (long long)count, 1);
}
extern "C" int vk_cast_f32_to_bf16(const float* x, void* y, int64_t count, void* stream) {
    if (!x || !y || count <= 0) return VK_EINVAL;
    EW_LAUNCH(silu_cast_kernel, count, x, (uint16_t*)y, (long long)count, 0);
}
extern "C" int vk_sampler_prepare(float* x, const float* cond_frame, const float* mask, const float* concat_uc, const float* concat_c,
                                  void* net_in, int32_t T, int32_t HW, int32_t Cpad, float c_in, int32_t replace, void* stream) {
    if (!x || !net_in || T <= 0 || HW <= 0 || Cpad < 8 || (Cpad % 8)) return VK_EINVAL;
    if (replace && (!cond_frame || !mask)) return VK_EINVAL;
    EW_LAUNCH(sampler_prepare_kernel, (long long)T * HW * (Cpad / 8), x, cond_frame, mask, concat_uc, concat_c, (uint4*)net_in, T, HW, Cpad,
              c_in, replace);
}
extern "C" int vk_sampler_update(float* x, const float* net_out, const float* scale, int32_t T, int32_t HW, int32_t ld, float c_out,
                                 float c_skip, float sigma, float sigma_next, void* stream) {
    if (!x || !net_out || !scale || T <= 0 || HW <= 0 || ld < 4 || sigma == 0.f) return VK_EINVAL;
    EW_LAUNCH(sampler_update_kernel, (long long)T * 4 * HW, x, net_out, scale, T, HW, ld, c_out, c_skip, sigma, sigma_next);
}
extern "C" int vk_denoiser_combine(const float* net, const float* x, const float* c_out, const float* c_skip, float* out, int32_t n_img,
                                   int32_t chw, void* stream) {
    if (!net || !x || !c_out || !c_skip || !out || n_img <= 0 || chw <= 0) return VK_EINVAL;
    EW_LAUNCH(denoiser_combine_kernel, (long long)n_img * chw, net, x, c_out, c_skip, out, n_img, chw);
}
extern "C" int vk_cfg_combine(const float* x2, const float* scale, float* out, int32_t T, int32_t chw, void* stream) {
    if (!x2 || !scale || !out || T <= 0 || chw <= 0) return VK_EINVAL;
    EW_LAUNCH(cfg_combine_kernel, (long long)T * chw, x2, scale, out, T, chw);
}
extern "C" int vk_euler_step(const float* x, const float* den, const float* sigma, const float* sigma_next, float* out, int32_t n_img,
                             int32_t chw, void* stream) {
    if (!x || !den || !sigma || !sigma_next || !out || n_img <= 0 || chw <= 0) return VK_EINVAL;
    EW_LAUNCH(euler_step_kernel, (long long)n_img * chw, x, den, sigma, sigma_next, out, n_img, chw);
}
extern "C" int vk_mask_replace(const float* x, const float* cond, const float* mask, float* out, int32_t n_img, int32_t chw, void* stream) {
    if (!x || !cond || !mask || !out || n_img <= 0 || chw <= 0) return VK_EINVAL;
    EW_LAUNCH(mask_replace_kernel, (long long)n_img * chw, x, cond, mask, out, n_img, chw);
}
extern "C" int vk_scale_rows(const float* x, const float* s, float* out, int32_t n_img, int32_t chw, void* stream) {
    if (!x || !s || !out || n_img <= 0 || chw <= 0) return VK_EINVAL;
    EW_LAUNCH(scale_rows_kernel, (long long)n_img * chw, x, s, out, n_img, chw);
}
extern "C" int vk_gaussian_sample(const float* moments, const float* noise, float* out, int32_t n_img, int32_t C, int32_t hw, float scale,
                                  void* stream) {
    if (!moments || !out || n_img <= 0 || C <= 0 || hw <= 0) return VK_EINVAL;
    EW_LAUNCH(gaussian_sample_kernel, (long long)n_img * C * hw, moments, noise, out, n_img, C * hw, scale);
}
extern "C" int vk_ensemble_variance_sum(const float* x, double* out, double* partial_ws, int32_t E, int64_t n, void* stream) {
    if (!x || !out || !partial_ws || E < 2 || n <= 0) return VK_EINVAL;
    const int nblk = 512;
    hipLaunchKernelGGL(ens_var_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, partial_ws, E, (long long)n);
    VK_CHECK_LAUNCH();
    hipLaunchKernelGGL(ens_var_final_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double*)partial_ws, out, nblk);
    VK_CHECK_LAUNCH();
    return VK_OK;
}
extern "C" int vk_clip_preprocess_patches(const float* img, void* out, int32_t n_img, int32_t H, int32_t W, int32_t out_hw, int32_t patch,
                                          int32_t ldo, float sigma_y, float sigma_x, int32_t ks_y, int32_t ks_x, const float* mean3,
                                          const float* std3, void* stream) {
    if (!img || !out || !mean3 || !std3 || n_img <= 0 || H < 2 || W < 2 || out_hw <= 0 || patch <= 0 || (out_hw % patch) != 0 ||
        ldo < 3 * patch * patch || ks_y < 1 || ks_x < 1 || ks_y > CLIP_MAX_KS || ks_x > CLIP_MAX_KS || !(ks_y & 1) || !(ks_x & 1) || !(sigma_y > 0.f) || !(sigma_x > 0.f))
        return VK_EINVAL;
    // the caller zero-fills `out` once (class-token rows and the K padding); this launch writes the 3*patch*patch image columns
    EW_LAUNCH(clip_preprocess_kernel, (long long)n_img * 3 * out_hw * out_hw, img, (uint16_t*)out, n_img, H, W, out_hw, patch, ldo, sigma_y, sigma_x,
              ks_y, ks_x, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
}
extern "C" int vk_abi_version(void) { return 7; }
extern "C" int vk_act_dtype(void) { return VK_F16 ? 1 : 0; }
