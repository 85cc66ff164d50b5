"""CPU fp32 ORACLE for Vista's denoising hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module -- as the checker,
never as the product.  It is a plain-PyTorch (fp32, CPU, NCHW like the reference) restatement of the reference
algorithm, written as pure functions over a *reference-keyed* state dict (`model.diffusion_model.*` minus the
prefix), each citing the reference file:line it follows (paths relative to the Vista tree).  It deliberately keeps
the reference's literal arithmetic -- e.g. the 1-token cross-attention runs q.k^T/softmax/v as written -- so the
product's algebraic shortcuts are checked against an independent statement of the maths.

Parity pinning: the reference ships no tests/golden vectors, so this oracle is pinned against the
reference ITSELF, imported on CPU through oracle/ref_shim.py in the build container: oracle/make_golden.py runs the
real `VideoUNet` / `Denoiser` / `EulerEDMSampler` on seeded inputs and commits their outputs under tests/golden/;
tests/test_oracle_cpu.py checks this file against those vectors (and, when /root/reference is present, against the
live reference modules).  The third-party kernels the reference calls (ATen conv/linear/norm, xformers attention)
are restated by their published definitions.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ helpers
def append_dims(x, target_dims):
    """vwm/util.py:180-188"""
    d = target_dims - x.ndim
    if d < 0:
        raise ValueError(f"Input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * d]


def timestep_embedding(timesteps, dim, max_period=10000):
    """vwm/modules/diffusionmodules/util.py:141-165"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat((torch.cos(args), torch.sin(args)), dim=-1)
    if dim % 2:
        emb = torch.cat((emb, torch.zeros_like(emb[:, :1])), dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn32(sd, p, x, eps=1e-5):
    """GroupNorm32: util.py:196-216 (32 groups, fp32); statistics over all non-(batch, group) dims, so a 5-D input
    normalises over (C/32, T, H, W)."""
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _mlp(sd, p, x):
    """Linear-SiLU-Linear (video_model.py:148-157,176-182; video_attention.py:227-231)"""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


# ------------------------------------------------------------------------------------------------ ResBlocks
def resblock(sd, p, x, emb, dims=2, exchange_temb_dims=False):
    """ResBlock._forward, openaimodel.py:258-284 (no up/down, no scale-shift norm: vista.yaml)."""
    h = _gn32(sd, p + ".in_layers.0", x)
    h = F.silu(h)
    w, b = sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"]
    h = F.conv2d(h, w, b, padding=1) if dims == 2 else F.conv3d(h, w, b, padding=(1, 0, 0))
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    while emb_out.ndim < h.ndim:
        emb_out = emb_out[..., None]
    if exchange_temb_dims:  # 'b t c ... -> b c t ...' (openaimodel.py:280-281)
        emb_out = emb_out.transpose(1, 2)
    h = h + emb_out
    h = F.silu(_gn32(sd, p + ".out_layers.0", h))
    w, b = sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"]
    h = F.conv2d(h, w, b, padding=1) if dims == 2 else F.conv3d(h, w, b, padding=(1, 0, 0))
    if (p + ".skip_connection.weight") in sd:  # 1x1 conv (openaimodel.py:241)
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def video_resblock(sd, p, x, emb, T):
    """VideoResBlock.forward, video_model.py:59-75"""
    x = resblock(sd, p, x, emb)
    n, c, hh, ww = x.shape
    b = n // T
    x5 = x.view(b, T, c, hh, ww).permute(0, 2, 1, 3, 4)  # (b t) c h w -> b c t h w
    xt = resblock(sd, p + ".time_stack", x5, emb.view(b, T, -1), dims=3, exchange_temb_dims=True)
    alpha = torch.sigmoid(sd[p + ".time_mixer.mix_factor"])  # AlphaBlender learned_with_images: util.py:304-318 (scalar)
    out = alpha * x5 + (1.0 - alpha) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


# ------------------------------------------------------------------------------------------------ attention
def _heads(t, heads):
    b, n, _ = t.shape
    return t.view(b, n, heads, -1).permute(0, 2, 1, 3).reshape(b * heads, n, -1)


def cross_attention(sd, p, x, context=None, action_control=False, context_dim=1024):
    """MemoryEfficientCrossAttention.forward, attention.py:326-421 (add_lora False); the xformers core
    (attention.py:400-407) restated by its definition softmax(q k^T / sqrt(d)) v."""
    heads = sd[p + ".to_q.weight"].shape[0] // 64
    ctx = x if context is None else context
    ctx_act = None
    if action_control:
        ctx, ctx_act = ctx[:, :, :context_dim], ctx[:, :, context_dim:]
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    if action_control:
        k = k + F.linear(ctx_act, sd[p + ".k_adapter_action_control.weight"])
        v = v + F.linear(ctx_act, sd[p + ".v_adapter_action_control.weight"])
    b = q.shape[0]
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    # same arithmetic per (image, head); the batch is walked in slices so the score matrix of the 9216-token level
    # (250 heads x 9216 x 9216 fp32 = 85 GB at once) stays below ~2 GB
    step = max(1, int(2 ** 29 // max(1, q.shape[1] * k.shape[1])))
    outs = []
    for i in range(0, q.shape[0], step):
        s = torch.matmul(q[i:i + step], k[i:i + step].transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        outs.append(torch.matmul(torch.softmax(s, dim=-1), v[i:i + step]))
    out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
    out = out.view(b, heads, out.shape[1], -1).permute(0, 2, 1, 3).reshape(b, out.shape[1], -1)
    return _lin(sd, p + ".to_out.0", out)


def feed_forward(sd, p, x):
    """FeedForward with GEGLU, attention.py:85-128 (exact-erf GELU)."""
    h = _lin(sd, p + ".net.0.proj", x)
    a, gate = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def basic_transformer_block(sd, p, x, context, action_control):
    """BasicTransformerBlock._forward, attention.py:514-524"""
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x)) + x
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, action_control) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def video_transformer_block(sd, p, x, context, T, action_control):
    """VideoTransformerBlock._forward, video_attention.py:111-141 (ff_in True, is_res True)."""
    bt, s, c = x.shape
    b = bt // T
    x = x.view(b, T, s, c).permute(0, 2, 1, 3).reshape(b * s, T, c)  # (b t) s c -> (b s) t c
    x = feed_forward(sd, p + ".ff_in", _ln(sd, p + ".norm_in", x)) + x
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x)) + x
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, action_control) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x.view(b, s, T, c).permute(0, 2, 1, 3).reshape(bt, s, c)


def spatial_video_transformer(sd, p, x, context, T, action_control):
    """SpatialVideoTransformer.forward, video_attention.py:239-296 (use_linear, use_spatial_context, depth 1)."""
    n, c, hh, ww = x.shape
    x_in = x
    # time_context = first frame's context of every clip, repeated for every pixel (video_attention.py:252-257)
    time_context = context[::T].repeat_interleave(hh * ww, dim=0)
    h = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)  # Normalize: attention.py:141-142
    h = h.permute(0, 2, 3, 1).reshape(n, hh * ww, c)
    h = _lin(sd, p + ".proj_in", h)
    frames = torch.arange(T).repeat(n // T)
    emb = _mlp(sd, p + ".time_pos_embed", timestep_embedding(frames, c))[:, None]
    depth = 0
    while (p + f".transformer_blocks.{depth}.norm1.weight") in sd:
        h = basic_transformer_block(sd, p + f".transformer_blocks.{depth}", h, context, action_control)
        h_mix = video_transformer_block(sd, p + f".time_stack.{depth}", h + emb, time_context, T, action_control)
        alpha = torch.sigmoid(sd[p + ".time_mixer.mix_factor"])
        h = alpha * h + (1.0 - alpha) * h_mix
        depth += 1
    h = _lin(sd, p + ".proj_out", h)
    h = h.view(n, hh, ww, c).permute(0, 3, 1, 2)
    return h + x_in


# ------------------------------------------------------------------------------------------------ UNet
def _block(sd, p, h, emb, context, T, action_control):
    """TimestepEmbedSequential dispatch, openaimodel.py:32-53; layer kinds recovered from the state-dict keys."""
    j = 0
    while True:
        q = f"{p}.{j}"
        if (q + ".in_layers.0.weight") in sd:
            h = video_resblock(sd, q, h, emb, T)
        elif (q + ".norm.weight") in sd:
            h = spatial_video_transformer(sd, q, h, context, T, action_control)
        elif (q + ".op.weight") in sd:  # Downsample, openaimodel.py:136,141-143
            h = F.conv2d(h, sd[q + ".op.weight"], sd[q + ".op.bias"], stride=2, padding=1)
        elif (q + ".conv.weight") in sd:  # Upsample, openaimodel.py:86-103
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[q + ".conv.weight"], sd[q + ".conv.bias"], padding=1)
        elif (q + ".weight") in sd:  # plain input conv, video_model.py:186-190
            h = F.conv2d(h, sd[q + ".weight"], sd[q + ".bias"], padding=1)
        else:
            return h
        j += 1


def unet_forward(sd, x, timesteps, context, y, cond_mask, num_frames, action_control=True):
    """VideoUNet.forward, video_model.py:442-503. x (N,8,H,W); context (N,1,ctx); y (N,adm); cond_mask (N,) float."""
    mc = sd["time_embed.0.weight"].shape[1]
    t_emb = timestep_embedding(timesteps, mc)
    if cond_mask is not None and bool(cond_mask.any()):
        m = cond_mask[..., None].float()
        emb = _mlp(sd, "cond_time_stack_embed", t_emb) * m + _mlp(sd, "time_embed", t_emb) * (1 - m)
    else:
        emb = _mlp(sd, "time_embed", t_emb)
    if num_frames > 1 and context.shape[0] != x.shape[0]:
        context = context.repeat_interleave(num_frames, dim=0)  # repeat_as_img_seq, vwm/util.py:63-75
    if num_frames > 1 and y.shape[0] != x.shape[0]:
        y = y.repeat_interleave(num_frames, dim=0)
    emb = emb + _mlp(sd, "label_emb.0", y)
    hs = []
    h = x.float()
    i = 0
    while (f"input_blocks.{i}.0.weight" in sd) or (f"input_blocks.{i}.0.in_layers.0.weight" in sd) or (f"input_blocks.{i}.0.op.weight" in sd):
        h = _block(sd, f"input_blocks.{i}", h, emb, context, num_frames, action_control)
        hs.append(h)
        i += 1
    h = _block(sd, "middle_block", h, emb, context, num_frames, action_control)
    i = 0
    while f"output_blocks.{i}.0.in_layers.0.weight" in sd:
        h = torch.cat((h, hs.pop()), dim=1)
        h = _block(sd, f"output_blocks.{i}", h, emb, context, num_frames, action_control)
        i += 1
    h = F.silu(_gn32(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def wrapper_forward(sd, x, t, c, cond_mask, num_frames, action_control=True):
    """OpenAIWrapper.forward, wrappers.py:25-40"""
    concat = c.get("concat")
    if concat is not None:
        if num_frames > 1 and concat.shape[0] != x.shape[0]:
            concat = concat.repeat_interleave(num_frames, dim=0)
        x = torch.cat((x, concat), dim=1)
    return unet_forward(sd, x, t, c.get("crossattn"), c.get("vector"), cond_mask, num_frames, action_control)


# ------------------------------------------------------------------------------------------------ denoiser / sampler
def vscaling_edm_cnoise(sigma):
    """VScalingWithEDMcNoise.__call__, denoiser_scaling.py:51-59"""
    c_skip = 1.0 / (sigma ** 2 + 1.0)
    c_out = -sigma / (sigma ** 2 + 1.0) ** 0.5
    c_in = 1.0 / (sigma ** 2 + 1.0) ** 0.5
    c_noise = 0.25 * sigma.log()
    return c_skip, c_out, c_in, c_noise


def denoiser_forward(sd, noised_input, sigma, cond, cond_mask, num_frames, action_control=True):
    """Denoiser.forward, denoiser.py:22-35"""
    sigma_shape = sigma.shape
    s = append_dims(sigma, noised_input.ndim)
    c_skip, c_out, c_in, c_noise = vscaling_edm_cnoise(s)
    c_noise = c_noise.reshape(sigma_shape)
    net = wrapper_forward(sd, noised_input * c_in, c_noise, cond, cond_mask, num_frames, action_control)
    return net * c_out + noised_input * c_skip


def edm_sigmas(n, sigma_min=0.002, sigma_max=700.0, rho=7.0, append_zero=True):
    """EDMDiscretization.get_sigmas + Discretization.__call__, discretizer.py:16-37 (defaults of sample_utils.py:156-159)."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat((sigmas, sigmas.new_zeros([1]))) if append_zero else sigmas


def linear_guider_scale(num_frames=25, max_scale=2.5, min_scale=1.0):
    """LinearPredictionGuider.__init__, guiders.py:50-61"""
    return torch.linspace(min_scale, max_scale, num_frames)


def triangle_guider_scale(num_frames=25, max_scale=2.5, min_scale=1.0, period=1.0):
    """TrianglePredictionGuider.__init__, guiders.py:87-118 (single period, period_fusing 'max')"""
    values = torch.linspace(0, 1, num_frames)
    tri = 2 * (values / period - torch.floor(values / period + 0.5)).abs()
    return tri * (max_scale - min_scale) + min_scale


def guider_combine(x, scale):
    """VanillaCFG.__call__ (scalar scale, guiders.py:23-26) / LinearPredictionGuider.__call__ (per-frame, :54-61)."""
    x_u, x_c = x.chunk(2)
    if not torch.is_tensor(scale):
        return x_u + scale * (x_c - x_u)
    T = scale.numel()
    b = x_u.shape[0] // T
    s = append_dims(scale.repeat(b), x_u.ndim)
    return x_u + s * (x_c - x_u)


def guider_prepare_inputs(x, s, c, cond_mask, uc):
    """VanillaCFG / LinearPredictionGuider.prepare_inputs, guiders.py:28-36,63-71"""
    c_out = {}
    for k in c:
        if k in ("vector", "crossattn", "concat"):
            c_out[k] = torch.cat((uc[k], c[k]), 0)
        else:
            c_out[k] = c[k]
    return torch.cat([x] * 2), torch.cat([s] * 2), c_out, torch.cat([cond_mask] * 2)


def euler_edm_sample(denoise_fn, x, cond, uc, cond_frame, cond_mask, num_steps, scale=2.5, sigma_max=700.0, sigma_min=0.002, rho=7.0,
                     guider="cfg", s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, noise_fn=None):
    """EulerEDMSampler.__call__, sampling.py:30-45,78-124. `denoise_fn(x, sigma, cond, cond_mask)` is the closure of
    sample_utils.py:314-315. guider: 'cfg' (VanillaCFG-like prepare/combine) or 'identity'. s_churn > 0: the stochastic branch of
    sampler_step (sampling.py:78-83): gamma = min(s_churn / (n_sigmas - 1), sqrt(2) - 1) while s_tmin <= sigma <= s_tmax (:109-113),
    sigma_hat = sigma (1 + gamma), x += noise_fn(x) * s_noise * sqrt(sigma_hat^2 - sigma^2); noise_fn defaults to torch.randn_like."""
    x = x.clone()
    sigmas = edm_sigmas(num_steps, sigma_min, sigma_max, rho)
    x *= torch.sqrt(1.0 + sigmas[0] ** 2)
    s_in = x.new_ones([x.shape[0]])
    replace = cond_mask is not None and bool(cond_mask.any())
    for i in range(len(sigmas) - 1):
        if replace:
            x = x * append_dims(1 - cond_mask, x.ndim) + cond_frame * append_dims(cond_mask, cond_frame.ndim)
        sigma, next_sigma = s_in * sigmas[i], s_in * sigmas[i + 1]
        gamma = min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= float(sigmas[i]) <= s_tmax else 0.0
        if gamma > 0:
            sigma_hat = sigma * (gamma + 1.0)
            eps = (torch.randn_like(x) if noise_fn is None else noise_fn(x)) * s_noise
            x = x + eps * append_dims(sigma_hat ** 2 - sigma ** 2, x.ndim) ** 0.5
            sigma = sigma_hat
        if guider == "identity":
            denoised = denoise_fn(x, sigma, cond, cond_mask)
        else:
            denoised = guider_combine(denoise_fn(*guider_prepare_inputs(x, sigma, cond, cond_mask, uc)), scale)
        d = (x - denoised) / append_dims(sigma, x.ndim)  # to_d, sampling_utils.py:46-47
        x = x + append_dims(next_sigma - sigma, x.ndim) * d  # euler_step, sampling.py:66-67
    if replace:
        x = x * append_dims(1 - cond_mask, x.ndim) + cond_frame * append_dims(cond_mask, cond_frame.ndim)
    return x
