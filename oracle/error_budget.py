"""Error budget of the HIP path's STORAGE FORMAT, measured in the oracle.  TEST INFRASTRUCTURE ONLY (as everything under oracle/).

The HIP path stores every activation and weight as bf16 and accumulates in fp32; the reference runs fp16 autocast with fp32 norms
(sample_utils.py:301-303, util.py:214-216).  This script restates oracle/vista_oracle.py with a rounding hook at every place the HIP
path writes a tensor, grouped in classes, so that the classes can be switched one at a time:

  W   weights of every Linear / conv (LayerNorm-folded ones as bf16(W * gamma): attention.py BasicTransformerBlock._pack of the product)
  H   the residual stream: block outputs, every `x + f(x)` sum inside the transformers, proj_in / up / down / input conv outputs
  OP  tensors that live between two kernels inside a block: GroupNorm+SiLU outputs, the first convolution's output, the 1x1 skip, q | k | v,
      the attention output, the GEGLU hidden activation
  P   the softmax numerators exp(s - max) handed to the P.V product (the row sum is taken from the unrounded exponentials, as the kernel's
      f32 adds do)

With every class off the restatement equals oracle.vista_oracle.unet_forward (checked on every run).  Rounding is `x.to(dtype).float()`
for dtype in {bf16, fp16}.  Run:  python -m oracle.error_budget [--tiny]  ->  profiles/r06_error_budget.txt (the table it prints).
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import vista_oracle as O  # noqa: E402

FMT = {}   # class -> torch dtype (absent = fp32, no rounding)


def R(x, cls):
    d = FMT.get(cls)
    return x if d is None else x.to(d).float()


def _w(sd, key):
    return R(sd[key], "W")


def _lin(sd, p, x):
    return F.linear(x, _w(sd, p + ".weight"), sd.get(p + ".bias"))


def _lin_lnfold(sd, p, norm, x, bias=True):
    """Linear(LayerNorm(x)) as the product computes it: the operand is x itself, the weight bf16(W * gamma), mean / rstd / beta applied in
    fp32 by the epilogue (gemm_common.h: rstd * (acc - mean * colsum) + bias)."""
    g, be = sd[norm + ".weight"], sd[norm + ".bias"]
    w = sd[p + ".weight"]
    wf = R(w * g[None, :], "W")
    mu = x.mean(-1, keepdim=True)
    rstd = (x.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    y = F.linear((x - mu) * rstd, wf) + F.linear(be, w)
    if bias and (p + ".bias") in sd:
        y = y + sd[p + ".bias"]
    return y


def _gn_silu(sd, p, x, eps=1e-5):
    return R(F.silu(F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)), "OP")


def resblock(sd, p, x, emb, dims=2, exchange=False, blend=None):
    h = _gn_silu(sd, p + ".in_layers.0", x)
    w, b = _w(sd, p + ".in_layers.2.weight"), sd[p + ".in_layers.2.bias"]
    h = F.conv2d(h, w, b, padding=1) if dims == 2 else F.conv3d(h, w, b, padding=(1, 0, 0))
    e = _lin(sd, p + ".emb_layers.1", R(F.silu(emb), "OP"))
    while e.ndim < h.ndim:
        e = e[..., None]
    if exchange:
        e = e.transpose(1, 2)
    h = R(h + e, "OP")
    h = _gn_silu(sd, p + ".out_layers.0", h)
    w, b = _w(sd, p + ".out_layers.3.weight"), sd[p + ".out_layers.3.bias"]
    h = F.conv2d(h, w, b, padding=1) if dims == 2 else F.conv3d(h, w, b, padding=(1, 0, 0))
    if (p + ".skip_connection.weight") in sd:
        x = R(F.conv2d(x, _w(sd, p + ".skip_connection.weight"), sd[p + ".skip_connection.bias"]), "OP")
    if blend is not None:   # time_stack: AlphaBlender folded into the second convolution's epilogue (x + (1 - alpha) * h)
        return R(x + (1.0 - blend) * h, "H")
    return R(x + h, "H")


def video_resblock(sd, p, x, emb, T):
    x = resblock(sd, p, x, emb)
    n, c, hh, ww = x.shape
    b = n // T
    x5 = x.view(b, T, c, hh, ww).permute(0, 2, 1, 3, 4)
    alpha = torch.sigmoid(sd[p + ".time_mixer.mix_factor"])
    out = resblock(sd, p + ".time_stack", x5, emb.view(b, T, -1), dims=3, exchange=True, blend=alpha)
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def _attn_core(q, k, v):
    s = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    e = torch.exp(s - s.amax(-1, keepdim=True))
    return torch.matmul(R(e, "P"), v) / e.sum(-1, keepdim=True)


def self_attention(sd, p, norm, x):
    heads = sd[p + ".to_q.weight"].shape[0] // 64
    q = R(_lin_lnfold(sd, p + ".to_q", norm, x, bias=False), "OP")
    k = R(_lin_lnfold(sd, p + ".to_k", norm, x, bias=False), "OP")
    v = R(_lin_lnfold(sd, p + ".to_v", norm, x, bias=False), "OP")
    b = q.shape[0]
    q, k, v = O._heads(q, heads), O._heads(k, heads), O._heads(v, heads)
    out = _attn_core(q, k, v)
    out = out.view(b, heads, out.shape[1], -1).permute(0, 2, 1, 3).reshape(b, out.shape[1], -1)
    return _lin(sd, p + ".to_out.0", R(out, "OP"))


def cross_attention_1tok(sd, p, context, action_control, context_dim=1024):
    """one-token context: softmax == 1, the output is to_out(to_v(ctx) + v_adapter(ctx_act)) (computed in fp32 from bf16 weights / context)"""
    ctx = R(context, "OP")
    v = F.linear(ctx[:, :, :context_dim], _w(sd, p + ".to_v.weight"))
    if action_control:
        v = v + F.linear(ctx[:, :, context_dim:], _w(sd, p + ".v_adapter_action_control.weight"))
    return _lin(sd, p + ".to_out.0", v)


def feed_forward(sd, p, norm, x):
    h = _lin_lnfold(sd, p + ".net.0.proj", norm, x)
    a, gate = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", R(a * F.gelu(gate), "OP"))


def basic_transformer_block(sd, p, x, context, action_control):
    x = R(self_attention(sd, p + ".attn1", p + ".norm1", x) + cross_attention_1tok(sd, p + ".attn2", context, action_control) + x, "H")
    return feed_forward(sd, p + ".ff", p + ".norm3", x) + x   # (the caller rounds: the frame-position embedding joins this epilogue)


def video_transformer_block(sd, p, x, context, T, action_control):
    bt, s, c = x.shape
    b = bt // T
    x = x.view(b, T, s, c).permute(0, 2, 1, 3).reshape(b * s, T, c)
    x = R(feed_forward(sd, p + ".ff_in", p + ".norm_in", x) + x, "H")
    x = R(self_attention(sd, p + ".attn1", p + ".norm1", x) + cross_attention_1tok(sd, p + ".attn2", context, action_control) + x, "H")
    x = feed_forward(sd, p + ".ff", p + ".norm3", x) + x
    return x.view(b, s, T, c).permute(0, 2, 1, 3).reshape(bt, s, c)


def spatial_video_transformer(sd, p, x, context, T, action_control):
    n, c, hh, ww = x.shape
    x_in = x
    time_context = context[::T].repeat_interleave(hh * ww, dim=0)
    h = R(F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6), "OP")
    h = h.permute(0, 2, 3, 1).reshape(n, hh * ww, c)
    h = R(_lin(sd, p + ".proj_in", h), "H")
    frames = torch.arange(T).repeat(n // T)
    emb = O._mlp(sd, p + ".time_pos_embed", O.timestep_embedding(frames, c))[:, None]
    depth = 0
    while (p + f".transformer_blocks.{depth}.norm1.weight") in sd:
        hs = basic_transformer_block(sd, p + f".transformer_blocks.{depth}", h, context, action_control)
        h_sp, h_t_in = R(hs, "H"), R(hs + emb, "H")
        h_mix = video_transformer_block(sd, p + f".time_stack.{depth}", h_t_in, time_context, T, action_control)
        alpha = torch.sigmoid(sd[p + ".time_mixer.mix_factor"])
        h = R(alpha * h_sp + (1.0 - alpha) * h_mix, "H")
        depth += 1
    h = _lin(sd, p + ".proj_out", h)
    h = h.view(n, hh, ww, c).permute(0, 3, 1, 2)
    return R(h + x_in, "H")


def _block(sd, p, h, emb, context, T, action_control):
    j = 0
    while True:
        q = f"{p}.{j}"
        if (q + ".in_layers.0.weight") in sd:
            h = video_resblock(sd, q, h, emb, T)
        elif (q + ".norm.weight") in sd:
            h = spatial_video_transformer(sd, q, h, context, T, action_control)
        elif (q + ".op.weight") in sd:
            h = R(F.conv2d(h, _w(sd, q + ".op.weight"), sd[q + ".op.bias"], stride=2, padding=1), "H")
        elif (q + ".conv.weight") in sd:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = R(F.conv2d(h, _w(sd, q + ".conv.weight"), sd[q + ".conv.bias"], padding=1), "H")
        elif (q + ".weight") in sd:
            h = R(F.conv2d(R(h, "OP"), _w(sd, q + ".weight"), sd[q + ".bias"], padding=1), "H")
        else:
            return h
        j += 1


def _mlp(sd, p, x):
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


def unet_forward(sd, x, timesteps, context, y, cond_mask, num_frames, action_control=True):
    mc = sd["time_embed.0.weight"].shape[1]
    t_emb = O.timestep_embedding(timesteps, mc)
    m = cond_mask[..., None].float()
    emb = _mlp(sd, "cond_time_stack_embed", t_emb) * m + _mlp(sd, "time_embed", t_emb) * (1 - m)
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
    h = _gn_silu(sd, "out.0", h)
    return F.conv2d(h, _w(sd, "out.2.weight"), sd["out.2.bias"], padding=1)


def rel(a, b):
    return ((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item()


def main():
    from oracle.make_golden import unet_inputs
    from vista_amd import synth
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    tiny = "--tiny" in sys.argv
    torch.set_grad_enabled(False)
    mc, T, H, W = (64, 3, 16, 32) if tiny else (320, 5, 16, 32)
    net = VideoUNet(**unet_kwargs(mc))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    del net
    sd = synth.seeded_state_dict(shapes, 0)
    x8, ts, ctx, y, mask = unet_inputs(T, H, W, seed=31, sigma=9.0)
    t0 = time.time()
    ref = O.unet_forward(sd, x8, ts, ctx, y, mask, T)
    print(f"# error budget of the storage format; {'tiny 64-channel' if tiny else 'full-width 1.65 B'} UNet, T = {T}, latent {H}x{W}, N = {2 * T} "
          f"(the inputs of tests/golden/unet_full_t5.pt); oracle forward {time.time() - t0:.0f} s")
    FMT.clear()
    base = unet_forward(sd, x8, ts, ctx, y, mask, T)
    print(f"restatement with every class off vs oracle.vista_oracle: rel-L2 {rel(base, ref):.2e} (must be ~1e-6)")
    assert rel(base, ref) < 1e-4
    if not tiny:
        gold = torch.load(os.path.join(ROOT, "tests", "golden", "unet_full_t5.pt"))
        print(f"oracle vs the reference's own output (golden): {rel(ref, gold['out']):.2e}")
    bf, hf = torch.bfloat16, torch.float16
    rows = [("bf16 everywhere (the shipped format)", {"W": bf, "H": bf, "OP": bf, "P": bf}),
            ("only W  bf16", {"W": bf}), ("only H  bf16", {"H": bf}), ("only OP bf16", {"OP": bf}), ("only P  bf16", {"P": bf}),
            ("fp16 everywhere (the reference's autocast width)", {"W": hf, "H": hf, "OP": hf, "P": hf}),
            ("bf16, H in fp16", {"W": bf, "H": hf, "OP": bf, "P": bf}),
            ("bf16, H in fp32", {"W": bf, "OP": bf, "P": bf}),
            ("bf16, H + OP in fp16 (weights, P bf16)", {"W": bf, "H": hf, "OP": hf, "P": bf}),
            ("fp16, P in bf16", {"W": hf, "H": hf, "OP": hf, "P": bf}),
            ("fp16, W in bf16", {"W": bf, "H": hf, "OP": hf, "P": hf})]
    res = {}
    for name, fmt in rows:
        FMT.clear()
        FMT.update(fmt)
        t0 = time.time()
        out = unet_forward(sd, x8, ts, ctx, y, mask, T)
        res[name] = rel(out, ref)
        print(f"{name:55s} rel-L2 vs fp32 {res[name]:.3e}   ({time.time() - t0:.0f} s)", flush=True)
    a = [res[f"only {c} bf16"] for c in ("W ", "H ", "OP", "P ")]
    print(f"root-sum-square of the four single-class figures: {sum(v * v for v in a) ** 0.5:.3e} (independent errors would add like this)")


if __name__ == "__main__":
    main()
