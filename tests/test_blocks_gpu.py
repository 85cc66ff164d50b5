"""Per-block parity AT THE BASELINE SHAPES (VERDICT r1 "weak" #1-#3): every distinct block type x UNet level of the shipped
1.65 B configuration, full channel widths, full 72x128 latent, T = 25 frames, against the CPU fp32 oracle restatement
(oracle/vista_oracle.py, pinned to the real reference by tests/test_oracle_cpu.py).

Whole-network tests cannot see a small systematic error (a wrong eps, a dropped bias) below the bf16 noise floor of ~100 layers;
one block has 4-25 bf16 roundings, so its floor is 2-6e-3 and the bound is tight enough to catch those. Stated tolerance per block
(bf16 storage / fp32 accumulation / bf16-rounded weights vs fp32):
    ResBlock family, Downsample, Upsample, out conv :  rel-L2 <= 5e-3, max|err| <= 2e-2 * max|ref|
    SpatialVideoTransformer (two transformer blocks, ~25 stored bf16 tensors, bf16 softmax probabilities) : rel-L2 <= 8e-3, max <= 3e-2
BASELINE config 5 (fp8 e4m3): the SAME blocks at the SAME shapes with the config's fp8 switches on, against the SAME
fp32 oracle output (computed once per block). Re-stated tolerances, fp8 storage of GEMM operands with 3 mantissa bits
(2^-4 = 6.3e-2 relative per element, averaged down by the K-sum) on residual branches:
    ResBlock family with its four convolutions in fp8            :  rel-L2 <= 4e-2, max|err| <= 6e-2 * max|ref|   (measured 2.7-2.8e-2 / 2.5-3.0e-2)
    SpatialVideoTransformer with fp8 FeedForwards (+ attention / projections when those switches exist) : rel-L2 <= 4e-2, max <= 6e-2
                                                                                                            (FeedForwards: 2.6-2.7e-2 / 2.5-2.8e-2)
Inputs are bf16-representable; weights are the seeded non-zero init of vista_amd.synth (every zero-init tensor re-randomised).
One window = 25 frames of ONE clip (b = 1): the CFG-doubled batch of the bench is two independent copies of this.
Measured values are appended to gpurun_out/block_parity.json when that directory exists."""
import json
import os
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
T = 25
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rand(shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).to(BF16).float()


def _tok(x):  # NCHW f32 -> (n, H*W, C) bf16 on the GPU
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n, h * w, c).to(BF16).cuda().contiguous()


def _nchw(t, n, H, W):
    return t.float().cpu().view(n, H, W, -1).permute(0, 3, 1, 2)


def _seed(mod, prefix, seed):
    """Seeded non-zero weights into `mod`; returns the oracle's state dict (keys prefixed like a slice of the UNet's)."""
    from vista_amd import synth
    shapes = {f"{prefix}.{k}": tuple(v.shape) for k, v in mod.state_dict().items()}
    sd = synth.seeded_state_dict(shapes, seed)
    mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()}, strict=True)
    return sd


def _fp8(keys):
    """Context manager: BASELINE config 5 switches `keys` on for the duration (whatever subset of them this build knows)."""
    import contextlib
    from vista_amd.modules import attention

    @contextlib.contextmanager
    def cm():
        saved = dict(attention.FP8)
        for k in keys:
            if k in attention.FP8:
                attention.FP8[k] = True
        try:
            yield
        finally:
            attention.FP8.update(saved)
    return cm()


def _report(name, out, ref, rl_tol, mx_tol, t_ref):
    err = (out - ref)
    rl = (err.pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item()
    mx = (err.abs().max() / ref.abs().max()).item()
    print(f"[block-parity] {name}: rel-L2 {rl:.3e} (<= {rl_tol:.0e})  max|err|/max|ref| {mx:.3e} (<= {mx_tol:.0e})  oracle {t_ref:.1f} s")
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "block_parity.json"), "a") as f:
            f.write(json.dumps({"block": name, "rel_l2": rl, "max_rel": mx, "oracle_s": round(t_ref, 2)}) + "\n")
    assert torch.isfinite(out).all(), name
    assert rl <= rl_tol and mx <= mx_tol, f"{name}: rel-L2 {rl:.3e} (tol {rl_tol}), max-rel {mx:.3e} (tol {mx_tol})"


def _emb(seed):
    return _rand((T, 1280), seed, 0.7)


# (name, Cin or (C_h, C_skip), Cout, H, W): every distinct VideoResBlock shape class of video_model.py:186-440 at the BASELINE latent
RESBLOCKS = [
    ("L0 320->320 @72x128", 320, 320, 72, 128),
    ("L0 cat(640+320)->320 @72x128", (640, 320), 320, 72, 128),       # 30 channels per group: a group straddles the two tensors
    ("L1 320->640 @36x64 (1x1 skip)", 320, 640, 36, 64),
    ("L2 640->1280 @18x32 (1x1 skip)", 640, 1280, 18, 32),
    ("L2 cat(1280+1280)->1280 @18x32", (1280, 1280), 1280, 18, 32),
    ("L3 1280->1280 @9x16 (S=144)", 1280, 1280, 9, 16),
]


@pytest.mark.parametrize("name,cin,cout,H,W", RESBLOCKS, ids=[r[0].split(" @")[0].replace(" ", "_") for r in RESBLOCKS])
def test_video_resblock_at_baseline_shape(name, cin, cout, H, W):
    """VideoResBlock.forward (video_model.py:59-75): 2-D ResBlock -> 3x1x1 temporal ResBlock over 25 frames -> learned blend."""
    from oracle import vista_oracle as O
    from vista_amd.modules.diffusionmodules.video_model import VideoResBlock
    parts = cin if isinstance(cin, tuple) else (cin,)
    blk = VideoResBlock(channels=sum(parts), emb_channels=1280, dropout=0.0, out_channels=cout, video_kernel_size=[3, 1, 1],
                        merge_strategy="learned_with_images", merge_factor=0.5, dims=2)
    sd = _seed(blk, "blk", 3)
    blk = blk.cuda().eval()
    xs = [_rand((T, c, H, W), 10 + i, 1.0, 0.25 * i) for i, c in enumerate(parts)]
    emb = _emb(5)
    t0 = time.time()
    with torch.no_grad():
        ref = O.video_resblock(sd, "blk", torch.cat(xs, 1), emb, T)
    t_ref = time.time() - t0
    toks = tuple(_tok(x) for x in xs)
    emb_silu = F.silu(emb).to(BF16).cuda()
    with torch.no_grad():
        out = blk(toks if len(toks) == 2 else toks[0], emb_silu, T, H, W)
    _report("VideoResBlock " + name, _nchw(out, T, H, W), ref, 5e-3, 2e-2, t_ref)
    with torch.no_grad(), _fp8(["conv"]):  # BASELINE config 5: the block's four convolutions on e4m3 GroupNorm output
        out8 = blk(toks if len(toks) == 2 else toks[0], emb_silu, T, H, W)
    assert not torch.equal(out8, out), "the fp8 switch did not change the path"
    _report("[fp8 conv] VideoResBlock " + name, _nchw(out8, T, H, W), ref, 4e-2, 6e-2, 0.0)
    with torch.no_grad():
        assert torch.equal(blk(toks if len(toks) == 2 else toks[0], emb_silu, T, H, W), out), "switching fp8 off restores the bf16 path bit for bit"


TRANSFORMERS = [("L0 C=320 S=9216", 320, 72, 128), ("L1 C=640 S=2304", 640, 36, 64), ("L2 C=1280 S=576", 1280, 18, 32),
                ("mid C=1280 S=144", 1280, 9, 16)]


@pytest.mark.parametrize("name,C,H,W", TRANSFORMERS, ids=[t[0].split(" ")[0] for t in TRANSFORMERS])
def test_spatial_video_transformer_at_baseline_shape(name, C, H, W):
    """SpatialVideoTransformer.forward (video_attention.py:239-296): GN -> proj_in -> spatial block (9216-token self-attention at
    level 0) -> + frame-pos-emb -> temporal block over 25 frames -> blend -> proj_out -> + x."""
    from oracle import vista_oracle as O
    from vista_amd import ops, synth
    from vista_amd.modules.video_attention import SpatialVideoTransformer
    blk = SpatialVideoTransformer(C, C // 64, 64, depth=1, context_dim=1024, use_linear=True, use_spatial_context=True, ff_in=True,
                                  merge_strategy="learned_with_images", merge_factor=0.5, attn_mode="softmax-xformers",
                                  action_control=True)
    sd = _seed(blk, "blk", 4)
    blk = blk.cuda().eval()
    x = _rand((T, C, H, W), 20, 1.0, 0.3)
    w = synth.window_inputs(T=T, H=2, W=2, seed=9, trajectory=[0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2])
    ctx = w["c"]["crossattn"].to(BF16).float()   # (T, 1, 3456): CLIP-like token + action sinusoids
    t0 = time.time()
    with torch.no_grad():
        ref = O.spatial_video_transformer(sd, "blk", x, ctx, T, True)
    t_ref = time.time() - t0
    frame_idx = torch.arange(T, dtype=torch.float32).cuda()
    with torch.no_grad():
        out = blk(_tok(x), ops.cast_to_bf16(ctx.reshape(T, -1).cuda()), frame_idx, T, H, W)
    _report("SpatialVideoTransformer " + name, _nchw(out, T, H, W), ref, 8e-3, 3e-2, t_ref)
    for keys in (["feedforward"], ["feedforward", "attention", "proj"]):  # BASELINE config 5: FeedForwards, then everything the config names
        from vista_amd.modules import attention
        if not all(k in attention.FP8 for k in keys):
            continue
        with torch.no_grad(), _fp8(keys):
            out8 = blk(_tok(x), ops.cast_to_bf16(ctx.reshape(T, -1).cuda()), frame_idx, T, H, W)
        if C == 320 and keys == ["feedforward"]:
            # level 0: config 5 keeps the fused bf16 FeedForward kernel (faster than the fp8 pair there and 15x closer to fp32, round 4)
            assert torch.equal(out8, out), "config 5 must leave the level-0 FeedForwards on the fused bf16 kernel"
            continue
        assert not torch.equal(out8, out), "the fp8 switch did not change the path"
        _report(f"[fp8 {'+'.join(keys)}] SpatialVideoTransformer " + name, _nchw(out8, T, H, W), ref, 4e-2, 6e-2, 0.0)
    with torch.no_grad():
        assert torch.equal(blk(_tok(x), ops.cast_to_bf16(ctx.reshape(T, -1).cuda()), frame_idx, T, H, W), out)


def test_downsample_upsample_outconv_at_baseline_shape():
    """Downsample (openaimodel.py:136: conv3x3 stride 2), Upsample (:100-102: nearest x2 then conv3x3), the final GroupNorm32 -> SiLU ->
    conv3x3 320->4 (video_model.py:434-440,502-503)."""
    from vista_amd import ops
    from vista_amd.modules.diffusionmodules.openaimodel import Downsample, Upsample
    n = 50  # the CFG-doubled batch, as in the bench
    down = Downsample(320, True, dims=2, out_channels=320)
    sd = _seed(down, "d", 6)
    x = _rand((n, 320, 72, 128), 30)
    t0 = time.time()
    ref = F.conv2d(x, sd["d.op.weight"], sd["d.op.bias"], stride=2, padding=1)
    t_ref = time.time() - t0
    out, Ho, Wo = down.cuda()(_tok(x), 72, 128)
    assert (Ho, Wo) == (36, 64)
    _report("Downsample 320 @72x128 -> 36x64", _nchw(out, n, 36, 64), ref, 5e-3, 2e-2, t_ref)

    up = Upsample(1280, True, dims=2, out_channels=1280)
    sd = _seed(up, "u", 7)
    x = _rand((n, 1280, 18, 32), 31)
    t0 = time.time()
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), sd["u.conv.weight"], sd["u.conv.bias"], padding=1)
    t_ref = time.time() - t0
    out, Ho, Wo = up.cuda()(_tok(x), 18, 32)
    assert (Ho, Wo) == (36, 64)
    _report("Upsample 1280 @18x32 -> 36x64", _nchw(out, n, 36, 64), ref, 5e-3, 2e-2, t_ref)

    from vista_amd import synth
    gw, gb = synth.seeded_tensor("out.0.weight", (320,), 8), synth.seeded_tensor("out.0.bias", (320,), 8)
    cw, cb = synth.seeded_tensor("out.2.weight", (4, 320, 3, 3), 8), synth.seeded_tensor("out.2.bias", (4,), 8)
    x = _rand((n, 320, 72, 128), 32, 1.0, 0.4)
    t0 = time.time()
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gw, gb, 1e-5)), cw, cb, padding=1)
    t_ref = time.time() - t0
    h = ops.groupnorm(_tok(x), gw.cuda(), gb.cuda(), 1e-5, silu=True)
    out, _, _ = ops.conv3x3(h, ops.pack_conv3x3(cw, cb), n, 72, 128, out_f32=True)
    _report("out: GN32+SiLU+conv3x3 320->4 @72x128", _nchw(out[:, :, :4], n, 72, 128), ref, 5e-3, 2e-2, t_ref)


def test_frame_position_rows_are_reused_only_for_the_same_unchanged_frame_index_tensor():
    """SpatialVideoTransformer keeps time_pos_embed(sinusoid(frame_idx)) next to its packed weights while the caller passes the SAME frame_idx
    tensor unchanged: the cached forward equals the uncached one bit for bit, an in-place change of the indices (version counter) or a
    changed time_pos_embed weight recomputes them."""
    from vista_amd import ops
    from vista_amd.modules.video_attention import SpatialVideoTransformer
    T, C, H, W = 3, 64, 4, 8
    blk = SpatialVideoTransformer(C, 1, 64, depth=1, context_dim=1024, use_linear=True, use_spatial_context=True, ff_in=True,
                                  merge_strategy="learned_with_images", merge_factor=0.5, attn_mode="softmax-xformers", action_control=False)
    _seed(blk, "blk", 11)
    blk = blk.cuda().eval()
    x = _tok(_rand((T, C, H, W), 3))
    ctx = ops.cast_to_bf16(_rand((T, 1024), 4).cuda())
    fi = torch.arange(T, dtype=torch.float32).cuda()
    with torch.no_grad():
        first = blk(x, ctx, fi, T, H, W)                       # fills the cache
        assert blk.packed()["_tpe_rows"][0] is fi
        again = blk(x, ctx, fi, T, H, W)                       # served from it
        fresh = blk(x, ctx, fi.clone(), T, H, W)               # another tensor object: recomputed
        assert torch.equal(first, again) and torch.equal(first, fresh)
        fi.add_(2.0)                                           # same object, new contents
        moved = blk(x, ctx, fi, T, H, W)
        assert not torch.equal(moved, first) and torch.equal(moved, blk(x, ctx, fi.clone(), T, H, W))
        blk.time_pos_embed[2].bias.add_(0.5)                   # a weight change drops the pack and the rows with it
        rew = blk(x, ctx, fi, T, H, W)
        assert not torch.equal(rew, moved) and torch.equal(rew, blk(x, ctx, fi.clone(), T, H, W))
