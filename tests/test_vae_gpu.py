"""Temporal VAE decoder on the MI355X: new kernels against PyTorch fp32 references, and the
vista_amd VideoDecoder / decode_first_stage against (a) the goldens produced by the REAL reference decoder on CPU
(tests/golden/vae_tiny.pt, oracle/make_golden_vae.py) and (b) the CPU oracle run on the spot.

Tolerance (stated): the reference decodes in fp32 (vista.yaml disable_first_stage_autocast: True); the HIP path keeps
activations in bf16 with fp32 accumulation through ~45 conv/norm layers. Measured stage by stage (tools/vae_stage_debug.py,
MI355X) the relative L2 error against the fp32 oracle grows smoothly by 0.5-1e-3 per block -- 1.9e-3 after conv_in, 9e-3 after
up.3, 1.8e-2 after up.0, 2.6e-2 at the output -- which is bf16 rounding noise, not a defect; a single layer fed the oracle's
own activations is within 4e-3. Stated tolerance for the whole decoder: relative L2 <= 4e-2.
"""
import os

import pytest
import torch
import torch.nn.functional as F


pytestmark = pytest.mark.gpu

def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).cuda()


def close(out, ref, name, rtol=1.6e-2, arel=2e-2):
    """|out - ref| <= arel*rms(ref) + rtol*|ref| elementwise (same rule as tests/test_kernels_gpu.py)."""
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape, f"{name}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    rms = ref.pow(2).mean().sqrt().item()
    err = (out - ref).abs()
    bad = err > arel * rms + rtol * ref.abs()
    assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance; max err {err.max().item():.4g} (rms ref {rms:.4g})"

GOLD = os.path.join(os.path.dirname(__file__), "golden")
BF16 = torch.bfloat16
TOL = 4e-2


def _ops():
    from vista_amd import ops
    return ops


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(1, 5, 8, 16, 64, 64), (2, 3, 6, 10, 128, 192), (1, 4, 16, 24, 64, 4), (1, 1, 8, 8, 64, 128)])
@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
def test_conv3d_3x3x3(B, T, H, W, Cin, Cout, cfg):
    ops = _ops()
    x = rnd(B * T, H * W, Cin)
    w = rnd(Cout, Cin, 3, 3, 3, scale=(27 * Cin) ** -0.5, seed=1)
    b = rnd(Cout, seed=2).float()
    r2 = rnd(B * T, H * W, Cout, seed=3)
    ops.TILE_CFG = cfg
    try:
        out = ops.conv3d(x, ops.pack_conv3d(w, b), T, H, W, res2=r2, alpha=0.7, beta=1.0)
        out32 = ops.conv3d(x, ops.pack_conv3d(w, b), T, H, W, out_f32=True)
    finally:
        ops.TILE_CFG = 0
    x5 = x.float().view(B, T, H, W, Cin).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(x5, w.float(), b, padding=1).permute(0, 2, 3, 4, 1).reshape(B * T, H * W, Cout)
    close(out32, ref, "conv3d f32", rtol=2e-3, arel=2e-3)
    close(out, 0.7 * ref + r2.float(), "conv3d blend epilogue")


def test_conv3d_rejects_bad_geometry():
    ops = _ops()
    x = rnd(4, 12, 64)
    pw = ops.pack_conv3d(rnd(64, 64, 3, 3, 3), None)
    with pytest.raises(Exception):
        ops.conv3d(x, pw, 0, 3, 4)          # T = 0
    with pytest.raises(ValueError):
        ops.conv3d(x, ops.pack_conv3x3(rnd(64, 64, 3, 3), None), 2, 3, 4)  # K mismatch


def test_conv_t3_padded_cin_f32_out():
    """AE3DConv.time_mix_conv: 3 real channels in a zero-padded 64-wide buffer, fp32 output."""
    ops = _ops()
    B, T, S = 1, 5, 96
    x3 = rnd(B * T, S, 3)
    xp = torch.zeros(B * T, S, 64, dtype=BF16, device="cuda")
    xp[..., :3] = x3
    w = rnd(3, 3, 3, 1, 1, scale=1 / 3, seed=1)
    b = rnd(3, seed=2).float()
    pw = ops.pack_conv_t3(w, b, cin_pad=64)
    assert pw.N == 4
    out = ops.conv_t3(xp, pw, T, S, out_f32=True)
    assert out.dtype == torch.float32 and out.shape == (B * T, S, 4)
    x5 = x3.float().view(B, T, S, 3).permute(0, 3, 1, 2)[..., None]
    ref = F.conv3d(x5, w.float(), b, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(B * T, S, 3)
    close(out[..., :3], ref, "time_mix_conv", rtol=2e-3, arel=2e-3)
    assert out[..., 3].abs().max().item() == 0.0


@pytest.mark.parametrize("rows,cols", [(128, 128), (300, 9216), (7, 16384), (64, 36), (5, 1028)])
def test_softmax_rows(rows, cols):
    ops = _ops()
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 4.0).cuda()
    x[0, : min(cols, 5)] = 60.0  # a dominant cluster: exercises the max subtraction
    out = ops.softmax_rows(x)
    ref = torch.softmax(x, -1)
    assert out.dtype == BF16
    close(out, ref, "softmax", rtol=8e-3, arel=1e-2)
    assert (out.float().sum(-1) - 1).abs().max().item() < 2e-2
    # strided source rows
    big = torch.zeros(rows, cols + 64, device="cuda")
    big[:, :cols] = x
    assert torch.equal(ops.softmax_rows(big[:, :cols]), out)
    with pytest.raises(Exception):
        ops.softmax_rows(torch.zeros(4, 16388, device="cuda"))  # > 16384 columns


def test_linear_vt_bias_and_out():
    ops = _ops()
    n, S, K, N = 3, 128, 256, 256
    x = rnd(n, S, K)
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    buf = torch.empty(n * N + 320, S, dtype=BF16, device="cuda")
    out = ops.linear_vt(x, ops.pack_linear(w, b), S, out=buf[: n * N].view(n, N, S))
    ref = (x.float() @ w.float().t() + b).transpose(1, 2)
    close(out, ref, "linear_vt bias")


def test_attn_block_single_head_matches_oracle():
    """AttnBlock (model.py:147-176) alone: 1 head of dim C over H*W tokens, against the oracle restatement on CPU."""
    from oracle import vae_oracle as V
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.model import AttnBlock
    C, n, H, W = 256, 3, 8, 16
    blk = AttnBlock(C)
    sd = synth.seeded_state_dict({k: tuple(v.shape) for k, v in blk.state_dict().items()}, 4)
    blk.load_state_dict(sd)
    blk.cuda()
    x = rnd(n, H * W, C)
    out = blk(x, H, W)
    xi = x.float().cpu().view(n, H, W, C).permute(0, 3, 1, 2)
    ref = V.attn_block({"a." + k: v for k, v in sd.items()}, "a", xi).permute(0, 2, 3, 1).reshape(n, H * W, C)
    close(out.cpu(), ref, "AttnBlock", rtol=2e-2, arel=2.5e-2)


# ------------------------------------------------------------------------------------------------ decoder
def _decoder(tag, seed=0):
    from oracle.make_golden_vae import TINY
    from vista_amd import synth
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    dec = VideoDecoder(video_kernel_size=[3, 1, 1] if tag == "k311" else 3, **TINY)
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    sd = synth.seeded_state_dict(shapes, seed)
    dec.load_state_dict(sd, strict=True)
    return dec.cuda(), sd, shapes


@pytest.mark.parametrize("tag", ["k311", "k333"])
def test_video_decoder_matches_reference_golden(tag):
    from oracle.make_golden_vae import latents
    from vista_amd import synth
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    dec, _, shapes = _decoder(tag)
    assert synth.shapes_digest(shapes) == g["digest_" + tag]
    z = latents(g["T"], g["H"], g["W"], g["seed_z"]).cuda()
    out = dec(z, timesteps=g["T"])
    assert out.dtype == torch.float32 and out.shape == g["out_" + tag].shape
    e = rel_l2(out, g["out_" + tag])
    print(f"VideoDecoder[{tag}] vs reference golden: rel-L2 {e:.3e}, max abs {(out.cpu() - g['out_' + tag]).abs().max().item():.3e}")
    assert e < TOL
    assert torch.equal(out, dec(z, timesteps=g["T"])), "decode must be deterministic"


def test_decode_first_stage_matches_reference_chunking():
    from oracle.make_golden_vae import latents
    from vista_amd.models.diffusion import decode_first_stage
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    dec, _, _ = _decoder("k311")
    z = (latents(11, g["H"], g["W"], 6) * 0.18215).cuda()
    for n, key in ((6, "dfs_11_n6"), (3, "dfs_11_n3")):
        out = decode_first_stage(dec, z, en_and_decode_n_samples_a_time=n)
        e = rel_l2(out, g[key])
        print(f"decode_first_stage n={n}: rel-L2 {e:.3e}")
        assert out.shape == g[key].shape and e < TOL
    with pytest.raises(ValueError):
        decode_first_stage(dec, z[:9], en_and_decode_n_samples_a_time=5)  # chunks of 2 < overlap 3: the reference fails too


def test_video_decoder_matches_oracle_other_shape():
    """A second geometry (T=3, latent 16x8, 2 clips in one call) against the CPU oracle run here."""
    from oracle import vae_oracle as V
    from vista_amd import synth
    dec, sd, _ = _decoder("k311", seed=2)
    z = synth.seeded_tensor("vae.z2", (6, 4, 16, 8), 9)
    with torch.no_grad():
        ref = V.video_decoder(sd, z, 3)
    out = dec(z.cuda(), timesteps=3)
    e = rel_l2(out, ref)
    print(f"VideoDecoder 2 clips x 3 frames vs oracle: rel-L2 {e:.3e}")
    assert e < TOL
    # clips are independent. Not bit for bit: the GEMM launcher may split K for the smaller problem (a different fp32 summation
    # order), so the two decodes agree to bf16 noise, and each of them is bitwise repeatable.
    alone = dec(z[3:].cuda(), timesteps=3)
    assert rel_l2(alone, out[3:]) < 2e-2
    assert torch.equal(alone, dec(z[3:].cuda(), timesteps=3))


def test_decoder_refuses_cpu():
    from oracle.make_golden_vae import TINY
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    dec = VideoDecoder(video_kernel_size=[3, 1, 1], **TINY)
    with pytest.raises(Exception):
        dec(torch.zeros(1, 4, 8, 8), timesteps=1)


# ------------------------------------------------------------------------------------------------ encoder
@pytest.mark.parametrize("n,H,W,C,Cout", [(2, 8, 16, 64, 64), (1, 10, 6, 128, 192), (3, 16, 16, 64, 128)])
def test_conv3x3_stride2_asymmetric_pad(n, H, W, C, Cout):
    """Downsample of the VAE encoder: F.pad(x, (0,1,0,1)) then conv 3x3 stride 2 pad 0 (model.py:77-81)."""
    ops = _ops()
    x = rnd(n, H * W, C)
    w = rnd(Cout, C, 3, 3, scale=(9 * C) ** -0.5, seed=1)
    b = rnd(Cout, seed=2).float()
    out, Ho, Wo = ops.conv3x3(x, ops.pack_conv3x3(w, b), n, H, W, stride=2, asym_pad=True)
    xi = x.float().view(n, H, W, C).permute(0, 3, 1, 2)
    ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w.float(), b, stride=2)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    close(out, ref.permute(0, 2, 3, 1).reshape(n, Ho * Wo, Cout), "asym-pad stride-2 conv")


def test_gaussian_sample_kernel():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    mom = torch.randn(3, 8, 5, 7, generator=g)
    mom[0, 4:] = 50.0    # logvar clamps at 20
    mom[1, 4:] = -50.0   # and at -30
    noise = torch.randn(3, 4, 5, 7, generator=g)
    mean, logvar = mom[:, :4], mom[:, 4:].clamp(-30, 20)
    ref = (mean + torch.exp(0.5 * logvar) * noise) * 0.18215
    out = ops.gaussian_sample(mom.cuda(), noise.cuda(), 0.18215).cpu()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(ops.gaussian_sample(mom.cuda(), None, 2.0).cpu(), mean * 2.0, rtol=1e-6, atol=0)


def test_encoder_and_encode_first_stage_match_reference_golden():
    from oracle.make_golden_vae import TINY, images
    from vista_amd import synth
    from vista_amd.models.autoencoder import AutoencodingEngine
    from vista_amd.models.diffusion import encode_first_stage
    from vista_amd.modules.diffusionmodules.model import Encoder
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    enc = Encoder(**TINY)
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert synth.shapes_digest(shapes) == g["digest_enc"]
    enc.load_state_dict(synth.seeded_state_dict(shapes, 0), strict=True)
    enc.cuda()
    x = images(5, 64, 128, 7).cuda()
    m = enc(x)
    e = rel_l2(m, g["enc_moments"])
    print(f"Encoder moments vs reference golden: rel-L2 {e:.3e}")
    assert m.shape == g["enc_moments"].shape and e < TOL
    fs = AutoencodingEngine(encoder=enc)
    torch.manual_seed(1234)
    noise = torch.randn(5, 4, 8, 16)
    fs.regularization.noise_fn = lambda shape, device: noise[:shape[0]].to(device)  # the draw the reference made under this seed
    z = encode_first_stage(fs, x[:3], scale_factor=0.18215, en_and_decode_n_samples_a_time=3)
    ez = rel_l2(z, g["enc_z_sampled"][:3] * 0.18215)
    print(f"encode_first_stage (sampled posterior) vs reference: rel-L2 {ez:.3e}")
    assert z.shape == (3, 4, 8, 16) and ez < TOL
    fs.regularization.sample = False
    assert rel_l2(encode_first_stage(fs, x, 1.0, 2), g["enc_z_mode"]) < TOL  # 3 chunks (2, 2, 1), posterior mode
    with pytest.raises(Exception):
        enc(torch.zeros(1, 3, 16, 16))
