"""Conditioner on the MI355X: the OpenCLIP ViT-H/14 image tower, the sinusoid and first-stage embedders and the
GeneralConditioner routing, through the C ABI, against oracle/clip_oracle.py (transformers.CLIPVisionModelWithProjection fed the same
open_clip-named seeded weights; kornia-0.6.9 preprocessing restated) and the reference-generated goldens.

Stated tolerances (bf16 storage / fp32 accumulation vs fp32): preprocessing |err| <= 1.2e-2 (bf16 rounding of values up to ~2.7);
one kernel (attention, GELU epilogue) as tests/test_kernels_gpu.py; the 2-layer tower rel-L2 <= 1e-2; the full 32-layer ViT-H/14 tower
rel-L2 <= 2.5e-2 (the UNet's whole-network tolerance: ~130 stored bf16 tensors on the residual path); first-stage mode <= 4e-2 (tests/test_vae_gpu.py)."""
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
BF16 = torch.bfloat16


def rel_l2(a, b):
    return ((a.float() - b.float()).pow(2).sum().sqrt() / b.float().pow(2).sum().sqrt()).item()


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("H,W", [(96, 160), (576, 1024), (224, 224), (150, 300)])
def test_clip_preprocess_patchify_vs_restated_kornia(H, W):
    """resize (antialias blur + bicubic, align_corners) + (x+1)/2 + mean/std + im2col in one kernel vs the torch restatement."""
    from oracle import clip_oracle as CO
    from vista_amd import ops
    img = torch.tanh(_rnd(2, 3, H, W, seed=3) * 1.5)
    want = CO.preprocess(img, 224)                                          # (2, 3, 224, 224)
    got = ops.clip_preprocess_patches(img.cuda(), out_hw=224, patch=14).float().cpu()   # (2*257, 640)
    assert got.shape == (2 * 257, 640)
    g = got.view(2, 257, 640)
    assert float(g[:, 0].abs().max()) == 0.0 and float(g[:, :, 588:].abs().max()) == 0.0   # class-token rows and the K padding stay zero
    patches = want.unfold(2, 14, 14).unfold(3, 14, 14)                      # (2, 3, 16, 16, 14, 14)
    patches = patches.permute(0, 2, 3, 1, 4, 5).reshape(2, 256, 588)        # [py*16 + px][c*196 + ky*14 + kx]
    err = (g[:, 1:, :588] - patches).abs().max().item()
    print(f"[parity] clip preprocess {H}x{W} -> 224: max |err| {err:.3e} (values up to {patches.abs().max().item():.2f})")
    assert err <= 1.2e-2


def _unpatch(got, n):
    """(n*257, 640) patch rows -> (n, 3, 224, 224) image (the inverse of the kernel's im2col)."""
    g = got.view(n, 257, 640)[:, 1:, :588].reshape(n, 16, 16, 3, 14, 14)
    return g.permute(0, 3, 1, 4, 2, 5).reshape(n, 3, 224, 224)


def _norm(v, c):
    from vista_amd import ops
    return ((v + 1.0) / 2.0 - ops.CLIP_MEAN[c]) / ops.CLIP_STD[c]


@pytest.mark.parametrize("H,W", [(448, 448), (576, 1024), (2240, 4096)])   # the last: blur kernel 35 x 63 (the old 15-tap cap refused it)
def test_clip_preprocess_properties_constant_and_ramp(H, W):
    """Properties no restatement is needed for (the kornia-0.6.9 resize is otherwise pinned only to oracle/clip_oracle.py):
    a normalised Gaussian and a bicubic with align_corners=True both reproduce constants exactly and linear ramps exactly away from the
    reflect-padded border, so resize(constant) = constant and resize(ramp)(i) = ramp(i (W-1)/223)."""
    from vista_amd import ops
    const = torch.full((1, 3, H, W), 0.37)
    got = _unpatch(ops.clip_preprocess_patches(const.cuda()).float().cpu(), 1)
    for c in range(3):
        assert (got[0, c] - _norm(0.37, c)).abs().max().item() <= 8e-3     # one bf16 step at |value| ~ 1.5
    ramp = torch.linspace(-0.9, 0.9, W).view(1, 1, 1, W).expand(1, 3, H, W).contiguous()
    got = _unpatch(ops.clip_preprocess_patches(ramp.cuda()).float().cpu(), 1)
    xs = torch.arange(224, dtype=torch.float64) * (W - 1) / 223.0
    want = -0.9 + 1.8 * xs / (W - 1)
    _, ks = ops.antialias_blur_params(W, 224)
    margin = int(ks / 2 * 224 / W) + 3                                      # output columns whose blur window touches the reflected border
    for c in range(3):
        e = (got[0, c, :, margin:224 - margin] - _norm(want[margin:224 - margin], c).float()).abs().max().item()
        assert e <= 1.2e-2, (c, e)


@pytest.mark.parametrize("f", [2, 4])
def test_clip_preprocess_integer_downscale_vs_independent_float64(f):
    """Integer-factor downscale against an implementation that shares no code with this package or its oracle: scipy's separable Gaussian
    (mode='mirror' = torch's 'reflect', truncated to kornia 0.6.9's kernel size int(max(4 sigma, 3)) made odd, sigma = (f - 1) / 2) followed by
    torch's own float64 bicubic (a = -0.75, align_corners=True, clamped taps)."""
    import numpy as np
    from scipy.ndimage import gaussian_filter1d
    from vista_amd import ops
    H = W = 224 * f
    img = torch.tanh(_rnd(1, 3, H, W, seed=11) * 1.5)
    sigma, ks = ops.antialias_blur_params(H, 224)
    assert ks % 2 == 1 and ks == (int(max(4 * sigma, 3)) | 1) and abs(sigma - (f - 1) / 2) < 1e-12
    a = img.double().numpy()
    a = gaussian_filter1d(a, sigma, axis=2, mode="mirror", radius=ks // 2)
    a = gaussian_filter1d(a, sigma, axis=3, mode="mirror", radius=ks // 2)
    want = F.interpolate(torch.from_numpy(np.ascontiguousarray(a)), size=(224, 224), mode="bicubic", align_corners=True)
    got = _unpatch(ops.clip_preprocess_patches(img.cuda()).float().cpu(), 1)
    for c in range(3):
        e = (got[0, c] - _norm(want[0, c], c).float()).abs().max().item()
        print(f"[parity] clip preprocess x{f} downscale vs scipy + torch float64, channel {c}: max |err| {e:.3e}")
        assert e <= 1.2e-2


@pytest.mark.parametrize("n,heads,S,D", [(2, 4, 257, 80), (1, 16, 257, 80), (3, 2, 50, 64), (1, 2, 300, 128), (2, 1, 5, 80)])
def test_attn_small_vs_sdpa(n, heads, S, D):
    from vista_amd import ops
    c = heads * D
    qkv = _rnd(n * S, 3 * c, seed=1).to(BF16)
    o = ops.attn_small(qkv.cuda(), n, heads, S, D).float().cpu()
    q, k, v = (qkv[:, i * c:(i + 1) * c].float().view(n, S, heads, D).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n * S, c)
    e = (o - ref).abs()
    assert (e <= 2e-2 * ref.pow(2).mean().sqrt() + 1.6e-2 * ref.abs()).all(), e.max().item()
    with pytest.raises(Exception):
        ops.attn_small(qkv.cuda(), n, heads, S, 72)


@pytest.mark.parametrize("M,N,K", [(257, 5120, 1280), (514, 1280, 320), (100, 64, 128)])
def test_linear_gelu_epilogue(M, N, K):
    """act='gelu': exact-erf GELU of (acc + bias) in the LINEAR epilogue (both the LDS-staged and the generic path), with and without
    a folded LayerNorm in front; residual added AFTER the activation."""
    from vista_amd import ops
    x = _rnd(M, K, seed=2).to(BF16).cuda()
    w, b = _rnd(N, K, seed=3, scale=K ** -0.5).to(BF16), _rnd(N, seed=4)
    pw = ops.pack_linear(w, b)
    res = _rnd(M, N, seed=5).to(BF16).cuda()
    ref = F.gelu(x.float().cpu() @ w.float().t() + b)
    for cfg in (0, 1):
        ops.TILE_CFG = cfg
        try:
            out = ops.linear(x, pw, act="gelu").float().cpu()
            out_r = ops.linear(x, pw, act="gelu", res1=res).float().cpu()
        finally:
            ops.TILE_CFG = 0
        for o, r in ((out, ref), (out_r, ref + res.float().cpu())):
            e = (o - r).abs()
            assert (e <= 2e-2 * r.pow(2).mean().sqrt() + 1.6e-2 * r.abs()).all(), (cfg, e.max().item())
    with pytest.raises(ValueError):
        ops.linear(x, pw, act="relu")


def _tower(arch, seed):
    from vista_amd import synth
    from vista_amd.modules.encoders.modules import FrozenOpenCLIPImageEmbedder
    emb = FrozenOpenCLIPImageEmbedder(arch=arch)
    shapes = {k: tuple(v.shape) for k, v in emb.state_dict().items()}
    sd = synth.seeded_state_dict(shapes, seed)
    emb.load_state_dict(sd, strict=True)
    return emb.cuda().eval(), sd, shapes


def test_image_tower_tiny_vs_hf_golden():
    """2 layers, 4 heads of dim 80, 257 tokens: golden from transformers.CLIPVisionModelWithProjection (oracle/make_golden_cond.py)."""
    from vista_amd import synth
    g = torch.load(os.path.join(GOLD, "clip_tiny.pt"))
    emb, _, shapes = _tower(g["geometry"], g["seed"])
    assert synth.shapes_digest(shapes) == g["digest"]
    img = torch.tanh(synth.seeded_tensor("clip.img", g["img_shape"], g["seed"]) * 1.5)
    img[1] = img[1].flip(-1) * 0.7
    out = emb(img.cuda()).cpu()
    r = rel_l2(out, g["embed"])
    print(f"[parity] OpenCLIP image tower (tiny: 2 layers, 4 x 80-dim heads, 257 tokens) vs HF golden: rel-L2 {r:.3e}")
    assert out.dtype == torch.float32 and out.shape == g["embed"].shape and r <= 1e-2
    # identical images share one tower pass (a batch-of-one pass may pick other GEMM tiles than the batch above: equal to bf16 rounding)
    rep = emb(img[:1].repeat(3, 1, 1, 1).cuda()).cpu()
    assert torch.equal(rep[0], rep[1]) and torch.equal(rep[0], rep[2]) and rel_l2(rep[0], out[0]) <= 5e-3
    with pytest.raises(Exception):
        emb(img)  # CPU tensor: no fallback


def test_image_tower_vit_h14_full_geometry_vs_hf_oracle():
    """The shipped geometry -- ViT-H/14: width 1280, 32 layers, 16 heads of dim 80, MLP 5120, 1024-d projection, 632 M parameters -- on one
    576x1024 conditioning frame, against the HF model run here on the host in fp32 with the same seeded weights."""
    import json
    from oracle import clip_oracle as CO
    from vista_amd.modules.encoders.modules import OPENCLIP_VISION_GEOMETRY, FrozenOpenCLIPImagePredictionEmbedder
    from vista_amd import synth
    geo = OPENCLIP_VISION_GEOMETRY["ViT-H-14"]
    pe = FrozenOpenCLIPImagePredictionEmbedder({"target": "vwm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder", "params": {"freeze": True}},
                                               n_cond_frames=1, n_copies=1)
    shapes = {k: tuple(v.shape) for k, v in pe.state_dict().items()}
    assert sum(math_prod(s) for s in shapes.values()) > 6.3e8
    sd = synth.seeded_state_dict(shapes, 5)
    pe.load_state_dict(sd, strict=True)
    pe = pe.cuda().eval()
    img = torch.tanh(synth.seeded_tensor("clip.img576", (1, 3, 576, 1024), 5) * 1.2)
    want = CO.image_embed(sd, geo, img, prefix="open_clip.model.visual.")
    got = pe(img.cuda()).cpu()
    r = rel_l2(got[:, 0], want)
    print(f"[parity] OpenCLIP ViT-H/14 image tower (632 M parameters, 32 layers) on a 576x1024 frame vs HF oracle: rel-L2 {r:.3e}")
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        json.dump({"what": "ViT-H/14 image tower vs transformers.CLIPVisionModelWithProjection, seeded weights", "rel_l2": r},
                  open(os.path.join(d, "parity_clip_vit_h14.json"), "w"))
    assert got.shape == (1, 1, 1024) and r <= 2.5e-2


def math_prod(s):
    p = 1
    for v in s:
        p *= int(v)
    return p


def test_concat_timestep_embedder_fp32():
    from vista_amd import synth
    from vista_amd.modules.encoders.modules import ConcatTimestepEmbedderND
    x = torch.tensor([[0.5, 0.0, 1.0, 0.0, 1.5, 0.1, 2.0, 0.2], [127.0, 9.0, 0.02, 3.0, -1.0, 40.0, 0.0, 7.5]])
    e = ConcatTimestepEmbedderND(128, num_features=8, add_sequence_dim=True)(x.cuda()).cpu()
    want = synth.concat_timestep_embed(x, 128)[:, None]
    assert e.dtype == torch.float32 and e.shape == (2, 1, 1024)
    assert torch.allclose(e, want, atol=3e-5, rtol=0), float((e - want).abs().max())   # fp32 sin/cos of arguments up to 127 rad
    v = ConcatTimestepEmbedderND(256)(torch.tensor([9.0, 9.0]).cuda()).cpu()
    assert v.shape == (2, 256) and torch.allclose(v, synth.concat_timestep_embed(torch.tensor([9.0, 9.0]), 256), atol=3e-5, rtol=0)


def test_mode_only_autoencoder_matches_oracle():
    """AutoencoderKLModeOnly.encode = mean half of quant_conv(encoder(x)) (vwm/models/autoencoder.py:467-488,519-529), quant_conv composed
    into conv_out at pack time; oracle: the pinned encoder restatement + a literal 1x1 conv."""
    from oracle import vae_oracle as VO
    from oracle.make_golden_vae import TINY, images
    from vista_amd import synth
    from vista_amd.models.autoencoder import AutoencoderKLModeOnly
    ae = AutoencoderKLModeOnly(embed_dim=4, ddconfig=dict(TINY), monitor="val/rec_loss", loss_config={"target": "torch.nn.Identity"})
    shapes = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    assert "quant_conv.weight" in shapes and "encoder.conv_out.weight" in shapes
    sd = synth.seeded_state_dict(shapes, 2)
    ae.load_state_dict(sd, strict=True)
    ae = ae.cuda().eval()
    x = images(3, 64, 128, 7)
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    mom = VO.encoder(enc_sd, x, num_resolutions=len(TINY["ch_mult"]), num_res_blocks=TINY["num_res_blocks"])
    want = F.conv2d(mom, sd["quant_conv.weight"], sd["quant_conv.bias"])[:, :4]
    got = ae.encode(x.cuda()).cpu()
    r = rel_l2(got, want)
    print(f"[parity] AutoencoderKLModeOnly.encode vs oracle: rel-L2 {r:.3e}")
    assert got.shape == want.shape and r <= 4e-2
    assert rel_l2(ae.encode(x.cuda(), scale=0.5).cpu(), 0.5 * want) <= 4e-2


def test_general_conditioner_on_gpu_matches_reference_golden():
    """GeneralConditioner + get_batch + get_condition with the HIP ConcatTimestepEmbedderND (and the fixture's stand-ins for the two
    weight-carrying embedders) against the golden the REAL reference classes produced (oracle/make_golden_cond.py)."""
    from oracle import cond_fixture as CF
    from vista_amd.modules.encoders.modules import AbstractEmbModel, GeneralConditioner
    from vista_amd.sample_utils import VistaPipeline, get_condition
    m = types.ModuleType("cond_stub_g")

    class StubImageEmbedder(AbstractEmbModel):
        def __init__(self, dim):
            super().__init__()
            self.dim = dim

        def forward(self, img):
            return CF.stub_image_embed(img, self.dim)

    class StubLatentEmbedder(AbstractEmbModel):
        def forward(self, z):
            return z * 1.0

    m.StubImageEmbedder, m.StubLatentEmbedder = StubImageEmbedder, StubLatentEmbedder
    sys.modules["cond_stub_g"] = m
    cond = GeneralConditioner(CF.emb_models("cond_stub_g"))   # the vwm.* ConcatTimestepEmbedderND targets resolve to this package
    g = torch.load(os.path.join(GOLD, "cond_general.pt"))
    c, uc = get_condition(VistaPipeline(None, None, conditioner=cond), CF.value_dict("cuda"), CF.N, CF.FORCE_UC_ZERO, "cuda")
    for name, got, want in (("c", c, g["c"]), ("uc", uc, g["uc"])):
        for k in want:
            assert got[k].is_cuda and got[k].shape == want[k].shape
            assert torch.allclose(got[k].cpu(), want[k], atol=5e-5, rtol=0), (name, k, float((got[k].cpu() - want[k]).abs().max()))


def test_full_conditioner_end_to_end_from_the_shipped_config():
    """The shipped conditioner_config (configs/inference/vista_mi355x.yaml = vista.yaml:42-140) at full size: ViT-H/14 tower + first-stage
    encoder + sinusoid embedders -> (c, uc) of the shapes the sampler consumes for a 25-frame window, from one 576x1024 frame; the CLIP
    segment of c equals the tower's own output and the concat entry the encoder's mode."""
    import yaml
    from vista_amd import synth
    from vista_amd.config import CONFIG_PATH
    from vista_amd.sample_utils import VistaPipeline, get_condition
    from vista_amd.util import instantiate_from_config
    cond = instantiate_from_config(yaml.safe_load(open(CONFIG_PATH))["model"]["params"]["conditioner_config"])
    shapes = {k: tuple(v.shape) for k, v in cond.state_dict().items()}
    cond.load_state_dict(synth.seeded_state_dict(shapes, 9), strict=True)
    cond = cond.cuda().eval()
    img = torch.tanh(synth.seeded_tensor("cond.img", (1, 3, 576, 1024), 9)).cuda()
    vd = {"cond_frames_without_noise": img, "cond_frames": img + 0.02 * torch.randn_like(img), "fps_id": 9.0, "motion_bucket_id": 127.0, "cond_aug": 0.02,
          "trajectory": torch.tensor([0.5, 0.0, 1.0, 0.0, 1.5, 0.1, 2.0, 0.2])}
    T = 25
    c, uc = get_condition(VistaPipeline(None, None, conditioner=cond), vd, T, ["cond_frames", "cond_frames_without_noise", "command", "trajectory",
                                                                                "speed", "angle", "goal"], "cuda")
    assert c["crossattn"].shape == (T, 1, 3456) and c["vector"].shape == (T, 768) and c["concat"].shape == (T, 4, 72, 128)
    assert all(torch.isfinite(v).all() for v in c.values())
    clip = cond.embedders[0](img)                                   # (1, 1, 1024)
    assert torch.equal(c["crossattn"][7, 0, :1024], clip[0, 0]) and float(c["crossattn"][0, 0, 1024:1152].abs().max()) == 0.0   # no `command`
    assert float(uc["crossattn"].abs().max()) == 0.0 and float(uc["concat"].abs().max()) == 0.0 and torch.equal(uc["vector"], c["vector"])
    mode = cond.embedders[3].encoder.encode(vd["cond_frames"])
    assert torch.equal(c["concat"][3], mode[0])
