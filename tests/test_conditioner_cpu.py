"""Conditioner, host side: the key-routing logic of vista_amd's GeneralConditioner / get_batch / get_condition against the
golden produced by the REAL reference classes (oracle/make_golden_cond.py), the open_clip -> HF name map of the CLIP oracle, the kornia-0.6.9
blur parameters, and the state-dict contract of the image tower. No GPU: embedders that need kernels are replaced by torch stand-ins."""
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _stub_module():
    """Embedders over vista_amd's AbstractEmbModel whose arithmetic is plain torch (the HIP ones are tested on the GPU box)."""
    from oracle import cond_fixture as CF
    from vista_amd import synth
    from vista_amd.modules.encoders.modules import AbstractEmbModel
    m = types.ModuleType("cond_stub_v")

    class StubImageEmbedder(AbstractEmbModel):
        def __init__(self, dim):
            super().__init__()
            self.dim = dim

        def forward(self, img):
            return CF.stub_image_embed(img, self.dim)

    class StubLatentEmbedder(AbstractEmbModel):
        def forward(self, z):
            return z * 1.0

    class TorchConcatTimestepEmbedderND(AbstractEmbModel):
        def __init__(self, outdim, num_features=None, add_sequence_dim=False):
            super().__init__()
            self.outdim, self.num_features, self.add_sequence_dim = outdim, num_features, add_sequence_dim

        def forward(self, x):
            e = synth.concat_timestep_embed(x.float().cpu(), self.outdim).to(x.device)
            return e[:, None] if self.add_sequence_dim else e

    m.StubImageEmbedder, m.StubLatentEmbedder, m.TorchConcatTimestepEmbedderND = StubImageEmbedder, StubLatentEmbedder, TorchConcatTimestepEmbedderND
    sys.modules["cond_stub_v"] = m
    return m


def test_general_conditioner_routing_matches_reference_golden():
    from oracle import cond_fixture as CF
    from vista_amd.modules.encoders.modules import GeneralConditioner
    from vista_amd.sample_utils import VistaPipeline, get_condition
    _stub_module()
    cfgs = CF.emb_models("cond_stub_v")
    for c in cfgs:
        if c["target"].endswith("ConcatTimestepEmbedderND"):
            c["target"] = "cond_stub_v.TorchConcatTimestepEmbedderND"
    cond = GeneralConditioner(cfgs)
    g = torch.load(os.path.join(GOLD, "cond_general.pt"))
    pipe = VistaPipeline(None, None, conditioner=cond)
    c, uc = get_condition(pipe, CF.value_dict(), CF.N, CF.FORCE_UC_ZERO, "cpu")
    assert pipe.condition_fn is get_condition
    for name, got, want in (("c", c, g["c"]), ("uc", uc, g["uc"])):
        assert set(got) == set(want) == {"crossattn", "vector", "concat"}
        for k in want:
            assert got[k].shape == want[k].shape, (name, k)
            assert torch.allclose(got[k], want[k], atol=2e-6, rtol=0), (name, k, float((got[k] - want[k]).abs().max()))
    # what the routing must have produced: absent actions (command, angle, goal) are zero segments, present ones sinusoids; uc zeroes crossattn/concat
    assert c["crossattn"].shape == (CF.N, 1, 1024 + 128 * 19)
    assert float(c["crossattn"][0, 0, 1024:1024 + 128].abs().max()) == 0.0 and float(c["crossattn"][0, 0, 1024 + 128:1024 + 128 + 1024].abs().max()) > 0.5
    assert float(uc["crossattn"].abs().max()) == 0.0 and float(uc["concat"].abs().max()) == 0.0 and torch.equal(uc["vector"], c["vector"])


def test_general_conditioner_accepts_training_ucg_rates_like_the_reference():
    """the reference's training YAMLs set ucg_rate 0.15 / legacy_ucg_value on embedders (encoders/modules.py:85-104); they must
    instantiate here, sampling (get_unconditional_conditioning) must run with the dropout disabled and restore the rates (:172-180), and a
    bare forward() -- where the reference would apply training dropout -- must refuse rather than silently skip it."""
    from oracle import cond_fixture as CF
    from vista_amd.modules.encoders.modules import GeneralConditioner
    _stub_module()
    cfgs = CF.emb_models("cond_stub_v")
    for c in cfgs:
        if c["target"].endswith("ConcatTimestepEmbedderND"):
            c["target"] = "cond_stub_v.TorchConcatTimestepEmbedderND"
    plain = GeneralConditioner([dict(c) for c in cfgs])
    cfgs[0]["ucg_rate"] = 0.15
    cfgs[1]["ucg_rate"] = 0.15
    cfgs[1]["legacy_ucg_value"] = None
    cond = GeneralConditioner(cfgs)
    assert [e.ucg_rate for e in cond.embedders][:2] == [0.15, 0.15]
    g = torch.load(os.path.join(GOLD, "cond_general.pt"))
    batch = {k: v for k, v in g["batch"].items()} if "batch" in g else None
    if batch is not None:
        c1, u1 = cond.get_unconditional_conditioning(batch, force_uc_zero_embeddings=["cond_frames"])
        c0, u0 = plain.get_unconditional_conditioning(batch, force_uc_zero_embeddings=["cond_frames"])
        assert all(torch.equal(c1[k], c0[k]) and torch.equal(u1[k], u0[k]) for k in c0)
        with pytest.raises(NotImplementedError):
            cond(batch)
    assert [e.ucg_rate for e in cond.embedders][:2] == [0.15, 0.15], "rates restored after sampling"


def test_conditioner_config_of_the_reference_instantiates_this_package():
    """configs/inference/vista.yaml's conditioner_config (vwm.* targets) builds vista_amd classes with the reference's state-dict names."""
    import yaml
    from vista_amd.config import CONFIG_PATH
    from vista_amd.modules.encoders import modules as M
    from vista_amd.util import instantiate_from_config
    cfg = yaml.safe_load(open(CONFIG_PATH))["model"]["params"]["conditioner_config"]
    # shrink the two weight-carrying embedders (632 M + 34 M parameters) to miniature geometries: the test is about plumbing and names
    cfg["params"]["emb_models"][0]["params"]["open_clip_embedding_config"]["params"]["arch"] = dict(width=128, layers=1, heads=2, mlp=256, patch=14,
                                                                                                     image=28, embed=32)
    dd = cfg["params"]["emb_models"][3]["params"]["encoder_config"]["params"]["ddconfig"]
    dd.update(ch=64, ch_mult=[1, 2], num_res_blocks=1)
    cond = instantiate_from_config(cfg)
    assert isinstance(cond, M.GeneralConditioner) and len(cond.embedders) == 10
    assert [type(e).__name__ for e in cond.embedders[:4]] == ["FrozenOpenCLIPImagePredictionEmbedder", "ConcatTimestepEmbedderND",
                                                              "ConcatTimestepEmbedderND", "VideoPredictionEmbedderWithEncoder"]
    keys = set(cond.state_dict())
    for k in ("embedders.0.open_clip.model.visual.conv1.weight", "embedders.0.open_clip.model.visual.class_embedding",
              "embedders.0.open_clip.model.visual.positional_embedding", "embedders.0.open_clip.model.visual.ln_pre.weight",
              "embedders.0.open_clip.model.visual.transformer.resblocks.0.attn.in_proj_weight",
              "embedders.0.open_clip.model.visual.transformer.resblocks.0.attn.out_proj.bias",
              "embedders.0.open_clip.model.visual.transformer.resblocks.0.mlp.c_fc.weight",
              "embedders.0.open_clip.model.visual.transformer.resblocks.0.mlp.c_proj.bias", "embedders.0.open_clip.model.visual.ln_post.bias",
              "embedders.0.open_clip.model.visual.proj", "embedders.3.encoder.encoder.conv_in.weight", "embedders.3.encoder.quant_conv.weight",
              "embedders.3.encoder.encoder.mid.block_1.norm1.weight"):
        assert k in keys, k
    assert [e.input_key for e in cond.embedders] == ["cond_frames_without_noise", "fps_id", "motion_bucket_id", "cond_frames", "cond_aug", "command",
                                                     "trajectory", "speed", "angle", "goal"]
    with pytest.raises(NotImplementedError):
        M.FrozenOpenCLIPImageEmbedder(arch="ViT-H-14", output_tokens=True)
    assert M.OPENCLIP_VISION_GEOMETRY["ViT-H-14"] == dict(width=1280, layers=32, heads=16, mlp=5120, patch=14, image=224, embed=1024)


def test_clip_oracle_is_pinned_to_hf_and_regenerates_the_golden():
    """The oracle = transformers.CLIPVisionModelWithProjection fed open_clip-named weights; a plain-torch restatement agrees (name map) and the
    committed golden is reproduced."""
    from oracle import clip_oracle as CO
    from oracle.make_golden_cond import TINY
    from vista_amd import synth
    from vista_amd.modules.encoders.modules import FrozenOpenCLIPImageEmbedder
    g = torch.load(os.path.join(GOLD, "clip_tiny.pt"))
    assert g["geometry"] == TINY
    shapes = {k: tuple(v.shape) for k, v in FrozenOpenCLIPImageEmbedder(arch=TINY).state_dict().items()}
    assert synth.shapes_digest(shapes) == g["digest"]
    sd = synth.seeded_state_dict(shapes, g["seed"])
    img = torch.tanh(synth.seeded_tensor("clip.img", g["img_shape"], g["seed"]) * 1.5)
    img[1] = img[1].flip(-1) * 0.7
    pix = CO.preprocess(img, TINY["image"])
    assert torch.allclose(pix[:, :, ::16, ::16], g["pixels_sample"], atol=1e-5)
    out = CO.image_embed(sd, TINY, img)
    assert torch.allclose(out, g["embed"], atol=2e-4, rtol=1e-4)
    assert torch.allclose(CO.restated_visual(sd, TINY, pix), out, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("f", [2, 3])
def test_clip_oracle_resize_agrees_with_an_independent_float64_implementation(f):
    """The kornia-0.6.9 antialias resize is restated in oracle/clip_oracle.py (kornia is not installed: "parity unpinned" against kornia
    itself). What CAN be pinned: the restatement against an implementation that shares no code with it -- scipy's separable Gaussian
    (mode='mirror' = torch 'reflect', radius = kornia's kernel size // 2) + torch's float64 bicubic (a = -0.75, align_corners=True)."""
    import numpy as np
    import torch.nn.functional as F
    from scipy.ndimage import gaussian_filter1d
    from oracle import clip_oracle as CO
    from vista_amd import ops
    H = W = 224 * f
    g = torch.Generator().manual_seed(f)
    img = torch.tanh(torch.randn(1, 3, H, W, generator=g) * 1.5)
    sigma, ks = ops.antialias_blur_params(H, 224)
    a = img.double().numpy()
    a = gaussian_filter1d(a, sigma, axis=2, mode="mirror", radius=ks // 2)
    a = gaussian_filter1d(a, sigma, axis=3, mode="mirror", radius=ks // 2)
    want = F.interpolate(torch.from_numpy(np.ascontiguousarray(a)), size=(224, 224), mode="bicubic", align_corners=True)
    want = torch.stack([((want[0, c] + 1) / 2 - ops.CLIP_MEAN[c]) / ops.CLIP_STD[c] for c in range(3)])
    got = CO.preprocess(img, 224)[0].double()
    assert (got - want).abs().max().item() <= 1e-3


def test_antialias_blur_parameters_follow_kornia_0_6_9():
    from vista_amd import ops
    # 576x1024 -> 224: vertical factor 2.571 -> sigma 0.7857, ks int(max(3.14, 3)) = 3; horizontal 4.571 -> sigma 1.7857, ks int(7.14) = 7
    sy, ky = ops.antialias_blur_params(576, 224)
    sx, kx = ops.antialias_blur_params(1024, 224)
    assert (ky, kx) == (3, 7) and abs(sy - (576 / 224 - 1) / 2) < 1e-12 and abs(sx - (1024 / 224 - 1) / 2) < 1e-12
    assert ops.antialias_blur_params(896, 224) == (1.5, 7)     # 4 sigma = 6 -> made odd
    assert ops.antialias_blur_params(100, 224) == (0.001, 3)   # upscaled axis of a resize that blurs because the OTHER axis shrinks: a delta kernel
