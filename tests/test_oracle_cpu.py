"""Pins the CPU oracle (oracle/vista_oracle.py) against the golden vectors produced by the REAL reference
(oracle/make_golden.py -> tests/golden/) and, where /root/reference is mounted, against the live reference modules.
fp32 on both sides: tolerance 2e-4 of the output rms (summation-order noise only)."""
import json
import os

import pytest
import torch

from oracle import ref_shim, vista_oracle as O
from oracle.make_golden import unet_inputs
from vista_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return ((a - b).abs().max() / b.pow(2).mean().sqrt()).item()


def test_scalar_kats():
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    for n in (10, 50):
        assert torch.allclose(O.edm_sigmas(n), torch.tensor(kat[f"edm_sigmas_{n}"]), rtol=1e-6, atol=0)
    assert torch.allclose(O.edm_sigmas(7, append_zero=False), torch.tensor(kat["edm_sigmas_7_noappend"]), rtol=1e-6)
    # the values quoted in SURVEY.md 8c
    s10 = O.edm_sigmas(10)
    assert abs(s10[0].item() - 700.00012) < 1e-3 and abs(s10[1].item() - 352.99237) < 1e-3 and s10[-1].item() == 0.0
    for s, ref in kat["vscaling"].items():
        got = [float(v) for v in O.vscaling_edm_cnoise(torch.tensor(float(s)))]
        assert torch.allclose(torch.tensor(got), torch.tensor(ref), rtol=1e-6, atol=1e-9), s
    assert torch.allclose(O.linear_guider_scale(25), torch.tensor(kat["linear_guider_25"]))
    assert torch.allclose(O.triangle_guider_scale(25), torch.tensor(kat["triangle_guider_25"]), atol=1e-6)
    tri = O.triangle_guider_scale(25)
    assert abs(tri[0].item() - 1.0) < 1e-6 and abs(tri[12].item() - 2.5) < 1e-6 and abs(tri[1].item() - 1.125) < 1e-6
    te = O.timestep_embedding(torch.tensor([0.25 * float(torch.tensor(700.0).log()), 0.0, 3.0]), 320)
    assert torch.allclose(te, torch.tensor(kat["timestep_embedding_320"]), atol=1e-6)
    assert abs(kat["x0_scale"] - 700.00073) < 1e-3


def _tiny_sd(model_channels=64):
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    from vista_amd.config import unet_kwargs
    net = VideoUNet(**unet_kwargs(model_channels))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    return synth.seeded_state_dict(shapes, 0), shapes


@pytest.mark.parametrize("tag", ["t5", "t25"])
def test_oracle_unet_matches_reference_golden(tag):
    g = torch.load(os.path.join(GOLD, f"unet_tiny_{tag}.pt"))
    sd, shapes = _tiny_sd()
    assert synth.shapes_digest(shapes) == g["digest"], "state-dict names/shapes drifted from the reference VideoUNet"
    x8, ts, ctx, y, mask = unet_inputs(g["T"], g["H"], g["W"], seed=g["seed_x"], sigma=g["sigma"])
    with torch.no_grad():
        out = O.unet_forward(sd, x8, ts, ctx, y, mask, g["T"])
    assert _rel(out, g["out"]) < 2e-4


def test_oracle_stochastic_sampler_step_matches_reference_golden():
    """The gamma > 0 branch of sampler_step (sampling.py:78-83) with the reference's own recorded noise draws injected
    (oracle/make_golden_churn.py): two of the four steps lie inside [s_tmin, s_tmax] and churn."""
    g = torch.load(os.path.join(GOLD, "sampler_churn_tiny.pt"))
    sd, _ = _tiny_sd()
    T, H, W, prm = g["T"], g["H"], g["W"], g["params"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1, trajectory=[0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2])
    draws = iter(d.float() for d in g["draws"])
    with torch.no_grad():
        out = O.euler_edm_sample(lambda x, sigma, cond, m: O.denoiser_forward(sd, x, sigma, cond, m, T), w["noise"], w["c"], w["uc"], w["cond_frame"],
                                 w["cond_mask"], prm["num_steps"], scale=2.5, s_churn=prm["s_churn"], s_tmin=prm["s_tmin"], s_tmax=prm["s_tmax"],
                                 s_noise=prm["s_noise"], noise_fn=lambda x: next(draws))
        assert next(draws, None) is None, "not every recorded draw was consumed"
        assert _rel(out, g["out"]) < 5e-4
        plain = O.euler_edm_sample(lambda x, sigma, cond, m: O.denoiser_forward(sd, x, sigma, cond, m, T), w["noise"], w["c"], w["uc"],
                                   w["cond_frame"], w["cond_mask"], prm["num_steps"], scale=2.5)
    assert _rel(plain, g["out"]) > 5e-2, "the churn must matter for the fixture to test anything"


def test_oracle_sampler_matches_reference_golden():
    g = torch.load(os.path.join(GOLD, "sampler_tiny.pt"))
    sd, _ = _tiny_sd()
    T, H, W = g["T"], g["H"], g["W"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1, trajectory=[0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2])

    def denoise(x, sigma, cond, cond_mask):
        return O.denoiser_forward(sd, x, sigma, cond, cond_mask, T)

    with torch.no_grad():
        for name, scale, guider in (("vanilla", 2.5, "cfg"), ("linear", O.linear_guider_scale(T), "cfg"),
                                    ("triangle", O.triangle_guider_scale(T), "cfg"), ("identity", None, "identity")):
            out = O.euler_edm_sample(denoise, w["noise"], w["c"], w["uc"], w["cond_frame"], w["cond_mask"], 3, scale=scale, guider=guider)
            assert _rel(out, g[name]) < 5e-4, name
        # rollout-style window: 3 carried-over cond frames, triangle guidance, another trajectory (BASELINE config 4)
        w3 = synth.window_inputs(T=T, H=H, W=W, seed=22, n_cond=3, trajectory=[1.0, 0.2, 2.0, 0.5, 3.0, 0.9, 4.0, 1.4])
        out = O.euler_edm_sample(denoise, w3["noise"], w3["c"], w3["uc"], w3["cond_frame"], w3["cond_mask"], 3, scale=O.triangle_guider_scale(T))
        assert _rel(out, g["rollout3"]) < 5e-4
        assert torch.equal(out[:3], w3["cond_frame"][:3])
        sig = torch.full((T,), 5.0)
        x2, s2, c2, m2 = O.guider_prepare_inputs(w["noise"] * 5.0, sig, w["c"], w["cond_mask"], w["uc"])
        assert _rel(denoise(x2, s2, c2, m2), g["denoiser_out"]) < 2e-4
    assert torch.allclose(g["vanilla_noise_after"], w["noise"] * torch.sqrt(1.0 + O.edm_sigmas(3)[0] ** 2))


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not mounted (GPU box)")
def test_oracle_matches_live_reference_block():
    """Live check of one full-width transformer + resblock pair against the reference modules themselves."""
    import contextlib
    import io
    ref_shim.install()
    with contextlib.redirect_stdout(io.StringIO()):
        from vwm.modules.diffusionmodules.video_model import VideoResBlock
        from vwm.modules.video_attention import SpatialVideoTransformer
        rb = VideoResBlock(channels=320, emb_channels=1280, dropout=0.0, out_channels=640, video_kernel_size=[3, 1, 1],
                           merge_strategy="learned_with_images").eval()
        st = SpatialVideoTransformer(320, 5, 64, depth=1, context_dim=1024, use_linear=True, use_spatial_context=True, ff_in=True,
                                     merge_strategy="learned_with_images", attn_mode="softmax-xformers", action_control=True).eval()
    T, n, hh, ww = 3, 6, 4, 8
    x = torch.randn(n, 320, hh, ww)
    emb = torch.randn(n, 1280)
    ctx = torch.randn(n, 1, synth.CTX_DIM)
    with torch.no_grad():
        for mod, name in ((rb, "rb"), (st, "st")):
            sd = synth.seeded_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()}, 3)
            mod.load_state_dict(sd)
            sdp = {f"{name}.{k}": v for k, v in sd.items()}
            if name == "rb":
                assert _rel(O.video_resblock(sdp, "rb", x, emb, T), mod(x, emb, T)) < 2e-4
            else:
                assert _rel(O.spatial_video_transformer(sdp, "st", x, ctx, T, True), mod(x, ctx, None, T)) < 2e-4


def test_oracle_config1_miniature_matches_reference_golden():
    """BASELINE config 1 in miniature (1 cond frame -> 25 frames, 10 EDM steps, CFG 2.5) from the real reference sampler."""
    g = torch.load(os.path.join(GOLD, "config1_tiny.pt"))
    sd, _ = _tiny_sd()
    T, H, W = g["T"], g["H"], g["W"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1)
    with torch.no_grad():
        out = O.euler_edm_sample(lambda x, s, c, m: O.denoiser_forward(sd, x, s, c, m, T), w["noise"], w["c"], w["uc"], w["cond_frame"],
                                 w["cond_mask"], g["steps"], scale=2.5)
    assert _rel(out, g["out"].float()) < 5e-3  # golden stored in fp16


# ------------------------------------------------------------------------------------ temporal VAE decoder
def _vae_sd(tag):
    from oracle.make_golden_vae import TINY
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    dec = VideoDecoder(video_kernel_size=[3, 1, 1] if tag == "k311" else 3, **TINY)
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    return synth.seeded_state_dict(shapes, 0), shapes


@pytest.mark.parametrize("tag", ["k311", "k333"])
def test_vae_oracle_matches_reference_decoder_golden(tag):
    from oracle import vae_oracle as V
    from oracle.make_golden_vae import latents
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    sd, shapes = _vae_sd(tag)
    assert synth.shapes_digest(shapes) == g["digest_" + tag], "decoder state-dict names/shapes drifted from the reference VideoDecoder"
    with torch.no_grad():
        out = V.video_decoder(sd, latents(g["T"], g["H"], g["W"], g["seed_z"]), g["T"])
    assert _rel(out, g["out_" + tag]) < 2e-4


def test_vae_oracle_decode_first_stage_matches_reference_chunking():
    from oracle import vae_oracle as V
    from oracle.make_golden_vae import latents
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    sd, _ = _vae_sd("k311")
    z = latents(11, g["H"], g["W"], 6) * 0.18215
    with torch.no_grad():
        assert _rel(V.decode_first_stage(sd, z, n_samples=6), g["dfs_11_n6"].float()) < 5e-3   # golden stored in fp16
        assert _rel(V.decode_first_stage(sd, z, n_samples=3), g["dfs_11_n3"].float()) < 5e-3


def test_vae_oracle_encoder_matches_reference_golden():
    from oracle import vae_oracle as V
    from oracle.make_golden_vae import TINY, images
    from vista_amd.modules.diffusionmodules.model import Encoder
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    shapes = {k: tuple(v.shape) for k, v in Encoder(**TINY).state_dict().items()}
    assert synth.shapes_digest(shapes) == g["digest_enc"], "encoder state-dict names/shapes drifted from the reference Encoder"
    sd = synth.seeded_state_dict(shapes, 0)
    with torch.no_grad():
        m = V.encoder(sd, images(5, 64, 128, 7))
    assert _rel(m, g["enc_moments"]) < 2e-4
    torch.manual_seed(1234)
    noise = torch.randn(5, 4, 8, 16)  # what DiagonalGaussianDistribution.sample drew under the same seed
    assert _rel(V.gaussian_sample(m, noise), g["enc_z_sampled"]) < 2e-4
    assert _rel(V.gaussian_sample(m), g["enc_z_mode"]) < 2e-4
