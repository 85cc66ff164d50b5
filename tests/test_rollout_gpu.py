"""BASELINE config 4 in miniature on the MI355X: multi-round autoregressive rollout (3 windows of 6 frames, last-3-frame
carry, trajectory action conditioning, triangle guidance, chunked VAE decode between and after the rounds) through
vista_amd.sample_utils.do_sample, against the golden produced by the REAL reference `do_sample` on CPU
(oracle/make_golden_rollout.py; conditioner and RNG stand-ins shared through oracle/rollout_fixture.py).

The driver's plumbing itself (frame indices, carry, re-conditioning, chunked decode) is pinned to fp32 accuracy on CPU by
tests/test_rollout_cpu.py (same do_sample, oracle stand-ins for the GPU parts). Here the whole thing runs on the HIP path.

Tolerance (stated): every round adds the bf16 noise of 3 EDM steps of the UNet and inherits the previous rounds' through the
carried latents and the re-computed conditioning (decoded frame -> image embedding), so the error compounds additively --
measured 3.8e-2 / 6.3e-2 / 9.1e-2 on the new frames of rounds 0 / 1 / 2 (a plumbing slip would be O(1)). Bounds: round r new
frames <= 4.5e-2 * (r + 1); all latents <= 8e-2; decoded frames <= 9e-2. Single windows are held to 4e-2 by
tests/test_model_gpu.py and the decoder to 4e-2 by tests/test_vae_gpu.py.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


def _pipeline():
    from oracle import rollout_fixture as RF
    from oracle.make_golden_vae import TINY
    from test_model_gpu import _sampler, tiny_unet
    from vista_amd import synth
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    from vista_amd.sample_utils import VistaPipeline
    net, _ = tiny_unet()
    wrapper = OpenAIWrapper(net)
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=RF.T)
    dec = VideoDecoder(video_kernel_size=[3, 1, 1], **TINY)
    dec.load_state_dict(synth.seeded_state_dict({k: tuple(v.shape) for k, v in dec.state_dict().items()}, 0))
    pipe = VistaPipeline(wrapper, den, decoder=dec.cuda(), encode_fn=lambda x: x, scale_factor=RF.SCALE, en_and_decode_n_samples_a_time=6)
    P = "vwm.modules.diffusionmodules.guiders."
    sampler = _sampler({"target": P + "TrianglePredictionGuider", "params": {"num_frames": RF.T, "max_scale": 2.5, "min_scale": 1.0}}, RF.STEPS)
    return pipe, sampler, RF


@pytest.mark.parametrize("fused", [False, True])
def test_rollout_three_rounds_vs_reference_do_sample(fused):
    from vista_amd.sample_utils import do_sample
    g = torch.load(os.path.join(GOLD, "rollout_tiny.pt"))
    pipe, sampler, RF = _pipeline()
    vd = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in RF.value_dict0().items()}
    samples, samples_z, _ = do_sample(RF.initial_latents().cuda(), pipe, sampler, vd, RF.ROUNDS, RF.T,
                                      force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"], initial_cond_indices=[0],
                                      device="cuda", get_condition=RF.get_condition, noise_fn=RF.noise_stream(), fused=fused)
    assert samples_z.shape == g["samples_z"].shape == (RF.ROUNDS * (RF.T - 3) + 3, 4, RF.H, RF.W)
    assert samples.shape == g["samples"].shape and samples.min().item() >= 0.0 and samples.max().item() <= 1.0
    ez = rel_l2(samples_z, g["samples_z"])
    # the golden images live in [0,1]; compare on the decoder's [-1,1] scale so that the mean offset does not flatter the number
    ex = rel_l2(samples * 2 - 1, g["samples"].float() * 2 - 1)
    per_round = [rel_l2(samples_z[3 + r * 3: 6 + r * 3], g["samples_z"][3 + r * 3: 6 + r * 3]) for r in range(RF.ROUNDS)]
    print(f"[parity] rollout ({'fused' if fused else 'generic'}): samples_z rel-L2 {ez:.3e} (new frames per round {per_round}), frames rel-L2 {ex:.3e}")
    assert all(e <= 4.5e-2 * (r + 1) for r, e in enumerate(per_round)), per_round
    assert ez <= 8e-2 and ex <= 9e-2
    assert torch.equal(samples_z[0].cpu(), RF.initial_latents()[0]), "frame 0 is the conditioning latent itself (sample[0] = z[0])"


def test_do_sample_needs_a_conditioner():
    from vista_amd.sample_utils import VistaPipeline, do_sample
    pipe = VistaPipeline(None, None, encode_fn=lambda x: x)
    with pytest.raises(ValueError):
        do_sample(torch.zeros(6, 4, 8, 8), pipe, None, {}, 1, 6)
