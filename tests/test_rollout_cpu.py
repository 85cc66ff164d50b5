"""The rollout driver (vista_amd.sample_utils.do_sample / fill_latent / VistaPipeline: host logic, device-agnostic) on CPU
with the fp32 ORACLE standing in for the GPU parts (sampler+denoiser: oracle.vista_oracle, decoder: oracle.vae_oracle), against
the golden produced by the REAL reference `do_sample` (oracle/make_golden_rollout.py). fp32 on both sides -> 2e-3."""
import os

import torch

from oracle import rollout_fixture as RF, vae_oracle as V, vista_oracle as O
from vista_amd import synth
from vista_amd.sample_utils import VistaPipeline, do_sample, fill_latent

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return ((a.float() - b.float()).pow(2).sum() / b.float().pow(2).sum()).sqrt().item()


def test_fill_latent():
    cond = torch.arange(2 * 3 * 1 * 1, dtype=torch.float32).view(2, 3, 1, 1)
    out = fill_latent(cond, 5, [0, 3], "cpu")
    assert out.shape == (5, 3, 1, 1) and torch.equal(out[[0, 3]], cond) and out[[1, 2, 4]].abs().sum() == 0


def test_do_sample_plumbing_matches_reference_driver():
    from oracle.make_golden_vae import TINY
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    g = torch.load(os.path.join(GOLD, "rollout_tiny.pt"))
    usd = synth.seeded_state_dict({k: tuple(v.shape) for k, v in VideoUNet(**unet_kwargs(64)).state_dict().items()}, 0)
    dsd = synth.seeded_state_dict({k: tuple(v.shape) for k, v in VideoDecoder(video_kernel_size=[3, 1, 1], **TINY).state_dict().items()}, 0)

    class OracleDecoder:
        is_video_decoder = True

        def __call__(self, z, timesteps):
            return V.video_decoder(dsd, z, timesteps)

    def oracle_sampler(denoiser, x, cond, uc=None, cond_frame=None, cond_mask=None):
        # ignores the closure: the oracle's sampler + denoiser restate EulerEDMSampler/Denoiser/VideoUNet themselves
        return O.euler_edm_sample(lambda xx, s, c, m: O.denoiser_forward(usd, xx, s, c, m, RF.T), x, cond, uc, cond_frame, cond_mask,
                                  RF.STEPS, scale=O.triangle_guider_scale(RF.T))

    pipe = VistaPipeline(None, None, decoder=OracleDecoder(), encode_fn=lambda x: x, scale_factor=RF.SCALE, en_and_decode_n_samples_a_time=6)
    with torch.no_grad():
        samples, samples_z, _ = do_sample(RF.initial_latents(), pipe, oracle_sampler, RF.value_dict0(), RF.ROUNDS, RF.T,
                                          force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"], initial_cond_indices=[0],
                                          device="cpu", get_condition=RF.get_condition, noise_fn=RF.noise_stream(), fused=False)
    assert _rel(samples_z, g["samples_z"]) < 2e-3
    assert _rel(samples, g["samples"]) < 2e-3  # golden images stored in fp16
