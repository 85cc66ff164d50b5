"""Shared fixture of the rollout (BASELINE config 4) parity case. TEST INFRASTRUCTURE ONLY.

The reference's conditioner (OpenCLIP ViT-H image tower + VAE-encoder embedder) cannot run offline, so both the golden
generator (oracle/make_golden_rollout.py, real reference `do_sample`) and the GPU test use this deterministic stand-in for
`get_condition`: it is a pure function of `value_dict`, so everything `do_sample` feeds back between rounds (the decoded frame
-> "CLIP" vector, the carried latent -> concat conditioning, the trajectory) still flows through the real plumbing.
"""
import torch
import torch.nn.functional as F

from vista_amd import synth

T, H, W = 6, 16, 32          # frames per window, latent size
ROUNDS, STEPS = 3, 3
SCALE = 0.18215
CLIP_SEED = 41


def initial_latents():
    return synth.seeded_tensor("rollout.z0", (T, 4, H, W), 3) * 0.8


def value_dict0():
    z = initial_latents()
    return {"cond_frames_without_noise": torch.tanh(synth.seeded_tensor("rollout.img0", (1, 3, 8 * H, 8 * W), 3)),
            "cond_frames": z[[0]] / SCALE, "trajectory": torch.tensor([1.0, 0.2, 2.0, 0.5, 3.0, 0.9, 4.0, 1.4]),
            "fps_id": 9.0, "motion_bucket_id": 127.0, "cond_aug": 0.0}


def get_condition(model, value_dict, num_samples, force_uc_zero_embeddings, device):
    """Stand-in for sample_utils.get_condition (:255-276): same signature, returns (c, uc) with crossattn (n,1,3456),
    vector (n,768), concat (n,4,h,w); uc zeroes crossattn and concat like force_uc_zero_embeddings does."""
    img = value_dict["cond_frames_without_noise"].float()
    proj = synth.seeded_tensor("rollout.clip_proj", (1024, 3 * 8 * 16), CLIP_SEED).to(img.device)
    pooled = F.adaptive_avg_pool2d(img, (8, 16)).reshape(1, -1)             # (1, 384)
    clip = torch.tanh(pooled @ proj.t()) * 1.5                               # (1, 1024) "image embedding"
    act = torch.zeros(1, synth.CTX_DIM - 1024, device=img.device)
    act[:, 128:128 + 1024] = synth.concat_timestep_embed(value_dict["trajectory"][None].float().cpu(), 128).to(img.device)
    cross = torch.cat([clip, act], 1)[:, None, :].repeat(num_samples, 1, 1)
    vec = torch.cat([synth.concat_timestep_embed(torch.tensor([float(value_dict[k])]), 256) for k in ("fps_id", "motion_bucket_id", "cond_aug")],
                    1).repeat(num_samples, 1).to(img.device)
    concat = value_dict["cond_frames"].float().to(img.device).repeat(num_samples, 1, 1, 1)
    c = {"crossattn": cross.to(device), "vector": vec.to(device), "concat": concat.to(device)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"]), "vector": c["vector"].clone(), "concat": torch.zeros_like(c["concat"])}
    return c, uc


def noise_stream():
    """The n-th `torch.randn_like` call of do_sample, reproducible on any device."""
    state = {"n": 0}

    def randn_like(t):
        state["n"] += 1
        return synth.seeded_tensor(f"rollout.noise{state['n']}", tuple(t.shape), 5).to(t.device)
    return randn_like
