"""Deterministic synthetic weights and conditioning for tests and bench (no dataset / checkpoint is reachable offline).

Everything is generated from numpy's PCG64 keyed on (seed, crc32(name)), so any machine regenerates bit-identical
tensors from names + shapes alone: the GPU box rebuilds exactly the weights the golden vectors were produced with.

* seeded_state_dict: replaces the reference's default init, whose 403 zero-initialised tensors (zero_module convs,
  proj_out, action adapters) would make the network output identically zero and parity vacuous.
* window_inputs: the shapes the conditioner hands to the sampler, including the exact
  parameter-free sinusoid embeddings of ConcatTimestepEmbedderND (vwm/modules/encoders/modules.py:402-425).
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def seeded_tensor(name, shape, seed=0):
    shape = tuple(int(s) for s in shape)
    g = _rng(seed, name)
    x = g.standard_normal(shape, dtype=np.float32)
    if name.endswith("mix_factor"):
        x = 0.5 * x                                    # sigmoid(mix) spread around 0.5
    elif len(shape) >= 2:
        x = x * (float(np.prod(shape[1:])) ** -0.5)    # fan-in scaling keeps activations O(1)
    elif name.endswith(".weight"):
        x = 1.0 + 0.1 * x                              # norm gains
    else:
        x = 0.1 * x                                    # biases
    return torch.from_numpy(np.ascontiguousarray(x))


def seeded_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})."""
    return {k: seeded_tensor(k, s, seed) for k, s in shapes.items()}


def shapes_digest(shapes):
    h = 0
    for k in sorted(shapes):
        h = zlib.crc32((k + str(tuple(int(s) for s in shapes[k]))).encode(), h)
    return h


def sinusoid(values, outdim, max_period=10000.0):
    """Timestep(outdim) of vwm/modules/diffusionmodules/openaimodel.py:287-293 applied to a 1-D float tensor."""
    half = outdim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = values[:, None].float() * freqs[None]
    return torch.cat((torch.cos(args), torch.sin(args)), dim=-1)


def concat_timestep_embed(x, outdim):
    """ConcatTimestepEmbedderND.forward: (b, d) -> (b, d*outdim)."""
    if x.ndim == 1:
        x = x[:, None]
    b, d = x.shape
    return sinusoid(x.reshape(-1), outdim).reshape(b, d * outdim)


CTX_DIM = 1024 + 128 * 19  # CLIP 1024 + command 128 + trajectory 1024 + speed 512 + angle 512 + goal 256


def window_inputs(T=25, H=72, W=128, seed=0, n_cond=1, trajectory=None, fps_id=9.0, motion_bucket=127.0, cond_aug=0.0):
    """One 25-frame sampling window worth of sampler inputs (CPU fp32):
    noise (T,4,H,W); cond_frame z (T,4,H,W); cond_mask (T,); c / uc dicts with crossattn (T,1,3456), vector (T,768),
    concat (T,4,H,W). uc = crossattn and concat zeroed, vector identical (sample.py:243)."""
    def randn(name, shape, s=1.0):
        return torch.from_numpy(_rng(seed, name).standard_normal(shape, dtype=np.float32)) * s
    noise = randn("noise", (T, 4, H, W))
    z = randn("cond_frame", (T, 4, H, W), 0.8)  # VAE latents x scale_factor have roughly this spread
    mask = torch.zeros(T)
    mask[:n_cond] = 1.0
    clip = randn("clip", (1, 1024))
    act = torch.zeros(1, CTX_DIM - 1024)
    if trajectory is not None:
        traj = concat_timestep_embed(torch.tensor([trajectory], dtype=torch.float32), 128)  # (1, 8*128)
        act[:, 128:128 + 1024] = traj
    cross = torch.cat([clip, act], dim=1)[:, None, :].repeat(T, 1, 1)
    vec = torch.cat([concat_timestep_embed(torch.tensor([v]), 256) for v in (fps_id, motion_bucket, cond_aug)], dim=1).repeat(T, 1)
    concat = (z[:1] / 0.18215).repeat(T, 1, 1, 1)
    c = {"crossattn": cross, "vector": vec, "concat": concat}
    uc = {"crossattn": torch.zeros_like(cross), "vector": vec.clone(), "concat": torch.zeros_like(concat)}
    return {"noise": noise, "cond_frame": z, "cond_mask": mask, "c": c, "uc": uc}
