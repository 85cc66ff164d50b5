"""Window / rollout driver around the sampler (reference: sample_utils.py:279-375, `fill_latent` and `do_sample`).

BASELINE config 4 (long-horizon rollout): round 0 predicts a 25-frame window from the initial conditioning frame(s); every
later round re-conditions on the last three frames of the previous window (their latents are carried over as cond frames,
the decoded third-from-last frame feeds the image embedder) and contributes `num_frames - 3` new frames.

`model` is duck-typed like the reference's DiffusionEngine -- `VistaPipeline` below bundles the vista_amd pieces that way:
    model.model / model.denoiser        network wrapper and Denoiser (called as model.denoiser(model.model, x, sigma, cond, mask))
    model.decode_first_stage(z)         latents -> images            model.encode_first_stage(x)   images -> latents
    model.scale_factor                  model.conditioner.embedders  (skip_encode toggling)        model.ema_scope(...)
The conditioner (vista_amd.modules.encoders.modules.GeneralConditioner: OpenCLIP image tower, first-stage encoder, sinusoid embedders --
SURVEY.md 8f rank 2) plugs in as `VistaPipeline(conditioner=...)`: `get_condition` / `get_batch` below are the reference's
(sample_utils.py:232-276). A caller may still pass its own `get_condition` (same signature) or give the pipeline a `condition_fn`.
"""
import math
import contextlib

import torch

from .models.diffusion import decode_first_stage as _decode_first_stage
from .modules.diffusionmodules.denoiser import Denoiser
from .modules.diffusionmodules.sampling import FusedDenoiser


def fill_latent(cond, length, cond_indices, device):
    """A `length`-frame latent stack that is zero except for `cond` at `cond_indices` (sample_utils.py:279-282)."""
    latent = torch.zeros((length,) + tuple(cond.shape[1:]), device=device, dtype=cond.dtype)
    latent[cond_indices] = cond.to(device)
    return latent


class VistaPipeline:
    """The attributes of DiffusionEngine that `do_sample` touches, over vista_amd modules."""

    def __init__(self, network, denoiser, decoder=None, encode_fn=None, condition_fn=None, scale_factor=0.18215,
                 en_and_decode_n_samples_a_time=14, conditioner=None):
        self.model, self.denoiser = network, denoiser
        self.decoder, self.encode_fn, self.condition_fn = decoder, encode_fn, condition_fn
        self.scale_factor = scale_factor
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        self.conditioner = conditioner if conditioner is not None else _NoEmbedders()
        if conditioner is not None and condition_fn is None:
            self.condition_fn = get_condition

    def ema_scope(self, *_a, **_k):
        return contextlib.nullcontext()

    def decode_first_stage(self, z, overlap=3):
        if self.decoder is None:
            raise RuntimeError("VistaPipeline: no first-stage decoder was given")
        return _decode_first_stage(self.decoder, z, self.scale_factor, self.en_and_decode_n_samples_a_time, overlap)

    def encode_first_stage(self, x):
        if self.encode_fn is None:
            raise RuntimeError("VistaPipeline: no first-stage encoder was given (pass latents through encode_fn=lambda x: x)")
        return self.encode_fn(x)


class _NoEmbedders:
    embedders = ()


def get_batch(keys, value_dict, N, device="cuda"):
    """sample_utils.py:232-253: per-key batch tensors of the demo setups -- scalars (fps_id, motion_bucket_id, cond_aug) repeated prod(N)
    times, action vectors and conditioning frames repeated N[0] times; batch_uc is a clone."""
    batch = {}
    for key in keys:
        if key not in value_dict:
            continue
        if key in ("fps", "fps_id", "motion_bucket_id", "cond_aug"):
            batch[key] = torch.tensor([value_dict[key]]).to(device).repeat(math.prod(N))
        elif key in ("command", "trajectory", "speed", "angle", "goal"):
            v = value_dict[key][None].to(device)
            batch[key] = v.repeat(N[0], *([1] * (v.dim() - 1)))
        elif key in ("cond_frames", "cond_frames_without_noise"):
            v = value_dict[key]
            batch[key] = v.repeat(N[0], *([1] * (v.dim() - 1)))
        else:
            raise NotImplementedError(key)
    batch_uc = {k: torch.clone(v) for k, v in batch.items() if isinstance(v, torch.Tensor)}
    return batch, batch_uc


def get_condition(model, value_dict, num_samples, force_uc_zero_embeddings, device):
    """sample_utils.py:256-276 (without the load_model / unload_model host<->device shuffling: 288 GB of HBM keep the conditioner resident)."""
    keys = list({e.input_key for e in model.conditioner.embedders if getattr(e, "input_key", None) is not None})
    batch, batch_uc = get_batch(keys, value_dict, [num_samples], device)
    c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc, force_uc_zero_embeddings=force_uc_zero_embeddings)
    for k in c:
        if isinstance(c[k], torch.Tensor):
            c[k], uc[k] = c[k][:num_samples].to(device), uc[k][:num_samples].to(device)
            if c[k].shape[0] < num_samples:
                c[k] = c[k][[0]]
            if uc[k].shape[0] < num_samples:
                uc[k] = uc[k][[0]]
    return c, uc


def _set_skip_encode(model, flag):
    for emb in model.conditioner.embedders:
        if hasattr(emb, "skip_encode"):
            emb.skip_encode = flag


@torch.no_grad()
def do_sample(images, model, sampler, value_dict, num_rounds, num_frames, force_uc_zero_embeddings=None, initial_cond_indices=None,
              device="cuda", get_condition=None, noise_fn=None, fused=True):
    """-> (samples in [0,1], samples_z, images). Same contract as the reference's do_sample; two optional hooks on top:
    `get_condition(model, value_dict, num_frames, force_uc_zero_embeddings, device) -> (c, uc)` and `noise_fn(like)`
    (default torch.randn_like) so a test can feed the very noise a CPU reference run drew. `fused`: hand the sampler a
    FusedDenoiser (its one-pass prepare/combine/Euler kernels) instead of the reference's opaque closure; same arithmetic."""
    initial_cond_indices = [0] if initial_cond_indices is None else initial_cond_indices
    force_uc_zero_embeddings = [] if force_uc_zero_embeddings is None else force_uc_zero_embeddings
    get_condition = get_condition or getattr(model, "condition_fn", None)
    if get_condition is None:
        raise ValueError("do_sample: no conditioner -- build the pipeline with conditioner=GeneralConditioner(...) or pass get_condition=")
    noise_fn = noise_fn or torch.randn_like
    carry = 3                                  # frames handed from one window to the next
    fresh = num_frames - carry                 # new frames every later round contributes

    def denoiser(x, sigma, cond, cond_mask):  # the reference's closure (sample_utils.py:314-315)
        return model.denoiser(model.model, x, sigma, cond, cond_mask)
    if fused and isinstance(model.denoiser, Denoiser):
        denoiser = FusedDenoiser(model.denoiser, model.model)

    with model.ema_scope("Sampling"):
        c, uc = get_condition(model, value_dict, num_frames, force_uc_zero_embeddings, device)
        z = model.encode_first_stage(images)
        samples_z = torch.zeros((num_rounds * fresh + carry,) + tuple(z.shape[1:]), device=device, dtype=z.dtype)

        first_mask = torch.zeros(num_frames, device=device)
        first_mask[initial_cond_indices] = 1
        carry_mask = torch.zeros(num_frames, device=device)
        carry_mask[:carry] = 1

        # round 0: the window grows out of the initial conditioning frame(s); the sampler rescales cond_frame itself
        sample = sampler(denoiser, noise_fn(z), cond=c, uc=uc, cond_frame=z, cond_mask=first_mask)
        sample[0] = z[0]
        samples_z[:num_frames] = sample

        for n in range(1, num_rounds):
            # re-condition: decoded third-from-last frame -> image embedder, its latent -> concat conditioning (no re-encode)
            tail_images = model.decode_first_stage(sample[-14:])
            value_dict["cond_frames_without_noise"] = tail_images[[-carry]]
            value_dict["cond_frames"] = sample[[-carry]] / model.scale_factor
            _set_skip_encode(model, True)
            try:
                c, uc = get_condition(model, value_dict, num_frames, force_uc_zero_embeddings, device)
            finally:
                _set_skip_encode(model, False)
            seeded = fill_latent(sample[-carry:], num_frames, list(range(carry)), device)
            sample = sampler(denoiser, noise_fn(seeded), cond=c, uc=uc, cond_frame=seeded, cond_mask=carry_mask)
            lo = n * fresh + carry
            samples_z[lo:lo + fresh] = sample[carry:]

        samples_x = model.decode_first_stage(samples_z)
    samples = torch.clamp((samples_x + 1.0) / 2.0, min=0.0, max=1.0)
    return samples, samples_z, images
