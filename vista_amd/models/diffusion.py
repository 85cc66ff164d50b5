"""Latents -> frames: the clip-chunked first-stage decode of the reference's DiffusionEngine
(vwm/models/diffusion.py:149-180, `decode_first_stage`), as a function over a vista_amd VideoDecoder.

Only this method of the engine is rebuilt; the Lightning module around it is out of scope.
"""
import torch

from ..modules.autoencoding.temporal_ae import VideoDecoder


@torch.no_grad()
def decode_first_stage(decoder, z, scale_factor=0.18215, en_and_decode_n_samples_a_time=14, overlap=3):
    """z (frames, 4, h, w) fp32 CUDA latents -> (frames, 3, 8h, 8w) fp32 images.

    Same chunking as the reference: clips of `en_and_decode_n_samples_a_time` frames; when `overlap` < clip length,
    consecutive clips share `overlap` latent frames and the two decodings of the shared frames are averaged.
    (vista.yaml: scale_factor 0.18215, 14 frames at a time; `AutoencodingEngine.decode` is just `self.decoder(z, **kwargs)`,
    vwm/models/autoencoder.py:206-208.)"""
    z = z / scale_factor
    n_samples = z.shape[0] if en_and_decode_n_samples_a_time is None else en_and_decode_n_samples_a_time
    video = isinstance(decoder, VideoDecoder) or getattr(decoder, "is_video_decoder", False)  # the flag lets a test stand in an oracle
    all_out = []
    if overlap < n_samples:
        previous_z = z[:overlap]
        for current_z in z[overlap:].split(n_samples - overlap, dim=0):
            context_z = torch.cat((previous_z, current_z), dim=0)
            kwargs = {"timesteps": current_z.shape[0] + overlap} if video else {}
            if video and context_z.shape[0] != kwargs["timesteps"]:
                # the reference fails here too (einops shape error): a ragged chunk shorter than `overlap` can only be the last one
                raise ValueError(f"decode_first_stage: clip of {context_z.shape[0]} frames but timesteps={kwargs['timesteps']}")
            previous_z = current_z[-overlap:]
            out = decoder(context_z, **kwargs)
            if not all_out:
                all_out.append(out)
            else:
                all_out[-1][-overlap:] = (all_out[-1][-overlap:] + out[:overlap]) / 2
                all_out.append(out[overlap:])
    else:
        for current_z in z.split(n_samples, dim=0):
            out = decoder(current_z, **({"timesteps": current_z.shape[0]} if video else {}))
            all_out.append(out)
    return torch.cat(all_out, dim=0)


@torch.no_grad()
def encode_first_stage(first_stage_model, x, scale_factor=0.18215, en_and_decode_n_samples_a_time=14):
    """Images (frames, 3, H, W) in [-1, 1] -> latents (frames, 4, H/8, W/8) * scale_factor, `en_and_decode_n_samples_a_time`
    frames per encoder call (vwm/models/diffusion.py:182-195). `first_stage_model.encode(x)` as in autoencoder.py:188-204."""
    n_samples = x.shape[0] if en_and_decode_n_samples_a_time is None else en_and_decode_n_samples_a_time
    outs = [first_stage_model.encode(x[i:i + n_samples], scale=scale_factor) for i in range(0, x.shape[0], n_samples)]
    return torch.cat(outs, dim=0)
