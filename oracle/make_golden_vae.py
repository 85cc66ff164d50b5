"""Generate tests/golden/vae_tiny.pt from the REAL reference temporal VAE decoder (build container only).

    python oracle/make_golden_vae.py

* `VideoDecoder` (vwm/modules/autoencoding/temporal_ae.py) is imported unmodified through oracle/ref_shim.py and run on CPU
  in fp32 with vista_amd.synth seeded weights (ch=64 miniature of configs/inference/vista.yaml's decoder_config), once with the
  shipped video_kernel_size [3,1,1] and once with the class default 3 (3x3x3 time_stack / time_mix_conv).
* `DiffusionEngine.decode_first_stage` (vwm/models/diffusion.py:149-180) cannot be imported (pytorch_lightning is absent),
  so its FunctionDef is extracted from the source file with `ast` and executed AS IS against a stand-in `self` carrying
  the real decoder: the overlap-3 clip chunking of the golden is the reference's own code.
Only outputs are stored; weights and latents are regenerated from their seeds where the golden is consumed.
"""
import ast
import contextlib
import io
import os
import sys
import types
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from vista_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TINY = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
            num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def latents(n, h, w, seed):
    return synth.seeded_tensor("vae.z", (n, 4, h, w), seed)


def images(n, h, w, seed):
    return torch.tanh(synth.seeded_tensor("vae.img", (n, 3, h, w), seed))


def ref_decoder(video_kernel_size, seed=0):
    ref_shim.install()
    with contextlib.redirect_stdout(io.StringIO()):
        from vwm.modules.autoencoding.temporal_ae import VideoDecoder
        dec = VideoDecoder(video_kernel_size=video_kernel_size, **TINY).eval()
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    dec.load_state_dict(synth.seeded_state_dict(shapes, seed), strict=True)
    return dec, shapes, VideoDecoder


def ref_decode_first_stage_fn(VideoDecoder):
    src = open(os.path.join(ref_shim.REF_ROOT, "vwm", "models", "diffusion.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "decode_first_stage")
    fn.decorator_list = []
    ns = {"torch": torch, "VideoDecoder": VideoDecoder, "default": lambda v, d: d if v is None else v}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "diffusion.py:decode_first_stage", "exec"), ns)
    return ns["decode_first_stage"]


def main():
    torch.set_grad_enabled(False)
    os.makedirs(GOLD, exist_ok=True)
    res = {"T": 5, "H": 8, "W": 16, "seed_w": 0, "seed_z": 5}
    z = latents(5, 8, 16, 5)
    for tag, vks in (("k311", [3, 1, 1]), ("k333", 3)):
        dec, shapes, VideoDecoder = ref_decoder(vks)
        res["digest_" + tag] = synth.shapes_digest(shapes)
        res["out_" + tag] = dec(z, timesteps=5).clone()
        print(tag, "out", tuple(res["out_" + tag].shape), "rms", float(res["out_" + tag].pow(2).mean().sqrt()))
        if tag == "k311":
            fn = ref_decode_first_stage_fn(VideoDecoder)
            me = types.SimpleNamespace(scale_factor=0.18215, en_and_decode_n_samples_a_time=6, disable_first_stage_autocast=True,
                                       first_stage_model=types.SimpleNamespace(decoder=dec, decode=lambda zz, **kw: dec(zz, **kw)))
            z9 = latents(11, 8, 16, 6) * 0.18215
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res["dfs_11_n6"] = fn(me, z9).clone()         # clips of 6, 6 and (ragged) 5 frames sharing 3 frames each
                me.en_and_decode_n_samples_a_time = 3          # overlap == n_samples -> the plain split branch
                res["dfs_11_n3"] = fn(me, z9).clone()
            print("decode_first_stage", tuple(res["dfs_11_n6"].shape), float(res["dfs_11_n6"].pow(2).mean().sqrt()),
                  float((res["dfs_11_n6"] - res["dfs_11_n3"]).abs().max()))
    # ---- encoder (vwm.modules.diffusionmodules.model.Encoder) + DiagonalGaussianRegularizer, composed as
    # AutoencodingEngine.encode does (autoencoder.py:193-204; the engine class itself needs pytorch_lightning)
    with contextlib.redirect_stdout(io.StringIO()):
        from vwm.modules.autoencoding.regularizers import DiagonalGaussianRegularizer
        from vwm.modules.diffusionmodules.model import Encoder
        enc = Encoder(**TINY).eval()
    eshapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(synth.seeded_state_dict(eshapes, 0), strict=True)
    x = images(5, 64, 128, 7)
    res["digest_enc"] = synth.shapes_digest(eshapes)
    res["enc_moments"] = enc(x).clone()
    torch.manual_seed(1234)
    res["enc_z_sampled"] = DiagonalGaussianRegularizer(sample=True)(res["enc_moments"])[0].clone()
    res["enc_z_mode"] = DiagonalGaussianRegularizer(sample=False)(res["enc_moments"])[0].clone()
    print("encoder moments", tuple(res["enc_moments"].shape), "rms", float(res["enc_moments"].pow(2).mean().sqrt()))
    torch.save({k: (v.half() if k.startswith("dfs") else v) for k, v in res.items()}, os.path.join(GOLD, "vae_tiny.pt"))  # clip outputs fp32
    print("vae_tiny.pt written", os.path.getsize(os.path.join(GOLD, "vae_tiny.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
