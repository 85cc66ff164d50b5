"""Golden for the multi-round rollout (BASELINE config 4 in miniature) from the REAL reference driver:
`do_sample` and `fill_latent` are extracted from /root/reference/sample_utils.py with `ast` and executed unmodified on CPU
against the real reference UNet / Denoiser / EulerEDMSampler (triangle guider) / VideoDecoder / decode_first_stage; only the
things that cannot run offline are stood in: the conditioner (oracle/rollout_fixture.get_condition), the engine object
(a namespace exposing the attributes do_sample touches) and torch.randn_like (seeded stream).   python oracle/make_golden_rollout.py
"""
import ast
import contextlib
import io
import os
import sys
import types
from typing import List, Optional

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, rollout_fixture as RF  # noqa: E402
from oracle.make_golden import CFG, seeded_ref_unet  # noqa: E402
from oracle.make_golden_vae import ref_decode_first_stage_fn, ref_decoder  # noqa: E402


def extract(path, names):
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    for f in fns:
        f.decorator_list = []
    return ast.Module(body=fns, type_ignores=[])


class TorchProxy:
    """`torch` as do_sample sees it: everything real except a reproducible randn_like."""

    def __init__(self, randn_like):
        self.randn_like = randn_like

    def __getattr__(self, k):
        return getattr(torch, k)


def main():
    torch.set_grad_enabled(False)
    c = ref_shim.ref_classes()
    net, _ = seeded_ref_unet(64, seed=0)
    wrapper = c["OpenAIWrapper"](net)
    den = c["Denoiser"](scaling_config={"target": CFG + "denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=RF.T)
    dec, _, VideoDecoder = ref_decoder([3, 1, 1], seed=0)
    dfs = ref_decode_first_stage_fn(VideoDecoder)
    engine = types.SimpleNamespace(scale_factor=RF.SCALE, en_and_decode_n_samples_a_time=6, disable_first_stage_autocast=True,
                                   first_stage_model=types.SimpleNamespace(decoder=dec, decode=lambda z, **kw: dec(z, **kw)),
                                   conditioner=types.SimpleNamespace(embedders=[]), denoiser=den, model=wrapper,
                                   ema_scope=lambda *_a, **_k: contextlib.nullcontext(), encode_first_stage=lambda x: x)
    engine.decode_first_stage = lambda z, **kw: dfs(engine, z, **kw)
    sampler = c["EulerEDMSampler"](
        num_steps=RF.STEPS, discretization_config={"target": CFG + "discretizer.EDMDiscretization", "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
        guider_config={"target": CFG + "guiders.TrianglePredictionGuider", "params": {"num_frames": RF.T, "max_scale": 2.5, "min_scale": 1.0}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cpu")
    ns = {"torch": TorchProxy(RF.noise_stream()), "Optional": Optional, "List": List, "default": lambda v, d: d if v is None else v,
          "autocast": lambda *_a, **_k: contextlib.nullcontext(), "tqdm": lambda **_k: types.SimpleNamespace(update=lambda n: None),
          "load_model": lambda m: None, "unload_model": lambda m: None, "get_condition": RF.get_condition}
    exec(compile(extract(os.path.join(ref_shim.REF_ROOT, "sample_utils.py"), {"do_sample", "fill_latent"}), "sample_utils.py", "exec"), ns)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        samples, samples_z, _ = ns["do_sample"](RF.initial_latents(), engine, sampler, RF.value_dict0(), RF.ROUNDS, RF.T,
                                                force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"],
                                                initial_cond_indices=[0], device="cpu")
    print("samples", tuple(samples.shape), "samples_z", tuple(samples_z.shape), "rms z", float(samples_z.pow(2).mean().sqrt()),
          "mean img", float(samples.mean()))
    torch.save({"samples": samples.half(), "samples_z": samples_z.clone(), "T": RF.T, "rounds": RF.ROUNDS, "steps": RF.STEPS},
               os.path.join(ROOT, "tests", "golden", "rollout_tiny.pt"))
    print("rollout_tiny.pt", os.path.getsize(os.path.join(ROOT, "tests", "golden", "rollout_tiny.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
