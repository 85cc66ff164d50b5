"""Golden for the reward-estimation path from the REAL reference driver: `do_sample` of
/root/reference/reward_utils.py (:284-341) is extracted with `ast` and executed unmodified on CPU against the real reference
UNet / Denoiser / EulerEDMSampler (VanillaCFG 2.5, as reward.py:236 sets it); stand-ins only for the engine object, the
conditioner and torch.randn_like (oracle/rollout_fixture.py).       python oracle/make_golden_reward.py
"""
import contextlib
import io
import json
import os
import sys
import types
from typing import List, Optional

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, rollout_fixture as RF  # noqa: E402
from oracle.make_golden import CFG, seeded_ref_unet  # noqa: E402
from oracle.make_golden_rollout import TorchProxy, extract  # noqa: E402

ENSEMBLE, N_CONDS = 3, 2


def main():
    torch.set_grad_enabled(False)
    c = ref_shim.ref_classes()
    net, _ = seeded_ref_unet(64, seed=0)
    wrapper = c["OpenAIWrapper"](net)
    den = c["Denoiser"](scaling_config={"target": CFG + "denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=RF.T)
    engine = types.SimpleNamespace(scale_factor=RF.SCALE, first_stage_model=None, conditioner=types.SimpleNamespace(embedders=[]),
                                   denoiser=den, model=wrapper, ema_scope=lambda *_a, **_k: contextlib.nullcontext(),
                                   encode_first_stage=lambda x: x)
    sampler = c["EulerEDMSampler"](
        num_steps=RF.STEPS, discretization_config={"target": CFG + "discretizer.EDMDiscretization", "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
        guider_config={"target": CFG + "guiders.VanillaCFG", "params": {"scale": 2.5}},
        s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cpu")
    ns = {"torch": TorchProxy(RF.noise_stream()), "Optional": Optional, "List": List, "default": lambda v, d: d if v is None else v,
          "autocast": lambda *_a, **_k: contextlib.nullcontext(), "load_model": lambda m: None, "unload_model": lambda m: None,
          "get_condition": RF.get_condition}
    exec(compile(extract(os.path.join(ref_shim.REF_ROOT, "reward_utils.py"), {"do_sample"}), "reward_utils.py", "exec"), ns)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        _, reward = ns["do_sample"](RF.initial_latents(), engine, sampler, RF.value_dict0(), RF.T, ensemble_size=ENSEMBLE,
                                    force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"],
                                    initial_cond_indices=list(range(N_CONDS)), device="cpu")
    out = {"reward": float(reward), "neg_log_reward": float(-torch.log(reward)), "ensemble_size": ENSEMBLE, "n_conds": N_CONDS, "T": RF.T, "steps": RF.STEPS}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "reward_tiny.json"), "w"))
    print(out)


if __name__ == "__main__":
    main()
