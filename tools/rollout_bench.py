"""Full-size end-to-end run of the rollout path (BASELINE config 4) on cuda:0: vista_amd.sample_utils.do_sample with the shipped
1.65 B VideoUNet and the shipped temporal VAE decoder (seeded random weights, synthetic conditioning), 25-frame 576x1024 windows.

    python tools/rollout_bench.py [--rounds 2] [--steps 5]
Reports wall time per phase and peak device memory (the reference needs ~66 GB for the decode alone, docs/ISSUES.md:5-10)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import bench
    from tools.vae_bench import SHIPPED
    from vista_amd import synth
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    from vista_amd.sample_utils import VistaPipeline, do_sample
    torch.cuda.set_device(0)
    T, H, W = 25, 72, 128
    net = bench.build_model(320)
    dec = VideoDecoder(**SHIPPED)
    dec.load_state_dict(synth.seeded_state_dict({k: tuple(v.shape) for k, v in dec.state_dict().items()}, 0))
    dec.cuda()
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    P = "vwm.modules.diffusionmodules."
    sampler = EulerEDMSampler(num_steps=a.steps, discretization_config={"target": P + "discretizer.EDMDiscretization",
                                                                        "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                              guider_config={"target": P + "guiders.TrianglePredictionGuider", "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}},
                              device="cuda")
    w = synth.window_inputs(T=T, H=H, W=W, seed=0, n_cond=1, trajectory=[1.0, 0.2, 2.0, 0.5, 3.0, 0.9, 4.0, 1.4])
    c = {k: v.cuda() for k, v in w["c"].items()}
    uc = {k: v.cuda() for k, v in w["uc"].items()}
    timings = {"condition_calls": 0}

    def get_condition(model, value_dict, n, force_uc, device):  # synthetic conditioner: the CLIP tower is out of scope
        timings["condition_calls"] += 1
        cc = dict(c)
        cc["concat"] = value_dict["cond_frames"].float().to(device).repeat(n, 1, 1, 1) if "cond_frames" in value_dict else c["concat"]
        return cc, uc

    pipe = VistaPipeline(OpenAIWrapper(net), den, decoder=dec, encode_fn=lambda x: x, condition_fn=get_condition)
    z0 = w["cond_frame"].cuda()
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples, samples_z, _ = do_sample(z0, pipe, sampler, {}, a.rounds, T, device="cuda")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert samples.shape == (a.rounds * (T - 3) + 3, 3, 8 * H, 8 * W) and torch.isfinite(samples).all()
    print(json.dumps({"rounds": a.rounds, "steps_per_round": a.steps, "frames_out": samples.shape[0], "wall_s": round(dt, 2),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "condition_calls": timings["condition_calls"],
                      "note": "first round includes one-time weight packing"}))


if __name__ == "__main__":
    main()
