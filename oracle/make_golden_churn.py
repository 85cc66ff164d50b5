"""Golden for the STOCHASTIC branch of EulerEDMSampler.sampler_step (reference sampling.py:78-83, gamma > 0), which Vista's own configs never
enable (s_churn = 0, sample_utils.py:212) but the sampler API carries. Runs only where /root/reference exists.

    python oracle/make_golden_churn.py

The reference's EulerEDMSampler (imported unmodified through oracle/ref_shim.py, CPU fp32, seeded 64-channel VideoUNet) runs 4 steps with
s_churn = 1.2 (gamma = min(1.2 / 4, sqrt(2) - 1) = 0.3), s_tmin = 0.05, s_tmax = 400, s_noise = 1.003 -- so the first step (sigma = 700 >
s_tmax) and the last one (sigma = 0.002 < s_tmin) are deterministic and the two in between draw noise. `torch.randn_like` is wrapped for the
duration of the call so that every tensor the reference draws is RECORDED; the fixture stores the draws (fp16-exact: they are rounded to fp16
BEFORE the reference uses them) and the output, and the consumers (oracle CPU test, HIP GPU test) inject the same draws through the
sampler's `noise_fn` hook. tests/golden/sampler_churn_tiny.pt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from oracle.make_golden import CFG, GOLD, seeded_ref_unet  # noqa: E402
from vista_amd import synth  # noqa: E402

PARAMS = {"num_steps": 4, "s_churn": 1.2, "s_tmin": 0.05, "s_tmax": 400.0, "s_noise": 1.003}
T, H, W, SEED_X = 5, 16, 32, 31


def main():
    c = ref_shim.ref_classes()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    net, shapes = seeded_ref_unet(64, seed=0)
    w = synth.window_inputs(T=T, H=H, W=W, seed=SEED_X, n_cond=1, trajectory=[0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2])
    wrapper = c["OpenAIWrapper"](net)
    den = c["Denoiser"](scaling_config={"target": CFG + "denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    sampler = c["EulerEDMSampler"](discretization_config={"target": CFG + "discretizer.EDMDiscretization",
                                                          "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                                   guider_config={"target": CFG + "guiders.VanillaCFG", "params": {"scale": 2.5}}, verbose=False, device="cpu", **PARAMS)
    draws = []
    real = torch.randn_like
    gen = torch.Generator().manual_seed(77)

    def recording_randn_like(x, **kw):
        e = torch.randn(x.shape, generator=gen).half().float()   # fp16-exact so the fixture can store it in half the bytes, losslessly
        draws.append(e.clone())
        return e.to(x.dtype)
    torch.randn_like = recording_randn_like
    try:
        out = sampler(lambda x, s, cc, m: den(wrapper, x, s, cc, m), w["noise"].clone(), cond={k: v.clone() for k, v in w["c"].items()},
                      uc={k: v.clone() for k, v in w["uc"].items()}, cond_frame=w["cond_frame"], cond_mask=w["cond_mask"])
    finally:
        torch.randn_like = real
    assert len(draws) == 2, f"expected 2 stochastic steps, the reference drew {len(draws)}"
    torch.save({"out": out.clone(), "draws": [d.half() for d in draws], "params": PARAMS, "T": T, "H": H, "W": W, "seed_x": SEED_X,
                "digest": synth.shapes_digest(shapes)}, os.path.join(GOLD, "sampler_churn_tiny.pt"))
    print(f"sampler_churn_tiny: {len(draws)} draws, out rms {out.pow(2).mean().sqrt():.4f}")


if __name__ == "__main__":
    main()
