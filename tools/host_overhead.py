"""Host-side cost of enqueueing ONE denoise step (full BASELINE config): wall time of loop.step() from an idle stream to return,
before any synchronisation, plus a cProfile of the same call. The GPU needs ~200 ms for the step, so anything below that is hidden on
one GPU -- but a frame-sharded rank of an 8-GPU run has ~30 ms of GPU work per step, so the enqueue time is what bounds scaling."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler, FusedDenoiser, FusedLoop
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    T, H, W = 25, 72, 128
    net = bench.build_model(320)
    w = synth.window_inputs(T=T, H=H, W=W, seed=0)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    sampler = EulerEDMSampler(num_steps=50, discretization_config=bench.EDM,
                              guider_config={"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}}, device="cuda")
    x, sigmas, _, cond, uc = sampler.prepare_sampling_loop(w["noise"].cuda(), cu(w["c"]), cu(w["uc"]))
    loop = FusedLoop(sampler, FusedDenoiser(den, OpenAIWrapper(net)), x.float().clone(), cond, uc, w["cond_frame"].cuda(), w["cond_mask"].cuda(), True,
                     [float(s) for s in sigmas])
    for i in range(2):
        loop.step(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(2, 6):
        t0 = time.perf_counter()
        loop.step(i)
        ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print("host enqueue ms per step (idle stream):", [round(t, 2) for t in ts])
    pr = cProfile.Profile()
    pr.enable()
    loop.step(6)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
