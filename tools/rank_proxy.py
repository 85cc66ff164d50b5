"""Compute-only proxy of ONE rank of an N-GPU run on a single GPU: the rank's FrameShard is built with a communicator that
moves no data (receive buffers are zero-filled locally), so kernel shapes, launch counts and host work are those of the real
rank while the collectives cost nothing. Tells how far per-rank compute is from (single-GPU step time / ideal speed-up).

    python tools/rank_proxy.py --world 8 [--rank 0] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class MockComm:
    def __init__(self, ranks, rank):
        self.world, self.rank = len(ranks), ranks.index(rank)

    def all_to_all(self, recv, send, out_splits, in_splits, async_op=False):
        recv.zero_()

    def all_reduce_sum(self, t):
        t.mul_(self.world)

    def all_gather_list(self, t, counts):
        return [t[:c] if c <= t.shape[0] else torch.zeros((c,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for c in counts]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--mode", default="hybrid")
    ap.add_argument("--cprofile", action="store_true")
    ap.add_argument("--chunks", type=int, default=1, help="VISTA_A2A_CHUNKS: temporal block on pixel sub-ranges (compute-side cost of the overlap option)")
    ap.add_argument("--graph", action="store_true", help="replay the UNet forward from one captured hipGraph (VISTA_HIPGRAPH=force: the mock collectives are plain kernels)")
    ap.add_argument("--torch-profile", action="store_true", help="per-kernel table of the timed steps only (no model-build / packing kernels)")
    a = ap.parse_args()
    os.environ["VISTA_A2A_CHUNKS"] = str(a.chunks)
    if a.graph:
        os.environ["VISTA_HIPGRAPH"] = "force"
    import bench
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler, FusedDenoiser, FusedLoop
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    from vista_amd.parallel import make_shard
    torch.cuda.set_device(0)
    T, H, W = 25, 72, 128
    shard = make_shard(T, a.world, a.rank, mode=a.mode, make_group=lambda ranks: MockComm(ranks, a.rank) if a.rank in ranks else None)
    net = bench.build_model(320)
    w = synth.window_inputs(T=T, H=H, W=W, seed=0)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    sampler = EulerEDMSampler(num_steps=50, discretization_config={"target": "vwm.modules.diffusionmodules.discretizer.EDMDiscretization",
                                                                   "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                              guider_config={"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}}, device="cuda")
    x, sigmas, _, cond, uc = sampler.prepare_sampling_loop(w["noise"].cuda(), cu(w["c"]), cu(w["uc"]))
    loop = FusedLoop(sampler, FusedDenoiser(den, OpenAIWrapper(net)), x.float().clone(), cond, uc, w["cond_frame"].cuda(), w["cond_mask"].cuda(),
                     True, [float(s) for s in sigmas], shard=shard)
    loop.step(0)
    torch.cuda.synchronize()
    if a.cprofile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(1, 1 + a.steps):
            loop.step(i)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        return
    if a.torch_profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(1, 1 + a.steps):
                loop.step(i)
            torch.cuda.synchronize()
        rows = sorted(((e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[2])
        tot = sum(r[2] for r in rows)
        print(f"# {a.steps} steps, world {a.world} mode {a.mode}: kernel time {tot / 1e3 / a.steps:.2f} ms/step, {sum(r[1] for r in rows) / a.steps:.0f} launches/step")
        for k, c, t in rows[:60]:
            print(f"{t / 1e3 / a.steps:8.3f} ms/step {c / a.steps:7.1f} calls/step  avg {t / c:8.1f} us  {k[:140]}")
        return
    t0 = time.perf_counter()
    for i in range(1, 1 + a.steps):
        loop.step(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"world": a.world, "rank": a.rank, "mode": a.mode, "frames_local": shard.t_local, "cfg_half": shard.cfg_half,
                      "ms_per_step_compute_only": round(dt * 1e3 / a.steps, 2), "host_enqueue_ms_per_step": round(t_host * 1e3 / a.steps, 2)}))


if __name__ == "__main__":
    main()
