"""Headline benchmark: denoise steps/s of the EulerEDM x VideoUNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one EulerEDMSampler.sampler_step (reference sampling.py:78-89): mask replace -> CFG-doubled UNet forward on
50 images (2 x 25 frames, latent 4x72x128 = 576x1024 pixels) -> guider combine -> Euler update, on the 50-step
sigma schedule. Inputs are synthetic and resident in HBM before the timed region; weights are random-init for the shipped
1.65 B-parameter configuration (every tensor non-zero). Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP_CFG = 1.604e14   # algorithmic FLOP of one CFG step at N=50, 72x128 (torch flop counter on the reference)
ATTN_L0_FLOP = 4.0 * 250 * 9216 * 9216 * 64  # spatial self-attention, level 0: (B*h, N, N, d) = (250, 9216, 9216, 64)
MFMA_BF16_PEAK = 2.5e15        # dense bf16 MFMA peak, MI355X_MICROARCH.md
METRIC = "denoise steps/sec, 25-frame 576x1024 latent, 50-step EDM @ 1/2/4/8 GPU"


def build_model(model_channels, seed=0):
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    with torch.device("cuda"):
        net = VideoUNet(**unet_kwargs(model_channels))
    net = net.cuda()
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():  # re-randomise EVERY tensor: the default init zeroes 403 of them and the net would output 0
        for name, p in net.named_parameters():
            if name.endswith("mix_factor"):
                p.normal_(0, 0.5, generator=g)
            elif p.dim() >= 2:
                p.normal_(0, float(p[0].numel()) ** -0.5, generator=g)
            elif name.endswith(".weight"):
                p.normal_(1.0, 0.1, generator=g)
            else:
                p.normal_(0, 0.1, generator=g)
    return net.eval()


def cpu_baseline(net, T, sample_hw, seed):
    """Times the CPU oracle (fp32 restatement of the reference, oracle/vista_oracle.py) on the host cores for one CFG
    UNet forward at a reduced latent, counts its FLOPs with torch's flop counter and extrapolates to the full-size step
    by FLOP ratio. Also returns the GPU-vs-oracle parity at that sample."""
    from torch.utils.flop_counter import FlopCounterMode
    from oracle import vista_oracle as O
    from oracle.make_golden import unet_inputs
    h, w = sample_hw
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    x8, ts, ctx, y, mask = unet_inputs(T, h, w, seed=seed, sigma=7.0)
    cores = torch.get_num_threads()
    with torch.no_grad():
        with FlopCounterMode(display=False) as fc:
            t0 = time.perf_counter()
            ref = O.unet_forward(sd, x8, ts, ctx, y, mask, T)
            dt = time.perf_counter() - t0
    flops = float(fc.get_total_flops())
    out = net(x8.cuda(), timesteps=ts.cuda(), context=ctx.cuda(), y=y.cuda(), cond_mask=mask.cuda(), num_frames=T).cpu()
    rel = ((out - ref).pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item()
    steps_per_s = (flops / dt) / FLOP_PER_STEP_CFG
    return {"value": steps_per_s, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"1 CFG UNet forward (N={2*T} images, full-width 1.65B weights) at latent {h}x{w} on the host: {dt:.1f} s, "
                      f"{flops/1e12:.2f} TFLOP counted -> {flops/dt/1e12:.3f} TFLOP/s, extrapolated to the 1.604e14-FLOP full-size step",
            "parity_rel_l2_at_sample": rel}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--latent-h", type=int, default=72)
    ap.add_argument("--latent-w", type=int, default=128)
    ap.add_argument("--model-channels", type=int, default=320)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=["hybrid", "frames"], default="hybrid")
    ap.add_argument("--cpu-sample", type=str, default="16x32")
    ap.add_argument("--fp8-ff", action="store_true",
                    help="BASELINE config 5 experiment, NOT the headline: FeedForward GEMMs in fp8 e4m3 (reported dtype says so)")
    args = ap.parse_args()
    if args.fp8_ff:
        from vista_amd.modules import attention as _att
        _att.FP8["feedforward"] = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VISTA_DIST_BACKEND", "nccl")  # "gloo" + VISTA_FORCE_DEVICE=0: dry run of the N>1 path on one GPU
    if "VISTA_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["VISTA_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    shard = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    from vista_amd import _lib, ops, synth
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler, FusedDenoiser, FusedLoop
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    _lib.load()

    T, H, W = args.frames, args.latent_h, args.latent_w
    if world > 1:
        from vista_amd.parallel import DistComm, make_shard

        def make_group(ranks):  # collective: every rank creates every group, members get a communicator
            g = dist.new_group(ranks=ranks)
            return DistComm(g) if rank in ranks else None
        # 'hybrid' (default): CFG halves x frame groups -- 8 GPUs = 2 x (7/6/6/6); 'frames': 4/3/3/3/3/3/3/3 (BASELINE config 3)
        shard = make_shard(T, world, rank, mode=args.shard, make_group=make_group)
    net = build_model(args.model_channels)
    w = synth.window_inputs(T=T, H=H, W=W, seed=0)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    sampler = EulerEDMSampler(num_steps=50, discretization_config={"target": "vwm.modules.diffusionmodules.discretizer.EDMDiscretization",
                                                                   "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                              guider_config={"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}}, device="cuda")
    noise = w["noise"].cuda()
    x, sigmas, _, cond, uc = sampler.prepare_sampling_loop(noise, cu(w["c"]), cu(w["uc"]))
    sig = [float(s) for s in sigmas]
    loop = FusedLoop(sampler, FusedDenoiser(den, OpenAIWrapper(net)), x.float().clone(), cond, uc, w["cond_frame"].cuda(),
                     w["cond_mask"].cuda(), True, sig, shard=shard)
    nsteps = len(sig) - 1
    assert args.warmup + args.steps <= nsteps, "at most 50 steps per window"
    for i in range(args.warmup):
        loop.step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    ops.PROFILE_ATTN = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loop.step(i)
    t_enqueue = time.perf_counter() - t0  # host time to enqueue the steps (before any sync)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:  # MAX over ranks
        tdt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
    prof, ops.PROFILE_ATTN = ops.PROFILE_ATTN, None
    assert torch.isfinite(loop.xw).all(), "non-finite latents"
    ms_per_step = dt * 1e3 / args.steps
    value = args.steps / dt
    full = (T, H, W, args.model_channels) == (25, 72, 128, 320)

    l0 = [(e0.elapsed_time(e1), nbh) for (S, nbh, e0, e1) in prof if S == H * W]
    roofline = None
    if l0:
        nbh = l0[0][1]  # (local images) x heads of this rank's level-0 launches
        l0 = [t for t, _ in l0]
        avg_ms = sum(l0) / len(l0)
        flop = 4.0 * nbh * float(H * W) ** 2 * 64
        ach = flop / (avg_ms * 1e-3) / 1e12
        traffic = None  # HBM bytes per launch from the separate PMC passes (profiles/r01_attn_traffic.json), full config only
        tp = os.path.join(ROOT, "profiles", "r01_attn_traffic.json")
        if full and world == 1 and os.path.exists(tp):
            traffic = json.load(open(tp))["traffic_bytes_per_launch"]
        roofline = {"kernel": "attn_spatial_kernel (level-0 spatial self-attention)", "bound": "mfma", "achieved": ach,
                    "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / (MFMA_BF16_PEAK / 1e12), "traffic": traffic,
                    "launches_timed": len(l0), "avg_ms": avg_ms, "flop_per_launch": flop}
    res = {
        "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16 + fp8(e4m3) FeedForward GEMMs [config 5 experiment]" if args.fp8_ff else "bf16",
        "data": "synthetic" if backend == "nccl" or world == 1 else "synthetic (DRY RUN: gloo host-staged transport, ranks share one GPU -- not a result)",
        "config": {"workload": (f"{world}xMI355X" + ((" CFG-split x2 x" if shard.cfg_half is not None else "") + " frame-sharded " +
                                                     "/".join(str(c) for c in shard.t_counts) if shard else "") +
                                ": 25x576x1024 (latent 25x4x72x128), 50-step EulerEDM, VanillaCFG 2.5 (N=50 images per UNet call), "
                                "bf16, random-init 1.65B VideoUNet, synthetic latents") if full else
                   f"REDUCED (not the BASELINE config): T={T} latent {H}x{W} model_channels={args.model_channels}",
                   "frames": T, "latent": [4, H, W], "cfg_images_per_call": 2 * T, "sampler": "EulerEDM s_churn=0, 50-step schedule",
                   "windows_per_s": value / 50.0,
                   "parallelism": "single GPU" if shard is None else
                   (("CFG halves on 2 rank groups (one output exchange per step) x " if shard.cfg_half is not None else "") +
                    f"frame-shard x{shard.P} (spatial half) + pixel-shard x{shard.P} (temporal half), 2 RCCL all-to-alls per block pair, "
                    "weights replicated")},
        "roofline": roofline,
        "host_enqueue_ms_per_step": t_enqueue * 1e3 / args.steps,
        "step_mfma_frac": (FLOP_PER_STEP_CFG / (ms_per_step * 1e-3) / (MFMA_BF16_PEAK * world)) if full else None,
    }
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        h, wd = (int(v) for v in args.cpu_sample.split("x"))
        res["cpu_baseline"] = cpu_baseline(net, T, (h, wd), seed=1)
    else:
        res["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
