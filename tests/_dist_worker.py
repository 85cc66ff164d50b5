"""Rank body for the multi-process tests (launched by `python -m torch.distributed.run ... tests/_dist_worker.py <mode>`):
the tiny UNet's 3-step CFG sampler, frame-sharded over the process group, against the reference's golden output.
Backend: VISTA_DIST_BACKEND = nccl (RCCL, one GPU per rank) or gloo (host-staged; ranks may share GPU 0 via VISTA_FORCE_DEVICE).
Mode "rccl1" (world 1): RCCL refuses two ranks on one GPU ("Duplicate GPU detected", tools/probes/rccl_one_gpu.py), so on a one-GPU box
the only way to put the sharded step's collectives through RCCL itself is a frame-shard group of ONE rank: make_shard's world-1 shortcut
is bypassed and every all_to_all_single / all_reduce / all_gather of the step (and of FrameShard.selfcheck) is issued on the nccl
process group -- call signatures, dtypes, split lists with zeros, async Work handles and stream ordering are RCCL's; no byte crosses xGMI."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "hybrid"
    backend = os.environ.get("VISTA_DIST_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("VISTA_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    from tests.test_model_gpu import _sampler, tiny_unet
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import FusedDenoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    from vista_amd.parallel import DistComm, FrameShard, make_shard
    g = torch.load(os.path.join(ROOT, "tests", "golden", "sampler_tiny.pt"))
    net, _ = tiny_unet()
    T, H, W = g["T"], g["H"], g["W"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1, trajectory=[0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2])
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    fused = FusedDenoiser(den, OpenAIWrapper(net))
    cfg = {"target": "vwm.modules.diffusionmodules.guiders.TrianglePredictionGuider", "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}}

    def make_group(ranks):
        grp = dist.new_group(ranks=ranks)
        return DistComm(grp) if rank in ranks else None
    s = _sampler(cfg)
    if mode == "rccl1":
        assert world == 1 and backend == "nccl"
        s.shard = FrameShard(T, DistComm(dist.group.WORLD, name="frames[0..0]"), B=2)
        s.shard.always_exchange = True   # the UNet forward re-shards through the group although every exchange is a copy to itself
        calls = {"all_to_all": 0, "all_reduce_sum": 0, "all_gather_list": 0}
        for name in calls:   # count what actually went through the process group
            def counted(*a, _f=getattr(s.shard.comm, name), _n=name, **k):
                calls[_n] += 1
                return _f(*a, **k)
            setattr(s.shard.comm, name, counted)
        steps = []
        s.shard.selfcheck("cuda", log=steps.append)
    else:
        s.shard = make_shard(T, world, rank, mode=mode, make_group=make_group)
    cu = lambda d: {k: v.clone().cuda() for k, v in d.items()}  # noqa: E731
    out = s(fused, w["noise"].clone().cuda(), cond=cu(w["c"]), uc=cu(w["uc"]), cond_frame=w["cond_frame"].cuda(),
            cond_mask=w["cond_mask"].cuda()).cpu()
    ref = g["triangle"].float()
    rel = ((out - ref).pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item()
    gathered = [None] * world
    dist.all_gather_object(gathered, (rel, bool(torch.equal(out[0], w["cond_frame"][0])), out.double().sum().item()))
    if rank == 0:
        print(json.dumps({"world": world, "backend": backend, "mode": mode, "t_counts": s.shard.t_counts, "rel_l2": [r[0] for r in gathered],
                          "cond_frame_exact": [r[1] for r in gathered], "checksums": [r[2] for r in gathered],
                          "selfcheck_steps": len(steps) if mode == "rccl1" else None, "collective_calls": calls if mode == "rccl1" else None, "a2a_chunks": s.shard.a2a_chunks}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    assert rel <= 4e-2, rel


if __name__ == "__main__":
    main()
