"""ONE UNet forward of the BASELINE configuration itself -- shipped 1.65 B-parameter network, 25 frames, latent 72x128 (9216 tokens at
level 0) -- on the MI355X against the CPU fp32 oracle (oracle/vista_oracle.py), one clip (n_img = 25: the CFG-doubled batch of the bench is
two independent clips through the same function). The oracle forward is ~8e13 FLOP on the host (minutes), so this is a tool run once per
round, not a test; the result goes to profiles/.   usage: python tools/full_size_parity.py [out.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import vista_oracle as O
    from oracle.make_golden import unet_inputs
    from vista_amd import synth
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    T, H, W = 25, 72, 128
    net = VideoUNet(**unet_kwargs(320))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.seeded_state_dict(shapes, 0)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    x8, ts, ctx, y, mask = unet_inputs(T, H, W, seed=11, sigma=9.0)
    sl = slice(T, 2 * T)  # the cond half: non-zero context / concat conditioning
    x8, ts, ctx, y, mask = x8[sl], ts[sl], ctx[sl], y[sl], mask[sl]
    t0 = time.time()
    out = net(x8.cuda(), timesteps=ts.cuda(), context=ctx.cuda(), y=y.cuda(), cond_mask=mask.cuda(), num_frames=T).cpu()
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    t0 = time.time()
    with torch.no_grad():
        ref = O.unet_forward(sd, x8, ts, ctx, y, mask, T)
    t_ref = time.time() - t0
    err = out - ref
    res = {"what": "VideoUNet forward, vista.yaml configuration (1.648 B params, seeded non-zero weights), 25 frames, latent 72x128, one clip (n_img = 25)",
           "rel_l2": (err.pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item(), "max_abs_over_max_ref": (err.abs().max() / ref.abs().max()).item(),
           "per_frame_rel_l2_max": max((err[i].pow(2).sum().sqrt() / ref[i].pow(2).sum().sqrt()).item() for i in range(T)),
           "oracle_seconds_on_host": round(t_ref, 1), "host_threads": torch.get_num_threads(), "gpu_seconds_first_call_incl_packing": round(t_gpu, 2),
           "finite": bool(torch.isfinite(out).all())}
    print(json.dumps(res), flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
