"""Diagnostic: input_blocks.1 (VideoResBlock + SpatialVideoTransformer) at a reduced size with ops.GN_EPI = 1 and 0 -- the output of every
GroupNorm call and of the block, path against path, next to what a one-ulp perturbation of 0.1 % of the block's input does to the same
quantities (the sensitivity of the block itself).   usage (GPU box): python tools/gnstat_block_diag.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402
from vista_amd.config import unet_kwargs  # noqa: E402
from vista_amd.modules.diffusionmodules.video_model import VideoUNet  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def main():
    torch.manual_seed(0)
    with torch.device("cuda"):
        net = VideoUNet(**unet_kwargs(320))
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for name, p in net.input_blocks[1].named_parameters():
            if name.endswith("mix_factor"):
                p.normal_(0, 0.5, generator=g)
            elif p.dim() >= 2:
                p.normal_(0, float(p[0].numel()) ** -0.5, generator=g)
            elif name.endswith(".weight"):
                p.normal_(1.0, 0.1, generator=g)
            else:
                p.normal_(0, 0.1, generator=g)
    net = net.eval()
    T, H, W = 4, 16, 16
    n = 2 * T
    gc = torch.Generator().manual_seed(5)
    x = torch.randn(n, H * W, 320, generator=gc).to(BF16).cuda()
    emb = (torch.randn(n, 1280, generator=gc) * 0.7).to(BF16).cuda()
    ctx = torch.randn(n, 3456, generator=gc).to(BF16).cuda()
    frame_idx = torch.arange(T, dtype=F32, device="cuda").repeat(2)
    blk = net.input_blocks[1]
    real_gn = ops.groupnorm
    log = []

    def spy(xx, *a, **k):
        had = k.get("gn") is not None and k["gn"].t is not None
        out = real_gn(xx, *a, **k)
        log.append((had, xx.clone(), out.clone()))
        return out
    ops.groupnorm = spy
    ops.TILE_CFG = 7
    rel = lambda a, b: ((a.float() - b.float()).pow(2).sum().sqrt() / b.float().pow(2).sum().sqrt().clamp_min(1e-30)).item()  # noqa: E731
    runs = {}
    xp = x.clone()
    idx = torch.randperm(x.numel(), generator=gc)[: x.numel() // 1000].cuda()
    flat = xp.view(torch.int16).view(-1)
    flat[idx] = flat[idx] ^ 1   # last mantissa bit of 0.1 % of the input elements
    for key, sw, inp in (("epi", 1, x), ("pass", 0, x), ("pass_perturbed", 0, xp)):
        ops.GN_EPI = sw
        log.clear()
        with torch.no_grad():
            o, _, _ = blk(inp, emb, ctx, frame_idx, T, H, W)
        runs[key] = (o.clone(), list(log))
    print(f"block output: epilogue statistics vs statistics pass {rel(runs['epi'][0], runs['pass'][0]):.3e}; "
          f"one-ulp input perturbation (0.1 % of elements), statistics pass both {rel(runs['pass_perturbed'][0], runs['pass'][0]):.3e}")
    for i, ((had, xi, oi), (_, xj, oj), (_, xk, ok)) in enumerate(zip(runs["epi"][1], runs["pass"][1], runs["pass_perturbed"][1])):
        d = (oi.float() - oj.float()).abs()
        print(f"groupnorm call {i}: shape {tuple(oi.shape)} from epilogue partials: {had};  input rel diff {rel(xi, xj):.3e}  output rel diff {rel(oi, oj):.3e} "
              f"(max abs {d.max().item():.3g}, changed {100.0 * (d > 0).float().mean().item():.3f} %);  perturbed run: input {rel(xk, xj):.3e} output {rel(ok, oj):.3e}")
    ops.groupnorm = real_gn


if __name__ == "__main__":
    main()
