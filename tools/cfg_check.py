"""Correctness spot-check of one forced GEMM tile config / flag set against torch (tuning helper)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops
cfg = int(sys.argv[1])
for M, N, K in ((4608, 640, 1280), (1000, 960, 320), (300, 256, 64), (5000, 320, 128)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N)
    pw = ops.pack_linear(w, b)
    ops.TILE_CFG = cfg
    out = ops.linear(x, pw).float()
    ops.TILE_CFG = 0
    ref = x.float() @ w.float().cuda().t() + b.cuda()
    print(cfg, (M, N, K), "max err", (out - ref).abs().max().item(), "rms", ref.pow(2).mean().sqrt().item())
