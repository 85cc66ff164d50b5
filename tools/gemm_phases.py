"""Per-wave phase timers of the DMA GEMM loop (s_memtime): where do the cycles of a K-step go?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def run(name, fn, nk, waves):
    dbg = torch.zeros(4096 * 16 * 4, dtype=torch.int64, device="cuda")
    ops.GEMM_DBG = dbg
    fn()
    torch.cuda.synchronize()
    ops.GEMM_DBG = None
    d = dbg.view(-1, 16, 4)[:, :waves].float()
    used = d[:, 0, 3] > 0
    d = d[used]
    m = d.mean(dim=(0, 1))
    print(f"{name}: sampled WGs {d.shape[0]}, K-steps {nk}: per K-step cycles  dma_issue {m[0]/nk:7.0f}  compute {m[1]/nk:7.0f}  barrier {m[2]/nk:7.0f}"
          f"  | loop total {m[3]:9.0f} cycles; wave spread of barrier wait: min {d[:, :, 2].min()/nk:.0f} max {d[:, :, 2].max()/nk:.0f}")


N = 50
for C, H, W in ((320, 72, 128), (1280, 18, 32)):
    M = N * H * W
    x = torch.randn(M, C, device="cuda").to(BF16)
    pw = ops.pack_linear(torch.randn(C, 4 * C) * (4 * C) ** -0.5, torch.randn(C))
    h = torch.randn(M, 4 * C, device="cuda").to(BF16)
    run(f"ff_out {M}x{C}x{4*C} (256x320)", lambda: ops.linear(h, pw), 4 * C // 64, 8)
    pc = ops.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5, torch.randn(C))
    x3 = x.view(N, H * W, C)
    run(f"conv3x3 {C}->{C} @{H}x{W} (256x320)", lambda: ops.conv3x3(x3, pc, N, H, W), 9 * C // 64, 8)
    pg = ops.pack_geglu(torch.randn(8 * C, C) * C ** -0.5, torch.randn(8 * C))
    run(f"geglu {M}x{8*C}x{C} (256x256)", lambda: ops.linear(x, pg), C // 64, 8)
    for flag, nm in ((16, "no-DMA"),):
        ops.TILE_CFG = flag
        run(f"  [{nm}] conv3x3 {C}", lambda: ops.conv3x3(x3, pc, N, H, W), 9 * C // 64, 8)
        ops.TILE_CFG = 0
