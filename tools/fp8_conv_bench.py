"""Same-process timing of the bf16 and fp8 forms of the ResBlock kernels at the BASELINE shapes (N = 50 images):
GroupNorm+SiLU (bf16 out / e4m3 out), conv3x3, temporal conv.   usage: python tools/fp8_conv_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
    N, T = 50, 25
    S = H * W
    x = torch.randn(N, S, C, device="cuda").to(BF16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    w = torch.randn(C, C, 3, 3) * (9 * C) ** -0.5
    wt = torch.randn(C, C, 3, 1, 1) * (3 * C) ** -0.5
    pc, pc8 = ops.pack_conv3x3(w, torch.zeros(C)), ops.pack_conv3x3_fp8(w, torch.zeros(C))
    pt, pt8 = ops.pack_conv_t3(wt, torch.zeros(C)), ops.pack_conv_t3_fp8(wt, torch.zeros(C))
    h = ops.groupnorm(x, g, b, 1e-5, True)
    h8, hs = ops.groupnorm_fp8(x, g, b, 1e-5, True)
    h8t, hst = ops.groupnorm_fp8(x, g, b, 1e-5, True, T)
    r = {
        "gn bf16": timeit(lambda: ops.groupnorm(x, g, b, 1e-5, True)),
        "gn fp8": timeit(lambda: ops.groupnorm_fp8(x, g, b, 1e-5, True)),
        "gn(T) bf16": timeit(lambda: ops.groupnorm(x, g, b, 1e-5, True, T)),
        "gn(T) fp8": timeit(lambda: ops.groupnorm_fp8(x, g, b, 1e-5, True, T)),
        "conv3x3 bf16": timeit(lambda: ops.conv3x3(h, pc, N, H, W)),
        "conv3x3 fp8": timeit(lambda: ops.conv3x3_fp8(h8, hs, pc8, N, H, W)),
        "conv_t3 bf16": timeit(lambda: ops.conv_t3(h, pt, T, S)),
        "conv_t3 fp8": timeit(lambda: ops.conv_t3_fp8(h8t, hst, pt8, T, S)),
    }
    print(f"C={C}: " + "  ".join(f"{k} {v:.4f}" for k, v in r.items()))
