"""A/B sweep of the GEMM block-tile / staging variants on the BASELINE shapes. One JSON line per (shape, variant)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=4, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,3,4,9,10,11,12".split(","))]
    N = 50
    shapes = []
    for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
        M = N * H * W
        shapes += [("linear", M, C, C), ("linear", M, 2 * C, C), ("ff_out", M, C, 4 * C), ("geglu", M, 8 * C, C),
                   ("conv3x3", (N, H, W), C, C), ("conv_t3", (N, H * W), C, C)]
    for kind, M, Nn, K in shapes:
        res = {}
        if kind in ("linear", "ff_out"):
            x = torch.randn(M, K, device="cuda").to(BF16)
            pw = ops.pack_linear(torch.randn(Nn, K) * K ** -0.5, torch.randn(Nn))
            fn = lambda: ops.linear(x, pw)  # noqa: E731
            flop = 2.0 * M * Nn * K
        elif kind == "geglu":
            x = torch.randn(M, K, device="cuda").to(BF16)
            pw = ops.pack_geglu(torch.randn(Nn, K) * K ** -0.5, torch.randn(Nn))
            fn = lambda: ops.linear(x, pw)  # noqa: E731
            flop = 2.0 * M * Nn * K
        elif kind == "conv3x3":
            n, H, W = M
            x = torch.randn(n, H * W, K, device="cuda").to(BF16)
            pw = ops.pack_conv3x3(torch.randn(Nn, K, 3, 3) * (9 * K) ** -0.5, torch.randn(Nn))
            fn = lambda: ops.conv3x3(x, pw, n, H, W)  # noqa: E731
            flop = 2.0 * n * H * W * Nn * 9 * K
        else:
            n, S = M
            x = torch.randn(n, S, K, device="cuda").to(BF16)
            pw = ops.pack_conv_t3(torch.randn(Nn, K, 3, 1, 1) * (3 * K) ** -0.5, torch.randn(Nn))
            fn = lambda: ops.conv_t3(x, pw, 25, S)  # noqa: E731
            flop = 2.0 * n * S * Nn * 3 * K
        best = {}
        for rnd in range(3):  # interleaved rounds, keep the best (least disturbed) time per variant
            for v in variants:
                if kind == "geglu" and (v & 7) == 4:
                    continue
                ops.TILE_CFG = v
                ms = timeit(fn, iters=10, warmup=1)
                best[v] = min(best.get(v, 1e9), ms)
        for v, ms in best.items():
            res[str(v)] = round(flop / ms / 1e9, 0)
        ops.TILE_CFG = 0
        print(json.dumps({"kind": kind, "M": M, "N": Nn, "K": K, "TFLOPs_by_variant": res}), flush=True)
        del x, pw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
