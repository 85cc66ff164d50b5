"""GroupNorm statistics from the convolution's epilogue (VkGemmDesc.gnstat_out) against the statistics pass it replaces, per launch, at the
BASELINE shapes: ms of the convolution without / with the emitting epilogue, of vk_groupnorm_stats_bf16 (statistics pass + fold: what the
three-launch GroupNorm runs) and of vk_groupnorm_finalize_partials (the fold alone: what is left). Alternated, HIP events on the launch stream.
usage (GPU box): python tools/gnstat_ab.py [images]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import _lib, ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    lib = _lib.load()
    rn = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
    rows = []
    for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
        S, T = H * W, 25 if N % 25 == 0 else N
        x = rn(N, S, C).to(BF16)
        res = rn(N, S, C).to(BF16)
        rv = rn(N, C)
        pc = ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C))
        pt = ops.pack_conv_t3(rn(C, C, 3, 1, 1) * (3 * C) ** -0.5, rn(C))
        cases = {
            "conv3x3+emb": lambda gn: ops.conv3x3(x, pc, N, H, W, rowvec=rv, gn=gn),
            "conv3x3+res": lambda gn: ops.conv3x3(x, pc, N, H, W, res1=res, gn=gn),
            "conv_t3+emb": lambda gn: ops.conv_t3(x, pt, T, S, rowvec=rv, gn=gn),
            "conv_t3+blend": lambda gn: ops.conv_t3(x, pt, T, S, alpha=0.4, res2=res, beta=1.0, gn=gn),
        }
        sums = torch.empty(N * 64, dtype=F32, device="cuda")
        part = torch.empty(N * ((S + 31) // 32) * 64, dtype=F32, device="cuda")
        slots = torch.zeros(N * (S // 64) * 64, dtype=F32, device="cuda")
        t_stats = timeit(lambda: lib.vk_groupnorm_stats_bf16(ops._p(x), ops._p(sums), ops._p(part), N, S, C, 1, ops._stream()))
        t_fold = timeit(lambda: lib.vk_groupnorm_finalize_partials(ops._p(slots), ops._p(sums), N, S // 64, 1, ops._stream()))
        t_fold5 = timeit(lambda: lib.vk_groupnorm_finalize_partials(ops._p(slots), ops._p(sums), N, S // 64, T, ops._stream()))
        for name, fn in cases.items():
            a, b = [], []
            for _ in range(3):   # alternate
                a.append(timeit(lambda: fn(None)))
                b.append(timeit(lambda: fn(ops.GnPartials())))
            g = ops.GnPartials()
            fn(g)
            rows.append({"C": C, "HxW": f"{H}x{W}", "images": N, "kernel": name, "emitted": g.t is not None, "ms_plain": min(a), "ms_emitting": min(b),
                         "ms_stats_pass_plus_fold": t_stats, "ms_fold_only": t_fold, "ms_fold_only_5d": t_fold5,
                         "net_us_saved_per_norm": 1e3 * (t_stats - t_fold - (min(b) - min(a)))})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
