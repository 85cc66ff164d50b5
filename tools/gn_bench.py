"""GroupNorm(+SiLU) passes at the BASELINE shapes vs plain streaming yardsticks on the same tensors (torch clone = read + write, torch sum = read only):
how far are vk_groupnorm_silu_bf16's statistics and apply passes from what the box's HBM delivers?   python tools/gn_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402


def best(fn, rounds=5, iters=10):
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        b = min(b, e0.elapsed_time(e1) / iters)
    return b


for C, S, fpg in ((320, 9216, 1), (320, 9216, 25), (640, 2304, 1), (1280, 576, 1), (1280, 144, 1)):
    n = 50
    x = torch.randn(n, S, C, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y = torch.empty_like(x)
    mb = x.numel() * 2 / 1e6
    lib = ops._lib.load()
    ws = torch.empty(((n // fpg) + n * ((S + 31) // 32)) * 64, dtype=torch.float32, device="cuda")
    sums = torch.empty((n // fpg) * 64, dtype=torch.float32, device="cuda")
    part = torch.empty(n * ((S + 31) // 32) * 64, dtype=torch.float32, device="cuda")
    t_full = best(lambda: ops.groupnorm(x, g, b, 1e-5, True, fpg, out=y))
    t_stats = best(lambda: ops.check(lib.vk_groupnorm_stats_bf16(ops._p(x), ops._p(sums), ops._p(part), n, S, C, fpg, ops._stream()), "stats"))
    t_apply = best(lambda: ops.check(lib.vk_groupnorm_apply_bf16(ops._p(x), ops._p(y), ops._p(g), ops._p(b), ops._p(sums), n, S, C, fpg, float(C // 32 * S * fpg), 1e-5, 1,
                                                                 ops._stream()), "apply"))
    t_clone = best(lambda: y.copy_(x))
    t_sum = best(lambda: x.view(torch.int16).sum(dtype=torch.int64))
    print(json.dumps({"C": C, "S": S, "fpg": fpg, "MB": round(mb), "ms": {"gn_full": round(t_full, 4), "stats(+finalize)": round(t_stats, 4), "apply": round(t_apply, 4),
                                                                        "torch_copy": round(t_clone, 4), "torch_sum": round(t_sum, 4)},
                      "TBps": {"stats_read": round(mb / t_stats / 1e3, 2), "apply_rw": round(2 * mb / t_apply / 1e3, 2), "copy_rw": round(2 * mb / t_clone / 1e3, 2),
                               "sum_read": round(mb / t_sum / 1e3, 2)}}), flush=True)
