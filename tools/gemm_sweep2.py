"""A/B sweep of GEMM block-tile choices (ops.TILE_CFG: 0 = auto, 1..5 = forced) on the BASELINE shapes, including the folded-LayerNorm consumers,
the row-sum emitting producers and the residual / row-vector epilogues. Interleaved rounds, best of 3 x 10 launches per variant; one JSON line per
shape: TFLOP/s and ms by variant.   usage: python tools/gemm_sweep2.py [cfg,cfg,...]   (tools/ab_sweep.sh alternates two LIBRARY builds on cfg 0)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16


class Norm:
    def __init__(self, C):
        self.weight, self.bias, self.eps = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 1e-5


def timeit(fn, iters=10):
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
    flags = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0".split(","))]  # values of ops.TILE_CFG to compare (0 = the launcher's own choice; 1..5 force a block tile)
    N = int(os.environ.get("SWEEP_IMAGES", "50"))  # 50 = the CFG-doubled 25-frame window; 7 / 8 = one rank of an 8-GPU run
    T = 25 if N % 25 == 0 else N
    rn = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
    for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
        M = N * H * W
        x = rn(M, C).to(BF16)
        res = rn(M, C).to(BF16)
        h4 = rn(M, 4 * C).to(BF16)
        st = ops.rowstats(x)
        nrm = Norm(C)
        x3 = x.view(N, H * W, C)
        cases = {
            "linear+res+stats": (lambda pw=ops.pack_linear(rn(C, C) * C ** -0.5, rn(C)): ops.linear(x, pw, res1=res, emit_stats=True), 2.0 * M * C * C),
            "qkv_lnfold": (lambda pw=ops.pack_linear_cat([rn(C, C) * C ** -0.5 for _ in range(3)], ln=nrm): ops.linear(x, pw, ln=st), 2.0 * M * 3 * C * C),
            "geglu_lnfold": (lambda pw=ops.pack_geglu(rn(8 * C, C) * C ** -0.5, rn(8 * C), ln=nrm): ops.linear(x, pw, ln=st), 2.0 * M * 8 * C * C),
            "geglu_plain": (lambda pw=ops.pack_geglu(rn(8 * C, C) * C ** -0.5, rn(8 * C)): ops.linear(x, pw), 2.0 * M * 8 * C * C),
            "ff_out+res+stats": (lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)): ops.linear(h4, pw, res1=res, emit_stats=True),
                                 2.0 * M * 4 * C * C),
            "conv3x3": (lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)): ops.conv3x3(x3, pw, N, H, W), 2.0 * M * 9 * C * C),
            "conv3x3+emb": (lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)), rv=rn(N, C): ops.conv3x3(x3, pw, N, H, W, rowvec=rv),
                            2.0 * M * 9 * C * C),
            "conv3x3+res": (lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)): ops.conv3x3(x3, pw, N, H, W, res1=x3), 2.0 * M * 9 * C * C),
            "ff_out+blend": (lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)), rv=rn(N, C): ops.linear(
                h4, pw, res1=res, alpha=0.4, res2=x, rowvec2=rv, beta=0.6, rows_per_vec=H * W), 2.0 * M * 4 * C * C),
            "attn_out+ctx": (lambda pw=ops.pack_linear(rn(C, C) * C ** -0.5, rn(C)), rv=rn(N, C): ops.linear(
                x, pw, res1=res, rowvec=rv, rows_per_vec=H * W, emit_stats=True), 2.0 * M * C * C),
            "conv_t3": (lambda pw=ops.pack_conv_t3(rn(C, C, 3, 1, 1) * (3 * C) ** -0.5, rn(C)): ops.conv_t3(x3, pw, T, H * W), 2.0 * M * 3 * C * C),
        }
        only = os.environ.get("SWEEP_KINDS")
        for name, (fn, flop) in cases.items():
            if only and not any(k in name for k in only.split(",")):
                continue
            best = {}
            for _ in range(3):
                for f in flags:
                    ops.TILE_CFG = f
                    best[f] = min(best.get(f, 1e9), timeit(fn))
            ops.TILE_CFG = 0
            print(json.dumps({"level_C": C, "kind": name, "M": M, "TFLOPs_by_flags": {str(f): round(flop / ms / 1e9) for f, ms in best.items()},
                              "ms_by_flags": {str(f): round(ms, 4) for f, ms in best.items()}}), flush=True)
        del x, res, h4, cases
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
