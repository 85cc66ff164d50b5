"""Same-process A/B of the two V operand forms of the spatial attention kernel (round 3): V^T tensor (EPI_TRANS GEMM output, ds_read_b128
fragments) vs V rows of the fused q|k|v GEMM (ds_read_b64_tr_b16 fragments); plus the projection GEMMs in front of each form.
Interleaved rounds, best of 4 x 5 launches, BASELINE shapes (N = 50 images)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402


class Norm:
    def __init__(self, C):
        self.weight, self.bias, self.eps = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 1e-5


def best_of(fns, rounds=4, iters=5):
    best = {k: 1e9 for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / iters)
    return best


def main():
    n = 50
    for C, S, heads in ((320, 9216, 5), (640, 2304, 10), (1280, 576, 20)):
        g = torch.Generator(device="cuda").manual_seed(0)
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
        x = rn(n * S, C).to(torch.bfloat16)
        st = ops.rowstats(x)
        nrm = Norm(C)
        wq, wk, wv = (rn(C, C) * C ** -0.5 for _ in range(3))
        p_qkv = ops.pack_linear_cat([wq, wk, wv], ln=nrm)
        p_qk = ops.pack_linear_cat([wq, wk], ln=nrm)
        p_v = ops.pack_linear(wv, None, ln=nrm)
        qkv = ops.linear(x, p_qkv, ln=st)
        vt = ops.linear_vt(x, p_v, S, ln=st)
        flop = 4.0 * n * heads * float(S) ** 2 * 64
        b = best_of({
            "attn_vT": lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], vt, n, heads, S),
            "attn_vrows": lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n, heads, S, v_rows=True),
            "proj_qk+vT": lambda: (ops.linear(x, p_qk, ln=st), ops.linear_vt(x, p_v, S, ln=st)),
            "proj_qkv": lambda: ops.linear(x, p_qkv, ln=st),
        })
        print(json.dumps({"C": C, "S": S, "ms": {k: round(v, 4) for k, v in b.items()},
                          "attn_TFLOPs": {k: round(flop / b[k] / 1e9) for k in ("attn_vT", "attn_vrows")},
                          "block_total_ms": {"round2_form": round(b["attn_vT"] + b["proj_qk+vT"], 4), "qkv_form": round(b["attn_vrows"] + b["proj_qkv"], 4)}}), flush=True)


if __name__ == "__main__":
    main()
