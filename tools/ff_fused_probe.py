"""Fused FeedForward (vk_ff_fused_bf16) vs the two-kernel form (GEGLU GEMM + out-projection GEMM) at the level-0 BASELINE shape:
correctness against a torch fp32 reference and against the two-kernel form, then interleaved timings, then (round-4 experiment) the
two-kernel form run in row chunks small enough for the hidden activation to stay in the 256 MB Infinity Cache.
usage: python tools/ff_fused_probe.py [--quick]
(round 5: the timing variants -- ops.FF_FUSED_DBG 1 / 2 / 4 / 8 -- exist only in a library built with -DFF_TIMING, e.g.
 hipcc ... -DFF_TIMING -c vista_amd/csrc/ff_fused.hip; the product library runs the one shipped instantiation whatever the value)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16


class Norm:
    def __init__(self, C):
        g = torch.Generator().manual_seed(7)
        self.weight = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
        self.bias = (0.1 * torch.randn(C, generator=g)).cuda()
        self.eps = 1e-5


def rn(*s, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(s))
    return (torch.randn(*s, generator=g) * scale).cuda()


def ref_ff(x, w1, b1, w2, b2, norm):
    xf = x.float()
    if norm is not None:
        xf = torch.nn.functional.layer_norm(xf, (xf.shape[1],), norm.weight, norm.bias, norm.eps)
    y = xf @ w1.float().t() + b1
    a, g = y.chunk(2, dim=1)
    h = (a * torch.nn.functional.gelu(g)).to(BF16).float()
    return h @ w2.float().t() + b2


def relerr(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def check(M, C=320, H=1280, ln=True, mode="res"):
    x = rn(M, C).to(BF16)
    w1 = rn(2 * H, C, scale=C ** -0.5, seed=1).to(BF16)
    b1 = rn(2 * H, seed=2) * 0.5
    w2 = rn(C, H, scale=H ** -0.5, seed=3).to(BF16)
    b2 = rn(C, seed=4)
    norm = Norm(C) if ln else None
    pin = ops.pack_geglu(w1, b1, ln=norm)
    pout = ops.pack_linear(w2, b2)
    poutp = ops.pack_ff_out(w2, b2)
    st = ops.rowstats(x) if ln else None
    kw = {}
    ref = ref_ff(x, w1, b1, w2, b2, norm)
    if mode == "res":
        kw = dict(res1=x, emit_stats=True)
        ref = ref + x.float()
    elif mode == "blend":
        S = 100
        xm = rn(M, C, seed=5).to(BF16)
        rv2 = rn((M + S - 1) // S, C, seed=6)
        kw = dict(res1=x, alpha=0.4, res2=xm, rowvec2=rv2, beta=0.6, rows_per_vec=S)
        ref = 0.4 * (ref + x.float()) + 0.6 * (xm.float() + rv2.repeat_interleave(S, 0)[:M])
    elif mode == "rowvec":
        S = 64
        rv = rn((M + S - 1) // S, C, seed=8)
        kw = dict(res1=x, rowvec=rv, rows_per_vec=S, emit_stats=True)
        ref = ref + x.float() + rv.repeat_interleave(S, 0)[:M]
    two = ops.linear(ops.linear(x, pin, ln=st), pout, **kw)
    fus = ops.ff_fused(x, pin, poutp, ln=st, **kw)
    if isinstance(two, tuple):
        (two, st2), (fus, stf) = two, fus
        s2 = st2.t.sum(0)
        sf = stf.t.sum(0)
        serr = ((s2 - sf).abs().max() / s2.abs().max()).item()
    else:
        serr = 0.0
    torch.cuda.synchronize()
    r = {"M": M, "ln": ln, "mode": mode, "fused_vs_ref": relerr(fus, ref), "two_vs_ref": relerr(two, ref), "fused_vs_two": relerr(fus, two),
         "maxabs_fused_vs_two": (fus.float() - two.float()).abs().max().item(), "rowstat_rel": serr, "finite": bool(torch.isfinite(fus.float()).all())}
    print(json.dumps(r), flush=True)
    return r


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


def bench(M=460800, C=320, H=1280):
    x = rn(M, C).to(BF16)
    xm = rn(M, C, seed=5).to(BF16)
    w1 = rn(2 * H, C, scale=C ** -0.5, seed=1).to(BF16)
    b1 = rn(2 * H, seed=2)
    w2 = rn(C, H, scale=H ** -0.5, seed=3).to(BF16)
    b2 = rn(C, seed=4)
    norm = Norm(C)
    pin = ops.pack_geglu(w1, b1, ln=norm)
    pout = ops.pack_linear(w2, b2)
    poutp = ops.pack_ff_out(w2, b2)
    st = ops.rowstats(x)
    rv2 = rn(50, C, seed=6)
    flop = 2.0 * M * C * (2 * H) + 2.0 * M * H * C
    out = torch.empty(M, C, dtype=BF16, device="cuda")
    cases = {
        "res+stats": dict(res1=x, emit_stats=True),
        "blend": dict(res1=x, alpha=0.4, res2=xm, rowvec2=rv2, beta=0.6, rows_per_vec=M // 50),
    }
    for name, kw in cases.items():
        two = lambda: ops.linear(ops.linear(x, pin, ln=st), pout, out=out, **kw)  # noqa: E731
        fus = lambda: ops.ff_fused(x, pin, poutp, ln=st, out=out, **kw)  # noqa: E731
        geg = lambda: ops.linear(x, pin, ln=st)  # noqa: E731
        best = {"two": 1e9, "fused": 1e9, "geglu_only": 1e9}
        for _ in range(3):
            best["two"] = min(best["two"], timeit(two))
            best["fused"] = min(best["fused"], timeit(fus))
            best["geglu_only"] = min(best["geglu_only"], timeit(geg))
        print(json.dumps({"case": name, "M": M, "ms": {k: round(v, 4) for k, v in best.items()},
                          "TFLOPs": {"two": round(flop / best["two"] / 1e9), "fused": round(flop / best["fused"] / 1e9)}}), flush=True)
    for dbg, what in ((1, "no LDS-DMA in the steps"), (2, "gelu -> plain product"), (4, "no out-projection MFMAs")):
        ops.FF_FUSED_DBG = dbg
        t = min(timeit(lambda: ops.ff_fused(x, pin, poutp, ln=st, out=out, res1=x)) for _ in range(3))
        ops.FF_FUSED_DBG = 0
        print(json.dumps({"case": "fused, timing experiment (wrong results)", "dbg": dbg, "what": what, "ms": round(t, 4)}), flush=True)
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    ops.FF_FUSED_DBG, ops.FF_FUSED_DBG_BUF = 8, buf
    ops.ff_fused(x, pin, poutp, ln=st, out=out, res1=x)
    torch.cuda.synchronize()
    ops.FF_FUSED_DBG, ops.FF_FUSED_DBG_BUF = 0, None
    b = buf.view(8, 8).tolist()
    for w in range(8):
        n = max(b[w][4] if w < 4 else b[w][2], 1)
        if w < 4:
            print(json.dumps({"wave": w, "role": "in", "steps": n, "cycles_per_step": {"mfma": b[w][0] // n, "gelu": b[w][2] // n, "barrier": b[w][3] // n}}), flush=True)
        else:
            print(json.dumps({"wave": w, "role": "out", "steps": n, "cycles_per_step": {"dma_issue": b[w][4] // n, "dma+mfma": b[w][0] // n, "barrier": b[w][1] // n}, "epilogue_per_tile": b[w][3] // 15}), flush=True)
    if "--chunks" not in sys.argv:
        return
    # --- Infinity-Cache experiment: the two-kernel form over row chunks (h chunk = rows x 1280 x 2 B)
    kw = dict(res1=x)
    for rows in (460800, 131072, 65536, 32768):
        chunks = [(r0, min(r0 + rows, M)) for r0 in range(0, M, rows)]
        sts = [ops.rowstats(x[a:b]) for a, b in chunks]

        def run():
            for (a, b), s in zip(chunks, sts):
                ops.linear(ops.linear(x[a:b], pin, ln=s), pout, out=out[a:b], res1=x[a:b])
        t = min(timeit(run) for _ in range(3))
        print(json.dumps({"case": "two-kernel, row chunks", "rows": rows, "h_chunk_MB": round(rows * H * 2 / 1e6), "ms": round(t, 4)}), flush=True)


if __name__ == "__main__":
    ok = True
    for M in (128, 1000, 4096 + 77):
        for ln in (False, True):
            for mode in ("plain", "res", "blend", "rowvec"):
                r = check(M, ln=ln, mode=mode)
                ok &= r["finite"] and r["fused_vs_ref"] < 6e-3 and r["fused_vs_two"] < 3e-3 and r["rowstat_rel"] < 2e-3
    r = check(1000, H=640, ln=True, mode="res")
    ok &= r["finite"] and r["fused_vs_ref"] < 6e-3
    print("CORRECT" if ok else "MISMATCH", flush=True)
    if "--quick" not in sys.argv:
        bench()
