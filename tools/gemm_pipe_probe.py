"""Pipelined eight-wave 256x320 GEMM variant (ops.TILE_CFG = 7, vista_amd/csrc/gemm_pipe.hip) against the sixteen-wave 256x320 kernel (TILE_CFG = 4):
bitwise comparison on the BASELINE shapes and on ragged ones, then timing, plus the launcher's own choice (TILE_CFG = 0) beside them.
Environment: PROBE_FAST=n (first n shape sets), PROBE_SMALL=1 (level 3 / one rank's sizes), PROBE_KINDS=a,b (case filter), PROBE_SEED, PROBE_TIMING=1 (library
built with -DPIPE_TIMING: s_memtime phase sums), PROBE_CHK=1 (checksums of the inputs and of every packed weight after every launch: found the
out-of-bounds row-sum write of round 4).   usage: python tools/gemm_pipe_probe.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402
from tools.gemm_sweep2 import Norm, timeit  # noqa: E402

BF16 = torch.bfloat16


def flat(r):
    return [t for t in (r if isinstance(r, (tuple, list)) else (r,)) if torch.is_tensor(t)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rn = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
    torch.manual_seed(int(os.environ.get("PROBE_SEED", "0")))
    bad = 0
    shapes = ((320, 72, 128, N), (640, 36, 64, N), (1280, 18, 32, N), (320, 20, 24, 3), (640, 9, 13, 5),
              (1280, 9, 16, N), (320, 72, 128, 7), (640, 36, 64, 7), (1280, 18, 32, 7), (1280, 9, 16, 7))
    if os.environ.get("PROBE_SMALL"):
        shapes = shapes[5:]
    for C, H, W, n in (shapes[:int(os.environ["PROBE_FAST"])] if os.environ.get("PROBE_FAST") else shapes):
        T = 25 if n % 25 == 0 else n
        M = n * H * W
        x = rn(M, C).to(BF16)
        res = rn(M, C).to(BF16)
        h4 = rn(M, 4 * C).to(BF16)
        st = ops.rowstats(x)
        nrm = Norm(C)
        x3 = x.view(n, H * W, C)
        rv = rn(n, C)
        cases = {
            "linear+res+stats": (lambda pw=ops.pack_linear(rn(C, C) * C ** -0.5, rn(C)): ops.linear(x, pw, res1=res, emit_stats=True), 2.0 * M * C * C),
            "qkv_lnfold": (lambda pw=ops.pack_linear_cat([rn(C, C) * C ** -0.5 for _ in range(3)], ln=nrm): ops.linear(x, pw, ln=st), 2.0 * M * 3 * C * C),
            "ff_out+res+stats": (lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)): ops.linear(h4, pw, res1=res, emit_stats=True), 2.0 * M * 4 * C * C),
            "ff_out+blend": (lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)): ops.linear(
                h4, pw, res1=res, alpha=0.4, res2=x, rowvec2=rv, beta=0.6, rows_per_vec=H * W), 2.0 * M * 4 * C * C),
            "geglu_lnfold": (lambda pw=ops.pack_geglu(rn(8 * C, C) * C ** -0.5, rn(8 * C), ln=nrm): ops.linear(x, pw, ln=st), 2.0 * M * 8 * C * C),
            "geglu_plain": (lambda pw=ops.pack_geglu(rn(8 * C, C) * C ** -0.5, rn(8 * C)): ops.linear(x, pw), 2.0 * M * 8 * C * C),
            "conv3x3": (lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)): ops.conv3x3(x3, pw, n, H, W), 2.0 * M * 9 * C * C),
            "conv3x3+emb+res": (lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)): ops.conv3x3(x3, pw, n, H, W, rowvec=rv, res1=x3), 2.0 * M * 9 * C * C),
            "conv_t3": (lambda pw=ops.pack_conv_t3(rn(C, C, 3, 1, 1) * (3 * C) ** -0.5, rn(C)): ops.conv_t3(x3, pw, T, H * W), 2.0 * M * 3 * C * C),
        }
        if H % 2 == 0 and W % 2 == 0:
            cases["conv3x3_s2"] = (lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)): ops.conv3x3(x3, pw, n, H, W, stride=2), 2.0 * M / 4 * 9 * C * C)
        for name, (fn, flop) in cases.items():
            if os.environ.get("PROBE_KINDS") and not any(k in name for k in os.environ["PROBE_KINDS"].split(",")):
                continue
            out = {}
            if os.environ.get("PROBE_CHK"):
                torch.cuda.synchronize()
                print("   chk before", name, "x", x.float().sum().item(), "res", res.float().sum().item(), "h4", h4.float().sum().item(), "st", st.t.float().sum().item(), flush=True)
            if os.environ.get("PROBE_CHK") and fn.__defaults__:
                pwd = fn.__defaults__[0]
                print("   packed:", {k: (tuple(getattr(pwd, k).shape), bool(torch.isfinite(getattr(pwd, k).float()).all())) for k in ("wt", "bias", "colsum") if torch.is_tensor(getattr(pwd, k, None))}, flush=True)
            for cfg in (4, 7, 0, 0):   # 0 = the launcher's own choice (split-K, 128x160 tiles, the streaming kernel ... where its rules pick them), twice: run-to-run bitwise check
                ops.TILE_CFG = cfg
                out.setdefault(cfg, []).append([t.clone() for t in flat(fn())])
                if os.environ.get("PROBE_CHK"):
                    torch.cuda.synchronize()
                    for nm2, (fn2, _) in cases.items():
                        pw2 = fn2.__defaults__[0]
                        if not bool(torch.isfinite(pw2.wt.float()).all()):
                            print(f"   !! after {name} cfg {cfg}: packed weight of {nm2} is corrupted", flush=True)
            if os.environ.get("PROBE_CHK"):
                for k, v in out.items():
                    for r, ts in enumerate(v):
                        for t in ts:
                            nz = torch.isnan(t.float()).nonzero()
                            if len(nz):
                                print(f"   cfg {k} run {r}: shape {tuple(t.shape)} NaNs {len(nz)} first {nz[0].tolist()} last {nz[-1].tolist()} rows {sorted(set((nz[:, 0] // 256).tolist()))[:12]}", flush=True)
            out = {k: v for k, v in out.items()}
            auto_rep = all(torch.equal(a, b) for a, b in zip(out[0][0], out[0][1]))
            auto_err = max(((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9)).item() for a, b in zip(out[0][0], out[7][0]))
            out = {4: out[4][0], 7: out[7][0]}
            same = all(torch.equal(a, b) for a, b in zip(out[4], out[7]))
            bad += not same
            ms = {}
            for _ in range(3):
                for cfg in (4, 7, 0) + ((3,) if name.startswith("geglu") else ()):
                    ops.TILE_CFG = cfg
                    ms[cfg] = min(ms.get(cfg, 1e9), timeit(fn))
            ops.TILE_CFG = 0
            if os.environ.get("PROBE_TIMING"):   # library built with -DPIPE_TIMING: per-wave s_memtime sums of workgroup 8
                ops.TILE_CFG = 7
                fn()
                torch.cuda.synchronize()
                ws = ops._splitk_workspace(ops._stream())[:64].view(8, 8).cpu()
                ops.TILE_CFG = 0
                for w in (0, 5):   # per wave of workgroup 8: [own-DMA wait, barrier wait, K-steps timed, K-loops, epilogues, kernel] in s_memtime ticks
                    d, b, ks, loop, epi, tot = ws[w, :6].tolist()
                    ks = max(ks, 1.0)
                    print(f"      wave {w}: K-steps timed {int(ks)}  K-loop {loop / ks:.0f} ticks per K-step  own-DMA wait {d / ks:.0f}  barrier wait {b / ks:.0f};  "
                          f"K-loops {loop:.0f}  epilogues {epi:.0f}  kernel {tot:.0f} ticks")
            err = max((a.float() - b.float()).abs().max().item() for a, b in zip(out[4], out[7]))
            print(f"C {C:5d} M {M:7d} {name:18s} bitwise {'OK ' if same else 'DIFF'} max|d| {err:.3g}   cfg4 {ms[4]:.4f} ms  cfg7 {ms[7]:.4f} ms  "
                  f"{100 * (ms[4] / ms[7] - 1):+.1f} %   {flop / ms[7] / 1e9:.0f} TFLOP/s" + (f"   cfg3 {ms[3]:.4f} ms" if 3 in ms else "") +
                  f"   auto {ms[0]:.4f} ms ({100 * (ms[7] / ms[0] - 1):+.1f} % vs cfg7, rel err {auto_err:.1e}, repeatable {auto_rep})", flush=True)
            bad += (not auto_rep) or auto_err > 2e-2
        del x, res, h4, cases
        torch.cuda.empty_cache()
    print("MISMATCHES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
