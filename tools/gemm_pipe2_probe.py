"""Two-workgroups-per-CU pipelined GEMM (ops.TILE_CFG = 16, vista_amd/csrc/gemm_pipe2.hip: four waves, 128x320 tiles, 32-deep K-steps) against the
eight-wave 256x320 pipelined kernel (TILE_CFG = 7) and the launcher's own choice (0) on the dense shapes of the BASELINE step: bitwise comparison,
then interleaved timing (best of 3 x 10 launches).   usage: python tools/gemm_pipe2_probe.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402
from tools.gemm_sweep2 import Norm, timeit  # noqa: E402


def flat(r):
    return [t for t in (r if isinstance(r, (tuple, list)) else (r,)) if torch.is_tensor(t)] + ([r[1].t] if isinstance(r, tuple) and hasattr(r[1], "t") else [])


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rn = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
    torch.manual_seed(0)
    bad = 0
    for C, H, W, n in ((320, 72, 128, N), (640, 36, 64, N), (1280, 18, 32, N), (320, 20, 24, 3), (640, 9, 13, 5)):
        M = n * H * W
        x = rn(M, C).to(ops.ACT)
        res = rn(M, C).to(ops.ACT)
        h4 = rn(M, 4 * C).to(ops.ACT)
        st = ops.rowstats(x)
        nrm = Norm(C)
        rv = rn(n, C)
        cases = {
            "linear+res+stats": (lambda pw=ops.pack_linear(rn(C, C) * C ** -0.5, rn(C)): ops.linear(x, pw, res1=res, emit_stats=True), 2.0 * M * C * C),
            "linear+res": (lambda pw=ops.pack_linear(rn(C, C) * C ** -0.5, rn(C)): ops.linear(x, pw, res1=res), 2.0 * M * C * C),
            "qkv_lnfold": (lambda pw=ops.pack_linear_cat([rn(C, C) * C ** -0.5 for _ in range(3)], ln=nrm): ops.linear(x, pw, ln=st), 2.0 * M * 3 * C * C),
            "ff_out+res+stats": (lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)): ops.linear(h4, pw, res1=res, emit_stats=True), 2.0 * M * 4 * C * C),
            "ff_out+blend": (lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)): ops.linear(
                h4, pw, res1=res, alpha=0.4, res2=x, rowvec2=rv, beta=0.6, rows_per_vec=H * W), 2.0 * M * 4 * C * C),
            "geglu_lnfold": (lambda pw=ops.pack_geglu(rn(8 * C, C) * C ** -0.5, rn(8 * C), ln=nrm): ops.linear(x, pw, ln=st), 2.0 * M * 8 * C * C),
        }
        for name, (fn, flop) in cases.items():
            out = {}
            for cfg in (7, 16, 16):
                ops.TILE_CFG = cfg
                out.setdefault(cfg, []).append([t.clone() for t in flat(fn())])
            ops.TILE_CFG = 0
            same = all(torch.equal(a, b) for a, b in zip(out[7][0], out[16][0]))
            rep = all(torch.equal(a, b) for a, b in zip(out[16][0], out[16][1]))
            err = max(((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9)).item() for a, b in zip(out[16][0], out[7][0]))
            bad += (not same) or (not rep)
            ms = {}
            for _ in range(3):
                for cfg in (7, 16, 0) + ((3,) if name.startswith("geglu") else ()):
                    ops.TILE_CFG = cfg
                    ms[cfg] = min(ms.get(cfg, 1e9), timeit(fn))
            ops.TILE_CFG = 0
            if os.environ.get("PROBE_TIMING"):   # library built with -DPIPE_TIMING (tools/build_variant.sh): s_memtime sums of workgroup 8's waves
                for cfg in (7, 16):
                    ops.TILE_CFG = cfg
                    ops._splitk_workspace(ops._stream())[:64].zero_()
                    fn()
                    torch.cuda.synchronize()
                    ws = ops._splitk_workspace(ops._stream())[:64].view(8, 8).cpu()
                    ops.TILE_CFG = 0
                    for w in (0, 3):
                        v = ws[w, :6].tolist()
                        if cfg == 7:   # [own-DMA wait, barrier wait, K-steps timed, K-loops, epilogues, kernel]
                            ks = max(v[2], 1.0)
                            print(f"      cfg7  wave {w}: {int(ks)} K-steps of 64 in the timed loops; per K-step: period {v[3] / ks:.0f} ticks, own-DMA wait {v[0] / ks:.0f}, barrier wait {v[1] / ks:.0f};  "
                                  f"K-loop {v[3]:.0f}  epilogue {v[4]:.0f}  kernel {v[5]:.0f} ticks (256-row tile; MFMA demand 2560 cycles per K-step and SIMD)")
                        else:          # [vmcnt + barrier wait, K-steps, K-loop, epilogue, kernel]
                            ks = max(v[1], 1.0)
                            print(f"      two-per-CU wave {w}: {int(ks)} K-steps of 32; per K-step: period {v[2] / ks:.0f} ticks, vmcnt + barrier wait {v[0] / ks:.0f};  "
                                  f"K-loop {v[2]:.0f}  epilogue {v[3]:.0f}  kernel {v[4]:.0f} ticks (128-row tile; MFMA demand 640 cycles per K-step and wave, 1280 with the partner workgroup's wave on the SIMD)")
            print(f"C {C:5d} M {M:7d} {name:18s} bitwise {'OK ' if same else 'DIFF'} (rel {err:.1e}) repeatable {rep}   cfg7 {ms[7]:.4f} ms  two-per-CU {ms[16]:.4f} ms  "
                  f"{100 * (ms[7] / ms[16] - 1):+.1f} %   {flop / ms[16] / 1e9:.0f} TFLOP/s   auto {ms[0]:.4f} ms" + (f"   cfg3 {ms[3]:.4f} ms" if 3 in ms else ""), flush=True)
        del x, res, h4, cases
        torch.cuda.empty_cache()
    print("MISMATCHES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
