"""Same-box A/B of the level-0 / level-1 spatial attention in its PRODUCT form (q | k | v column blocks of one GEMM, V rows, pre-scaled query)
across the launcher's VISTA_ATTN_PIPE modes: 0 = un-pipelined 8 x 64-row kernel (rounds 2-4), 1 = software-pipelined, four waves, two workgroups
per CU, 2 = four waves, one workgroup per CU, 3 = eight waves. Every mode runs in its own process (the hook is read once), alternated; also
checks every mode's output against mode 0.   usage: python tools/attn_pipe_ab.py [rounds] [modes, e.g. 0,1,3]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ((320, 9216, 5, 50), (640, 2304, 10, 50), (320, 9216, 5, 7))


def inner():
    import torch
    from vista_amd import ops
    out = {}
    for C, S, heads, n in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        qkv = torch.randn(n * S, 3 * C, device="cuda", generator=g).to(torch.bfloat16)
        qkv[:, :C] *= 0.18033688
        fn = lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n, heads, S, v_rows=True, q_log2=True)  # noqa: E731
        o = fn()
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        # a fixed sample of the output (every 97th row) for the cross-mode comparison
        out[f"S={S},n={n}"] = {"ms": best, "sample": o[::97].float().cpu().flatten()[:20000].tolist()[::7], "finite": bool(torch.isfinite(o.float()).all())}
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--inner":
        return inner()
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
    best, sample = {}, {}
    for _ in range(rounds):
        for m in modes:
            e = dict(os.environ, VISTA_ATTN_PIPE=str(m))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--inner"], env=e, capture_output=True, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                print(f"mode {m} FAILED:\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
                continue
            d = json.loads(lines[-1])
            for k, v in d.items():
                best.setdefault(k, {})[m] = min(best.get(k, {}).get(m, 1e9), v["ms"])
                sample.setdefault(k, {})[m] = (v["sample"], v["finite"])
    for k, per in best.items():
        base = per.get(0)
        flop = None
        S = int(k.split(",")[0][2:]); n = int(k.split("=")[-1])
        heads = 5 if S == 9216 else 10
        flop = 4.0 * n * heads * S * S * 64
        for m, ms in sorted(per.items()):
            err = ""
            if 0 in sample[k] and m != 0:
                a, b = sample[k][m][0], sample[k][0][0]
                num = sum((x - y) ** 2 for x, y in zip(a, b)) ** 0.5
                den = sum(y * y for y in b) ** 0.5
                err = f"  rel-L2 vs mode 0 {num / den:.2e} finite {sample[k][m][1]}"
            print(f"attn (product form) {k}: mode {m}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s" + (f"  {100 * (base / ms - 1):+.1f} % vs mode 0" if base and m else "") + err)


if __name__ == "__main__":
    main()
