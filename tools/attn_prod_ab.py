"""Same-box A/B of the PRODUCT form of the spatial attention (q | k | v column blocks of one GEMM, V rows, pre-scaled query) between two
library builds (VISTA_HIP_LIB selects the base build in a child process), alternated.  usage: python tools/attn_prod_ab.py <base.so> [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inner():
    import torch
    from vista_amd import ops
    out = {}
    for C, S, heads in ((320, 9216, 5), (640, 2304, 10), (1280, 576, 20)):
        n = 50
        g = torch.Generator(device="cuda").manual_seed(0)
        qkv = torch.randn(n * S, 3 * C, device="cuda", generator=g).to(torch.bfloat16)
        qkv[:, :C] *= 0.18033688
        fn = lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n, heads, S, v_rows=True, q_log2=True)  # noqa: E731
        fn(); fn()
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
        out[f"S={S}"] = best
    print(json.dumps(out))


def main():
    if sys.argv[1] == "--inner":
        return inner()
    base, rounds = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    best = {"base": {}, "new": {}}
    for _ in range(rounds):
        for tag, env in (("base", {"VISTA_HIP_LIB": base}), ("new", {})):
            e = dict(os.environ)
            e.pop("VISTA_HIP_LIB", None)
            e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--inner"], env=e, capture_output=True, text=True)
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            for k, v in d.items():
                best[tag][k] = min(best[tag].get(k, 1e9), v)
    for k in best["base"]:
        b, n = best["base"][k], best["new"][k]
        print(f"attn_spatial (product form) {k}: base {b:.4f} ms  new {n:.4f} ms  {100 * (b / n - 1):+.1f} %")


if __name__ == "__main__":
    main()
