"""Temporal attention kernel (csrc/attention.hip::attn_temporal_kernel) variants, same box, alternated processes: VISTA_ATTN_T bit 0 = wave-local
synchronisation instead of two workgroup barriers per problem, bit 1 = 16-byte output stores. Prints ms per launch at the three BASELINE levels,
the algorithmic HBM rate (q | k | v read once, o written once) and a checksum of the output (the variants are bitwise equal).
usage (GPU box): python tools/attn_t_ab.py            (spawns itself once per mode and round)"""
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
    for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
        S, heads, B, T = H * W, C // 64, 2, 25
        M = B * T * S
        torch.manual_seed(0)
        qkv = torch.randn(M, 3 * C, device="cuda").to(torch.bfloat16)
        o = ops.attn_temporal(qkv, B, T, S, heads)
        for _ in range(3):
            ops.attn_temporal(qkv, B, T, S, heads)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attn_temporal(qkv, B, T, S, heads)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out[f"C{C}"] = {"ms": ms, "TBps": 8.0 * M * C / ms / 1e9, "checksum": float(o.float().double().sum().item()), "absmax": float(o.float().abs().max().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    if "--inner" in sys.argv:
        inner()
    else:
        for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
            for mode in (0, 1, 2, 3):
                env = dict(os.environ, VISTA_ATTN_T=str(mode))
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "--inner"], env=env, capture_output=True, text=True)
                line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
                if not line:
                    print(f"mode {mode}: FAILED\n{res.stderr[-2000:]}")
                    continue
                d = json.loads(line[-1])
                print(f"mode {mode} round {r}: " + "  ".join(f"{k}: {v['ms']:.4f} ms {v['TBps']:.2f} TB/s sum {v['checksum']:.6f}" for k, v in d.items()), flush=True)
