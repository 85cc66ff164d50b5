"""GroupNorm apply pass with / without non-temporal hints (VISTA_GN_NT: bit 0 = loads, bit 1 = stores), one process per mode, alternated:
ms per launch of vk_groupnorm_apply_bf16 at the three BASELINE levels (50 images) + a checksum (the modes are bitwise equal), followed by the
consumer's view: the same apply immediately followed by the 3x3 convolution that reads its output (ms of the pair).
usage (GPU box): python tools/gn_nt_ab.py [rounds]"""
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
        N, S = 50, H * W
        torch.manual_seed(0)
        x = torch.randn(N, S, C, device="cuda").to(torch.bfloat16)
        gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        pc = ops.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5, torch.randn(C))
        y = torch.empty_like(x)

        def t(fn, n=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        ms_gn = t(lambda: ops.groupnorm(x, gamma, beta, 1e-5, True, out=y))
        ms_pair = t(lambda: ops.conv3x3(ops.groupnorm(x, gamma, beta, 1e-5, True, out=y), pc, N, H, W))
        out[f"C{C}"] = {"gn_ms": ms_gn, "gn_conv_ms": ms_pair, "checksum": float(y.float().double().sum().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    if "--inner" in sys.argv:
        inner()
    else:
        for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
            for mode in (0, 1, 2, 3):
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "--inner"], env=dict(os.environ, VISTA_GN_NT=str(mode)), capture_output=True, text=True)
                line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
                if not line:
                    print(f"mode {mode}: FAILED\n{res.stderr[-1500:]}")
                    continue
                d = json.loads(line[-1])
                print(f"mode {mode} round {r}: " + "  ".join(f"{k}: gn {v['gn_ms']:.4f} gn+conv {v['gn_conv_ms']:.4f} sum {v['checksum']:.4f}" for k, v in d.items()), flush=True)
