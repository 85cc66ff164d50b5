"""Summarise a rocprofv3 --kernel-trace result into a per-kernel table (the `--stats` view) as text.
Usage: python tools/prof_summary.py <results.db | out_kernel_trace.csv> [header text] > profiles/xxx.txt
(rocpd sqlite DB of the default output format, or the kernel-trace CSV of `--output-format csv`)"""
import csv
import re
import sqlite3
import sys


def short(name):
    if "gemm_pipe_kernel" in name:   # <AMODE, EPI, NT_A>: the eight-wave pipelined 256x320 kernel (round 4)
        m = re.search(r"gemm_pipe_kernel<([^>]*)>", name)
        a = [x.strip() for x in m.group(1).split(",")] if m else ["?", "?"]
        am = {"0": "dense", "1": "conv3x3", "2": "temporal3"}.get(a[0], a[0])
        ep = {"0": "linear", "1": "geglu"}.get(a[1], a[1])
        return f"gemm_pipe_kernel[{am},{ep},bf16,256x320 pipelined]"
    if "gemm_kernel" in name:
        m = re.search(r"gemm_kernel<([^>]*)>", name)
        a = [x.strip() for x in m.group(1).split(",")]
        am = {"0": "dense", "1": "conv3x3", "2": "temporal3"}[a[0]]
        ep = {"0": "linear", "1": "geglu", "2": "trans"}[a[1]]
        tile = f"{int(a[3])*int(a[5])*32}x{int(a[4])*int(a[6])*32}" if len(a) >= 7 else "128x128"
        dma = (" dma" if a[7] == "true" else " reg") if len(a) >= 8 else ""
        return f"gemm_kernel[{am},{ep},{'f32' if a[2]=='true' else 'bf16'},{tile}{dma}]"
    if "at::native" in name or "rocclr" in name:
        m = re.search(r"(\w+_kernel\w*|__amd_rocclr_\w+)", name)
        return "torch/runtime: " + (m.group(1) if m else name[:40])
    m = re.search(r"(\w+_kernel(?:<\d+>)?)", name)
    return m.group(1) if m else name[:60]


def family(name):
    """(table, label) of a kernel whose launches are split by geometry: the attention kernels (the roofline kernel is the level-0 launch of
    attn_spatial_pipe_kernel: the largest grid) and the whole GEMM family (tiled, pipelined, streaming, fused FeedForward)."""
    m = re.search(r"(attn_spatial_pipe_kernel|attn_spatial_kernel|attn_spatial_fp8qk_kernel|attn_temporal_kernel)<[^>]*>", name)
    if m:
        return "attn", m.group(0)
    m = re.search(r"(gemm_pipe_kernel|gemm_pipe2_kernel|gemm_stream_kernel|ff_fused_kernel|gemm_fp8_kernel)<([^>]*)>", name)
    if m:
        return "gemm", f"{m.group(1)}<{m.group(2).replace(' ', '')}>"
    if "gemm_kernel" in name:
        return "gemm", short(name)
    return None


def print_geometry(attn, gemm):
    print("\nattention kernels by launch geometry (grid_x = threads; the level-0 spatial launch is the largest grid):")
    for (n, gx, wx), (c, t) in sorted(attn.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:44s} wgs {gx // wx:6d} x {wx:4d}  calls {c:4d}  avg_us {t/c/1e3:9.2f}  total_ms {t/1e6:8.2f}")
    print("\nGEMM family by launch geometry (workgroups x workgroup size), top 40 by total time:")
    for (n, gx, wx), (c, t) in sorted(gemm.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {n:52s} wgs {gx // wx:6d} x {wx:4d}  calls {c:4d}  avg_us {t/c/1e3:9.2f}  total_ms {t/1e6:8.2f}")


def from_csv(path):
    """The same three tables from out_kernel_trace.csv (one row per dispatch, timestamps in ns)."""
    agg, attn, gemm = {}, {}, {}
    total = 0.0
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            total += d
            e = agg.setdefault(short(n), [0, 0.0])
            e[0] += 1; e[1] += d
            gx, wx = int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])
            fam = family(n)
            if fam is not None:
                g = (attn if fam[0] == "attn" else gemm).setdefault((fam[1], gx, wx), [0, 0.0])
                g[0] += 1; g[1] += d
    if len(sys.argv) > 2:
        print(sys.argv[2])
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:62s} {c:7d} {t/1e6:10.2f} {t/c/1e3:10.2f} {100*t/total:6.2f}")
    print(f"{'TOTAL':62s} {sum(v[0] for v in agg.values()):7d} {total/1e6:10.2f}")
    print_geometry(attn, gemm)


def main():
    if sys.argv[1].endswith(".csv"):
        return from_csv(sys.argv[1])
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    agg = {}
    for n, c, t, a, p in rows:
        k = short(n)
        e = agg.setdefault(k, [0, 0.0, 0.0])
        e[0] += c; e[1] += t; e[2] += p
    if len(sys.argv) > 2:
        print(sys.argv[2])
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for k, (c, t, p) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:62s} {c:7d} {t/1e3:10.2f} {t/c:10.2f} {p:6.2f}")
    print(f"{'TOTAL':62s} {sum(v[0] for v in agg.values()):7d} {sum(v[1] for v in agg.values())/1e3:10.2f}")
    # the attention kernels and the GEMM family split by launch geometry (one line per distinct problem shape class), so that the roofline
    # kernel's average duration (the level-0 launch) can be read next to bench.py's `roofline.avg_ms`
    attn, gemm = {}, {}
    try:
        rows = list(db.execute("select name, grid_x, workgroup_x, count(*), sum(duration) from kernels group by name, grid_x, workgroup_x"))
    except sqlite3.Error:
        rows = []
    for n, gx, wx, c, tot in rows:
        fam = family(n)
        if fam is not None:
            g = (attn if fam[0] == "attn" else gemm).setdefault((fam[1], gx, wx), [0, 0.0])
            g[0] += c; g[1] += tot
    if rows:
        print_geometry(attn, gemm)

if __name__ == "__main__":
    main()
