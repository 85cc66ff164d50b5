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
            for tab, key in ((attn, "attn_spatial_kernel"), (gemm, "gemm_kernel")):
                if key in n:
                    g = tab.setdefault((short(n) if key == "gemm_kernel" else re.search(r"attn_spatial_kernel<[^>]*>", n).group(0), gx, wx), [0, 0.0])
                    g[0] += 1; g[1] += d
    if len(sys.argv) > 2:
        print(sys.argv[2])
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:62s} {c:7d} {t/1e6:10.2f} {t/c/1e3:10.2f} {100*t/total:6.2f}")
    print(f"{'TOTAL':62s} {sum(v[0] for v in agg.values()):7d} {total/1e6:10.2f}")
    print("\nattn_spatial_kernel by launch geometry (grid_x = threads):")
    for (n, gx, wx), (c, t) in sorted(attn.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:28s} grid_x {gx:9d} wg {wx:4d}  calls {c:4d}  avg_us {t/c/1e3:9.2f}  total_ms {t/1e6:8.2f}")
    print("\ngemm_kernel by launch geometry (grid_x = threads = workgroups x workgroup size), top 28 by total time:")
    for (n, gx, wx), (c, t) in sorted(gemm.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  {n:44s} wgs {gx // wx:6d} x {wx:4d}  calls {c:4d}  avg_us {t/c/1e3:9.2f}  total_ms {t/1e6:8.2f}")


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
    # the roofline kernel, split by launch geometry (grid x = workgroups: level 0 of the UNet is the largest grid), so that its
    # average duration can be compared with bench.py's `roofline.avg_ms` (which times only the level-0 launches)
    try:
        rows = list(db.execute("select name, grid_x, workgroup_x, count(*), avg(duration), sum(duration) from kernels "
                               "where name like '%attn_spatial_kernel%' group by name, grid_x, workgroup_x order by sum(duration) desc"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\nattn_spatial_kernel by launch geometry (grid_x = threads):")
        for n, gx, wx, c, avg, tot in rows:
            print(f"  {short(n):28s} grid_x {gx:9d} wg {wx:4d}  calls {c:4d}  avg_us {avg/1e3:9.2f}  total_ms {tot/1e6:8.2f}")
    # the GEMM family by launch geometry (one line per distinct problem shape class): where the step's GEMM time goes
    try:
        rows = list(db.execute("select name, grid_x, workgroup_x, count(*), avg(duration), sum(duration) from kernels "
                               "where name like '%gemm_kernel%' group by name, grid_x, workgroup_x order by sum(duration) desc limit 28"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\ngemm_kernel by launch geometry (grid_x = threads = workgroups x workgroup size), top 28 by total time:")
        for n, gx, wx, c, avg, tot in rows:
            print(f"  {short(n):44s} wgs {gx // wx:6d} x {wx:4d}  calls {c:4d}  avg_us {avg/1e3:9.2f}  total_ms {tot/1e6:8.2f}")


if __name__ == "__main__":
    main()
