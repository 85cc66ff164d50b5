"""How much of the two-stream step do the two guidance halves really overlap?  Sweep over the dispatch intervals of a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras`) inside the window
spanned by the half graphs' level-0 attention launches (4500 workgroups of 256 threads, the longest launches of a half graph).
    python tools/stream_overlap.py <..._kernel_trace.csv>
Prints the share of the window with 0 / 1 / >= 2 kernels in flight, and per kernel family the time it spends alone vs sharing the chip."""
import csv
import re
import sys
from collections import defaultdict


def fam(n):
    m = re.search(r"(attn_spatial_pipe_kernel|attn_spatial_kernel|attn_temporal_kernel|gemm_pipe_kernel|gemm_stream_kernel|ff_fused_kernel|gemm_kernel|gn_apply_kernel|gn_stats_kernel|splitk_finish_kernel)", n)
    return m.group(1) if m else "other"


def main():
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    marks = [r for r in rows if "attn_spatial_pipe_kernel" in r[2] and r[3] == 4500 and r[1] - r[0] > 1_500_000]
    if len(marks) < 4:
        print("no half-graph attention launches found")
        return 1
    # bench.py --steps K --warmup W traces: 10 launches of the two eager warm-up forwards before the captures, 10 (W + K) of the replayed steps (back to back,
    # GPU-bound), 10 of the idle-stream step; argv[2:4] = (skip, take) select the replayed steps (default: everything)
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    take = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - skip
    marks = marks[skip:skip + take]
    t0, t1 = marks[0][0], marks[-1][1]
    ev = []
    for s, e, n, _ in rows:
        if e <= t0 or s >= t1:
            continue
        ev.append((max(s, t0), 1, n))
        ev.append((min(e, t1), -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    depth, last = 0, t0
    hist = defaultdict(int)
    active = defaultdict(int)
    alone, shared = defaultdict(int), defaultdict(int)
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            hist[min(depth, 3)] += dt
            for k, c in active.items():
                if c > 0:
                    (alone if depth == 1 else shared)[k] += dt
        last = t
        depth += d
        active[fam(n)] += d
    tot = t1 - t0
    print(f"window {tot / 1e6:.2f} ms ({len(marks)} half-graph level-0 attention launches): no kernel in flight {100 * hist[0] / tot:.1f} %, one {100 * hist[1] / tot:.1f} %, "
          f"two {100 * hist[2] / tot:.1f} %, three or more {100 * hist[3] / tot:.1f} %")
    print(f"{'family':28s} {'alone ms':>10s} {'sharing ms':>11s}")
    for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k])):
        print(f"{k:28s} {alone[k] / 1e6:10.2f} {shared[k] / 1e6:11.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
