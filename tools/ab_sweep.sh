#!/bin/bash
# Same-box A/B of two library builds: alternates base / new on the GEMM sweep and the bench.  usage: tools/ab_sweep.sh <base.so> [rounds]
base=$1; rounds=${2:-2}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  VISTA_HIP_LIB=$base python tools/gemm_sweep2.py > gpurun_out/ab_base_$r.jsonl 2>/dev/null
  python tools/gemm_sweep2.py > gpurun_out/ab_new_$r.jsonl 2>/dev/null
  VISTA_HIP_LIB=$base python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tr "," "\n" | grep ms_per_step | sed "s/^/base $r /"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tr "," "\n" | grep ms_per_step | sed "s/^/new  $r /"
done
python - <<'PY'
import glob, json
def load(pat):
    best = {}
    for f in sorted(glob.glob(pat)):
        for ln in open(f):
            d = json.loads(ln)
            k = (d["level_C"], d["kind"])
            best[k] = min(best.get(k, 1e9), d["ms_by_flags"]["0"])
    return best
b, n = load("gpurun_out/ab_base_*.jsonl"), load("gpurun_out/ab_new_*.jsonl")
for k in b:
    print(f"{k[0]:5d} {k[1]:18s} base {b[k]:.4f} ms  new {n[k]:.4f} ms  {100 * (b[k] / n[k] - 1):+.1f} %")
PY
