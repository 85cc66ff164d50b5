#!/bin/bash
# Same-box comparison of library variants built from one source with different -D flags (gpurun_var_<name>.so at the repo root; "cur" = the
# in-tree library): runs tools/gemm_pipe_probe.py on each, interleaved twice, and prints the cfg7 time per shape and variant.  usage: tools/var_sweep.sh v1 v2 ...
mkdir -p gpurun_out/var
for r in 1 2; do
  for v in cur "$@"; do
    if [ $v = cur ]; then unset VISTA_HIP_LIB; else export VISTA_HIP_LIB=$PWD/gpurun_var_$v.so; fi
    env ${PROBE_ENV:-PROBE_FAST=3} python tools/gemm_pipe_probe.py > gpurun_out/var/${v}_$r.txt 2>&1
  done
done
unset VISTA_HIP_LIB
python - "$@" <<'PY'
import glob, re, sys
names = ["cur"] + sys.argv[1:]
best = {}
order = []
for v in names:
    for f in glob.glob(f"gpurun_out/var/{v}_*.txt"):
        for ln in open(f):
            m = re.match(r"C\s+(\d+) M\s+(\d+) (\S+)\s+bitwise (\S+).*cfg4 ([\d.]+) ms\s+cfg7 ([\d.]+) ms", ln)
            if m:
                k = (m.group(1), m.group(2), m.group(3))
                if k not in order: order.append(k)
                best[(v, k)] = min(best.get((v, k), 1e9), float(m.group(6)))
                a = re.search(r"auto ([\d.]+) ms", ln)
                if a: best[("auto:" + v, k)] = min(best.get(("auto:" + v, k), 1e9), float(a.group(1)))
                best[("cfg4", k)] = min(best.get(("cfg4", k), 1e9), float(m.group(5)))
                if m.group(4) != "OK": print("MISMATCH", v, ln.strip())
print("shape".ljust(36), "cfg4".rjust(8), *[n.rjust(8) for n in names])
for k in order:
    print(f"{k[0]:>5} {k[1]:>7} {k[2]:20s}", f"{best[('cfg4', k)]:8.4f}", *[f"{best.get((n, k), 0):8.4f}" for n in names])
print("\nthe launcher's own choice (tile_cfg 0):")
for k in order:
    print(f"{k[0]:>5} {k[1]:>7} {k[2]:20s}", " " * 8, *[f"{best.get(('auto:' + n, k), 0):8.4f}" for n in names])
PY
