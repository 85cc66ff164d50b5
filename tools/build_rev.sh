#!/bin/bash
# Build libvista_hip.so of another git revision for same-box A/B timing:  tools/build_rev.sh <rev> <out.so>
# (boxes handed out by gpurun differ by +-5 %, so a kernel change is only measurable against its predecessor in the same call:
#  VISTA_HIP_LIB=<out.so> python tools/gemm_sweep2.py   vs the in-tree library). The .so is git-ignored.
set -e
rev=$1; out=$2
tmp=$(mktemp -d)
git archive "$rev" vista_amd/csrc include | tar -x -C "$tmp"
objs=""
for f in "$tmp"/vista_amd/csrc/*.hip; do
  o="$tmp/$(basename "$f" .hip).o"
  extra=""
  case "$(basename "$f")" in ff_fused.hip|attention.hip) extra="-fno-slp-vectorize";; esac   # vista_amd/build.py EXTRA_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -I"$tmp/include" -I"$tmp/vista_amd/csrc" -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
rm -rf "$tmp"
echo "$out"
