#!/bin/bash
# A/B build of libvista_hip.so with extra compiler flags:  tools/build_variant.sh <out.so> [flags...]
#   -DVK_EPI_NT_STORES=1      LDS-staged GEMM epilogues store with the non-temporal hint (round 5: +7.8 ms per step, not adopted)
#   -DVK_ATTN_NO_FALLBACK     attn_spatial_pipe_kernel without its inlined general-kernel fallback (round 6: is its scratch the cost?)
# then VISTA_HIP_LIB=<out.so> python bench.py ... against the in-tree library on the same box. The .so is git-ignored and built in a temp dir.
set -e
root=$(cd "$(dirname "$0")/.." && pwd); out=$(realpath -m "$1"); shift; tmp=$(mktemp -d); objs=""
for f in "$root"/vista_amd/csrc/*.hip; do
  o="$tmp/$(basename "$f" .hip).o"; extra=""
  case "$(basename "$f")" in ff_fused.hip|attention.hip) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -I"$root/include" -I"$root/vista_amd/csrc" -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
# link from inside the temp dir: the offload bundler drops its per-target intermediates into the cwd
(cd "$tmp" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs)
rm -rf "$tmp"; echo "$out"
