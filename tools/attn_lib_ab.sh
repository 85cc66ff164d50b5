# Same-box A/B of the product-form spatial attention between two library builds (tools/build_rev.sh <rev> base.so vs the in-tree library),
# alternated, each in its own process.   usage (GPU box): bash tools/attn_lib_ab.sh <base.so> [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
base=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for lib in "$base" ""; do
    VISTA_HIP_LIB=$lib python $R/tools/attn_pipe_ab.py --inner | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${lib:-in-tree}'.split('/')[-1].ljust(16), '  '.join(f'{k}: {v[\"ms\"]:.4f} ms' for k, v in d.items()))"
  done
done
