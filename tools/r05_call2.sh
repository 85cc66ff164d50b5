#!/bin/bash
# GPU call 2 (round 5): epilogue-statistics tests again (fixed case list), temporal attention variants A/B + its kernel tests per variant
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/call2; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gnstat_gpu.py -q --timeout 300 > $O/tests_gnstat.txt 2>&1; echo "gnstat tests rc=$?" | tee -a $O/summary.txt
tail -6 $O/tests_gnstat.txt
timeout 300 python tools/attn_t_ab.py 2 > $O/attn_t_ab.txt 2>&1; cat $O/attn_t_ab.txt
for m in 3 0; do VISTA_ATTN_T=$m timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "temporal" --timeout 250 > $O/tests_attn_t_$m.txt 2>&1; echo "attn_temporal tests mode $m rc=$?" | tee -a $O/summary.txt; tail -2 $O/tests_attn_t_$m.txt; done
