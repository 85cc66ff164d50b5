#!/bin/bash
# One GPU call for the epilogue-statistics GroupNorm (round 5): new kernel tests, per-launch A/B, step A/B (alternated), block parity.
# usage: gpurun --timeout 900 -- 'bash tools/r05_gnstat_call.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/gnstat; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gnstat_gpu.py -q -x --timeout 300 > $O/tests_gnstat.txt 2>&1; echo "gnstat tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/tests_gnstat.txt
timeout 200 python tools/gnstat_ab.py 50 > $O/ab_launch.jsonl 2> $O/ab_launch.err; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/ab_launch.jsonl | cut -c1-400
for r in 1 2; do for sw in 0 1; do
  VISTA_GN_EPI=$sw timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2> $O/bench_$sw_$r.err | tail -1 > $O/bench_${sw}_${r}.json
  python -c "import json,sys; d=json.load(open('$O/bench_${sw}_${r}.json')); print('GN_EPI=$sw run $r: %.2f ms/step, attn %.3f' % (d['ms_per_step'], d['roofline']['frac']))" | tee -a $O/summary.txt
done; done
timeout 500 python -m pytest tests/test_blocks_gpu.py -q --timeout 400 > $O/tests_blocks.txt 2>&1; echo "blocks rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests_blocks.txt


