#!/bin/bash
# Round 5, final measurement set in ONE gpurun call (ABI v6 tree: epilogue GroupNorm statistics, 16-byte stores in the temporal attention):
# HBM traffic of the level-0 attention (PMC passes), the whole -m gpu suite, smoke, the driver-form bench line, kernel trace, per-step kernel
# table, rank proxies.   usage: gpurun --timeout 1500 -- 'bash tools/r05_final_call.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_t
  VISTA_ATTN_PIPE=1 timeout 200 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_t -o p -- python $R/tools/one_kernel.py attn 0 > /tmp/pmc_t.log 2>&1
  db=$(find /tmp/pmc_t -name '*.db' | head -1)
  echo "== VISTA_ATTN_PIPE=1 level 0, counter: $set (tree with ABI v6)" >> $O/pmc_traffic.txt
  python $R/tools/pmc_summary.py "$db" attn_spatial 2>&1 | grep -v "^cols" >> $O/pmc_traffic.txt
done
cat $O/pmc_traffic.txt
cd $R
( time timeout 900 python -m pytest tests/ -q -m gpu --durations=15 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt; tail -4 $O/pytest_gpu.txt
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -3 $O/smoke.txt
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_form.txt 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
grep '^{' $O/bench_driver_form.txt > $O/r05_bench_driver_form.json; python -c "import json; d=json.load(open('$O/r05_bench_driver_form.json')); print('ms_per_step', d['ms_per_step'], 'attn', d['roofline']['frac'], d['roofline']['avg_ms'], 'fp8', d.get('config5_fp8',{}).get('ms_per_step'))" | tee -a $O/summary.txt
cd /tmp
rm -rf /tmp/prof_b; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/bench_b.log 2>&1
db=$(find /tmp/prof_b -name '*.db' | head -1)
grep '^{' /tmp/bench_b.log > $O/r05_kernel_trace_bench_line.json
python $R/tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (round 5, ABI v6 tree; traced: warm-up, 2 timed, 1 idle-stream enqueue step, 2 steps with HIP events around the level-0 attention launches): $(python -c "import json;d=json.loads(open('$O/r05_kernel_trace_bench_line.json').read());print('ms_per_step', round(d['ms_per_step'],2), 'attention avg_ms by HIP events', round(d['roofline']['avg_ms'],3))")" > $O/r05_kernel_trace.txt 2>&1
timeout 200 python $R/tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/r05_step_kernels_single.txt 2>/dev/null
( timeout 200 python $R/tools/rank_proxy.py --world 8 --mode hybrid --steps 3; timeout 200 python $R/tools/rank_proxy.py --world 8 --mode frames --steps 3 ) > $O/r05_rank_proxy.txt 2>&1
tail -3 $O/r05_rank_proxy.txt
echo done >> $O/summary.txt
