# round-6 GPU session 2: fp16 build first contact, full GPU suite with measured parity lines, attention traffic re-stamp, default bench, kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2; mkdir -p $O
cd $R; export TMPDIR=/tmp
VISTA_ACT_DTYPE=fp16 timeout 900 python tests/_f16_worker.py --full-size > $O/f16_worker.log 2>&1; echo "rc $?" >> $O/f16_worker.log
timeout 2400 python -m pytest tests -m gpu -q -s -x > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_t
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_t -o p -- python $R/tools/one_kernel.py attn 0 > /tmp/pmc_t.log 2>&1
  db=$(find /tmp/pmc_t -name '*.db' | head -1)
  echo "== level 0 attn_spatial_pipe_kernel, counter: $set" >> $O/r06_pmc_traffic.txt
  python $R/tools/pmc_summary.py "$db" attn_spatial 2>&1 | grep -v "^cols" >> $O/r06_pmc_traffic.txt
done
python $R/bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r06_bench_default.json
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/bench_b.log 2>&1
db=$(find /tmp/prof_b -name '*.db' | head -1)
grep '^{' /tmp/bench_b.log > $O/r06_kernel_trace_bench_line.json
python $R/tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (round 6, first tree)" > $O/r06_kernel_trace.txt 2>&1
python $R/tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/r06_step_kernels_single.txt 2>/dev/null
echo done > $O/done.txt
