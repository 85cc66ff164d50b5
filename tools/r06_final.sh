# round-6 final measurement set in ONE gpurun call: full GPU suite, smoke, driver-form bench (graph replay + cfg_streams default) and its --one-stream
# form on the same box, kernel trace, per-step table, rank proxies, rollout (config 4) with the graph / two-stream options
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q -x -s > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
cd /tmp
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.log 2>&1; grep '^{' $O/bench_driver_form.log > $O/r06_bench_driver_form.json
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --one-stream --no-cpu-baseline --no-extras > $O/bench_one_stream.log 2>&1; grep '^{' $O/bench_one_stream.log > $O/r06_bench_one_stream.json
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/bench_b.log 2>&1
db=$(find /tmp/prof_b -name '*.db' | head -1)
grep '^{' /tmp/bench_b.log > $O/r06_kernel_trace_bench_line.json
python $R/tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (round 6, final tree: two concurrent 25-image hipGraphs per step in the timed region -- their launches are the 4500 / 2250-workgroup attention rows and run SHARING the chip, so their durations are not standalone times; traced besides: eager warm-up forwards + captures, 1 idle-stream step, 3 eager 50-image steps around the attention timing = the 9000-workgroup rows the roofline object is computed from)" > $O/r06_kernel_trace.txt 2>&1
cd $R
python tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/r06_step_kernels.txt 2>/dev/null
( python tools/rank_proxy.py --world 8 --mode hybrid --steps 3; python tools/rank_proxy.py --world 8 --mode frames --steps 3 ) 2>/dev/null | grep "^{" > $O/r06_rank_proxy.txt
python tools/rank_proxy.py --world 8 --mode hybrid --steps 2 --torch-profile > $O/r06_rank_step_kernels.txt 2>/dev/null
python tools/rollout_bench.py --rounds 4 --steps 50 > $O/rollout_eager.log 2>&1
VISTA_HIPGRAPH=1 VISTA_CFG_STREAMS=1 python tools/rollout_bench.py --rounds 4 --steps 50 > $O/rollout_graph_streams.log 2>&1
echo done > $O/done.txt
