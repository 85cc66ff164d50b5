# rocprofv3 --kernel-trace --stats of the default bench command; summary -> gpurun_out/r03_kernel_trace_<tag>.txt (+ the bench line it produced)
R=$GRAFT_REPO_ROOT
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/bench_b.log 2>&1
db=$(find /tmp/prof_b -name '*.db' | head -1)
grep '^{' /tmp/bench_b.log > $R/gpurun_out/r03_kernel_trace_${TAG}_bench_line.json
python $R/tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (final tree of round 3; 4 steps traced: warm-up, 2 timed, 1 idle-stream enqueue step): $(python -c "import json;d=json.loads(open('$R/gpurun_out/r03_kernel_trace_${TAG}_bench_line.json').read());print('ms_per_step', round(d['ms_per_step'],2), 'attention avg_ms by HIP events', round(d['roofline']['avg_ms'],3))")" > $R/gpurun_out/r03_kernel_trace_$TAG.txt 2>&1
