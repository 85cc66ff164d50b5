# Round 5: the whole -m gpu suite (with the slowest tests listed) + smoke + the default bench line, one call
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
( time python -m pytest tests/ -x -q -m gpu --durations=25 ) > $O/pytest_gpu.txt 2>&1
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_form.txt 2>&1
grep '^{' $O/bench_driver_form.txt > $O/r05_bench_driver_form.json
echo done > $O/done.txt
