cd $GRAFT_REPO_ROOT; O=gpurun_out/c11; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -s -k "cfg_streams or hipgraph" > $O/test.log 2>&1; echo "rc $?" >> $O/test.log
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_two_$i.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --one-stream > $O/bench_one_$i.log 2>&1
done
grep -h '^{' $O/bench_*.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['cfg_streams'], d['ms_per_step'], d['roofline']['avg_ms'], d['step_mfma_frac'])
" > $O/ab.txt
