cd $GRAFT_REPO_ROOT; O=gpurun_out/c13; mkdir -p $O
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_def_$i.log 2>&1
VISTA_SPLITK_WS_MB=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_nosplitk_$i.log 2>&1
done
for f in $O/bench_*.log; do echo $f $(grep -h '^{' $f | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['cfg_streams'], d['ms_per_step'], d['roofline']['avg_ms'], d['step_mfma_frac'])
"); done > $O/ab.txt
