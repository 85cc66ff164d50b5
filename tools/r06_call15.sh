cd $GRAFT_REPO_ROOT; O=gpurun_out/c15; rm -rf $O; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
for i in 1 2; do
$B > $O/def_$i.log 2>&1
VISTA_GEGLU_WALK=0 $B > $O/geglu0_$i.log 2>&1
VISTA_FF_WALK=0 $B > $O/ff0_$i.log 2>&1
VISTA_GEGLU_WALK=0 VISTA_FF_WALK=0 $B > $O/both0_$i.log 2>&1
done
for f in $O/*.log; do echo $(basename $f) $(grep -h '^{' $f | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['roofline']['avg_ms'])
"); done > $O/ab.txt
