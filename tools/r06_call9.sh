# round-6 GPU session 9: temporal conv frame-fastest tile order: tests, per-launch A/B, step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c9; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_gnstat_gpu.py -k "conv_t3 or temporal or gnstat or pipe" -q -x > $O/tests_t3.log 2>&1; echo "rc $?" >> $O/tests_t3.log
for r in 1 2; do for m in 0 1; do
  echo "== VISTA_T3_ORDER=$m round $r" >> $O/t3_probe.log
  VISTA_T3_ORDER=$m PROBE_KINDS=conv_t3 PROBE_FAST=3 timeout 600 python tools/gemm_pipe_probe.py 2>&1 | grep "^C " | cut -c1-150 >> $O/t3_probe.log
done; done
for r in 1 2 3; do for m in 0 1; do
  VISTA_T3_ORDER=$m python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('T3_ORDER=$m', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))" >> $O/t3_bench_ab.log 2>&1
done; done
echo done > $O/done.txt
