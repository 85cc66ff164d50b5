# round-6 GPU session 7: the prefetch launch rule (<= one round of tiles) on the rank proxy and on the single-GPU step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c7; mkdir -p $O
cd $R; export TMPDIR=/tmp
for r in 1 2; do for m in 0 rule; do
  if [ $m = rule ]; then unset VISTA_GEMM_PF; else export VISTA_GEMM_PF=$m; fi
  echo "PF=$m" >> $O/pf_rank_ab.log
  ( python tools/rank_proxy.py --world 8 --mode hybrid --steps 4; python tools/rank_proxy.py --world 8 --mode frames --steps 4 ) 2>/dev/null | grep "^{" >> $O/pf_rank_ab.log
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PF=$m', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))" >> $O/pf_step_ab.log 2>&1
done; done
unset VISTA_GEMM_PF
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "gemm or linear or geglu or pipe" -q -x > $O/tests_gemm.log 2>&1; echo "rc $?" >> $O/tests_gemm.log
echo done > $O/done.txt
