# round-6 GPU session 6: L2 prefetch of the dense activation rows two K-steps ahead (gemm_pipe_kernel<..., PF>): tests, per-shape A/B, step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c6; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_kernels_gpu.py -k "gemm or linear or geglu or pipe" -q -x > $O/tests_gemm.log 2>&1; echo "rc $?" >> $O/tests_gemm.log
for r in 1 2; do for m in 0 1; do
  echo "== VISTA_GEMM_PF=$m round $r" >> $O/pf_probe.log
  VISTA_GEMM_PF=$m timeout 600 python tools/gemm_pipe2_probe.py 2>&1 | grep "^C " | cut -c1-120 >> $O/pf_probe.log
done; done
for r in 1 2 3; do for m in 0 1; do
  VISTA_GEMM_PF=$m python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PF=$m', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))" >> $O/pf_bench_ab.log 2>&1
done; done
echo done > $O/done.txt
