# round-6 GPU session 3: the two-per-CU pipelined GEMM (tests, probe, step A/B); fp16 vs bf16 per-kernel tables on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "pipe2" -q -x > $O/tests_pipe2.log 2>&1; echo "rc $?" >> $O/tests_pipe2.log
timeout 900 python tools/gemm_pipe2_probe.py > $O/pipe2_probe.log 2>&1
for r in 1 2; do for m in 0 1 2 3; do
  VISTA_GEMM_PIPE2=$m python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PIPE2=$m', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))" >> $O/pipe2_bench_ab.log 2>&1
done; done
python tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/step_kernels_bf16.txt 2>/dev/null
VISTA_ACT_DTYPE=fp16 python tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/step_kernels_fp16.txt 2>/dev/null
echo done > $O/done.txt
