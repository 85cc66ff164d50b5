# round-6 GPU session 4: pipe2 tests + phase timers, GroupNorm nt-mode step A/B, rank proxies + rank-step kernel table, hipGraph step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "pipe2" -q -x > $O/tests_pipe2.log 2>&1; echo "rc $?" >> $O/tests_pipe2.log
VISTA_HIP_LIB=$R/build_ab/libvista_timing.so PROBE_TIMING=1 timeout 900 python tools/gemm_pipe2_probe.py > $O/pipe2_timing.log 2>&1
for r in 1 2; do for m in rule 2; do
  if [ $m = rule ]; then unset VISTA_GN_NT; else export VISTA_GN_NT=$m; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GN_NT=$m', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))" >> $O/gn_nt_step_ab.log 2>&1
done; done
unset VISTA_GN_NT
( python tools/rank_proxy.py --world 8 --mode hybrid --steps 3; python tools/rank_proxy.py --world 8 --mode frames --steps 3 ) > $O/r06_rank_proxy.txt 2>&1
python tools/rank_proxy.py --world 8 --mode hybrid --steps 2 --torch-profile > $O/r06_rank_step_kernels.txt 2>/dev/null
python bench.py --graph --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_graph.json
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_eager.json
echo done > $O/done.txt
