# round-6 GPU session 1: GroupNorm fold tests + A/B, attention no-fallback A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gnstat_gpu.py tests/test_kernels_gpu.py -k "groupnorm or gn" -x -q > gpurun_out/c1_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c1_tests.log
bash tools/attn_lib_ab.sh $GRAFT_REPO_ROOT/build_ab/libvista_nofb.so 3 > gpurun_out/c1_attn_ab.log 2>&1
for r in 1 2; do for f in 0 1; do
  VISTA_GN_FOLD=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GN_FOLD=$f', d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> gpurun_out/c1_bench_ab.log 2>&1
done; done
for r in 1 2; do for lib in "$GRAFT_REPO_ROOT/build_ab/libvista_nofb.so" ""; do
  VISTA_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=${lib:-in-tree}', d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> gpurun_out/c1_bench_ab.log 2>&1
done; done
