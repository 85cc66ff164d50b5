# Round 5, GPU call 1: issue-rate probe for the pipelined attention unit, tail-split A/B per shape and per step, the new parity tests.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_1
mkdir -p $O
cd $R
( time ./tools/probes/attn_issue_probe ) > $O/attn_issue_probe.txt 2>&1
( time python -m pytest tests/test_kernels_gpu.py -q -x -k "tail_split" ) > $O/test_tail.txt 2>&1
( time PROBE_FAST=3 PROBE_KINDS=qkv,ff_out,conv python tools/gemm_pipe_probe.py ) > $O/gemm_probe_tail.txt 2>&1
( time python -m pytest tests/test_checkpoint_gpu.py tests/test_model_gpu.py -q -x -s -k "checkpoint or converted or stochastic or hipgraph" ) > $O/test_new.txt 2>&1
( time python -m pytest tests/test_fp8_gpu.py -q -x -s -k "tiny or full_width or 50_step" ) > $O/test_fp8.txt 2>&1
for tail in 0 40; do
  VISTA_GEMM_TAIL=$tail python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_tail$tail.json 2> $O/bench_tail$tail.err
done
VISTA_GEMM_TAIL=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_tail0_b.json 2> $O/bench_tail0_b.err
VISTA_GEMM_TAIL=40 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_tail40_b.json 2> $O/bench_tail40_b.err
echo done > $O/done.txt
