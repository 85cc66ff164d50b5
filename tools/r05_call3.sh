R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_3
mkdir -p $O
cd $R
for m in 0 1; do VISTA_ATTN_PIPE=$m python tools/attn_pipe_dbg.py 2304 > $O/dbg_mode$m.txt 2>&1; done
( VISTA_EPI_ROWS=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_pipe_is_bitwise or linear_emit_rowstats or layernorm_fold" ) > $O/test_epirows.txt 2>&1
( VISTA_EPI_ROWS=0 PROBE_FAST=3 PROBE_KINDS=qkv,ff_out,conv,linear python tools/gemm_pipe_probe.py ) > $O/probe_rows0.txt 2>&1
( VISTA_EPI_ROWS=1 PROBE_FAST=3 PROBE_KINDS=qkv,ff_out,conv,linear python tools/gemm_pipe_probe.py ) > $O/probe_rows1.txt 2>&1
for r in 0 1 0 1; do
  VISTA_ATTN_PIPE=0 VISTA_EPI_ROWS=$r python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras >> $O/bench_rows$r.json 2>> $O/bench_rows$r.err
done
echo done > $O/done.txt
