R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_4
mkdir -p $O
cd $R
VISTA_ATTN_PIPE=1 python tools/attn_pipe_dbg.py 2304 > $O/dbg_mode1.txt 2>&1
for m in 1 3; do
  ( VISTA_ATTN_PIPE=$m timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attn_spatial" ) > $O/test_attn_mode$m.txt 2>&1
done
( time timeout 900 python tools/attn_pipe_ab.py 2 0,1,3 ) > $O/attn_pipe_ab.txt 2>&1
echo done > $O/done.txt
