# Round 5, GPU call 2: the software-pipelined spatial attention kernel: parity tests under every launcher mode, same-box timing A/B, step A/B.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_2
mkdir -p $O
cd $R
for m in 1 3 2; do
  ( VISTA_ATTN_PIPE=$m timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn_spatial" ) > $O/test_attn_mode$m.txt 2>&1
done
( time timeout 900 python tools/attn_pipe_ab.py 2 0,1,2,3 ) > $O/attn_pipe_ab.txt 2>&1
( time python -m pytest tests/test_checkpoint_gpu.py -q -x -s ) > $O/test_ckpt.txt 2>&1
( time python -m pytest tests/test_fp8_gpu.py -q -x -s -k "unet_with_fp8_feedforward" ) > $O/test_fp8_tiny.txt 2>&1
for m in 0 1 3 0 1 3; do
  VISTA_ATTN_PIPE=$m python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras >> $O/bench_mode$m.json 2>> $O/bench_mode$m.err
done
echo done > $O/done.txt
