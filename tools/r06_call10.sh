cd $GRAFT_REPO_ROOT; O=gpurun_out/c10; mkdir -p $O
timeout 600 python -m pytest tests/test_parallel_gpu.py -q -x -s -k "single_rank_group_over_rccl" > $O/rccl1.log 2>&1; echo "rc $?" >> $O/rccl1.log
timeout 900 python tools/two_stream_probe.py --steps 6 > $O/two_stream.log 2>&1; echo "rc $?" >> $O/two_stream.log
