cd $GRAFT_REPO_ROOT; O=gpurun_out/c12; mkdir -p $O
timeout 900 python tools/two_stream_probe.py --steps 6 > $O/two_stream.log 2>&1; echo "rc $?" >> $O/two_stream.log
