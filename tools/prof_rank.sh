# rocprofv3 kernel trace of the compute-only rank proxy (tools/rank_proxy.py); summary -> gpurun_out/r03_rank_proxy_trace_<tag>.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
python $R/tools/rank_proxy.py --world 8 --steps 5 > $R/gpurun_out/r03_rank_proxy_$TAG.txt 2>/dev/null
python $R/tools/rank_proxy.py --world 8 --mode frames --steps 5 >> $R/gpurun_out/r03_rank_proxy_$TAG.txt 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_rank -o p -- python $R/tools/rank_proxy.py --world 8 --steps 5 > /tmp/rank.log 2>&1
db=$(find /tmp/prof_rank -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" "rank_proxy --world 8 --steps 5 (6 steps traced incl. the warm-up step): $(grep '^{' /tmp/rank.log)" > $R/gpurun_out/r03_rank_proxy_trace_$TAG.txt 2>&1
