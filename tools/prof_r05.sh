# Round-5 measurement set in ONE gpurun call: default bench line, rocprofv3 kernel trace of the bench, per-step kernel table, SQ counters and HBM
# traffic of the level-0 attention (new pipelined kernel AND the round-4 kernel), the rank proxies.   usage (GPU box): bash tools/prof_r05.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r05_bench_default.json
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/bench_b.log 2>&1
db=$(find /tmp/prof_b -name '*.db' | head -1)
grep '^{' /tmp/bench_b.log > $O/r05_kernel_trace_bench_line.json
python $R/tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (round 5; traced: warm-up, 2 timed, 1 idle-stream enqueue step, 2 steps with HIP events around the level-0 attention launches): $(python -c "import json;d=json.loads(open('$O/r05_kernel_trace_bench_line.json').read());print('ms_per_step', round(d['ms_per_step'],2), 'attention avg_ms by HIP events', round(d['roofline']['avg_ms'],3))")" > $O/r05_kernel_trace.txt 2>&1
python $R/tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/r05_step_kernels_single.txt 2>/dev/null
for mode in 1 0; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    rm -rf /tmp/pmc_a
    VISTA_ATTN_PIPE=$mode rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_a -o p -- python $R/tools/one_kernel.py attn 0 > /tmp/pmc_a.log 2>&1
    db=$(find /tmp/pmc_a -name '*.db' | head -1)
    echo "== VISTA_ATTN_PIPE=$mode (1 = attn_spatial_pipe_kernel<4>, 0 = attn_spatial_kernel<8,2,true,true>) counters: $set" >> $O/r05_pmc_attn_sq.txt
    python $R/tools/pmc_summary.py "$db" attn_spatial 2>&1 | grep -v "^cols" >> $O/r05_pmc_attn_sq.txt
  done
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pmc_t
    VISTA_ATTN_PIPE=$mode rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_t -o p -- python $R/tools/one_kernel.py attn 0 > /tmp/pmc_t.log 2>&1
    db=$(find /tmp/pmc_t -name '*.db' | head -1)
    echo "== VISTA_ATTN_PIPE=$mode level 0, counter: $set" >> $O/r05_pmc_traffic.txt
    python $R/tools/pmc_summary.py "$db" attn_spatial 2>&1 | grep -v "^cols" >> $O/r05_pmc_traffic.txt
  done
done
( python $R/tools/rank_proxy.py --world 8 --mode hybrid --steps 3; python $R/tools/rank_proxy.py --world 8 --mode frames --steps 3 ) > $O/r05_rank_proxy.txt 2>&1
echo done > $O/done.txt
