# Round-4 measurement set in ONE gpurun call: default bench line, rocprofv3 kernel trace of the bench, per-step kernel table, SQ counters of the
# fused FeedForward / level-0 GEMM family, HBM traffic of the level-0 attention and the fused FeedForward.   usage (GPU box): bash tools/prof_r04.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/r04_bench_default.json
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/bench_b.log 2>&1
db=$(find /tmp/prof_b -name '*.db' | head -1)
grep '^{' /tmp/bench_b.log > $O/r04_kernel_trace_bench_line.json
python $R/tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (round 4; 4 steps traced: warm-up, 2 timed, 1 idle-stream enqueue step): $(python -c "import json;d=json.loads(open('$O/r04_kernel_trace_bench_line.json').read());print('ms_per_step', round(d['ms_per_step'],2), 'attention avg_ms by HIP events', round(d['roofline']['avg_ms'],3))")" > $O/r04_kernel_trace.txt 2>&1
python $R/tools/rank_proxy.py --world 1 --steps 2 --torch-profile > $O/r04_step_kernels_single.txt 2>/dev/null
for kind in ffused conv ffout linear; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
    rm -rf /tmp/pmc_g
    rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_g -o p -- python $R/tools/one_kernel.py $kind 0 > /tmp/pmc_g.log 2>&1
    db=$(find /tmp/pmc_g -name '*.db' | head -1)
    echo "== $kind level 0, counters: $set" >> $O/r04_pmc_gemm_sq.txt
    python $R/tools/pmc_summary.py "$db" "$( [ $kind = ffused ] && echo ff_fused_kernel || ( [ $kind = linear ] && echo gemm_stream_kernel || echo gemm_pipe_kernel ) )" 2>&1 | grep -v "^cols" >> $O/r04_pmc_gemm_sq.txt
  done
done
for kind in attn ffused; do
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pmc_t
    rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_t -o p -- python $R/tools/one_kernel.py $kind 0 > /tmp/pmc_t.log 2>&1
    db=$(find /tmp/pmc_t -name '*.db' | head -1)
    echo "== $kind level 0, counter: $set" >> $O/r04_pmc_traffic.txt
    python $R/tools/pmc_summary.py "$db" "$( [ $kind = ffused ] && echo ff_fused_kernel || echo attn_spatial_kernel )" 2>&1 | grep -v "^cols" >> $O/r04_pmc_traffic.txt
  done
done
