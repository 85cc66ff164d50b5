R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tag in attn noattn; do
  extra=""; [ $tag = noattn ] && extra="--fp8-no-attn"
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py --fp8 $extra --steps 2 --warmup 1 --no-cpu-baseline > /tmp/bench_$tag.log 2>&1
  db=$(find /tmp/prof_$tag -name '*.db' | head -1)
  [ -z "$db" ] && db=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
  ls -R /tmp/prof_$tag | head -5 >> $R/gpurun_out/r03_fp8_kernel_trace_$tag.txt
  python $R/tools/prof_summary.py "$db" "bench.py --fp8 $extra --steps 2 --warmup 1: $(grep '^{' /tmp/bench_$tag.log | cut -c100-260)" > $R/gpurun_out/r03_fp8_kernel_trace_$tag.txt 2>&1
done
