R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
python -m pytest tests -q -m gpu -k "attn or attention" -x > $O/test_attn.txt 2>&1; tail -n 3 $O/test_attn.txt
bash tools/attn_lib_ab.sh $R/gpurun_base.so 4 > $O/attn_store_ab.txt 2>&1; cat $O/attn_store_ab.txt
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_t
  VISTA_ATTN_PIPE=1 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_t -o p -- python $R/tools/one_kernel.py attn 0 > /tmp/pmc_t.log 2>&1
  db=$(find /tmp/pmc_t -name '*.db' | head -1)
  echo "== VISTA_ATTN_PIPE=1 level 0, counter: $set" >> $O/pmc_traffic.txt
  python $R/tools/pmc_summary.py "$db" attn_spatial 2>&1 | grep -v "^cols" >> $O/pmc_traffic.txt
done
cat $O/pmc_traffic.txt
