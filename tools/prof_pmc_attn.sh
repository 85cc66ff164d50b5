# SQ counter passes for the level-0 spatial-attention launch (tools/one_kernel.py attn 0), plain (ATTN_LOG2=0) vs pre-scaled zero-base form.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r03_pmc_attn_sq.txt
: > $OUT
for mode in 0 1; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    rm -rf /tmp/pmc_a
    ATTN_LOG2=$mode rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_a -o p -- python $R/tools/one_kernel.py attn 0 > /tmp/pmc_a.log 2>&1
    db=$(find /tmp/pmc_a -name '*.db' | head -1)
    echo "== ATTN_LOG2=$mode counters: $set" >> $OUT
    python $R/tools/pmc_summary.py "$db" attn_spatial 2>&1 | grep -v "^cols" >> $OUT
  done
done
