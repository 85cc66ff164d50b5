# SQ counter passes for the level-0 GEMM-family launches (tools/one_kernel.py <kind> 0)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r03_pmc_gemm_sq.txt
: > $OUT
for kind in conv geglu ffout linear; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
    rm -rf /tmp/pmc_g
    rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_g -o p -- python $R/tools/one_kernel.py $kind 0 > /tmp/pmc_g.log 2>&1
    db=$(find /tmp/pmc_g -name '*.db' | head -1)
    echo "== $kind level 0, counters: $set" >> $OUT
    python $R/tools/pmc_summary.py "$db" gemm_kernel 2>&1 | grep -v "^cols" >> $OUT
  done
done
