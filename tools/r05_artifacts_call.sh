#!/bin/bash
# Round 5, evidence on the final tree in ONE gpurun call: reference-hosted checks (staged reference files), config-4 rollout, SQ counters of the
# level-0 conv3x3 with / without the statistics-emitting epilogue, full-size forward parity.   usage: tools/ship_reference_for_test.sh stage;
# gpurun --timeout 1000 -- 'bash tools/r05_artifacts_call.sh'; tools/ship_reference_for_test.sh clean
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
cd $R
VISTA_REFERENCE=$R/_ref_testdata timeout 300 python -m pytest tests/test_reference_hosted_gpu.py -q -s > $O/reference_hosted.txt 2>&1; echo "reference-hosted rc=$?" | tee $O/summary.txt
grep "reference" $O/reference_hosted.txt | cut -c1-300; tail -1 $O/reference_hosted.txt
timeout 300 python tools/rollout_bench.py --rounds 4 --steps 50 2> $O/rollout.err | tail -1 > $O/r05_rollout_full_size.json; cat $O/r05_rollout_full_size.json
cd /tmp
for kind in convemb convgn; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
    rm -rf /tmp/pmc_g
    timeout 150 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_g -o p -- python $R/tools/one_kernel.py $kind 0 > /tmp/pmc_g.log 2>&1
    db=$(find /tmp/pmc_g -name '*.db' | head -1)
    echo "== $kind level 0 (conv3x3 320->320 @72x128, 50 images, + per-image row vector), counters: $set" >> $O/r05_pmc_conv_gnstat_sq.txt
    python $R/tools/pmc_summary.py "$db" gemm_pipe 2>&1 | grep -v "^cols" >> $O/r05_pmc_conv_gnstat_sq.txt
  done
done
cat $O/r05_pmc_conv_gnstat_sq.txt
cd $R
timeout 600 python tools/full_size_parity.py $O/r05_full_size_parity.json > $O/full_size_parity.txt 2>&1; echo "full-size parity rc=$?" | tee -a $O/summary.txt; cat $O/r05_full_size_parity.json
