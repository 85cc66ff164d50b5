# Round 5: the whole -m gpu suite + smoke + the reference-hosted tests (reference files staged as temporary test data) + the rollout bench, one call
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05t
mkdir -p $O
cd $R
( time python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
if [ -d $R/_ref_testdata ]; then ( VISTA_REFERENCE=$R/_ref_testdata python -m pytest tests/test_reference_hosted_gpu.py -q -s ) > $O/reference_hosted.txt 2>&1; fi
( python tools/rollout_bench.py --rounds 4 --steps 50 ) > $O/rollout.txt 2>&1
echo done > $O/done.txt
