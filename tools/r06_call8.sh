# round-6 GPU session 8: mixed storage (first stage / conditioner in bf16 inside an fp16 process), bf16 process unaffected
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c8; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_f16_gpu.py -q -x -s > $O/tests_f16.log 2>&1; echo "rc $?" >> $O/tests_f16.log
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_conditioner_gpu.py tests/test_rollout_gpu.py tests/test_reward_gpu.py -q -x > $O/tests_stage.log 2>&1; echo "rc $?" >> $O/tests_stage.log
VISTA_ACT_DTYPE=fp16 timeout 900 python -m pytest tests/test_rollout_gpu.py -q -x -s > $O/tests_rollout_in_f16_process.log 2>&1; echo "rc $?" >> $O/tests_rollout_in_f16_process.log
echo done > $O/done.txt
