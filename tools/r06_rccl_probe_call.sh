cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rccl
for w in 1 2; do
  for ign in 0 1; do
    echo "== world $w NCCL_IGNORE_DUP $ign" >> gpurun_out/rccl/log.txt
    NCCL_DEBUG=WARN RCCL_IGNORE_DUPLICATE_GPU=$ign timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29700+w*2+ign)) tools/probes/rccl_one_gpu.py >> gpurun_out/rccl/log.txt 2>&1
    echo "rc $?" >> gpurun_out/rccl/log.txt
  done
done
