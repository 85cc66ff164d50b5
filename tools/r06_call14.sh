cd $GRAFT_REPO_ROOT; O=gpurun_out/c14; rm -rf $O; mkdir -p $O

export VISTA_DIST_BACKEND=nccl
VISTA_HIPGRAPH=force timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 tests/_dist_worker.py rccl1 > $O/rccl1_graph_force.log 2>&1; echo "rc $?" >> $O/rccl1_graph_force.log
VISTA_HIPGRAPH=force VISTA_A2A_CHUNKS=2 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29712 tests/_dist_worker.py rccl1 > $O/rccl1_graph_force_chunks2.log 2>&1; echo "rc $?" >> $O/rccl1_graph_force_chunks2.log
