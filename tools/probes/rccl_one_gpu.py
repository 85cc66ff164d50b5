"""Probe: does RCCL accept several ranks on ONE GPU (the test boxes have one)?  Launch:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 tools/probes/rccl_one_gpu.py
Prints one JSON line per rank; a refusal ("Duplicate GPU detected") is the expected negative."""
import json, os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
out = {"rank": rank, "world": world}
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x); torch.cuda.synchronize()
    out["all_reduce"] = float(x[0])
    a = torch.arange(world * 4, device="cuda", dtype=torch.float32) + 100 * rank
    b = torch.empty_like(a)
    dist.all_to_all_single(b, a); torch.cuda.synchronize()
    out["all_to_all"] = b.tolist()
    dist.barrier(); out["ok"] = True
except Exception as e:  # noqa: BLE001
    out["ok"] = False; out["error"] = repr(e)[:600]
print(json.dumps(out), flush=True)
try: dist.destroy_process_group()
except Exception: pass
