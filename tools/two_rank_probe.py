#!/usr/bin/env python3
"""Can two ranks share ONE GPU for a collective on this box?  (decides how the a15 GPU test is built)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/two_rank_probe.py"""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
for backend in sys.argv[1:] or ["nccl"]:
    try:
        dist.init_process_group(backend, device_id=torch.device("cuda", 0) if backend == "nccl" else None)
        t = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
        if backend == "gloo":
            h = t.cpu(); dist.all_reduce(h); t.copy_(h)
        else:
            dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"[rank {rank}] backend {backend}: all_reduce ok, value {float(t[0])}", flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(f"[rank {rank}] backend {backend}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
