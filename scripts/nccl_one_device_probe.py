"""Can the real RCCL branch of bench.py --gpus N (dist.init_process_group("nccl") + the state send / recv of sharding.run_handoff) be
exercised on a ONE-GPU box, with two processes sharing device 0?    Run as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/nccl_one_device_probe.py
Prints what happened; exits 0 either way (the answer is the output)."""
import os
import sys
import torch
import torch.distributed as dist

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    buf = torch.full((65536 + 64,), rank + 7, dtype=torch.uint8, device="cuda")
    if rank == 0:
        dist.send(buf, dst=1)
    else:
        dist.recv(buf, src=0)
    torch.cuda.synchronize()
    print("rank %d: nccl send/recv between two processes on ONE device worked, first byte %d" % (rank, int(buf[0].item())), flush=True)
    dist.barrier()
    dist.destroy_process_group()
except Exception as e:                                    # RCCL refuses two ranks on one device ("Duplicate GPU detected")
    print("rank %d: nccl with two processes on one device failed: %s: %s" % (rank, type(e).__name__, str(e).splitlines()[0][:300]), flush=True)
sys.exit(0)
