"""Can RCCL (torch.distributed backend "nccl") run a 2-rank group when both ranks sit on the SAME GPU?  (NCCL refuses
with "Duplicate GPU detected".)  Decides whether the N > 1 path can be exercised on RCCL on a one-GPU box."""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        t = torch.tensor([float(rank)], device="cuda:0")
        out = torch.empty(world, device="cuda:0")
        dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_gather over RCCL on one GPU -> {out.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(f"rank {rank}: RCCL on a shared GPU failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    try:
        mp.spawn(run, args=(2, 29911), nprocs=2, join=True)
    except Exception as e:
        print("probe: spawn ended with", type(e).__name__, str(e)[:200])
