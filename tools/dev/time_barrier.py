"""Dev timing (GPU box, under torch.distributed.run): cost of the bench's fence (barrier + synchronize) variants."""
import os, time
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
flag = torch.zeros(1, device="cuda")
def t(f, reps=200):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
def barrier_sync():
    dist.barrier(); torch.cuda.synchronize()
def allreduce_sync():
    dist.all_reduce(flag); torch.cuda.synchronize()
if rank == 0:
    print("dist.barrier() + synchronize: %.1f us" % t(barrier_sync))
    print("all_reduce(1 float) + synchronize: %.1f us" % t(allreduce_sync))
    print("synchronize alone: %.1f us" % t(torch.cuda.synchronize))
dist.destroy_process_group()
