"""Can gloo reduce DEVICE tensors between two processes that share one GPU?  (It can: sum, min, float64 - what tests/test_gpu_two_ranks.py builds on.)
    python tools/gloo_cuda_probe.py        # on an MI355X box"""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = torch.full((5,), float(rank + 1), device="cuda")
    dist.all_reduce(t); print(rank, "sum", t.tolist(), flush=True)
    m = torch.tensor([3.0 - rank, 1.0 + rank], device="cuda"); dist.all_reduce(m, op=dist.ReduceOp.MIN); print(rank, "min", m.tolist(), flush=True)
    d = torch.ones(4, dtype=torch.float64, device="cuda") * (rank + 1); dist.all_reduce(d); print(rank, "f64", d.tolist(), flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2, 29871), nprocs=2, join=True)
