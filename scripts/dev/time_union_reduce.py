"""2-rank gloo, CUDA tensors on one GPU: what the union-of-visibility all-reduce of average_gradients_sparse costs by dtype."""
import os, sys, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = 200_000
for dt in (torch.uint8, torch.int32, torch.float32):
    t = (torch.arange(n, device=dev) % 3 == dist.get_rank()).to(dt)
    for _ in range(3):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    if dist.get_rank() == 0:
        print(dt, f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per all_reduce(MAX) of {n} elements")
big = torch.zeros(62_000_000 // 4, device=dev)   # a 62 MB gradient bucket
dist.all_reduce(big); torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(5):
    dist.all_reduce(big)
torch.cuda.synchronize()
if dist.get_rank() == 0:
    print(f"62 MB float all_reduce(SUM): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
dist.destroy_process_group()
