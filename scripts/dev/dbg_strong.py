import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = dict(bench.CONFIGS["cfg2"])
e1 = bench.make_engine(cfg, 32, rank, world, torch)
for _ in range(3): e1.step()
torch.cuda.synchronize()
e2 = bench.make_engine(cfg, 16, rank, world, torch)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); e2.step(); torch.cuda.synchronize()
    if rank == 0: print("e2 step", i, round(1e3 * (time.perf_counter() - t0), 2), "ms")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); e1.step(); torch.cuda.synchronize()
    if rank == 0: print("e1 step", i, round(1e3 * (time.perf_counter() - t0), 2), "ms")
if rank == 0: print(e2.plan.tunes())
dist.destroy_process_group()
