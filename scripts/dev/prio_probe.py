"""Does a high-priority caller stream change the two-stream iteration?  (the plan's side stream is lowest priority already)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mfvi_dip_mia_amd.engine import ElboEngine
H = 256; K = 16
def run(stream):
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        eng = ElboEngine(H, H, K=K, temp=5.66e-7, sigma=1.46e-5)
        eng.set_target(torch.rand(H, H, device="cuda"))
        for _ in range(5): eng.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): eng.step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 100 * 1e3
lo, hi = -1, 0
try:
    print("priority range", torch.cuda.Stream.priority_range())
except Exception as e:
    print("no priority_range:", e)
for rep in range(int(os.environ.get("REPS", "2"))):
    print("default stream           : %.4f ms" % run(None), flush=True)
    print("own stream, priority  0  : %.4f ms" % run(torch.cuda.Stream(priority=0)), flush=True)
    print("own stream, priority -1  : %.4f ms" % run(torch.cuda.Stream(priority=-1)), flush=True)
