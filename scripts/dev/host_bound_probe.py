"""Is the K = 1 iteration bound by the host's enqueue rate or by the device?  Host time to ENQUEUE n iterations (no sync) against the
time until the device has finished them.  usage: host_bound_probe.py [K] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import mfvi_dip_mia_amd as M
from mfvi_dip_mia_amd.engine import ElboEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = ElboEngine(S, S, task="den", K=K, seed=3)
eng.set_target(torch.rand(S, S))
for _ in range(20):
    eng.step()
torch.cuda.synchronize()
for n in (20, 100):
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("K=%d %dx%d  n=%d: host enqueue %.3f ms / iteration, device done %.3f ms / iteration" % (K, S, S, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
