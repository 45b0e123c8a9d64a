"""Experiment: K=16 as one chain of 16-sample launches vs two concurrent chains of 8 samples on two streams (each with its own plan,
workspace and side stream).  Prints ms per 16-sample ELBO iteration for both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M
from mfvi_dip_mia_amd.engine import ElboEngine
import numpy as np

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NCH = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = 40

def mk(k):
    e = ElboEngine(H, H, K=k, temp=5.66e-7, sigma=1.46e-5)
    e.set_target(torch.rand(H, H, device="cuda"))
    return e

def timed(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

one = mk(K)
print("one chain  K=%d           : %.3f ms / iteration" % (K, timed(one.step, steps)))
del one
engs = [mk(K // NCH) for _ in range(NCH)]
streams = [torch.cuda.Stream() for _ in range(NCH)]
def both():
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.step()
print("%d chains x K=%d concurrent: %.3f ms / iteration" % (NCH, K // NCH, timed(both, steps)))
def serial():
    for e in engs: e.step()
print("%d chains x K=%d serial    : %.3f ms / iteration" % (NCH, K // NCH, timed(serial, steps)))
