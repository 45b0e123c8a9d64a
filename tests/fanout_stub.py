"""Stub fit for the CPU test of the independent-fit fan-out (tests/test_host_logic.py): stands in for run_<task>_<method>."""
import math
import os
import time


def fit(temp, sigma, img="phantom", scale=1.0, **kw):
    time.sleep(0.05)
    if temp == 3.0:
        return float("nan")                      # a diverged candidate: must be dropped (bayesian_optimization.py:3777-3781)
    if temp == 5.0:
        raise RuntimeError("boom")               # a crashed fit must not take the other candidates of its device down
    if temp == 9.0:
        os._exit(3)                              # a worker that dies hard (GPU fault, SIGSEGV, OOM kill): no `finally`, no sentinel
    return {"psnr": scale * (10.0 * temp + sigma) + (100.0 if img == "b" else 0.0), "pid": os.getpid()}
