"""Independent fits over the GPUs of one node: the reference's candidate fan-out (bayesian_optimization.py:3709-3724 `f`, :3760-3781;
eval_result.py:19-58) — `zip(candidates, itertools.cycle(device_list))`, one process per candidate, an mp.Queue of (candidate, psnr),
NaN results dropped — re-cut for one process per GPU:

  * one worker process per DEVICE (not per candidate), started with the 'spawn' method so that it is a fresh interpreter that selects
    its GPU before anything touches HIP; the candidates dealt to a device by the same round-robin run back to back in that worker
    (a fit fills an MI355X by itself; concurrent fits on one GPU would only interleave), results are identical either way;
  * no collective and no shared state: the fits are independent; the only exchange is the final gather of
    (job index, candidate, psnr) through a queue, after which NaNs are filtered exactly like the reference does.

Jobs are (image, candidate) pairs, so "8 independent images, one per GPU" (BASELINE configs[3] / [4]) and "N hyper-parameter candidates
over the node's GPUs" (the reference's use) are the same call."""
import importlib
import itertools
import math
import os


def assign(n_jobs, devices):
    """Job i -> devices[i % len(devices)], as zip(jobs, itertools.cycle(device_list)) does (bayesian_optimization.py:3762)."""
    if not devices:
        raise ValueError("no devices")
    per = {d: [] for d in devices}
    for i, d in zip(range(n_jobs), itertools.cycle(devices)):
        per[d].append(i)
    return per


def _resolve(fn):
    if callable(fn):
        return fn
    mod, _, name = fn.partition(":")
    return getattr(importlib.import_module(mod), name)


def _worker(device, jobs, fn, run_params, queue):
    """Body of one per-device process.  jobs: [(index, job kwargs)]; fn: 'module:function' (or a picklable callable) returning either
    the PSNR or a dict with key 'psnr' (the runners of mfvi_dip_mia_amd.runner)."""
    dev = str(device)
    try:
        try:
            if dev.startswith("cuda"):
                import torch                               # fresh interpreter: the first GPU call of this process picks its device
                torch.cuda.set_device(int(dev.split(":")[1]) if ":" in dev else 0)
            f = _resolve(fn)
        except Exception as e:                             # a device that is not on the box, an import error: every job of this worker
            for idx, _ in jobs:                            # is reported with the reason instead of a bare 'worker died'
                queue.put((idx, dev, os.getpid(), float("nan"), "worker setup failed: %s: %s" % (type(e).__name__, e)))
            return
        for idx, kw in jobs:
            try:
                res = f(**kw, **run_params)
                psnr = float(res["psnr"] if isinstance(res, dict) else res)
                queue.put((idx, dev, os.getpid(), psnr, None))
            except Exception as e:                         # one failed fit must not take the other candidates of this device down
                queue.put((idx, dev, os.getpid(), float("nan"), "%s: %s" % (type(e).__name__, e)))
    finally:
        queue.put(("done", dev))                           # this worker is done (never sent by a process that dies hard)


def run_jobs(jobs, devices, fn, run_params=None, start_method="spawn", poll_seconds=0.5):
    """Run fn(**job, **run_params) for every job, job i on devices[i % len(devices)], one fresh process per device.
    Returns (results, dropped): results = [(job index, job, psnr)] with NaN results removed (bayesian_optimization.py:3777-3781),
    sorted by job index; dropped = [(job index, job, error message or 'nan')].

    A worker that dies without running its `finally` (GPU fault, abort, SIGSEGV, the OOM killer) sends no sentinel: the gather polls the
    queue with a timeout and checks the processes, so such a worker counts as finished and the jobs it had not reported are dropped with
    its exit code — the reference joins its children the same way (bayesian_optimization.py:3764-3775)."""
    import multiprocessing as mp
    import queue as queue_mod
    run_params = dict(run_params or {})
    devices = [str(d) for d in devices]
    per = assign(len(jobs), devices)
    ctx = mp.get_context(start_method)
    queue = ctx.Queue()
    procs = {}
    for d in devices:
        if not per[d]:
            continue
        p = ctx.Process(target=_worker, args=(d, [(i, jobs[i]) for i in per[d]], fn, run_params, queue))
        p.start()
        procs[d] = p
    got, finished, died = {}, set(), {}
    while len(finished) < len(procs):
        try:
            item = queue.get(timeout=poll_seconds)
        except queue_mod.Empty:
            for d, p in procs.items():
                if d not in finished and not p.is_alive():
                    # drain what the dead worker managed to send before deciding which of its jobs are lost
                    try:
                        while True:
                            late = queue.get(timeout=0.05)
                            if late[0] == "done":
                                finished.add(late[1])
                            else:
                                got[late[0]] = late
                    except queue_mod.Empty:
                        pass
                    if d not in finished:
                        finished.add(d); died[d] = p.exitcode
            continue
        if item[0] == "done":
            finished.add(item[1])
            continue
        got[item[0]] = item
    for p in procs.values():
        p.join()
    results, dropped = [], []
    for i in range(len(jobs)):
        if i not in got:
            d = devices[i % len(devices)]
            dropped.append((i, jobs[i], "worker on %s died (exit code %s)" % (d, died.get(d, procs[d].exitcode)))); continue
        _, dev, pid, psnr, err = got[i]
        if err is not None or math.isnan(psnr):
            dropped.append((i, jobs[i], err or "nan"))
        else:
            results.append((i, jobs[i], psnr))
    run_jobs.last_placement = {i: (got[i][1], got[i][2]) for i in got}     # job -> (device, pid), for inspection / tests
    return results, dropped


def print_table(results, keys):
    """The table eval_result.py:55-58 prints."""
    print()
    print("      ".join(keys) + "       psnr")
    for _, job, y in results:
        print("  ".join("%.6f" % job[k] for k in keys) + "  %.6f" % y)
