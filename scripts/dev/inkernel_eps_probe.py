"""Which layers of a config the autotuner puts on the in-kernel-eps (generic) kernels, and the per-op times with / without the candidate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from mfvi_dip_mia_amd import _lib as L
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
eng = bench.make_engine(cfg, cfg["k"], 0, 1, torch)
l = L.lib()
for i, o in enumerate(eng.prog.ops):
    if o["type"] != 1:
        continue
    t = [l.mfvi_plan_get_tune(eng.plan.handle, i, w) for w in range(3)]
    if any(x & (1 << 27) for x in t):
        print("op", i, "%dx%d %d->%d @%d" % (o["ksize"], o["ksize"], eng.prog.tensors[o["in0"]]["C"], eng.prog.tensors[o["out"]]["C"], eng.prog.tensors[o["out"]]["H"]), ["%#x" % x for x in t])


def per_op(label):
    eng.plan.side_stream(False); eng.plan.profile(1)
    for _ in range(5):
        eng.step()
    torch.cuda.synchronize()
    by = {}
    for op, ps, ms in eng.plan.profile_read():
        by.setdefault((op, ps), []).append(ms)
    eng.plan.profile(0); eng.plan.side_stream(True)
    return {k: sorted(v)[len(v) // 2] * 1e3 for k, v in by.items()}


tiny = [i for i, o in enumerate(eng.prog.ops) if o["type"] == 1 and eng.prog.tensors[o["out"]]["C"] * eng.prog.tensors[o["in0"]]["C"] * o["ksize"] ** 2 <= 2560]
base = per_op("tuned")
saved = {i: [l.mfvi_plan_get_tune(eng.plan.handle, i, w) for w in range(3)] for i in tiny}
for i in tiny:
    for w in (0, 2):
        L.check(l.mfvi_plan_set_tune(eng.plan.handle, i, w, 1 << 27))
gen = per_op("generic")
print("op: layer | forward us (matrix-core kernel reading the slab -> in-kernel eps) | backward-weight us")
for i in tiny:
    o = eng.prog.ops[i]
    print("op %2d %dx%d %3d->%-3d @%-3d s%d | fwd %6.1f -> %6.1f | bwd_weight %6.1f -> %6.1f" % (i, o["ksize"], o["ksize"], eng.prog.tensors[o["in0"]]["C"], eng.prog.tensors[o["out"]]["C"],
          eng.prog.tensors[o["out"]]["H"], o["stride"], base.get((i, 0), 0), gen.get((i, 0), 0), base.get((i, 1), 0), gen.get((i, 1), 0)))
