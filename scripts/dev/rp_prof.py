"""Phase attribution of the row-phase kernels (library built with -DRP_PROF: scripts/dev/build_rp_variant.sh out.so -DRP_PROF).
usage: MFVI_LIB_PATH=.../_lib_prof.so rp_prof.py cin cout hw pass mf r T rem [...more 'pass mf r T rem' groups]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M

RP = 1 << 24
K = int(os.environ.get("K", "16")); reps = int(os.environ.get("REPS", "5"))
cin, cout, hw = [int(v) for v in sys.argv[1:4]]
rest = [int(v) for v in sys.argv[4:]]
groups = [tuple(rest[i:i + 5]) for i in range(0, len(rest), 5)]
lib = M._lib.lib()
lib.mfvi_debug_rp_prof.restype = C.c_int; lib.mfvi_debug_rp_prof.argtypes = [C.c_void_p, C.c_int]
P = M.Program()
zin = P.tensor(cin, hw, hw)
x = P.tensor(cin, hw, hw); P.conv(zin, x, 1, 1); P.set_bn(x, act=True)
y = P.tensor(cout, hw, hw); P.conv(x, y, 3, 1); P.set_bn(y, act=True)
out = P.tensor(2, hw, hw); P.conv(y, out, 1, 1)
op = 1
plan = P.compile(zin, out, K)
mu = 0.1 * torch.randn(P.n_vi, device="cuda"); rho = -3 + 0.1 * torch.randn(P.n_vi, device="cuda")
bn = torch.ones(max(P.n_bn, 1), device="cuda"); z = torch.randn(cin * hw * hw, device="cuda")
plan.side_stream(False)
o = plan.forward(mu, rho, bn, z, 1, 0, 0, K)
dout = torch.randn_like(o); dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
dz = torch.empty(K * cin * hw * hw, device="cuda")
names = ["c.prologue", "c.mfma", "c.epilogue", "c.barrier", "c.loop", "blocks", "iters", "-", "p.prologue", "p.wait_loads", "p.store", "p.fetch", "p.barrier", "p.loop"]
for w, mf, r, T, rem in groups:
    # the other pass runs the round-2 kernel, so only the pass under test writes the counters
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1 - w, 1 | 8 << 8 | 1 << 16))
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, w, mf | r << 8 | rem << 12 | T << 16 | RP))
    for _ in range(2):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.mfvi_debug_rp_prof(buf, 1)
    plan.profile(1)
    for _ in range(reps):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    ms = sorted(m for o_, p_, m in plan.profile_read() if o_ == op and p_ == (0 if w == 0 else 2))
    plan.profile(0)
    lib.mfvi_debug_rp_prof(buf, 1)
    v = list(buf)
    nb = max(v[5], 1)
    print("%d->%d @%d pass %d mf=%d r=%d T=%d rem=%d: %.1f us (instrumented); blocks/launch %d, stages/block %.1f" % (cin, cout, hw, w, mf, r, T, rem, ms[len(ms) // 2] * 1e3, nb // reps, v[6] / nb))
    for i in (0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13):
        print("    %-14s %9.0f cycles/block  (%.0f per stage)" % (names[i], v[i] / nb, v[i] / max(v[6], 1)))
