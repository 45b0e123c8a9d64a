"""Phase attribution of the bf16x6 backward-data kernel (library built with -DX6B_PROF: scripts/dev/build_variant.sh conv_bwd_x6 out.so -DX6B_PROF).
usage: MFVI_LIB_PATH=out.so bwdx6_prof.py [cin cout hw T]...   cycles are s_memtime ticks (100 MHz x ... = shader cycles) of wave 0 of each kind, averaged per block"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M

K = int(os.environ.get("K", "16")); reps = int(os.environ.get("REPS", "5"))
a = [int(v) for v in sys.argv[1:]] or [36, 16, 256, 8, 68, 32, 128, 4, 132, 64, 64, 2]
lib = M._lib.lib()
lib.mfvi_debug_x6b_prof.restype = C.c_int; lib.mfvi_debug_x6b_prof.argtypes = [C.c_void_p, C.c_int]
names = {0: "m.prologue", 1: "m.weight reads", 2: "m.wait B2", 3: "m.rows", 4: "m.fold", 5: "m.wait B1", 6: "m.wait B3", 9: "m.total",
         12: "s.prologue", 13: "s.wait B2", 14: "s.weight copy", 15: "s.stage phase", 16: "s.wait B1", 17: "s.write+B3", 18: "s.fold", 19: "s.pro issue", 20: "s.pro tables", 21: "s.pro loads", 22: "s.fold lds", 23: "s.fold wait x", 10: "s.fold compute+st", 11: "s.fold reduce"}
for cin, cout, hw, T in [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]:
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
    sr = {16: 8, 32: 4, 64: 2}[cout]
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, T | sr << 8 | 1 << 25))
    for _ in range(2):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 24)()
    lib.mfvi_debug_x6b_prof(buf, 1)
    plan.profile(1)
    for _ in range(reps):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    ms = sorted(m for o_, p_, m in plan.profile_read() if o_ == op and p_ == 2)
    plan.profile(0)
    lib.mfvi_debug_x6b_prof(buf, 1)
    v = list(buf)
    nb = max(v[8], 1); npass = max(v[7], 1)
    print("%d->%d @%d T=%d: %.1f us (instrumented); blocks/launch %d, passes/block %.1f" % (cin, cout, hw, T, ms[len(ms) // 2] * 1e3, nb // reps, npass / nb))
    for i in sorted(names):
        print("    %-16s %9.0f ticks/block  (%.0f per pass)" % (names[i], v[i] / nb, v[i] / npass))
