"""Phase attribution of the strip-resident bf16x6 backward-data kernel (library built with -DX6S_PROF: scripts/dev/build_variant.sh conv_bwd_x6s out.so -DX6S_PROF).
usage: MFVI_LIB_PATH=out.so bwdx6s_prof.py [cin hw T]...   s_memtime ticks of matrix wave 0 / staging wave 0, averaged per block and per strip"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M

K = int(os.environ.get("K", "16")); reps = int(os.environ.get("REPS", "5"))
a = [int(v) for v in sys.argv[1:]] or [36, 256, 8]
lib = M._lib.lib()
lib.mfvi_debug_x6s_prof.restype = C.c_int; lib.mfvi_debug_x6s_prof.argtypes = [C.c_void_p, C.c_int]
names = {0: "m.prologue", 1: "m.main sweep", 2: "m.4-channel sweep", 3: "m.wait D", 4: "m.dump", 5: "m.wait E", 7: "m.total",
         10: "s.prologue", 11: "s.request A", 12: "s.fold tile 0", 13: "s.consume A + request B", 14: "s.fold tile 1", 15: "s.consume B", 16: "s.fold 4-ch tile",
         17: "s.request x", 18: "s.wait D", 19: "s.write window", 20: "s.wait E", 21: "s.last folds"}
for cin, hw, T in [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]:
    P = M.Program()
    zin = P.tensor(cin, hw, hw)
    x = P.tensor(cin, hw, hw); P.conv(zin, x, 1, 1); P.set_bn(x, act=os.environ.get("BWDX6_ACT", "0") == "1")
    y = P.tensor(16, hw, hw); P.conv(x, y, 3, 1); P.set_bn(y, act=True)
    out = P.tensor(2, hw, hw); P.conv(y, out, 1, 1)
    op = 1
    plan = P.compile(zin, out, K)
    mu = 0.1 * torch.randn(P.n_vi, device="cuda"); rho = -3 + 0.1 * torch.randn(P.n_vi, device="cuda")
    bn = torch.ones(max(P.n_bn, 1), device="cuda"); z = torch.randn(cin * hw * hw, device="cuda")
    plan.side_stream(False)
    o = plan.forward(mu, rho, bn, z, 1, 0, 0, K)
    dout = torch.randn_like(o); dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
    dz = torch.empty(K * cin * hw * hw, device="cuda")
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, T | 8 << 8 | 1 << 16 | 1 << 25))
    for _ in range(2):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 24)()
    lib.mfvi_debug_x6s_prof(buf, 1)
    plan.profile(1)
    for _ in range(reps):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    ms = sorted(m for o_, p_, m in plan.profile_read() if o_ == op and p_ == 2)
    plan.profile(0)
    lib.mfvi_debug_x6s_prof(buf, 1)
    v = list(buf)
    nb = max(v[8], 1); ns = max(v[6], 1)
    print("%d->16 @%d T=%d: %.1f us (instrumented); blocks/launch %d, strips/block %.1f" % (cin, hw, T, ms[len(ms) // 2] * 1e3, nb // reps, ns / nb))
    for i in sorted(names):
        print("    %-26s %9.0f ticks/block  (%.0f per strip)" % (names[i], v[i] / nb, v[i] / ns))
