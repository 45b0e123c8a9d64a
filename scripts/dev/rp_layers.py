"""Row-phase kernels (conv_rp.hip) against the autotuned round-2 kernels, one 3x3 layer at a time, every kernel alone on the stream.
usage: rp_layers.py [cin cout hw]...   (default: the three dominant layers of cfg2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M

RP = 1 << 24
K = int(os.environ.get("K", "16")); reps = int(os.environ.get("REPS", "10"))
specs = [(36, 16, 256), (68, 32, 128), (132, 64, 64)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]; specs = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
lib = M._lib.lib()


def measure(plan, P, op, bufs, which_pass):
    mu, rho, bn, z, o, dout, dmu, drho, dbn, dz = bufs
    for _ in range(2):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    plan.profile(1)
    for _ in range(reps):
        plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
    torch.cuda.synchronize()
    by = {}
    for o_, p_, ms in plan.profile_read():
        by.setdefault((o_, p_), []).append(ms)
    plan.profile(0)
    v = sorted(by[(op, which_pass)]); return v[len(v) // 2] * 1e3


for cin, cout, hw in specs:
    P = M.Program()
    zin = P.tensor(cin, hw, hw)
    x = P.tensor(cin, hw, hw); P.conv(zin, x, 1, 1); P.set_bn(x, act=True)
    y = P.tensor(cout, hw, hw); P.conv(x, y, 3, 1); P.set_bn(y, act=True)
    out = P.tensor(2, hw, hw); P.conv(y, out, 1, 1)
    op = 1
    plan = P.compile(zin, out, K)
    mu = 0.1 * torch.randn(P.n_vi, device="cuda"); rho = -3 + 0.1 * torch.randn(P.n_vi, device="cuda")
    bn = torch.ones(max(P.n_bn, 1), device="cuda"); z = torch.randn(cin * hw * hw, device="cuda")
    plan.autotune(mu, rho, bn, z, K)
    plan.side_stream(False)
    o = plan.forward(mu, rho, bn, z, 1, 0, 0, K)
    dout = torch.randn_like(o); dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
    dz = torch.empty(K * cin * hw * hw, device="cuda")
    bufs = (mu, rho, bn, z, o, dout, dmu, drho, dbn, dz)
    flops = 2.0 * K * cout * cin * 9 * hw * hw
    tf = lambda us: flops / (us * 1e-6) / 1e12
    base = [lib.mfvi_plan_get_tune(plan.handle, op, w) for w in range(3)]
    for w, name, ps in ((0, "fwd", 0), (1, "bwd_data", 2), (2, "bwd_weight", 1)):
        us = measure(plan, P, op, bufs, ps)
        print("%d->%d @%d %-10s autotuned %#x: %7.1f us %5.1f TF (%.3f)" % (cin, cout, hw, name, base[w], us, tf(us), tf(us) / 157.3), flush=True)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 0, 1 | 1 << 26))      # small-map forward (conv_small.hip)
    try:
        us = measure(plan, P, op, bufs, 0)
        if lib.mfvi_plan_last_kernel(plan.handle, op, 0) == 4:
            print("%d->%d @%d %-10s small-map kernel: %7.1f us %5.1f TF (%.3f)" % (cin, cout, hw, "fwd", us, tf(us), tf(us) / 157.3), flush=True)
    except M._lib.MfviError:
        pass
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 0, base[0]))
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, 1 | 1 << 26))      # small-map backward-data
    try:
        us = measure(plan, P, op, bufs, 2)
        if lib.mfvi_plan_last_kernel(plan.handle, op, 1) == 4:
            print("%d->%d @%d %-10s small-map kernel: %7.1f us %5.1f TF (%.3f)" % (cin, cout, hw, "bwd_data", us, tf(us), tf(us) / 157.3), flush=True)
    except M._lib.MfviError:
        pass
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, base[1]))
    rem_ok = (cin & 15) == 4
    only = os.environ.get("RP_ONLY")      # e.g. "0:1,4,1,0;1:2,1,2,1" = pass:mf,r,T,rem
    sel = None
    if only:
        sel = set()
        for item in only.split(";"):
            w_, rest = item.split(":"); vals = tuple(int(v) for v in rest.split(","))
            sel.add((int(w_),) + (vals if len(vals) == 5 else vals + (1,)))
    for w, name, ps in ((0, "fwd", 0), (1, "bwd_data", 2)):
        for mf, r in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (4, 1)):
            for rem in ((0, 1) if (w == 1 and rem_ok) else (0,)):
                for T, ks in [(T_, k_) for T_ in (1, 2, 4, 8, 16) for k_ in (1, 2, 4)]:
                    if sel is not None and (w, mf, r, T, rem, ks) not in sel:
                        continue
                    if sel is None and (T > 4 or (ks > 1 and not (hw == 16 and T == 1 and r == 1))):
                        continue
                    code = mf | r << 8 | rem << 12 | ks << 13 | T << 16 | RP
                    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, w, code))
                    try:
                        us = measure(plan, P, op, bufs, ps)
                    except M._lib.MfviError:
                        continue
                    finally:
                        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, w, base[w]))
                    print("%d->%d @%d %-10s rp mf=%d r=%d rem=%d T=%d ks=%d: %7.1f us %5.1f TF (%.3f)" % (cin, cout, hw, name, mf, r, rem, T, ks, us, tf(us), tf(us) / 157.3), flush=True)
