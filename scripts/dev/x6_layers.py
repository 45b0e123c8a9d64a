"""Backward-weight of one 3x3 layer at a time, alone on the stream: the autotuned fp32-MFMA kernels against the bf16x6 kernel
(csrc/conv_bww_x6.hip, tune w = 11) over its tilings (cof = output fragments per block, tb = block-count target / 256).
usage: x6_layers.py [cin cout hw]...   (default: the three dominant layers of cfg2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M

K = int(os.environ.get("K", "16")); reps = int(os.environ.get("REPS", "10"))
specs = [(36, 16, 256), (68, 32, 128), (132, 64, 64)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]; specs = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
lib = M._lib.lib()


def measure(plan, op, bufs, which_pass):
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
    if os.environ.get("X6_AUTOTUNE", "1") != "0":
        plan.autotune(mu, rho, bn, z, K)
    plan.side_stream(False)
    o = plan.forward(mu, rho, bn, z, 1, 0, 0, K)
    dout = torch.randn_like(o); dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
    dz = torch.empty(K * cin * hw * hw, device="cuda")
    bufs = (mu, rho, bn, z, o, dout, dmu, drho, dbn, dz)
    flops = 2.0 * K * cout * cin * 9 * hw * hw
    tf = lambda us: flops / (us * 1e-6) / 1e12
    base = lib.mfvi_plan_get_tune(plan.handle, op, 2)
    us = measure(plan, op, bufs, 1)
    print("%d->%d @%d bwd_weight autotuned %#x: %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak)" % (cin, cout, hw, base, us, tf(us), tf(us) / 157.3), flush=True)
    for code, name in ((2 | 10 << 8 | 1 << 16, "fp32 fragment-split"), (2 | 9 << 8 | 1 << 16, "fp32 specialised nb=2")):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 2, code))
        try:
            us = measure(plan, op, bufs, 1)
            print("%d->%d @%d bwd_weight %-22s: %7.1f us %5.1f TF (%.3f)" % (cin, cout, hw, name, us, tf(us), tf(us) / 157.3), flush=True)
        except M._lib.MfviError:
            pass
    for cof in (1, 2):
        for tb in (1, 2, 4):
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 2, cof | 11 << 8 | tb << 16))
            try:
                us = measure(plan, op, bufs, 1)
            except M._lib.MfviError:
                continue
            if lib.mfvi_plan_last_kernel(plan.handle, op, 2) != 3:
                print("%d->%d @%d bwd_weight bf16x6 cof=%d tb=%d: not served" % (cin, cout, hw, cof, tb), flush=True); continue
            print("%d->%d @%d bwd_weight bf16x6 cof=%d tb=%d: %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak, %.3f of 2.5 PF / 6)" % (cin, cout, hw, cof, tb, us, tf(us), tf(us) / 157.3, tf(us) / (2500.0 / 6)), flush=True)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 2, base))
    # forward: autotuned kernel, the row-phase fp32 kernel, the bf16x6 kernel (csrc/conv_x6.hip, tune bit 25)
    basef = lib.mfvi_plan_get_tune(plan.handle, op, 0)
    us = measure(plan, op, bufs, 0)
    print("%d->%d @%d fwd        autotuned %#x: %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak)" % (cin, cout, hw, basef, us, tf(us), tf(us) / 157.3), flush=True)
    for mf, T, mrg in [(m_, t_, 0) for m_ in (1, 2) for t_ in (1, 2, 4, 8)] + ([(1, t_, 1) for t_ in (1, 2, 4, 8, 16)] if cin % 32 == 4 else []):      # mrg: remainder plane on the last group's pass (bit 12)
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 0, mf | 8 << 8 | mrg << 12 | T << 16 | 1 << 25))
        try:
            us = measure(plan, op, bufs, 0)
        except M._lib.MfviError as e:
            print("fwd bf16x6 mf=%d: not served (%s)" % (mf, e)); continue
        if lib.mfvi_plan_last_kernel(plan.handle, op, 0) != 3:
            print("%d->%d @%d fwd        bf16x6 mf=%d T=%d: not served (fell back to kernel family %d)" % (cin, cout, hw, mf, T, lib.mfvi_plan_last_kernel(plan.handle, op, 0)), flush=True); continue
        print("%d->%d @%d fwd        bf16x6 mf=%d%s sr=8 T=%d: %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak, %.3f of 2.5 PF / 6)" % (cin, cout, hw, mf, " merged" if mrg else "", T, us, tf(us), tf(us) / 157.3, tf(us) / (2500.0 / 6)), flush=True)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 0, basef))
