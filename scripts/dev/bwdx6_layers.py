"""Backward-data (with the fold) of one 3x3 layer at a time, alone on the stream: the autotuned fp32 kernels (row-phase / round-2) against
the bf16x6 kernel (csrc/conv_bwd_x6.hip, tune bit 25) over strips per block.  usage: bwdx6_layers.py [cin cout hw]...
(default: the three dominant layers of cfg2 + their 512^2 twins).  K = samples, REPS = iterations measured (median)."""
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
    x = P.tensor(cin, hw, hw); P.conv(zin, x, 1, 1); P.set_bn(x, act=os.environ.get('BWDX6_ACT', '0') == '1')      # the net's big 3x3 layers read concat tensors: BatchNorm, no activation
    y = P.tensor(cout, hw, hw); P.conv(x, y, 3, 1); P.set_bn(y, act=True)
    out = P.tensor(2, hw, hw); P.conv(y, out, 1, 1)
    op = 1
    plan = P.compile(zin, out, K)
    mu = 0.1 * torch.randn(P.n_vi, device="cuda"); rho = -3 + 0.1 * torch.randn(P.n_vi, device="cuda")
    bn = torch.ones(max(P.n_bn, 1), device="cuda"); z = torch.randn(cin * hw * hw, device="cuda")
    only = os.environ.get("BWDX6_ONLY")
    if not only:
        plan.autotune(mu, rho, bn, z, K)
    plan.side_stream(False)
    o = plan.forward(mu, rho, bn, z, 1, 0, 0, K)
    dout = torch.randn_like(o); dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
    dz = torch.empty(K * cin * hw * hw, device="cuda")
    bufs = (mu, rho, bn, z, o, dout, dmu, drho, dbn, dz)
    flops = 2.0 * K * cout * cin * 9 * hw * hw
    byts = 4.0 * K * hw * hw * (2 * cout + 2 * cin)        # ga + y of the output read, raw x read, ga of the input written
    tf = lambda us: flops / (us * 1e-6) / 1e12
    base = lib.mfvi_plan_get_tune(plan.handle, op, 1)
    us = measure(plan, op, bufs, 2)
    print("%d->%d @%d bwd_data autotuned %#x (family %d): %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak) %5.0f GB/s incl. fold" % (
        cin, cout, hw, base, lib.mfvi_plan_last_kernel(plan.handle, op, 1), us, tf(us), tf(us) / 157.3, byts / us / 1e3), flush=True)
    # best fp32 tiling for comparison when the autotuner already chose bf16x6
    sr = {16: 8, 32: 4, 64: 2}.get(cout)
    if sr:
        for T in ((max(1, (hw // 64) * (hw // sr) * K // 256),) if only else (1, 2, 4, 8, 16, 32)):
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, T | sr << 8 | 1 << 25))
            try:
                us = measure(plan, op, bufs, 2)
            except M._lib.MfviError as e:
                print("bwd_data bf16x6 T=%d: not served (%s)" % (T, e)); continue
            if lib.mfvi_plan_last_kernel(plan.handle, op, 1) != 3:
                print("%d->%d @%d bwd_data bf16x6 T=%d: fell back to family %d" % (cin, cout, hw, T, lib.mfvi_plan_last_kernel(plan.handle, op, 1)), flush=True); continue
            print("%d->%d @%d bwd_data bf16x6 sr=%d T=%2d: %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak, %.3f of 2.5 PF / 6) %5.0f GB/s incl. fold" % (
                cin, cout, hw, sr, T, us, tf(us), tf(us) / 157.3, tf(us) / (2500.0 / 6), byts / us / 1e3), flush=True)
    if sr == 8 and cin in (32, 36):      # strip-resident form (conv_bwd_x6s.hip)
        for T in (2, 4, 8, 16, 32):
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, T | sr << 8 | 1 << 16 | 1 << 25))
            us = measure(plan, op, bufs, 2)
            if lib.mfvi_plan_last_kernel(plan.handle, op, 1) != 3:
                print("strip-resident T=%d: fell back" % T); continue
            print("%d->%d @%d bwd_data bf16x6 strip-resident T=%2d: %7.1f us %5.1f TF (%.3f of the fp32 MFMA peak, %.3f of 2.5 PF / 6) %5.0f GB/s incl. fold" % (
                cin, cout, hw, T, us, tf(us), tf(us) / 157.3, tf(us) / (2500.0 / 6), byts / us / 1e3), flush=True)
    for mf, r, T, rem in (() if only else ((1, 1, 8, 1), (2, 1, 4, 1), (2, 1, 2, 1), (1, 1, 4, 0), (2, 1, 4, 0))):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, mf | r << 8 | rem << 12 | T << 16 | 1 << 24))
        try:
            us = measure(plan, op, bufs, 2)
        except M._lib.MfviError:
            continue
        if lib.mfvi_plan_last_kernel(plan.handle, op, 1) == 2:
            print("%d->%d @%d bwd_data row-phase fp32 mf=%d r=%d T=%d rem=%d: %7.1f us %5.1f TF (%.3f)" % (cin, cout, hw, mf, r, T, rem, us, tf(us), tf(us) / 157.3), flush=True)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, op, 1, base))
