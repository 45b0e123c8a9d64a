"""Micro-benchmark of ONE reparameterised conv layer through the C ABI: forward, backward-data, backward-weight
timed separately with the plan's HIP-event hooks.  usage: bench_layer.py cin cout k stride H W [K] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mfvi_dip_mia_amd as M

cin, cout, k, stride, H, W = [int(v) for v in sys.argv[1:7]]
K = int(sys.argv[7]) if len(sys.argv) > 7 else 16
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
with_bn = int(os.environ.get("BN", "1"))
P = M.Program()
zin = P.tensor(cin, H, W)
if with_bn:   # z -> conv0(1x1, cin->cin) -> BN/act -> [conv under test] -> BN/act -> conv2 (1x1 -> 2) so all fused paths are live
    x = P.tensor(cin, H, W); P.conv(zin, x, 1, 1); P.set_bn(x, act=True)
    y = P.tensor(cout, *P.conv_out_hw(x, k, stride)); P.conv(x, y, k, stride); P.set_bn(y, act=True)
    out = P.tensor(2, P.tensors[y]["H"], P.tensors[y]["W"]); P.conv(y, out, 1, 1)
    op = 1
else:
    out = P.tensor(cout, *P.conv_out_hw(zin, k, stride)); P.conv(zin, out, k, stride); op = 0
plan = P.compile(zin, out, K)
mu = 0.1 * torch.randn(P.n_vi, device="cuda"); rho = -3 + 0.1 * torch.randn(P.n_vi, device="cuda")
bn = torch.ones(max(P.n_bn, 1), device="cuda"); z = torch.randn(cin * H * W, device="cuda")
if int(os.environ.get("AUTOTUNE", "1")):
    plan.autotune(mu, rho, bn, z, K)
if os.environ.get("MFVI_TUNE") or os.environ.get("MFVI_TUNE_W"):      # forced tilings (launcher env) apply where the plan holds none
    for w in range(3):
        if (w < 2 and os.environ.get("MFVI_TUNE")) or (w == 2 and os.environ.get("MFVI_TUNE_W")):
            M._lib.check(M._lib.lib().mfvi_plan_set_tune(plan.handle, op, w, 0))
if int(os.environ.get("ALONE", "1")):
    plan.side_stream(False)      # every kernel alone on the stream
o = plan.forward(mu, rho, bn, z, 1, 0, 0, K)
dout = torch.randn_like(o); dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
dz = torch.empty(K * cin * H * W, device="cuda")
for _ in range(3):
    plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
torch.cuda.synchronize()
plan.profile(1)
for _ in range(reps):
    plan.forward(mu, rho, bn, z, 1, 0, 0, K, out=o); plan.backward(mu, rho, bn, z, 1, 0, 0, K, dout, dmu, drho, dbn, dz=dz)
torch.cuda.synchronize()
recs = plan.profile_read()
by = {}
for o_, p_, ms in recs:
    by.setdefault((o_, p_), []).append(ms)
Ho, Wo = P.tensors[P.ops[op]["out"]]["H"], P.tensors[P.ops[op]["out"]]["W"]
flops = 2.0 * K * cout * cin * k * k * Ho * Wo
names = {0: "fwd", 1: "bwd_weight", 2: "bwd_data", 3: "fold"}
tn = [M._lib.lib().mfvi_plan_get_tune(plan.handle, op, w) for w in range(3)]
print("tunes fwd mf=%d th=%d T=%d | bwd-data mf=%d th=%d T=%d | bww nb=%d w=%d tgt=%d" % (tn[0] & 255, (tn[0] >> 8) & 255, tn[0] >> 16, tn[1] & 255, (tn[1] >> 8) & 255, tn[1] >> 16, tn[2] & 255, (tn[2] >> 8) & 255, tn[2] >> 16))
for p_ in (0, 2, 1, 3):
    if (op, p_) in by:
        v = sorted(by[(op, p_)]); med = v[len(v) // 2]
        print("%dx%d %d->%d s%d @%dx%d K=%d  %-10s %8.1f us  %6.1f TFLOP/s (%.1f%% of 157.3)" % (
            k, k, cin, cout, stride, Ho, Wo, K, names[p_], med * 1e3, flops / (med * 1e-3) / 1e12, flops / (med * 1e-3) / 1e12 / 1.573))
