"""Which op's row-phase backward-data differs from the round-2 kernel inside a small net?  (debug helper)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import mfvi_dip_mia_amd as M
kw = dict(H=64, W=64, input_depth=8, n_out=2, nd=(16, 16, 32), nu=(16, 16, 32), ns=(4, 4, 4))
P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
n = 2
plan = P.compile(zin, out_id, max_samples=n)
g = torch.Generator(device="cpu").manual_seed(1)
mu = (0.1 * torch.randn(P.n_vi, generator=g)).cuda(); rho = (-3 + 0.1 * torch.randn(P.n_vi, generator=g)).cuda()
bn = torch.ones(P.n_bn).cuda(); z = torch.randn(8 * 64 * 64, generator=g).cuda()
lib = M._lib.lib()
def run():
    o = plan.forward(mu, rho, bn, z, 1, 0, 0, n)
    dout = torch.ones_like(o) * 0.01 + 0.001 * torch.arange(o.numel(), device="cuda").reshape(o.shape) / o.numel()
    dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn); dz = torch.empty(n * 8 * 64 * 64, device="cuda")
    plan.backward(mu, rho, bn, z, 1, 0, 0, n, dout, dmu, drho, dbn, dz=dz)
    return dz.cpu().numpy(), dmu.cpu().numpy()
os.environ["X"] = "1"
base = run()
ops = [i for i, o in enumerate(P.ops) if o["type"] == M._lib.OP_CONV and o["ksize"] == 3 and o["stride"] == 1]
for i in ops:
    fam = [lib.mfvi_plan_last_kernel(plan.handle, i, w) for w in range(3)]
    o = P.ops[i]; t_in = P.tensors[o["in0"]]
    print("op", i, "in", (t_in["C"], t_in["H"], t_in["W"]), "-> C", P.tensors[o["out"]]["C"], "families", fam)
# all on round 2
for i in ops:
    lib.mfvi_plan_set_tune(plan.handle, i, 1, 1 | 8 << 8 | 1 << 16); lib.mfvi_plan_set_tune(plan.handle, i, 0, 1 | 8 << 8 | 1 << 16)
ref = run()
rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
print("default vs round-2: dz", rel(base[0], ref[0]), "dmu", rel(base[1], ref[1]))
for i in ops:
    for w in (0, 1):
        lib.mfvi_plan_set_tune(plan.handle, i, w, 0)
        got = run()
        print("op", i, "pass", w, "on default dispatch: dz", rel(got[0], ref[0]), "dmu", rel(got[1], ref[1]), "family", lib.mfvi_plan_last_kernel(plan.handle, i, w))
        lib.mfvi_plan_set_tune(plan.handle, i, w, 1 | 8 << 8 | 1 << 16)
