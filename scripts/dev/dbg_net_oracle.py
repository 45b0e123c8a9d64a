"""Small skip nets against the oracle (debug helper): which configuration disagrees?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import mfvi_dip_mia_amd as M
from oracle import oracle as O
from test_gpu_parity import dev, host, relerr, _net_params
def check(kw, seed=71):
    net = O.make_net(**kw); step, k0, n = 2, 1, 2
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    conv, bn, n_vi, n_bnp = O.net_table(net)
    plan = P.compile(zin, out_id, max_samples=n)
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * net.H * net.W)).reshape(net.input_depth, net.H, net.W)
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n)
    dout = O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    dz = torch.empty((n,) + z.shape, device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dev(dout), dmu, drho, dbn, dz=dz)
    oh = host(out); res = []
    r_dmu = np.zeros(n_vi)
    for i in range(n):
        ref, tape = O.net_forward(net, mu, rho, bnp, z, seed, step, k0 + i)
        a, b, c_, dzr = tape.backward(dout[i], n_vi, n_bnp, want_dz=True)
        r_dmu += a
        res.append((relerr(oh[i], ref), relerr(host(dz)[i], dzr)))
        tape.free()
    print(kw, "seed", seed, "out/dz per sample", res, "dmu", relerr(host(dmu), r_dmu), flush=True)
base = dict(H=64, W=64, input_depth=8, n_out=2, nd=(16, 16, 32), nu=(16, 16, 32), ns=(4, 4, 4))
check(base); check(base, seed=5)
check(dict(base, H=32, W=32)); check(dict(base, nd=(16, 16), nu=(16, 16), ns=(4, 4))); check(dict(base, nd=(16, 32, 32), nu=(16, 32, 32)))
check(dict(base, nd=(16, 16, 16), nu=(16, 16, 16))); check(dict(base, H=128, W=128))

def locate(kw, seed, i):
    net = O.make_net(**kw); step, k0, n = 2, 1, 2
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    conv, bn, n_vi, n_bnp = O.net_table(net)
    plan = P.compile(zin, out_id, max_samples=n)
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * net.H * net.W)).reshape(net.input_depth, net.H, net.W)
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n)
    dout = O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    dz = torch.empty((n,) + z.shape, device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dev(dout), dmu, drho, dbn, dz=dz)
    ref, tape = O.net_forward(net, mu, rho, bnp, z, seed, step, k0 + i)
    a, b, c_, dzr = tape.backward(dout[i], n_vi, n_bnp, want_dz=True)
    d = np.abs(host(dz)[i] - dzr)
    print("max diff", d.max(), "ref max", np.abs(dzr).max(), "mean diff", d.mean())
    idx = np.argwhere(d > 0.1 * d.max())
    print("elements above 10% of the max:", len(idx), "channels", sorted(set(idx[:, 0])), "rows", idx[:, 1].min(), idx[:, 1].max(), "cols", idx[:, 2].min(), idx[:, 2].max())
    print("elements above 1e-4 of ref max:", int((d > 1e-4 * np.abs(dzr).max()).sum()), "of", d.size)
locate(dict(base, nd=(16, 16), nu=(16, 16), ns=(4, 4)), 71, 0)
