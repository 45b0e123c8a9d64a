"""The backward-data of the 4-channel 1x1 skip convolution formed inside the fold of the tensor it shares with the scale's stride-2 convolution
(csrc/elementwise.hip finalize_dx_vec1_kernel; plan.hip books it as kernel family 5 on the skip op's backward-data slot): hour-glass nets
against the oracle's tape, every gradient.  Reference: models/skip.py:60-66 (skip branch: conv 1x1 -> bn -> act beside the deeper branch),
autograd of BayTorch/modules/reparam_layers.py:37 for both consumers, summed at the shared input."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _net_params      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("kw", [dict(H=32, W=32, input_depth=8, n_out=2, nd=(16, 32), nu=(16, 32), ns=(4, 4)),
                                dict(H=16, W=48, input_depth=4, n_out=1, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 2, 8)),
                                dict(H=64, W=64, input_depth=8, n_out=2, nd=(16, 32), nu=(16, 64), ns=(4, 4))])
def test_skip_backward_data_inside_the_fold(M, kw):
    net = O.make_net(**kw)
    seed, step, k0, n = 83, 2, 1, 2
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    conv, bn, n_vi, n_bnp = O.net_table(net)
    plan = P.compile(zin, out_id, max_samples=n)
    lib = M._lib.lib()
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * net.H * net.W)).reshape(net.input_depth, net.H, net.W)
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n)
    dout = O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    dz = torch.empty((n,) + z.shape, device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dev(dout), dmu, drho, dbn, dz=dz)
    # the skip convolutions: 1x1, at most 8 output channels, their input shared with one other convolution
    n_cons = {}
    for o in P.ops:
        for key in ("in0", "in1"):
            if o.get(key, -1) >= 0:
                n_cons[o[key]] = n_cons.get(o[key], 0) + 1
    skips = [i for i, o in enumerate(P.ops) if o["type"] == 1 and o["ksize"] == 1 and P.tensors[o["out"]]["C"] <= 8 and n_cons.get(o["in0"], 0) == 2 and P.tensors[o["in0"]]["W"] % 4 == 0
             and o["w_off"] % 4 == 0 and P.tensors[o["in0"]]["C"] % 4 == 0]      # (a layer outside the sampled-weight slab keeps its own generic launch)
    assert skips
    fams = [lib.mfvi_plan_last_kernel(plan.handle, i, 1) for i in skips]
    assert all(f == 5 for f in fams), ("skip ops %r ran as families %r" % (skips, fams))
    r_dmu = np.zeros(n_vi); r_drho = np.zeros(n_vi); r_dbn = np.zeros(n_bnp)
    for i in range(n):
        ref, tape = O.net_forward(net, mu, rho, bnp, z, seed, step, k0 + i)
        a, b, c_, dzr = tape.backward(dout[i], n_vi, n_bnp, want_dz=True)
        r_dmu += a; r_drho += b; r_dbn += c_
        assert relerr(host(dz)[i], dzr) < 2e-4, ("dz", i)
        tape.free()
    assert relerr(host(dmu), r_dmu) < 2e-4
    assert relerr(host(drho), r_drho) < 2e-4
    assert relerr(host(dbn), r_dbn) < 2e-4
