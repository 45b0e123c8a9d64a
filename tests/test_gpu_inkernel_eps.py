""""Draws eps, forms w and convolves in one launch" (BASELINE north_star; BayTorch/modules/module.py:82-85 + reparam_layers.py:28-37) as a tiling of
the plan: tune bit 27 puts a pass of a layer on the generic kernels, which sample inside the kernel instead of reading the sampled-weight
slab.  eps is bit-identical by the RNG spec, so every result must agree with the slab-reading matrix-core kernels to rounding."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _net_params      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GENERIC = 1 << 27


def test_inkernel_eps_layers_match_the_slab_path(M):
    kw = dict(H=32, W=64, input_depth=16, n_out=2, nd=(16, 16), nu=(16, 16), ns=(4, 4))
    net = O.make_net(**kw)
    seed, step, k0, n = 79, 3, 1, 3
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    plan = P.compile(zin, out_id, max_samples=n)
    lib = M._lib.lib()
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * net.H * net.W)).reshape(net.input_depth, net.H, net.W)
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)

    def run():
        out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n)
        dout = dev(O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape)))
        dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
        plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dout, dmu, drho, dbn)
        return host(out), host(dmu), host(drho), host(dbn)
    ref = run()
    tiny = [i for i, o in enumerate(P.ops) if o["type"] == 1 and P.tensors[o["out"]]["C"] * P.tensors[o["in0"]]["C"] * o["ksize"] ** 2 <= 2560]
    assert len(tiny) >= 4
    for i in tiny:
        for which in (0, 2):      # forward and backward-weight draw inside the kernel; backward-data keeps the fused-fold matrix-core kernel
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, i, which, GENERIC))
    got = run()
    assert all(lib.mfvi_plan_last_kernel(plan.handle, i, 0) == 0 and lib.mfvi_plan_last_kernel(plan.handle, i, 2) == 0 for i in tiny), "the generic kernels did not run"
    assert relerr(got[0], ref[0]) < 1e-5          # (the generic kernels evaluate softplus with libm, the weight draw with the hardware exp / log: 3e-6 per weight)
    for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dbn")):
        assert relerr(a, b) < 2e-5, name
