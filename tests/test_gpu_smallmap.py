"""Small-map forward kernel (csrc/conv_small.hip): 3x3 stride-1 layers on 8- and 16-wide maps with the block's whole reduction in LDS
(one stage, all 8 waves on the matrix cores), against the oracle and — behind BatchNorm + LeakyReLU, with the output's BN statistics feeding
the next layer and the backward pass — against the round-2 tiling.  Reference op: BayTorch/modules/reparam_layers.py:26-37 behind
models/common.py:100-135; the `deeper` / `up` layers of the 8x8 and 16x16 scales of models/skip.py."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _conv_bn_plan, _run_plan      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SM = 1 | 1 << 26


def enc(a, b, c):
    return a | b << 8 | c << 16


@pytest.mark.parametrize("case", [(128, 128, 8, 8), (132, 128, 16, 16), (128, 64, 16, 16), (64, 32, 32, 16), (36, 16, 16, 16), (16, 32, 8, 8), (4, 16, 4, 16),
                                  (20, 48, 24, 8), (128, 16, 4, 16), (144, 16, 8, 16)])
def test_smallmap_forward_against_oracle(M, case):
    cin, cout, H, W = case
    seed, step, k0, n = 2600 + cin + cout + H, 3, 2, 3
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * 9
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, SM))
    y = plan.forward(dev(mu), dev(rho), torch.zeros(1, device="cuda"), dev(x), seed, step, k0, n)
    assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 4
    yh = host(y)
    for i in range(n):
        ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
        w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 3, 3); b = O.reparam(mu[nw:], rho[nw:], eb)
        assert relerr(yh[i], O.conv_fwd(x, w, b, 1)) < 2e-6, ("fwd", i)


@pytest.mark.parametrize("shape", [(128, 128, 8, 8), (132, 128, 16, 16), (36, 16, 16, 16), (64, 64, 32, 16)])
def test_smallmap_forward_inside_a_plan(M, shape):
    """conv -> BN+act -> 3x3 (under test) -> BN+act -> conv: deferred BN + LeakyReLU on load, BN statistics of the output, all gradients."""
    cin, cout, H, W = shape
    n, seed = 2, 97
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, SM))
    got = _run_plan(plan, P, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 0) == 4
    assert relerr(got[0], ref[0]) < 3e-6          # (another summation order — two halves of the channel groups — behind a BatchNorm over 64 pixels)
    for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
        assert relerr(a, b) < 2e-5, name


def test_smallmap_tiling_on_a_wide_map_falls_back(M):
    """A 64-wide map is not served: the plan takes the generic kernel for that launch (family 0), not a wrong tile."""
    P = M.Program()
    zin = P.tensor(16, 16, 64); out = P.tensor(16, 16, 64); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=1)
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, SM))
    z = torch.zeros
    plan.forward(z(P.n_vi, device="cuda"), z(P.n_vi, device="cuda"), z(1, device="cuda"), z(16 * 16 * 64, device="cuda"), 1, 0, 0, 1)
    assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 0


@pytest.mark.parametrize("shape", [(128, 128, 8, 8), (132, 128, 16, 16), (128, 128, 16, 16), (36, 16, 16, 16), (64, 64, 32, 16), (16, 32, 8, 8), (20, 16, 24, 8),
                                   (68, 32, 4, 16), (16, 16, 64, 16)])
def test_smallmap_backward_data_inside_a_plan(M, shape):
    """Fused-fold backward-data of the small-map kernel: BN-backward on load, zero padding, the reflection adjoint through spare window rows /
    columns (rows 1 and H-2 in one tile on 8-row maps, in the first / last tile otherwise; corners), a partial last output fragment
    (132 = 8 x 16 + 4 channels), LeakyReLU' + BN-backward sums of the input tensor in the epilogue.  Against the round-2 tiling."""
    cin, cout, H, W = shape
    n, seed = 2, 101
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, enc(1, 8, 1))); M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, SM))
    got = _run_plan(plan, P, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) == 4
    assert relerr(got[0], ref[0]) < 1e-6
    for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
        assert relerr(a, b) < 2e-5, name
