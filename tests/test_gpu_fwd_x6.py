"""Forward of the 3x3 stride-1 layers on the bf16 matrix cores with three-way split operands (csrc/conv_x6.hip, tune bit 25) against the
oracle's convolution and against the fp32-MFMA kernels on the same plan.  Reference op: BayTorch/modules/reparam_layers.py:26-37 behind
models/common.py:100-135 (ReflectionPad2d + Conv2d)."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _conv_bn_plan, _run_plan      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

X6 = 1 << 25


def enc(a, b, c):
    return a | b << 8 | c << 16


@pytest.mark.parametrize("case", [(32, 16, 8, 64, 2), (36, 16, 16, 64, 2), (68, 32, 8, 128, 2), (36, 48, 8, 64, 3), (64, 20, 16, 64, 1)])
def test_x6_forward_against_oracle(M, case):
    """Single layer from the plan's input (no BatchNorm): every sample's output against the oracle's convolution with the drawn weights."""
    cin, cout, H, W, n = case
    seed, step, k0 = 2100 + cin + cout, 4, 1
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * 9
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    d_mu, d_rho, d_x = dev(mu), dev(rho), dev(x)
    bn = torch.zeros(1, device="cuda")
    lib = M._lib.lib()
    # (third field: bit 12 of the tiling = the remainder plane rides on the last group's pass — one output fragment per wave, Cin = 32 n + 4)
    for mf, T, mrg in ((1, 1, 0), (2, 1, 0), (1, 2, 0), (2, 3, 0)) + (((1, 1, 1), (1, 3, 1)) if cin % 32 == 4 else ()):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, mf | 8 << 8 | mrg << 12 | T << 16 | X6))
        yh = host(plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, n))
        assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 3, "the bf16x6 kernel did not run"
        for i in range(n):
            ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
            w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 3, 3); b = O.reparam(mu[nw:], rho[nw:], eb)
            assert relerr(yh[i], O.conv_fwd(x, w, b, 1)) < 2e-6, ("fwd", mf, i)
    # eval branch (w = mu, no weight draw, so no scratch for the pieces): the tiling stays set, the layer runs on its fp32 default
    y0 = plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, 1, sample_weights=False)
    assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 2, "expected the row-phase fp32 kernel"
    assert relerr(host(y0)[0], O.conv_fwd(x, mu[:nw].reshape(cout, cin, 3, 3), mu[nw:], 1)) < 2e-6


@pytest.mark.parametrize("shape", [(36, 16, 16, 64), (36, 16, 24, 128), (68, 32, 8, 128), (32, 32, 8, 64), (132, 64, 8, 64), (100, 24, 16, 64)])
def test_x6_forward_against_fp32_mfma(M, shape):
    """conv -> BN+act -> 3x3 (under test) -> BN+act -> conv: deferred BN + LeakyReLU on load, reflected rows / columns in every position,
    the remainder plane, BN statistics of the output (they feed the next layer and the whole backward pass)."""
    cin, cout, H, W = shape
    n, seed = 2, 87
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    for mf, T, mrg in ((1, 1, 0), (2, 1, 0), (1, 2, 0), (2, 4, 0)) + (((1, 1, 1), (1, 2, 1), (1, 8, 1)) if cin % 32 == 4 else ()):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, mf | 8 << 8 | mrg << 12 | T << 16 | X6))
        got = _run_plan(plan, P, seed, n, z, dout)
        assert lib.mfvi_plan_last_kernel(plan.handle, 1, 0) == 3, "the bf16x6 kernel did not run"
        assert relerr(got[0], ref[0]) < 2e-6, ("out", mf)
        for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
            assert relerr(a, b) < 2e-5, (name, mf)
