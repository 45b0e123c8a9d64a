"""Backward-weight of the 3x3 stride-1 layers on the bf16 matrix cores with three-way split operands (csrc/conv_bww_x6.hip, tune
w = 11) against the fp32-MFMA kernels on the same plan (which the oracle / reference goldens pin, test_gpu_parity.py) and against a
float64 restatement of the gradient.  Reference op: autograd of BayTorch/modules/reparam_layers.py:37 behind models/common.py:100-135."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _conv_bn_plan, _run_plan      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def enc(a, b, c):
    return a | b << 8 | c << 16


@pytest.mark.parametrize("shape", [(36, 16, 16, 64), (68, 32, 8, 128), (16, 16, 32, 64), (32, 32, 8, 64), (20, 16, 4, 64), (132, 64, 8, 64),
                                   (48, 16, 6, 128), (52, 48, 12, 64), (100, 24, 8, 64),
                                   # maps 32 (96) wide: bands of 32 pixels, four padded rows per stage (H + 2 not a multiple of 4: zero rows below the map)
                                   (36, 16, 8, 32), (68, 32, 32, 32), (64, 64, 12, 32), (132, 48, 6, 96), (20, 16, 4, 32)])
def test_x6_backward_weight_against_fp32_mfma(M, shape):
    """conv -> BN+act -> 3x3 (under test) -> BN+act -> conv.  Every channel grouping (one / two input fragments per block, with and without
    the 4-channel remainder fragment, several groups, output channels not a multiple of the block's), one and two output fragments per
    block, one strip and several (zero rows above / below the map, ring wrap-around), bias gradient on group 0."""
    cin, cout, H, W = shape
    n, seed = 2, 95
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 2, enc(1, 4, 1)))       # plain 4-wave fp32 variant, one input tile per block
    ref = _run_plan(plan, P, seed, n, z, dout)
    tried = 0
    for cof in (1, 2):
        for tgt in (1, 4):
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 2, enc(cof, 11, tgt)))
            got = _run_plan(plan, P, seed, n, z, dout)
            assert lib.mfvi_plan_last_kernel(plan.handle, 1, 2) == 3, "the bf16x6 kernel did not run"
            tried += 1
            assert np.array_equal(got[0], ref[0])
            for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
                assert relerr(a, b) < 2e-5, (name, cof, tgt)
    assert tried == 4


@pytest.mark.parametrize("case", [(36, 16, 8, 64, 2), (16, 32, 12, 128, 2), (36, 32, 10, 32, 2), (68, 16, 6, 64, 3), (32, 40, 4, 96, 1)])
def test_x6_backward_weight_against_float64(M, case):
    """Single layer from the plan's input (no BatchNorm on either side, the input shared by all samples): d mu / d rho against the gradient
    restated in float64; one, two and three samples (launch orders with and without the XCD banding)."""
    cin, cout, H, W, n = case
    seed, step, k0 = 3000 + cin + cout, 5, 0
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * 9
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    dy = O.normal_fill(seed, 2, 3, 0, 0, n * cout * H * W).reshape(n, cout, H, W)
    d_mu, d_rho, d_x, d_dy = dev(mu), dev(rho), dev(x), dev(dy)
    bn = torch.zeros(1, device="cuda")
    lib = M._lib.lib()
    want_mu = np.zeros(nw + cout); want_rho = np.zeros(nw + cout)
    sig = 1.0 / (1.0 + np.exp(-rho.astype(np.float64)))
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)), mode="reflect")
    for i in range(n):
        dw = np.zeros((cout, cin, 3, 3))
        for ky in range(3):
            for kx in range(3):
                dw[:, :, ky, kx] = np.einsum("ohw,ihw->oi", dy[i].astype(np.float64), xp[:, ky:ky + H, kx:kx + W])
        db = dy[i].astype(np.float64).sum(axis=(1, 2))
        gw = np.concatenate([dw.ravel(), db])
        e = np.concatenate([O.eps(seed, step, k0 + i, 0, 0, nw), O.eps(seed, step, k0 + i, 0, 1, cout)]).astype(np.float64)
        want_mu += gw; want_rho += gw * e * sig
    for cof in (1, 2):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 2, enc(cof, 11, 2)))
        plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, n)
        dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(bn)
        plan.backward(d_mu, d_rho, bn, d_x, seed, step, k0, n, d_dy, dmu, drho, dbn)
        assert lib.mfvi_plan_last_kernel(plan.handle, 0, 2) == 3, "the bf16x6 kernel did not run"
        assert relerr(host(dmu), want_mu) < 5e-6, cof
        assert relerr(host(drho), want_rho) < 5e-6, cof
