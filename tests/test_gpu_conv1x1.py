"""One-stage 1x1 kernels (csrc/conv_1x1.hip, tune bit 26 on a 1x1 layer): the `up` 1x1 convolutions of skip() with 32 ... 128 channels — forward
against the oracle; forward (deferred BN + LeakyReLU on load, BN statistics of the output) and backward-data with the fold (BN-backward on
load, LeakyReLU', BN-backward sums of the input, ga) inside a plan against the staged tiling, which the oracle / reference goldens pin
(test_gpu_parity.py); a small hour-glass net against the oracle's tape.  Reference op: BayTorch/modules/reparam_layers.py:26-37 with a 1x1
filter behind models/common.py:100-135; models/skip.py:110-119."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _net_params      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SM = 1 | 1 << 26


# (cin, cout, H, W): 2 / 4 / 8 output fragments = 4 / 2 / 1 pixel parts per fragment; rectangular layers in both directions;
# one tile per sample (8x8) up to 64 tiles; the BASELINE shapes 128 @16^2 / 32^2, 64 @64^2, 32 @128^2 (a band of it)
CASES = [(128, 128, 16, 16), (128, 128, 32, 32), (64, 64, 64, 64), (32, 32, 16, 128), (128, 64, 8, 8), (64, 128, 8, 16), (16, 32, 8, 8),
         (48, 64, 4, 16), (64, 112, 8, 24)]      # 48 / 112 reduction channels (3 / 7 groups): forward of the first, backward-data of the second


def served(mout):
    return mout // 16 in (2, 4, 8)


@pytest.mark.parametrize("case", [c for c in CASES if served(c[1])])
def test_conv1x1_forward_against_oracle(M, case):
    cin, cout, H, W = case
    seed, step, k0, n = 3100 + cin + cout + H, 3, 2, 3
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 1, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, SM))
    y = plan.forward(dev(mu), dev(rho), torch.zeros(1, device="cuda"), dev(x), seed, step, k0, n)
    assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 4
    yh = host(y)
    for i in range(n):
        ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
        w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 1, 1); b = O.reparam(mu[nw:], rho[nw:], eb)
        assert relerr(yh[i], O.conv_fwd(x, w, b, 1)) < 2e-6, ("fwd", i)


def _plan_1x1(M, cin, cout, H, W, n):
    """z -> 3x3 conv -> BN+act -> 1x1 (under test) -> BN+act -> 1x1 conv -> out"""
    P = M.Program()
    zin = P.tensor(8, H, W)
    x = P.tensor(cin, H, W); P.conv(zin, x, 3, 1); P.set_bn(x, act=True)
    y = P.tensor(cout, H, W); P.conv(x, y, 1, 1); P.set_bn(y, act=True)
    out = P.tensor(2, H, W); P.conv(y, out, 1, 1)
    return P, P.compile(zin, out, n), zin, out


def _run(plan, P, seed, n, z, dout):
    mu = dev(0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi)); rho = dev(-3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi))
    bn = dev(1 + 0.1 * O.normal_fill(seed, 2, 4, 0, 0, max(P.n_bn, 1)))
    o = plan.forward(mu, rho, bn, z, seed, 3, 0, n)
    dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
    dz = torch.empty((n,) + tuple(z.shape), device="cuda")
    plan.backward(mu, rho, bn, z, seed, 3, 0, n, dout, dmu, drho, dbn, dz=dz)
    return host(o), host(dmu), host(drho), host(dbn), host(dz)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("which", [0, 1])
def test_conv1x1_inside_a_plan(M, case, which):
    cin, cout, H, W = case
    n, seed = 2, 101
    P, plan, zin, out = _plan_1x1(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, 8 * H * W).reshape(8, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    ref = _run(plan, P, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, which) == 1
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, SM))
    got = _run(plan, P, seed, n, z, dout)
    # 3, 5, 6, 7 output fragments are not served: the plan runs that launch on the generic kernel (family 0), same numbers
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, which) == (4 if served(cin if which else cout) else 0), "the one-stage 1x1 kernel did not run"
    if which == 1:
        assert relerr(got[0], ref[0]) == 0.0          # the forward pass is untouched
    else:
        assert relerr(got[0], ref[0]) < 3e-6
    for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dbn", "dz")):
        assert relerr(a, b) < 2e-5, name


def test_conv1x1_unserved_shapes_fall_back(M):
    """36 input channels (not a multiple of 16) / a 6x6 map (36 pixels): the tune bit is refused for that launch, the plan runs the generic
    kernel (family 0) — never a wrong tile."""
    lib = M._lib.lib()
    for cin, cout, H, W in ((36, 32, 8, 8), (32, 32, 6, 6), (32, 144, 8, 8)):
        P = M.Program()
        zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 1, 1)
        plan = P.compile(zin, out, max_samples=1)
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, SM))
        z = torch.zeros
        plan.forward(z(P.n_vi, device="cuda"), z(P.n_vi, device="cuda"), z(1, device="cuda"), z(cin * H * W, device="cuda"), 1, 0, 0, 1)
        assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 0


def test_conv1x1_small_net_against_oracle(M):
    """An hour-glass net with 32- and 64-channel `up` 1x1 layers, every 1x1 layer the kernel serves switched to it in both directions:
    output and all gradients against the oracle's tape."""
    kw = dict(H=32, W=32, input_depth=8, n_out=2, nd=(32, 64), nu=(32, 64), ns=(4, 4))
    net = O.make_net(**kw)
    seed, step, k0, n = 79, 2, 1, 2
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    conv, bn, n_vi, n_bnp = O.net_table(net)
    plan = P.compile(zin, out_id, max_samples=n)
    lib = M._lib.lib()
    cand = []
    for i, o in enumerate(P.ops):
        tin, tout = P.tensors[o["in0"]] if "in0" in o else None, P.tensors[o["out"]]
        if o["type"] == 1 and o["ksize"] == 1 and tout["C"] % 16 == 0 and tin is not None and tin["C"] % 16 == 0 and (tout["H"] * tout["W"]) % 64 == 0:
            for which in (0, 1):
                M._lib.check(lib.mfvi_plan_set_tune(plan.handle, i, which, SM))
            cand.append(i)
    assert cand
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * net.H * net.W)).reshape(net.input_depth, net.H, net.W)
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n)
    assert all(lib.mfvi_plan_last_kernel(plan.handle, i, 0) == 4 for i in cand), "a 1x1 layer did not run on the one-stage kernel"
    dout = O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    dz = torch.empty((n,) + z.shape, device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dev(dout), dmu, drho, dbn, dz=dz)
    assert any(lib.mfvi_plan_last_kernel(plan.handle, i, 1) == 4 for i in cand), "no backward-data ran on the one-stage kernel"
    r_dmu = np.zeros(n_vi); r_drho = np.zeros(n_vi); r_dbn = np.zeros(n_bnp)
    for i in range(n):
        ref, tape = O.net_forward(net, mu, rho, bnp, z, seed, step, k0 + i)
        assert relerr(host(out)[i], ref) < 1e-4, ("out", i)
        a, b, c_, dzr = tape.backward(dout[i], n_vi, n_bnp, want_dz=True)
        r_dmu += a; r_drho += b; r_dbn += c_
        assert relerr(host(dz)[i], dzr) < 2e-4, ("dz", i)
        tape.free()
    assert relerr(host(dmu), r_dmu) < 2e-4
    assert relerr(host(drho), r_drho) < 2e-4
    assert relerr(host(dbn), r_dbn) < 2e-4


ST = 1 | 1 << 28


# streaming forward of the narrow 1x1 layers (conv1_stream_kernel, tune bit 28): reduction depths 4 ... 16, 32, 64; 16 / 4 / 2 / 3 output channels;
# maps from one 64-pixel group per block to several unrolled batches with a ragged tail
ST_CASES = [(16, 16, 32, 64), (16, 4, 64, 64), (16, 2, 16, 48), (4, 16, 8, 8), (8, 3, 8, 24), (12, 16, 16, 16), (32, 4, 32, 32), (64, 4, 16, 16), (32, 32, 16, 32), (16, 24, 8, 8)]      # the last two: two output fragments


@pytest.mark.parametrize("case", ST_CASES)
def test_conv1x1_streaming_forward_against_oracle(M, case):
    cin, cout, H, W = case
    seed, step, k0, n = 3300 + cin + cout + H, 3, 2, 3
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 1, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, ST))
    y = plan.forward(dev(mu), dev(rho), torch.zeros(1, device="cuda"), dev(x), seed, step, k0, n)
    assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 6
    yh = host(y)
    for i in range(n):
        ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
        w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 1, 1); b = O.reparam(mu[nw:], rho[nw:], eb)
        assert relerr(yh[i], O.conv_fwd(x, w, b, 1)) < 2e-6, ("fwd", i)


@pytest.mark.parametrize("case", [(16, 16, 32, 64), (16, 4, 16, 48), (32, 4, 16, 16), (32, 32, 16, 16)])
def test_conv1x1_streaming_forward_inside_a_plan(M, case):
    """z -> 3x3 -> BN+act -> 1x1 (under test, streaming) -> BN+act -> 1x1 -> out: deferred BN + LeakyReLU in the register, the output's BN statistics
    (they feed the next layer and the whole backward pass): every gradient against the staged kernel."""
    cin, cout, H, W = case
    n, seed = 2, 103
    P, plan, zin, out = _plan_1x1(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, 8 * H * W).reshape(8, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    ref = _run(plan, P, seed, n, z, dout)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, ST))
    got = _run(plan, P, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 0) == 6, "the streaming kernel did not run"
    assert relerr(got[0], ref[0]) < 3e-6
    for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dbn", "dz")):
        assert relerr(a, b) < 2e-5, name


def test_conv1x1_streaming_unserved_shapes_fall_back(M):
    lib = M._lib.lib()
    for cin, cout, H, W in ((20, 4, 8, 8), (16, 48, 8, 8), (64, 32, 8, 8), (16, 4, 3, 5)):
        P = M.Program()
        zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 1, 1)
        plan = P.compile(zin, out, max_samples=1)
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, ST))
        z = torch.zeros
        plan.forward(z(P.n_vi, device="cuda"), z(P.n_vi, device="cuda"), z(1, device="cuda"), z(cin * H * W, device="cuda"), 1, 0, 0, 1)
        assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) == 0
