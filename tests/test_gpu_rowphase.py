"""Row-phase 3x3 kernels (csrc/conv_rp.hip): forward and fused-fold backward-data of the stride-1 layers on maps whose width is a
multiple of 64, against the oracle and against the round-2 kernels (conv_mfma.hip) on the same plan.  Reference op:
BayTorch/modules/reparam_layers.py:26-37 behind models/common.py:100-135 (ReflectionPad2d + Conv2d)."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _conv_bn_plan, _run_plan, _net_params      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

RP = 1 << 24


def enc_rp(mf, r, T=1, rem=0, ks=1):
    return mf | r << 8 | rem << 12 | ks << 13 | T << 16 | RP


def enc(a, b, c):
    return a | b << 8 | c << 16


@pytest.mark.parametrize("case", [(16, 16, 64, 64), (36, 16, 16, 64), (8, 32, 24, 128), (20, 48, 8, 192)])
def test_rowphase_forward_against_oracle(M, case):
    """Single layer from the plan's input: forward (and the shared backward-weight / generic backward-data) vs the oracle's conv."""
    cin, cout, H, W = case
    seed, step, k0, n = 2000 + cin + cout, 4, 1, 2
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * 9
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    d_mu, d_rho, d_x = dev(mu), dev(rho), dev(x)
    bn = torch.zeros(1, device="cuda")
    lib = M._lib.lib()
    tried = 0
    for mf, r, T, ks in [(1, 1, 1, 1), (1, 2, 1, 1), (1, 4, 2, 1), (2, 1, 1, 1), (2, 2, 3, 1), (4, 1, 1, 1)]:
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, enc_rp(mf, r, T, ks=ks)))
        try:
            y = plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, n)
        except M._lib.MfviError:
            continue          # tiling not valid for the shape (-3)
        tried += 1
        yh = host(y)
        for i in range(n):
            ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
            w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 3, 3); b = O.reparam(mu[nw:], rho[nw:], eb)
            assert relerr(yh[i], O.conv_fwd(x, w, b, 1)) < 2e-6, ("fwd", mf, r, T, ks, i)
    assert tried >= 3


@pytest.mark.parametrize("shape", [(36, 16, 16, 64), (68, 32, 8, 128), (16, 16, 32, 64), (32, 32, 8, 64), (20, 16, 4, 64), (132, 64, 8, 64)])
def test_rowphase_tilings_against_round2_kernels(M, shape):
    """conv -> BN+act -> 3x3 (under test) -> BN+act -> conv: every row-phase tiling of the forward and of the fused-fold backward-data
    (border rows 1 / H-2 and columns 1 / W-2 in every position, 4-channel remainder on the 4x4x1 instruction, several tiles per block)
    against the round-2 rectangular tiling, which the oracle / reference goldens pin (test_gpu_parity.py)."""
    cin, cout, H, W = shape
    n, seed = 2, 83
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, enc(1, 8, 1))); M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    tried = [0, 0]
    fwd = [(1, 1, 1, 0, 1), (1, 2, 1, 0, 1), (1, 4, 1, 0, 1), (2, 1, 2, 0, 1), (2, 2, 1, 0, 1), (4, 1, 1, 0, 1), (1, 1, 3, 0, 1),
           ]       # last field: k-steps per stage
    bwd = fwd + [(1, 1, 1, 1, 1), (1, 2, 1, 1, 1), (2, 1, 1, 1, 1), (2, 1, 2, 1, 1)]
    for which, cands in ((0, fwd), (1, bwd)):
        for mf, r, T, rem, ks in cands:
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, enc_rp(mf, r, T, rem, ks)))
            try:
                got = _run_plan(plan, P, seed, n, z, dout)
            except M._lib.MfviError:
                continue
            finally:
                M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, enc(1, 8, 1)))
            tried[which] += 1
            assert relerr(got[0], ref[0]) < 1e-6, ("out", which, mf, r, T, rem, ks)
            for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
                assert relerr(a, b) < 2e-5, (name, which, mf, r, T, rem, ks)
    assert tried[0] >= 2 and tried[1] >= 2, tried


@pytest.mark.parametrize("kw", [dict(H=32, W=64, input_depth=8, n_out=2, nd=(16,), nu=(16,), ns=(4,)),
                                dict(H=16, W=128, input_depth=4, n_out=1, nd=(16,), nu=(32,), ns=(4,))])
def test_rowphase_small_nets_against_oracle(M, kw):
    """Hour-glass nets whose top scale is 64 / 128 wide: the default dispatch puts its 3x3 stride-1 layers on the row-phase kernels
    (concat input 16 + 4 channels: the remainder path in backward-data); every gradient against the oracle."""
    net = O.make_net(**kw)
    seed, step, k0, n = 71, 2, 1, 2
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
    oh = host(out)
    r_dmu = np.zeros(n_vi); r_drho = np.zeros(n_vi); r_dbn = np.zeros(n_bnp)
    for i in range(n):
        ref, tape = O.net_forward(net, mu, rho, bnp, z, seed, step, k0 + i)
        assert relerr(oh[i], ref) < 2e-5, ("out", i)
        a, b, c_, dzr = tape.backward(dout[i], n_vi, n_bnp, want_dz=True)
        r_dmu += a; r_drho += b; r_dbn += c_
        assert relerr(host(dz)[i], dzr) < 2e-4, ("dz", i)
        tape.free()
    assert relerr(host(dmu), r_dmu) < 2e-4
    assert relerr(host(drho), r_drho) < 2e-4
    assert relerr(host(dbn), r_dbn) < 2e-4


@pytest.mark.parametrize("case", [(16, 16, 16, 32), (36, 16, 8, 32), (64, 64, 32, 32), (20, 48, 16, 32), (132, 32, 16, 32),
                                  (128, 128, 16, 16), (132, 64, 16, 16), (36, 16, 32, 16), (16, 48, 48, 16)])
def test_rowphase_forward_on_32_wide_maps(M, case):
    """Maps exactly 32 (16) wide: two (four) image rows side by side in the 64-pixel strip (conv_rp.hip, WSH) — every group carries both
    image borders and sits TH rows below the one before.  Single layer against the oracle's convolution, then behind BatchNorm + LeakyReLU against the
    round-2 tiling (output, BN statistics of the output through the following layer, gradients)."""
    cin, cout, H, W = case
    seed, step, k0, n = 2300 + cin + cout, 4, 1, 2
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * 9
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    d_mu, d_rho, d_x = dev(mu), dev(rho), dev(x)
    bn = torch.zeros(1, device="cuda")
    lib = M._lib.lib()
    tried = 0
    for mf, r, T, ks in [(1, 1, 1, 1), (1, 2, 1, 1), (2, 1, 2, 1), (2, 2, 1, 1), (4, 1, 1, 1), (1, 4, 1, 1), (1, 1, 3, 1), (1, 1, 1, 2), (1, 1, 1, 4), (2, 1, 1, 2),
                         (2, 1, 1, 4)]:       # (last field: 4-channel k-steps per stage, 16-wide maps only)
        if H % ((64 // W) * 4 * r) or (ks > 1 and (W != 16 or cin % (4 * ks))):       # (the tile: 64 / W groups of 4 r image rows)
            continue
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 0, 0, enc_rp(mf, r, T, ks=ks)))
        y = plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, n)
        if lib.mfvi_plan_last_kernel(plan.handle, 0, 0) != 2:
            continue          # tiling not valid for the shape: fell back
        tried += 1
        yh = host(y)
        for i in range(n):
            ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
            w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 3, 3); b = O.reparam(mu[nw:], rho[nw:], eb)
            assert relerr(yh[i], O.conv_fwd(x, w, b, 1)) < 2e-6, ("fwd", mf, r, T, ks, i)
    assert tried >= (3 if H % ((64 // W) * 8) == 0 else 2)
    # behind BatchNorm + LeakyReLU, with the BN statistics of the output feeding the next layer and the backward pass
    P2, plan2, zin2, out2 = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    M._lib.check(lib.mfvi_plan_set_tune(plan2.handle, 1, 0, enc(1, 8, 1)))
    ref = _run_plan(plan2, P2, seed, n, z, dout)
    M._lib.check(lib.mfvi_plan_set_tune(plan2.handle, 1, 0, enc_rp(1, 1, 1)))
    got = _run_plan(plan2, P2, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan2.handle, 1, 0) == 2
    assert relerr(got[0], ref[0]) < 1e-6
    for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
        assert relerr(a, b) < 2e-5, name


@pytest.mark.parametrize("shape", [(36, 16, 8, 32), (36, 16, 16, 32), (68, 32, 32, 32), (20, 16, 24, 32), (132, 64, 16, 32), (16, 48, 48, 32),
                                   (132, 128, 16, 16), (128, 128, 16, 16), (36, 16, 32, 16), (20, 16, 48, 16), (68, 32, 64, 16)])
def test_rowphase_backward_data_on_32_wide_maps(M, shape):
    """Fused-fold backward-data on maps 32 / 16 wide: image rows 1 and H-2 (the reflection adjoint's spare window rows) live in the first
    group of the first tile and the last group of the last one, every group carries both border columns; one tile that is first AND last
    (H = 8 with 4-row halves, H = 16 with 8-row halves), several tiles per block, 4-channel remainder.  Against the round-2 tiling."""
    cin, cout, H, W = shape
    n, seed = 2, 89
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, enc(1, 8, 1))); M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    tried = 0
    for mf, r, T, rem, ks in [(1, 1, 1, 0, 1), (1, 2, 1, 0, 1), (2, 1, 1, 0, 1), (1, 1, 2, 0, 1), (1, 1, 3, 0, 1), (1, 1, 1, 1, 1), (1, 2, 1, 1, 1), (2, 1, 2, 1, 1),
                              (1, 1, 1, 0, 2), (1, 1, 1, 0, 4), (2, 1, 1, 0, 2), (1, 1, 1, 1, 2), (1, 1, 1, 1, 4), (2, 1, 1, 1, 2)]:
        if H % ((64 // W) * 4 * r) or (ks > 1 and W != 16):
            continue
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc_rp(mf, r, T, rem, ks)))
        try:
            got = _run_plan(plan, P, seed, n, z, dout)
        except M._lib.MfviError:
            continue
        finally:
            fam = lib.mfvi_plan_last_kernel(plan.handle, 1, 1)
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
        if fam != 2:
            continue
        tried += 1
        assert relerr(got[0], ref[0]) < 1e-6, ("out", mf, r, T, rem, ks)
        for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
            assert relerr(a, b) < 2e-5, (name, mf, r, T, rem, ks)
    assert tried >= (4 if W == 16 and cout % 16 == 0 and cout >= 64 else 2), tried


def test_rowphase_small_net_32_wide_against_oracle(M):
    """Hour-glass net whose top scale is 32 wide: default dispatch (row-phase forward and backward-data, W32) against the oracle."""
    test_rowphase_small_nets_against_oracle(M, dict(H=32, W=32, input_depth=8, n_out=2, nd=(16, 32), nu=(16, 32), ns=(4, 4)))


def test_rowphase_small_net_16_wide_against_oracle(M):
    """Top scale 64 wide, third scale 16 wide: default dispatch puts 64-, 32- and 16-wide layers on the row-phase kernels; against the oracle."""
    # (nd = (16, 16, 32) with this seed has a LeakyReLU input within fp32 rounding of zero at the 16 x 16 level: the float64 oracle and any
    #  float32 kernel — round-2 kernels included — then take different branches of the derivative; scripts/dev/dbg_net_oracle.py)
    test_rowphase_small_nets_against_oracle(M, dict(H=64, W=64, input_depth=8, n_out=2, nd=(16, 32, 32), nu=(16, 32, 32), ns=(4, 4, 4)))


@pytest.mark.parametrize("case", [(16, 64, 40, 32), (16, 128, 24, 32), (16, 64, 48, 16)])
def test_heuristic_tiling_never_strands_a_layer(M, case):
    """No tiling set on the plan (the heuristic of rp_default_tune answers) on narrow maps whose height the taller row-phase tiles do
    not divide (W = 32, H = 40: tiles of 8 r rows need H % 16 == 0 for r = 2), many samples: the layer must stay on an MFMA kernel —
    the heuristic's own tiling when valid, else the round-2 tiles — never drop to the generic fp32 kernels."""
    cin, cout, H, W = case
    seed, step, k0, n = 2500 + H + W, 1, 0, 16
    P = M.Program()
    zin = P.tensor(cin, H, W); out = P.tensor(cout, H, W); P.conv(zin, out, 3, 1)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * 9
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    lib = M._lib.lib()
    y = host(plan.forward(dev(mu), dev(rho), torch.zeros(1, device="cuda"), dev(x), seed, step, k0, n))
    assert lib.mfvi_plan_last_kernel(plan.handle, 0, 0) in (1, 2), "the layer fell off the MFMA kernels"
    for i in (0, n - 1):
        ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
        w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, 3, 3); b = O.reparam(mu[nw:], rho[nw:], eb)
        assert relerr(y[i], O.conv_fwd(x, w, b, 1)) < 2e-6, i
