"""Backward-data WITH the fold of the 3x3 stride-1 layers on the bf16 matrix cores with three-way split operands (csrc/conv_bwd_x6.hip,
tune bit 25 on the backward-data slot) against the fp32-MFMA tilings on the same plan (which the oracle / reference goldens pin,
test_gpu_parity.py) and against the oracle on small hour-glass nets.  Reference op: autograd of BayTorch/modules/reparam_layers.py:37
behind models/common.py:100-135 (ReflectionPad2d + Conv2d), then LeakyReLU' and the BatchNorm-backward sums of the layer's input."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import M, dev, host, relerr, _conv_bn_plan, _run_plan, _net_params      # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

X6 = 1 << 25


def enc(a, b, c):
    return a | b << 8 | c << 16


def x6_tune(cout, T):
    return T | {16: 8, 32: 4, 64: 2}[cout] << 8 | X6


# (cin, cout, H, W): 16 / 32 / 64 output channels = the three forms of the kernel (K = [piece | piece] of 16 channels; one group; two groups);
# 36 / 68 / 132 / 20 input channels: a padded last fragment; 16 / 8 input channels: ONE fragment (one or two passes per strip: the staging
# phases collapse into the last pass); W = 64: both image borders in one band, 128: two border bands, 192: an interior band as well;
# H = rows of one strip (first AND last strip: both row adjoints in one block), two and three strips
SHAPES = [(36, 16, 16, 64), (36, 16, 24, 128), (32, 16, 8, 192), (16, 16, 8, 64), (68, 32, 8, 128), (68, 32, 12, 64), (20, 32, 4, 64),
          (8, 32, 8, 64), (132, 64, 8, 64), (16, 64, 4, 64), (36, 64, 4, 128)]


@pytest.mark.parametrize("shape", SHAPES)
def test_x6_backward_data_against_fp32_tilings(M, shape):
    """conv -> BN+act -> 3x3 (under test) -> BN+act -> conv: BN-backward on load, zero padding, the reflection adjoint in rows (first /
    last strip) and columns (first / last band), the padded last fragment, the fold (LeakyReLU', BN-backward sums of x, ga) — every
    gradient of the plan against the round-2 rectangular tiling; several strips per block, also a count that does not divide the strips."""
    cin, cout, H, W = shape
    n, seed = 2, 91
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) in (1, 2)
    for T in (1, 2, 3, 16):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, x6_tune(cout, T)))
        got = _run_plan(plan, P, seed, n, z, dout)
        assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) == 3, "the bf16x6 backward-data kernel did not run"
        assert relerr(got[0], ref[0]) == 0.0, ("out", T)            # the forward pass is untouched
        for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
            assert relerr(a, b) < 2e-5, (name, T)


# strip-resident form (csrc/conv_bwd_x6s.hip, bit 16 of the tiling): 36 -> 16 and 32 -> 16; one strip (first AND last: both row adjoints in one
# sweep), two, four strips; one band (both column adjoints), two, three bands
S_SHAPES = [(36, 16, 8, 64), (36, 16, 16, 64), (36, 16, 32, 128), (36, 16, 16, 192), (32, 16, 16, 64), (32, 16, 24, 128)]


@pytest.mark.parametrize("shape", S_SHAPES)
@pytest.mark.parametrize("act", [True, False])
def test_x6_strip_resident_form_against_fp32_tilings(M, shape, act):
    """Weights resident in registers, one sweep per strip, the last 4 input channels as ONE (tap row, channel) operand whose columns are
    shifted onto their output rows: every gradient of the plan against the round-2 rectangular tiling; the layer's input with and without
    an activation behind its BatchNorm (the net's 36-channel concat tensor has none); strips per block that do not divide the strips."""
    cin, cout, H, W = shape
    n, seed = 2, 95
    P = M.Program()
    zin = P.tensor(cin, H, W)
    x = P.tensor(cin, H, W); P.conv(zin, x, 1, 1); P.set_bn(x, act=act)
    y = P.tensor(cout, H, W); P.conv(x, y, 3, 1); P.set_bn(y, act=True)
    out = P.tensor(2, H, W); P.conv(y, out, 1, 1)
    plan = P.compile(zin, out, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) in (1, 2)
    for T in (1, 2, 3, 16):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, x6_tune(cout, T) | 1 << 16))
        got = _run_plan(plan, P, seed, n, z, dout)
        assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) == 3, "the bf16x6 backward-data kernel did not run"
        assert relerr(got[0], ref[0]) == 0.0, ("out", T)
        for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
            assert relerr(a, b) < 2e-5, (name, T)


def test_x6_strip_resident_form_at_full_size(M):
    """36 -> 16 at 256 x 512 (the band / strip counts of the BASELINE 256^2 and 512^2 configs): 32 strips of 8 rows, 8 bands, the block
    counts the autotuner picks from (8 and 32 strips per block) against the fp32 tiling."""
    cin, cout, H, W, n, seed = 36, 16, 256, 512, 2, 99
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    for T in (8, 32):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, x6_tune(cout, T) | 1 << 16))
        got = _run_plan(plan, P, seed, n, z, dout)
        assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) == 3
        for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
            assert relerr(a, b) < 2e-5, (name, T)


def test_x6_strip_resident_form_refuses_other_shapes(M):
    """68 -> 32 with bit 16 set: the tiling is not valid for the shape (-3), the plan runs that launch on the generic kernel."""
    P, plan, zin, out = _conv_bn_plan(M, 68, 32, 8, 64, 1)
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, x6_tune(32, 1) | 1 << 16))
    z = dev(O.normal_fill(3, 2, 2, 0, 0, 68 * 8 * 64).reshape(68, 8, 64)); dout = dev(O.normal_fill(3, 2, 3, 0, 0, 2 * 8 * 64).reshape(1, 2, 8, 64))
    _run_plan(plan, P, 3, 1, z, dout)
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) == 0


def test_x6_backward_data_without_a_weight_draw_falls_back(M):
    """w = mu (sample_weights = 0, the siblings' path): no weight-piece scratch is handed over, the tiling stays set and the layer runs on
    its fp32 default — same gradients."""
    cin, cout, H, W = 36, 16, 16, 64
    n, seed = 2, 93
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    mu = dev(0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi)); rho = dev(-3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi))
    bn = torch.ones(max(P.n_bn, 1), device="cuda")

    def run():
        plan.forward(mu, rho, bn, z, seed, 3, 0, n, sample_weights=False)
        dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
        plan.backward(mu, rho, bn, z, seed, 3, 0, n, dout, dmu, drho, dbn, sample_weights=False)
        return host(dmu), host(dbn)
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = run()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, x6_tune(cout, 2)))
    got = run()
    assert lib.mfvi_plan_last_kernel(plan.handle, 1, 1) in (1, 2), "expected an fp32 kernel"
    for a, b in zip(got, ref):
        assert relerr(a, b) < 2e-5


@pytest.mark.parametrize("kw", [dict(H=32, W=64, input_depth=8, n_out=2, nd=(16,), nu=(16,), ns=(4,)),
                                dict(H=16, W=128, input_depth=4, n_out=1, nd=(16,), nu=(32,), ns=(4,)),
                                dict(H=64, W=64, input_depth=8, n_out=2, nd=(16, 32), nu=(16, 64), ns=(4, 4))])
def test_x6_backward_data_small_nets_against_oracle(M, kw):
    """Hour-glass nets whose top scale is 64 / 128 wide, every 3x3 stride-1 layer the kernel serves switched to it: all gradients
    (d mu, d rho, d BN, dz) against the oracle's tape."""
    net = O.make_net(**kw)
    seed, step, k0, n = 73, 2, 1, 2
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    conv, bn, n_vi, n_bnp = O.net_table(net)
    plan = P.compile(zin, out_id, max_samples=n)
    lib = M._lib.lib()
    cand = []
    for i, o in enumerate(P.ops):
        if o["type"] == 1 and o["ksize"] == 3 and o["stride"] == 1 and P.tensors[o["out"]]["C"] in (16, 32, 64) and P.tensors[o["out"]]["W"] % 64 == 0:
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, i, 1, x6_tune(P.tensors[o["out"]]["C"], 2)))
            cand.append(i)
    assert cand
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * net.H * net.W)).reshape(net.input_depth, net.H, net.W)
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n)
    dout = O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    dz = torch.empty((n,) + z.shape, device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dev(dout), dmu, drho, dbn, dz=dz)
    assert any(lib.mfvi_plan_last_kernel(plan.handle, i, 1) == 3 for i in cand), "no layer ran on the bf16x6 backward-data kernel"
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
