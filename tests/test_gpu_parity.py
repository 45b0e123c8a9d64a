"""GPU parity tests: every kernel of libmfvi_hip is driven through the C ABI (ctypes) and compared
with the CPU oracle on the same seeded inputs, plus the committed golden vectors of the reference.
Run on an MI355X with `pytest -m gpu`."""
import os

import numpy as np
import pytest

from conftest import note_margin as _note

from oracle import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    _v = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    _note(_v, 'relerr')
    return _v


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available(), "these tests need the GPU"
    M_._lib.lib()          # fails loudly if the HIP library is missing
    return M_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------------------------------
def test_rng_bit_exact(M):
    """The device RNG (csrc/common.h) and the oracle's independent restatement agree bit for bit."""
    L = M._lib
    for (seed, dom, stream, sample, step, n) in [(1, 0, 0, 0, 0, 1024), (0xDEADBEEFCAFE, 0, 51, 7, 123456, 4099),
                                                 (42, 1, 0, 0, 3, 100001), (7, 2, 1, 0, 0, 5)]:
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        L.check(L.lib().mfvi_normal_fill(seed, dom, stream, sample, step, n, 0.0, 1.0, L.ptr(out), L.stream_ptr()))
        ref = O.normal_fill(seed, dom, stream, sample, step, n)
        assert np.array_equal(host(out).view(np.uint32), ref.view(np.uint32)), (seed, dom, stream)
    out = torch.empty(5003, dtype=torch.float32, device="cuda")
    L.check(L.lib().mfvi_uniform_fill(9, 2, 1, 4, 5003, 1.0, L.ptr(out), L.stream_ptr()))
    assert np.array_equal(host(out), O.uniform_fill(9, 2, 1, 4, 5003))
    z0 = dev(O.uniform_fill(3, 0, 0, 0, 777)); z = torch.empty_like(z0)
    L.check(L.lib().mfvi_perturb_input(L.ptr(z0), 3, 11, 777, 0.1, L.ptr(z), L.stream_ptr()))
    assert relerr(host(z), host(z0) + 0.1 * O.normal_fill(3, 1, 0, 0, 11, 777)) < 1e-6


# --------------------------------------------------------------------------------------------------
CONV_CASES = [
    # cin, cout, k, stride, H, W
    (16, 4, 1, 1, 12, 12), (16, 16, 3, 2, 16, 16), (16, 16, 3, 1, 10, 14), (36, 16, 3, 1, 8, 8), (8, 32, 3, 2, 12, 20),
    (32, 2, 1, 1, 6, 6), (132, 8, 3, 1, 8, 8), (5, 3, 3, 1, 33, 47), (7, 21, 3, 2, 18, 70), (3, 19, 1, 1, 40, 36),
    (128, 128, 3, 1, 2, 2), (16, 16, 3, 1, 64, 64), (12, 20, 3, 2, 2, 2),
    # several rectangular tiles with a partial last tile column / row: fused fold by the producer waves, border patches in every position
    (36, 16, 3, 1, 40, 80), (68, 32, 3, 1, 24, 96), (20, 16, 3, 1, 36, 44),
    # stride 2 with Wo % 4 == 0 (phase-decomposed backward-data): tiny, ragged and multi-tile domains
    (16, 16, 3, 2, 8, 8), (32, 32, 3, 2, 8, 24), (16, 24, 3, 2, 72, 40), (64, 16, 3, 2, 24, 136),
    # 5x5 filters of the inpainting variant (bayesian_optimization.py:2970-2998): reflection pad 2, stride 1 and 2
    (16, 16, 5, 2, 20, 20), (8, 12, 5, 1, 12, 16), (32, 16, 5, 1, 8, 8), (5, 7, 5, 2, 9, 13), (16, 32, 5, 2, 3, 3),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_single_conv_fwd_bwd(M, case):
    cin, cout, k, stride, H, W = case
    seed, step, k0, n = 1000 + cin + cout, 5, 3, 2
    P = M.Program()
    zin = P.tensor(cin, H, W)
    out = P.tensor(cout, *P.conv_out_hw(zin, k, stride))
    P.conv(zin, out, k, stride)
    plan = P.compile(zin, out, max_samples=n)
    nw = cout * cin * k * k
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, nw + cout); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, nw + cout)
    x = O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W)
    d_mu, d_rho, d_x = dev(mu), dev(rho), dev(x)
    bn = torch.zeros(1, device="cuda")
    y = plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, n)
    dy = O.normal_fill(seed, 2, 3, 0, 0, n * y[0].numel()).reshape(tuple(y.shape))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros(1, device="cuda")
    dz = torch.empty((n, cin, H, W), device="cuda")
    plan.backward(d_mu, d_rho, bn, d_x, seed, step, k0, n, dev(dy), dmu, drho, dbn, dz=dz)
    yh, dzh = host(y), host(dz)
    sig = 1 / (1 + np.exp(-rho.astype(np.float64)))
    ref_dmu = np.zeros(nw + cout); ref_drho = np.zeros(nw + cout)
    for i in range(n):
        ew = O.eps(seed, step, k0 + i, 0, 0, nw); eb = O.eps(seed, step, k0 + i, 0, 1, cout)
        w = O.reparam(mu[:nw], rho[:nw], ew).reshape(cout, cin, k, k); b = O.reparam(mu[nw:], rho[nw:], eb)
        assert relerr(yh[i], O.conv_fwd(x, w, b, stride)) < 2e-6, ("fwd", i)
        dx, dw, db = O.conv_bwd(x, w, stride, dy[i])
        assert relerr(dzh[i], dx) < 5e-6, ("dx", i)
        ref_dmu[:nw] += dw.ravel(); ref_dmu[nw:] += db
        ref_drho[:nw] += dw.ravel() * ew * sig[:nw]; ref_drho[nw:] += db * eb * sig[nw:]
    assert relerr(host(dmu), ref_dmu) < 2e-5
    assert relerr(host(drho), ref_drho) < 2e-5
    # eval branch: w = mu, no rho gradient
    y0 = plan.forward(d_mu, d_rho, bn, d_x, seed, step, k0, 1, sample_weights=False)
    assert relerr(host(y0)[0], O.conv_fwd(x, mu[:nw].reshape(cout, cin, k, k), mu[nw:], stride)) < 2e-6


# --------------------------------------------------------------------------------------------------
def _net_params(net, seed):
    mu, rho, bnp = O.init_params(net, seed)
    conv, bn, n_vi, n_bnp = O.net_table(net)
    g = O.normal_fill(seed, 2, 7, 0, 0, n_bnp)
    for c, off in bn:
        bnp[off:off + c] = 1.0 + 0.1 * g[off:off + c]; bnp[off + c:off + 2 * c] = 0.1 * g[off + c:off + 2 * c]
    return mu, rho, bnp


NET_CASES = {
    "one_scale_16x16": dict(H=16, W=16, input_depth=8, n_out=2, nd=(8,), nu=(8,), ns=(4,)),
    "one_scale_20x44": dict(H=20, W=44, input_depth=6, n_out=2, nd=(12,), nu=(8,), ns=(4,)),
    "two_scale_40x72": dict(H=40, W=72, input_depth=8, n_out=1, nd=(8, 16), nu=(8, 16), ns=(4, 4)),
    "three_scale_32": dict(H=32, W=32, input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4)),
    "three_scale_skip2_48x64": dict(H=48, W=64, input_depth=4, n_out=2, nd=(8, 12, 20), nu=(8, 12, 20), ns=(2, 3, 5)),
    # sizes not divisible by 2^n_scales: Concat centre-crops the up-sampled branch (models/common.py:31-41).  36 -> 18 -> 9 -> 5 rows,
    # 44 -> 22 -> 11 -> 6 columns; 35 x 45 is odd from the first scale on
    "crop_three_scale_36x44": dict(H=36, W=44, input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4)),
    "crop_two_scale_35x45": dict(H=35, W=45, input_depth=4, n_out=1, nd=(8, 12), nu=(8, 12), ns=(4, 2)),
}


@pytest.mark.parametrize("name", list(NET_CASES))
def test_small_nets_fwd_bwd(M, name):
    """Whole hour-glass: every fused path (deferred BN/LeakyReLU, stats epilogue, concat+upsample, reflection
    fold, BN backward on load, multi-consumer gradient sum) against the oracle, n MC samples with k0 > 0."""
    kw = NET_CASES[name]
    net = O.make_net(**kw)
    seed, step, k0, n = 77, 9, 2, 3
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, names = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    conv, bn, n_vi, n_bnp = O.net_table(net)
    assert P.n_vi == n_vi and P.n_bn == n_bnp
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
        if i == 0:   # layer-wise: raw conv outputs and BN sums of the first sample
            tids = []
            for s in range(net.n_scales):
                tids += [names[s]["skip"], names[s]["d1"], names[s]["d2"]]
            for lid, tid in enumerate(tids[:3]):
                y_ref, m_ref, _ = tape.conv_out(lid)
                assert relerr(host(plan.read_tensor(tid, 0, 0)).reshape(y_ref.shape), y_ref) < 1e-5, ("layer", lid)
                st = host(plan.read_tensor(tid, 0, 2)); hw = y_ref[0].size
                assert np.abs(st[:, 0] / hw - m_ref).max() < 1e-5 * np.abs(y_ref).max(), ("bn mean", lid)
        assert relerr(oh[i], ref) < 2e-5, ("out", i)
        a, b, c_, dzr = tape.backward(dout[i], n_vi, n_bnp, want_dz=True)
        r_dmu += a; r_drho += b; r_dbn += c_
        assert relerr(host(dz)[i], dzr) < 2e-4, ("dz", i)
        tape.free()
    assert relerr(host(dmu), r_dmu) < 2e-4
    assert relerr(host(drho), r_drho) < 2e-4
    assert relerr(host(dbn), r_dbn) < 2e-4
    # gradients accumulate (+=) across calls
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, k0, n, dev(dout), dmu, drho, dbn)
    assert relerr(host(dmu), 2 * r_dmu) < 2e-4


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name,size", [("full_den_64_k1", 64), ("full_den_128_k1", 128)])
def test_full_net_against_reference_golden(M, golden_dir, name, size):
    """The 26-layer den net vs golden vectors produced by the reference's own modules (fp32 and float64 runs),
    tolerance 1e-4 relative on the output image and the ELBO as BASELINE.json's north_star states."""
    g = _load(golden_dir, name)
    net = O.make_net(size, size)
    seed, step = int(g["seed"]), int(g["step"])
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, _ = M.skip_program(size, size)
    plan = P.compile(zin, out_id, max_samples=1)
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, 16 * size * size)).reshape(16, size, size)
    tgt = O.noisy(O.phantom(size, size, seed), 0.1, seed)
    d_mu, d_rho, d_bn, d_z, d_t = dev(mu), dev(rho), dev(bnp), dev(z), dev(tgt)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, 0, 1)
    assert relerr(host(out), g["out"]) < 1e-4            # fp32 reference
    assert relerr(host(out), g["out_f64"]) < 1e-4        # float64 run of the reference
    L = M._lib
    nll = torch.zeros(1, dtype=torch.float64, device="cuda"); klv = torch.zeros(1, dtype=torch.float64, device="cuda")
    dout = torch.empty_like(out)
    L.check(L.lib().mfvi_gaussian_nll(L.ptr(out), L.ptr(d_t), 1, size, size, 1, 1.0, L.ptr(dout), L.ptr(nll), L.stream_ptr()))
    L.check(L.lib().mfvi_kl(L.ptr(d_mu), L.ptr(d_rho), P.n_vi, 0.0, float(g["prior_sigma"]), L.ptr(klv), L.stream_ptr()))
    temp = float(g["temp"])
    elbo = float(nll) + temp * float(klv)
    assert abs(float(nll) - float(g["nll"])) < 1e-4 * abs(float(g["nll"]))
    assert abs(float(klv) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    assert abs(elbo - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, 0, 1, dout, dmu, drho, dbn)
    L.check(L.lib().mfvi_kl_backward(L.ptr(d_mu), L.ptr(d_rho), P.n_vi, 0.0, float(g["prior_sigma"]), temp, L.ptr(dmu), L.ptr(drho), L.stream_ptr()))
    gm, gr = host(dmu), host(drho)
    st = max(1, gm.size // 4096)
    # Gradients of the full-depth net.  LeakyReLU has a kink: a BN output within fp32 rounding of 0 (a handful of the
    # ~1e6 activations) takes slope 1 on one side and 0.2 on the other, and train-mode BN backward spreads that single
    # pixel over its whole channel (scripts/diag_grad.py shows exactly this: 1 element of 262144 off by a factor 5.000).
    # The fp32 REFERENCE has the same noise against its own float64 run (printed below).  So: relative L2 error tight,
    # max-norm loose.
    def rel2(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        _v = float(np.linalg.norm(a - b) / np.linalg.norm(b))
        _note(_v, 'rel-L2')
        return _v
    gms, grs = gm[::st][:4096], gr[::st][:4096]
    print("full-net grad err vs f64 reference, max-norm: dmu %.2e drho %.2e dbn %.2e | L2: dmu %.2e drho %.2e | fp32 reference itself: max %.2e L2 %.2e" % (
        relerr(gms, g["dmu_s_f64"]), relerr(grs, g["drho_s_f64"]), relerr(host(dbn), g["dbn_f64"]), rel2(gms, g["dmu_s_f64"]),
        rel2(grs, g["drho_s_f64"]), relerr(g["dmu_s"], g["dmu_s_f64"]), rel2(g["dmu_s"], g["dmu_s_f64"])))
    # at 64x64 the deepest BatchNorm normalises over 2x2 pixels: the net is numerically degenerate there
    # measured (profiles/r04_parity_margins.txt): 128^2 L2 3.9e-5, max-norm 4.2e-5 (the fp32 reference itself: 4.8e-4 / 5.2e-4); 64^2 L2 4.9e-4,
    # max-norm 5.0e-4.  Tolerance = 3x the measurement at 64^2, 2x the reference's own fp32 noise at 128^2 (one flipped kink costs that much)
    l2_tol = 1.5e-3 if size == 64 else 1e-3
    assert rel2(gms, g["dmu_s_f64"]) < l2_tol and rel2(grs, g["drho_s_f64"]) < l2_tol and rel2(host(dbn), g["dbn_f64"]) < 2 * l2_tol
    assert relerr(gms, g["dmu_s_f64"]) < l2_tol and relerr(grs, g["drho_s_f64"]) < l2_tol
    # RNG-free anchor
    out_eval = plan.forward(d_mu, d_rho, d_bn, d_z, seed, 0, 0, 1, sample_weights=False)
    assert relerr(host(out_eval)[0], g["out_eval"]) < 1e-4


@pytest.mark.parametrize("name,task", [("small_den_k2", 0), ("small_sr_k1", 1), ("small_ct_k1", 2), ("crop_den_36x44_k1", 0)])
def test_small_golden_elbo_grad(M, golden_dir, name, task):
    """den / SR / CT losses on small nets vs the reference goldens (K-sample loop, loss averaged over K)."""
    from test_oracle_golden import NETS
    g = _load(golden_dir, name)
    kw, _ = NETS[name]
    net = O.make_net(**kw)
    seed, K, step = int(g["seed"]), int(g["K"]), int(g["step"])
    mu, rho, bnp = _net_params(net, seed)
    P, zin, out_id, _ = M.skip_program(kw["H"], kw["W"], kw["input_depth"], kw["n_out"], kw["nd"], kw["nu"], kw["ns"])
    plan = P.compile(zin, out_id, max_samples=K)
    H, W = net.H, net.W
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, net.input_depth * H * W)).reshape(net.input_depth, H, W)
    img = O.phantom(H, W, seed); tgt = O.noisy(img, 0.1, seed)
    L = M._lib
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, 0, K)
    assert relerr(host(out), g["out"]) < 1e-4
    lossv = torch.zeros(1, dtype=torch.float64, device="cuda"); dout = torch.empty_like(out)
    if task == 0:
        d_t = dev(tgt)
        L.check(L.lib().mfvi_gaussian_nll(L.ptr(out), L.ptr(d_t), K, H, W, 1, 1.0 / K, L.ptr(dout), L.ptr(lossv), L.stream_ptr()))
    elif task == 1:
        d_t = dev(tgt[::4, ::4])
        L.check(L.lib().mfvi_gaussian_nll(L.ptr(out), L.ptr(d_t), K, H, W, 4, 1.0 / K, L.ptr(dout), L.ptr(lossv), L.stream_ptr()))
    else:
        theta = np.arange(0, 180., 4., dtype=np.float32)
        sino = dev(g["sino_target"]); scratch = torch.empty(K * theta.size * W, device="cuda"); d_th = dev(theta)
        L.check(L.lib().mfvi_radon_mse(L.ptr(out), L.ptr(sino), L.ptr(d_th), K, H, W, theta.size, 1.0 / K, L.ptr(scratch),
                                       L.ptr(dout), L.ptr(lossv), L.stream_ptr()))
    nll = float(lossv) / K
    assert abs(nll - float(g["nll"])) < 1e-4 * abs(float(g["nll"]))
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, 0, K, dout, dmu, drho, dbn)
    L.check(L.lib().mfvi_kl_backward(L.ptr(d_mu), L.ptr(d_rho), P.n_vi, 0.0, float(g["prior_sigma"]), float(g["temp"]),
                                     L.ptr(dmu), L.ptr(drho), L.stream_ptr()))
    assert relerr(host(dmu), g["dmu"]) < 3e-4 and relerr(host(drho), g["drho"]) < 3e-4 and relerr(host(dbn), g["dbn"]) < 3e-4


# --------------------------------------------------------------------------------------------------
def test_losses_kl_adam_metrics_radon(M):
    L = M._lib
    # KL + gradient
    n = 100003
    mu = 0.1 * O.normal_fill(9, 2, 0, 0, 0, n); rho = -3 + 0.5 * O.normal_fill(9, 2, 1, 0, 0, n)
    s0 = np.float32(1.0109e-6)
    d_mu, d_rho = dev(mu), dev(rho)
    klv = torch.zeros(1, dtype=torch.float64, device="cuda")
    L.check(L.lib().mfvi_kl(L.ptr(d_mu), L.ptr(d_rho), n, 0.0, float(s0), L.ptr(klv), L.stream_ptr()))
    ref, rdm, rdr = O.kl(mu, rho, s0, scale=0.37, want_grad=True)
    assert abs(float(klv) - ref) < 1e-6 * abs(ref)
    dmu = torch.zeros(n, device="cuda"); drho = torch.zeros(n, device="cuda")
    L.check(L.lib().mfvi_kl_backward(L.ptr(d_mu), L.ptr(d_rho), n, 0.0, float(s0), 0.37, L.ptr(dmu), L.ptr(drho), L.stream_ptr()))
    assert relerr(host(dmu), rdm) < 1e-5 and relerr(host(drho), rdr) < 1e-5
    # gaussian_nll incl. clamped entries, den and SR
    H, W, K = 32, 48, 3
    o = O.normal_fill(8, 2, 0, 0, 0, K * 2 * H * W).reshape(K, 2, H, W).copy(); o[0, 1, 0, :4] = [25.0, -25.0, 20.0, -20.0]
    for f in (1, 4):
        t = O.uniform_fill(8, 1, 0, 0, (H // f) * (W // f)).reshape(H // f, W // f)
        acc = torch.zeros(1, dtype=torch.float64, device="cuda"); dout = torch.full((K, 2, H, W), 7.0, device="cuda")
        d_o, d_t = dev(o), dev(t)
        L.check(L.lib().mfvi_gaussian_nll(L.ptr(d_o), L.ptr(d_t), K, H, W, f, 0.5, L.ptr(dout), L.ptr(acc), L.stream_ptr()))
        tot = 0.0; dref = np.zeros_like(o)
        for k in range(K):
            v, dm, ds = O.gaussian_nll(o[k, 0, ::f, ::f], o[k, 1, ::f, ::f], t, scale=0.5, want_grad=True)
            tot += v; dref[k, 0, ::f, ::f] = dm; dref[k, 1, ::f, ::f] = ds
        assert abs(float(acc) - tot) < 1e-5 * abs(tot) and relerr(host(dout), dref) < 1e-5
    # Adam
    p = O.normal_fill(12, 2, 0, 0, 0, 5000).copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    dp, dm_, dv = dev(p), dev(m), dev(v)
    for t_ in range(3):
        g = O.normal_fill(12, 2, 1 + t_, 0, 0, 5000).copy()
        O.adam(p, g, m, v, 1e-3, t_ + 1)
        d_g = dev(g)
        L.check(L.lib().mfvi_adam_step(L.ptr(dp), L.ptr(d_g), L.ptr(dm_), L.ptr(dv), 5000, 1e-3, 0.9, 0.999, 1e-8, t_ + 1, L.stream_ptr()))
    assert relerr(host(dp), p) < 1e-6
    # PSNR / SSIM
    a = O.phantom(48, 40, 3); b = O.noisy(a, 0.1, 3)
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    d_a, d_b = dev(a), dev(b)
    L.check(L.lib().mfvi_sq_err_sum(L.ptr(d_a), L.ptr(d_b), a.size, L.ptr(acc), L.stream_ptr()))
    assert abs(10 * np.log10(1.0 / (float(acc) / a.size)) - O.psnr(a, b)) < 1e-4
    L.check(L.lib().mfvi_ssim_sum(L.ptr(d_a), L.ptr(d_b), 48, 40, L.ptr(acc), L.stream_ptr()))
    assert abs(float(acc) / a.size - O.ssim(a, b)) < 2e-5
    # Radon forward / adjoint, incl. <Ax, y> == <x, A^T y>
    theta = np.arange(0, 180., 4., dtype=np.float32)
    for Hs in (64, 256):
        img = O.phantom(Hs, Hs, 11)
        sino = torch.empty((1, theta.size, Hs), device="cuda")
        d_img, d_th = dev(img), dev(theta)
        L.check(L.lib().mfvi_radon_forward(L.ptr(d_img), L.ptr(d_th), 1, Hs, Hs, theta.size, L.ptr(sino), L.stream_ptr()))
        sref = O.radon_fwd(img, theta)
        assert relerr(host(sino)[0], sref) < 2e-5
        r = O.normal_fill(11, 2, 5, 0, 0, sref.size).reshape(sref.shape)
        adj = torch.empty((1, Hs, Hs), device="cuda")
        d_r = dev(r)
        L.check(L.lib().mfvi_radon_adjoint(L.ptr(d_r), L.ptr(d_th), 1, Hs, Hs, theta.size, L.ptr(adj), L.stream_ptr()))
        assert relerr(host(adj)[0], O.radon_adj(r, theta, Hs, Hs)) < 2e-5
        lhs = float((host(sino)[0].astype(np.float64) * r).sum()); rhs = float((host(adj)[0].astype(np.float64) * img).sum())
        assert abs(lhs - rhs) < 1e-5 * abs(lhs)


def test_argument_errors(M):
    """Shape/argument errors come back as negative status + message, never as a crash."""
    L = M._lib
    P = M.Program(); a = P.tensor(4, 8, 8); b = P.tensor(4, 9, 9); P.conv(a, b, 3, 1)
    with pytest.raises(L.MfviError, match="spatial size"):
        P.compile(a, b, 1)
    P = M.Program(); a = P.tensor(4, 8, 8); b = P.tensor(4, 8, 8); P.conv(a, b, 7, 1)
    with pytest.raises(L.MfviError, match="not supported"):
        P.compile(a, b, 1)
    P = M.Program(); a = P.tensor(4, 8, 8); b = P.tensor(4, 8, 8); P.conv(a, b, 3, 1)
    plan = P.compile(a, b, 2)
    z = torch.zeros(4 * 8 * 8, device="cuda"); mu = torch.zeros(P.n_vi, device="cuda")
    with pytest.raises(L.MfviError, match="n_samples"):
        plan.forward(mu, mu, mu, z, 1, 0, 0, 3)
    with pytest.raises(L.MfviError):
        L.check(L.lib().mfvi_radon_forward(L.ptr(z), L.ptr(z), 1, 8, 16, 4, L.ptr(z), L.stream_ptr()))


# --------------------------------------------------------------------------------------------------
def _conv_bn_plan(M, cin, cout, H, W, n):
    """z -> 1x1 conv -> BN+act -> 3x3 conv (under test) -> BN+act -> 1x1 conv -> out: every fused path of the MFMA kernels is live."""
    P = M.Program()
    zin = P.tensor(cin, H, W)
    x = P.tensor(cin, H, W); P.conv(zin, x, 1, 1); P.set_bn(x, act=True)
    y = P.tensor(cout, H, W); P.conv(x, y, 3, 1); P.set_bn(y, act=True)
    out = P.tensor(2, H, W); P.conv(y, out, 1, 1)
    return P, P.compile(zin, out, n), zin, out


def _run_plan(plan, P, seed, n, z, dout):
    mu = dev(0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi)); rho = dev(-3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi))
    bn = torch.ones(max(P.n_bn, 1), device="cuda")
    o = plan.forward(mu, rho, bn, z, seed, 3, 0, n)
    dmu = torch.zeros_like(mu); drho = torch.zeros_like(rho); dbn = torch.zeros_like(bn)
    dz = torch.empty((n,) + tuple(z.shape), device="cuda")
    plan.backward(mu, rho, bn, z, seed, 3, 0, n, dout, dmu, drho, dbn, dz=dz)
    return host(o), host(dmu), host(drho), host(dz)


@pytest.mark.parametrize("shape", [(36, 16, 64, 64), (68, 32, 32, 32), (132, 64, 16, 16), (36, 16, 40, 80), (36, 32, 20, 64)])
def test_tilings_do_not_change_results(M, shape):
    """Every tiling the autotuner may pick (rectangular / FLAT tiles, fragments, tiles per block, backward-weight variants)
    computes the same numbers: forward bit-identical, gradients to summation-order rounding."""
    cin, cout, H, W = shape
    n, seed = 2, 77
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    ref = _run_plan(plan, P, seed, n, z, dout)
    enc = lambda a, b, c: a | b << 8 | c << 16
    fwd_bwd = [(1, 8, 1), (1, 16, 1), (2, 8, 2), (3, 8, 1), (1, 8 | 128, 1), (1, 16 | 128, 1), (2, 8 | 128, 2), (1, 4 | 128, 1), (2, 2 | 128, 1), (4, 4 | 128, 1)]
    bww = [(1, 4, 1), (1, 8, 2), (2, 9, 1), (3, 9, 1), (3, 4, 2), (2, 10, 1), (2, 10, 4)]      # w = 10: fragment-split variant (full-width tiles)
    # backward-data with the layer's last 4 input channels on the 4x4x1 matrix instruction (tile-height bit 64): 36 = 2*16+4, 68 = 4*16+4 ...
    rem = [(mf, 8 | 64, T) for mf, T in ((1, 1), (2, 2), (4, 1)) if (cin - 4) % (16 * mf) == 0]
    assert len(rem) >= 2
    for which, cands in ((0, fwd_bwd), (1, fwd_bwd + rem), (2, bww)):
        for cand in cands:
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, enc(*cand)))
            got = _run_plan(plan, P, seed, n, z, dout)
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, 0))
            assert relerr(got[0], ref[0]) < 1e-6, ("out", which, cand)          # BN statistics are summed in a different order
            for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
                assert relerr(a, b) < 2e-5, (name, which, cand)


@pytest.mark.parametrize("shape", [(16, 16, 16, 32), (20, 16, 24, 32), (32, 32, 16, 64), (48, 16, 16, 32), (52, 32, 12, 64), (96, 16, 8, 32), (100, 16, 8, 32)])
def test_split_backward_weight_channel_groups(M, shape):
    """Fragment-split backward-weight variant (w = 10) over every input-channel grouping it serves: one or two 16-channel tiles per block,
    with and without the 4-channel remainder, several groups (100 = 32 + 32 + 36), partial last row tile (H = 12), bias on group 0 only."""
    cin, cout, H, W = shape
    n, seed = 2, 91
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 2, 1 | 4 << 8 | 1 << 16))       # plain 4-wave variant, one input tile per block
    ref = _run_plan(plan, P, seed, n, z, dout)
    for tgt in (1, 3):
        M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 2, 2 | 10 << 8 | tgt << 16))
        got = _run_plan(plan, P, seed, n, z, dout)
        assert np.array_equal(got[0], ref[0])
        for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
            assert relerr(a, b) < 2e-5, (name, tgt)


@pytest.mark.parametrize("shape", [(132, 128, 8, 8), (128, 128, 16, 16), (68, 64, 8, 16)])
def test_small_map_tilings_do_not_change_results(M, shape):
    """The 8x8 / 16x16 maps at the bottom of the hour-glass: FLAT tiles of 2 / 4 / 8 rows-worth of pixels with 32-channel stages in both
    passes (backward-data with the fold in its epilogue included, two map rows per 16-pixel fragment at width 8), against the
    rectangular 8-row tiling."""
    cin, cout, H, W = shape
    n, seed = 3, 79
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    lib = M._lib.lib()
    enc = lambda a, b, c: a | b << 8 | c << 16
    M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 0, enc(1, 8, 1))); M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, 1, enc(1, 8, 1)))
    ref = _run_plan(plan, P, seed, n, z, dout)
    tried = 0
    cands = [(mf, th | 128, 1) for th in (2, 4, 8) for mf in (1, 2, 4)] + [(2, 8, 1)]
    for which in (0, 1):
        for cand in cands:
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, enc(*cand)))
            try:
                got = _run_plan(plan, P, seed, n, z, dout)
            except M._lib.MfviError:          # a tiling this shape does not admit (-3)
                continue
            finally:
                M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, enc(1, 8, 1)))
            tried += 1
            assert relerr(got[0], ref[0]) < 1e-6, ("out", which, cand)
            for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
                assert relerr(a, b) < 2e-5, (name, which, cand)
    assert tried >= 12


@pytest.mark.parametrize("shape", [(16, 32, 64, 64), (64, 128, 32, 32)])
def test_stride2_tilings_do_not_change_results(M, shape):
    """Stride-2 layers: the phase-decomposed backward-data on rectangular tiles (every fragment count / tile height / tiles per block)
    and the zero-stuffed formulation on FLAT tiles agree with the default tiling, the forward tilings too.  (The default tiling itself
    is pinned against the oracle by test_single_conv_fwd_bwd's stride-2 cases and the small-net / golden tests.)"""
    cin, cout, H, W = shape
    n, seed = 2, 78
    P = M.Program()
    zin = P.tensor(cin, H, W)
    x = P.tensor(cin, H, W); P.conv(zin, x, 1, 1); P.set_bn(x, act=True)
    y = P.tensor(cout, H // 2, W // 2); P.conv(x, y, 3, 2); P.set_bn(y, act=True)
    out = P.tensor(2, H // 2, W // 2); P.conv(y, out, 1, 1)
    plan = P.compile(zin, out, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * (H // 2) * (W // 2)).reshape(n, 2, H // 2, W // 2))
    lib = M._lib.lib()
    ref = _run_plan(plan, P, seed, n, z, dout)
    enc = lambda a, b, c: a | b << 8 | c << 16
    tried = 0
    for which, cands in ((0, [(1, 8, 1), (2, 8, 1), (1, 8, 2), (1, 8 | 128, 1), (2, 4 | 128, 1)]),
                         (1, [(1, 8, 1), (2, 8, 1), (3, 8, 1), (1, 16, 1), (2, 16, 1), (1, 8, 2), (1, 8 | 128, 1), (2, 8 | 128, 1), (1, 16 | 128, 1), (2, 4 | 128, 1)])):
        for cand in cands:
            M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, enc(*cand)))
            try:
                got = _run_plan(plan, P, seed, n, z, dout)
            except M._lib.MfviError:          # a tiling this shape does not admit (-3): the launcher refuses it, nothing ran
                continue
            finally:
                M._lib.check(lib.mfvi_plan_set_tune(plan.handle, 1, which, 0))
            tried += 1
            assert relerr(got[0], ref[0]) < 1e-6, ("out", which, cand)
            for a, b, name in zip(got[1:], ref[1:], ("dmu", "drho", "dz")):
                assert relerr(a, b) < 2e-5, (name, which, cand)
    assert tried >= 8


def test_autotune_cache_and_reproducible_gradients(M, tmp_path):
    cin, cout, H, W, n, seed = 36, 16, 32, 32, 4, 5
    P, plan, zin, out = _conv_bn_plan(M, cin, cout, H, W, n)
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, cin * H * W).reshape(cin, H, W))
    dout = dev(O.normal_fill(seed, 2, 3, 0, 0, n * 2 * H * W).reshape(n, 2, H, W))
    before = _run_plan(plan, P, seed, n, z, dout)
    mu = dev(0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi)); rho = dev(-3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi))
    bn = torch.ones(P.n_bn, device="cuda")
    cache = str(tmp_path / "tunes.json")
    plan.autotune(mu, rho, bn, z, n, cache=cache)
    tuned = plan.tunes()
    assert os.path.exists(cache) and all(t is not None for t in tuned[1])       # the 3x3 layer got a tiling for all three passes
    after = _run_plan(plan, P, seed, n, z, dout)
    again = _run_plan(plan, P, seed, n, z, dout)
    for a, b in zip(after, before):
        assert relerr(a, b) < 2e-5
    # weight gradients come from slabs + grad_finalize (no atomics); only the fp64 BN-statistic atomics can reorder
    assert relerr(again[1], after[1]) < 1e-6 and relerr(again[2], after[2]) < 1e-6
    P2, plan2, _, _ = _conv_bn_plan(M, cin, cout, H, W, n)
    plan2.autotune(mu, rho, bn, z, n, cache=cache)                              # served from the cache
    assert plan2.tunes() == tuned
