"""GPU parity of the non-Bayesian siblings (DIP, MC dropout, SGLD — bayesian_optimization.py:1064-1237,
1447-1860) on the same HIP layer program: Dropout2d folded into the deferred BatchNorm, one-channel MSE, AdamW with decoupled
weight decay, SGLD parameter noise.  Checked against the CPU oracle and the golden run of the reference's own skip() net."""
import os

import numpy as np
import pytest

from conftest import note_margin as _note

from oracle import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SIB_NET = dict(input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4))


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    _v = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    _note(_v, 'relerr')
    return _v


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available(), "these tests need the GPU"
    M_._lib.lib()
    return M_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _params(net, seed):
    mu, _, bnp = O.init_params(net, seed)
    _, bn, _, n_bnp = O.net_table(net)
    g = O.normal_fill(seed, 2, 7, 0, 0, n_bnp)
    for c, off in bn:
        bnp[off:off + c] = 1.0 + 0.1 * g[off:off + c]; bnp[off + c:off + 2 * c] = 0.1 * g[off + c:off + 2 * c]
    return mu, bnp


def _engine(method, g, K=1):
    from mfvi_dip_mia_amd.engine import SiblingEngine
    H, W, seed = int(g["H"]), int(g["W"]), int(g["seed"])
    kw = dict(nd=SIB_NET["nd"], nu=SIB_NET["nu"], ns=SIB_NET["ns"])
    eng = SiblingEngine(H, W, method=method, task="den", K=K, input_depth=SIB_NET["input_depth"], lr=float(g["lr"]), seed=seed,
                        weight_decay=float(g[method + "_wd"]), dropout_p=float(g[method + "_p"]) or 0.3, gamma=float(g[method + "_gamma"]),
                        net_kwargs=kw, autotune=False)
    p = float(g[method + "_p"])
    net = O.make_net(H, W, drop_down=p, drop_up=p, **SIB_NET)
    mu, bnp = _params(net, seed)
    eng.mu.copy_(dev(mu)); eng.bn.copy_(dev(bnp))
    eng.set_target(dev(O.noisy(O.phantom(H, W, seed), 0.1, seed)))
    return eng, net, mu, bnp


@pytest.mark.parametrize("method", ["dip", "mcd", "sgld"])
def test_sibling_trajectory_matches_reference_golden(M, golden_dir, method):
    g = np.load(os.path.join(golden_dir, "siblings.npz"))
    eng, net, mu, bnp = _engine(method, g)
    z0 = host(eng.z0)
    assert np.array_equal(z0.ravel(), (0.1 * O.uniform_fill(int(g["seed"]), 0, 0, 0, z0.size)).astype(np.float32))
    losses = []
    for it in range(int(g["steps"])):
        eng.step()
        losses.append(eng.losses()[0])
        if it == 0:
            assert relerr(host(eng.out)[0], g[method + "_out0"]) < 1e-4          # the image: north_star's 1e-4 relative
            assert relerr(host(eng.dmu), g[method + "_dmu0"]) < 2e-4 and relerr(host(eng.dbn), g[method + "_dbn0"]) < 2e-4
            assert float(host(eng.drho).max()) == 0.0 and float(host(eng.drho).min()) == 0.0
    gl = g[method + "_loss"]
    assert np.abs(np.array(losses) - gl).max() < 2e-4 * np.abs(gl).max(), (losses, gl)
    bound = 2 * int(g["steps"]) * float(g["lr"])      # sign flips of noise-floor gradients: up to lr per step in either direction
    assert np.abs(host(eng.mu) - g[method + "_mu"]).max() < bound and np.abs(host(eng.mu) - g[method + "_mu"]).mean() < 3e-6
    assert np.abs(host(eng.bn) - g[method + "_bn"]).max() < bound
    assert float(host(eng.rho).max()) == 0.0                                     # RHO does not exist for these methods


@pytest.mark.parametrize("method,loss", [("dip", "mse0"), ("mcd", "gnll")])
def test_sibling_gradient_matches_oracle_k3(M, golden_dir, method, loss):
    """K = 3 forwards (for MC dropout: three different mask draws, keyed by the global sample index) against the oracle."""
    g = np.load(os.path.join(golden_dir, "siblings.npz"))
    eng, net, mu, bnp = _engine(method, g, K=3)
    eng.grad_only(step=5, perturb=True, with_kl=False)
    seed = int(g["seed"])
    z0 = host(eng.z0)
    z = z0 + 0.1 * O.normal_fill(seed, 1, 0, 0, 5, z0.size).reshape(z0.shape)
    r = O.sibling_grad(net, mu, bnp, z, host(eng.target), loss=loss, seed=seed, step=5, K=3, want_out=True)
    assert relerr(host(eng.out), r["out"]) < 2e-5
    assert abs(eng.losses()[0] - r["loss"]) < 2e-5 * abs(r["loss"])
    assert relerr(host(eng.dmu), r["dmu"]) < 2e-4 and relerr(host(eng.dbn), r["dbn"]) < 2e-4
    if method == "mcd":
        a, b = host(eng.out)[0], host(eng.out)[1]
        assert np.abs(a - b).max() > 1e-3                                        # different masks per sample


def test_dropout_off_is_identity_and_masks_follow_the_spec(M):
    """mfvi_plan_set_dropout(0) = nn.Dropout2d in eval mode; with dropout on, a dropped channel's BN output is beta."""
    H = W = 16
    P, zin, zout, names = M.skip_program(H, W, 4, 2, nd=(8,), nu=(8,), ns=(4,), drop_down=0.5, drop_up=0.5)
    Q, _, _, _ = M.skip_program(H, W, 4, 2, nd=(8,), nu=(8,), ns=(4,))
    seed = 3
    net = O.make_net(H, W, input_depth=4, n_out=2, nd=(8,), nu=(8,), ns=(4,), drop_down=0.5, drop_up=0.5)
    mu, bnp = _params(net, seed)
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, 4 * H * W)).reshape(4, H, W)
    d_mu, d_bn, d_z = dev(mu), dev(bnp), dev(z)
    rho = torch.zeros_like(d_mu)
    plan, plain = P.compile(zin, zout, 2), Q.compile(zin, zout, 2)
    on = host(plan.forward(d_mu, rho, d_bn, d_z, seed, 4, 1, 2, False))
    for k in range(2):
        ref, tape = O.net_forward(net, mu, np.zeros_like(mu), bnp, z, seed, 4, 1 + k, sample_weights=False)
        tape.free()
        assert relerr(on[k], ref) < 2e-5
    M._lib.check(M._lib.lib().mfvi_plan_set_dropout(plan.handle, 0))
    off = host(plan.forward(d_mu, rho, d_bn, d_z, seed, 4, 1, 2, False))
    assert np.array_equal(off, host(plain.forward(d_mu, rho, d_bn, d_z, seed, 4, 1, 2, False)))
    assert np.abs(on - off).max() > 1e-3


def test_adamw_mse_channel_sgld_noise(M):
    lib, L = M._lib.lib(), M._lib
    sp = L.stream_ptr()
    n = 5000
    p = O.normal_fill(1, 2, 40, 0, 0, n).copy(); g_ = O.normal_fill(1, 2, 41, 0, 0, n).copy()
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    dp, dg, dm, dv = dev(p), dev(g_), dev(m), dev(v)
    for t in (1, 2, 3):
        L.check(lib.mfvi_adamw_step(L.ptr(dp), L.ptr(dg), L.ptr(dm), L.ptr(dv), n, 3e-3, 0.9, 0.999, 1e-8, t, 0.05, sp))
        O.adamw(p, g_, m, v, 3e-3, t, 0.05)
    assert np.abs(host(dp) - p).max() < 2e-6
    # one-channel MSE with and without the SR projection
    out = O.normal_fill(2, 2, 42, 0, 0, 2 * 3 * 16 * 24).reshape(2, 3, 16, 24)
    for f in (1, 4):
        tgt = O.normal_fill(2, 2, 43, 0, 0, (16 // f) * (24 // f)).reshape(16 // f, 24 // f)
        acc = torch.zeros(1, dtype=torch.float64, device="cuda"); dout = torch.full((2, 3, 16, 24), 7.0, device="cuda")
        d_out, d_tgt = dev(out), dev(tgt)                              # keep the tensors alive across the call
        L.check(lib.mfvi_mse_channel(L.ptr(d_out), L.ptr(d_tgt), 2, 3, 16, 24, 1, f, 0.5, L.ptr(dout), L.ptr(acc), sp))
        ref = 0.0; dref = np.zeros_like(out)
        for k in range(2):
            v_, d_ = O.mse(out[k, 1, ::f, ::f], tgt, 0.5, want_grad=True)
            ref += v_; dref[k, 1, ::f, ::f] = d_
        assert abs(float(acc) - ref) < 1e-6 * abs(ref)
        assert np.abs(host(dout) - dref).max() < 1e-7
    # SGLD noise: x += std * N(0,1) from RNG domain 4
    x = O.normal_fill(3, 2, 44, 0, 0, 1001).copy(); dx = dev(x)
    L.check(lib.mfvi_add_normal(L.ptr(dx), 9, 6, 11, 1001, 6e-4, sp))
    assert np.abs(host(dx) - (x + np.float32(6e-4) * O.normal_fill(9, 4, 6, 0, 11, 1001))).max() < 1e-7
    # ranged uniform fill (nn.Conv2d's kaiming-uniform bound)
    u = torch.empty(777, device="cuda")
    L.check(lib.mfvi_uniform_fill_range(9, 20, 0, 0, 777, -0.25, 0.25, L.ptr(u), sp))
    assert np.abs(host(u) - (-0.25 + 0.5 * O.uniform_fill(9, 20, 0, 0, 777))).max() < 1e-7
    assert host(u).min() >= -0.25 and host(u).max() < 0.25


def test_fused_elbo_update_equals_kl_plus_adam(M):
    """mfvi_elbo_update (one launch) == mfvi_kl + mfvi_kl_backward + mfvi_adam_step: same per-element arithmetic up to the
    compiler's fma contraction (a few ulp); the KL sum differs only in its (now fixed) summation order."""
    lib, L = M._lib.lib(), M._lib
    sp = L.stream_ptr()
    n_vi, n_bn = 300_007, 1234
    n = 2 * n_vi + n_bn
    torch.manual_seed(3)
    p0 = torch.cat([0.1 * torch.randn(n_vi), -3 + 0.1 * torch.randn(n_vi), 1 + 0.1 * torch.randn(n_bn)]).cuda()
    g0 = (1e-3 * torch.randn(n)).cuda()
    scratch = torch.zeros(lib.mfvi_elbo_update_scratch_bytes(), dtype=torch.uint8, device="cuda")
    pa, ga, ma, va = p0.clone(), g0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, gb, mb, vb = p0.clone(), g0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    kla = torch.zeros(1, dtype=torch.float64, device="cuda"); klb = torch.zeros_like(kla)
    ps, temp = 1.0000110e-06, 5.6e-7
    for t in (1, 2, 3):
        ga.copy_(g0); gb.copy_(g0)
        L.check(lib.mfvi_elbo_update(L.ptr(pa), L.ptr(ga), L.ptr(ma), L.ptr(va), n_vi, n_bn, 0.0, ps, temp, 1e-3, 0.9, 0.999, 1e-8, t, L.ptr(kla),
                                     L.ptr(scratch), sp))
        L.check(lib.mfvi_kl(L.ptr(pb), L.ptr(pb[n_vi:]), n_vi, 0.0, ps, L.ptr(klb), sp))
        L.check(lib.mfvi_kl_backward(L.ptr(pb), L.ptr(pb[n_vi:]), n_vi, 0.0, ps, temp, L.ptr(gb), L.ptr(gb[n_vi:]), sp))
        L.check(lib.mfvi_adam_step(L.ptr(pb), L.ptr(gb), L.ptr(mb), L.ptr(vb), n, 1e-3, 0.9, 0.999, 1e-8, t, sp))
        close = lambda a, b: float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
        assert close(ga, gb) and close(ma, mb) and close(va, vb), t
        assert float((pa - pb).abs().max()) < 1e-6, t                 # far below one Adam step (lr = 1e-3)
        assert abs(float(kla) - float(klb)) < 1e-10 * abs(float(klb))      # different fp64 summation order
    ref = O.kl(host(p0[:n_vi]), host(p0[n_vi:2 * n_vi]), ps)
    # first-iteration KL of the oracle on the initial parameters
    kl0 = torch.zeros(1, dtype=torch.float64, device="cuda")
    pc = p0.clone(); gc = g0.clone()
    mc, vc = torch.zeros_like(p0), torch.zeros_like(p0)
    L.check(lib.mfvi_elbo_update(L.ptr(pc), L.ptr(gc), L.ptr(mc), L.ptr(vc), n_vi, n_bn, 0.0, ps, temp, 1e-3,
                                 0.9, 0.999, 1e-8, 1, L.ptr(kl0), L.ptr(scratch), sp))
    assert abs(float(kl0) - ref) < 2e-6 * abs(ref)


def test_inp_dip_loss_kernel(M, golden_dir):
    """mfvi_mse_sigmoid_masked vs the reference's torch expression (tests/golden/inp_dip_loss.npz) and the oracle, n = 2 samples."""
    lib, L = M._lib.lib(), M._lib
    g = np.load(os.path.join(golden_dir, "inp_dip_loss.npz"))
    for mc in (1, 3):
        o1 = g["out%d" % mc]; H, W = o1.shape[-2:]
        out = np.stack([o1, o1[::-1].copy()])
        acc = torch.zeros(1, dtype=torch.float64, device="cuda"); dout = torch.full((2, 4, H, W), 7.0, device="cuda")
        d_out, d_img, d_mask = dev(out), dev(g["img%d" % mc]), dev(g["mask%d" % mc])      # keep the tensors alive across the call
        L.check(lib.mfvi_mse_sigmoid_masked(L.ptr(d_out), L.ptr(d_img), L.ptr(d_mask), mc, 2, H, W, 1.0, L.ptr(dout), L.ptr(acc), L.stream_ptr()))
        v1, d1 = O.mse_sigmoid_masked(out[1], g["img%d" % mc], g["mask%d" % mc], 1.0, want_grad=True)
        assert abs(float(acc) - (float(g["loss%d" % mc]) + v1)) < 1e-5 * abs(float(acc))
        assert relerr(host(dout)[0], g["grad%d" % mc]) < 1e-5 and relerr(host(dout)[1], d1) < 1e-5


@pytest.mark.parametrize("task,method,loss", [("sr", "dip", "mse0"), ("sr", "mcd", "gnll"), ("sr", "sgld", "gnll"), ("ct", "dip", "radon"),
                                              ("ct", "mcd", "radon")])
def test_sibling_losses_on_sr_and_ct(M, task, method, loss):
    """The siblings' data terms on the other tasks against the oracle: one-channel MSE / Gaussian NLL behind the SR projection
    (bayesian_optimization.py:1983-1985, 2406, 2614), MSE of the Radon transform (:377, :789, :991)."""
    from mfvi_dip_mia_amd.engine import SiblingEngine
    H = W = 32; seed = 9; K = 2
    small = dict(nd=(8, 16), nu=(8, 16), ns=(4, 4))
    n_out = 1 if task == "ct" else 2
    p = 0.25 if method == "mcd" else 0.0
    eng = SiblingEngine(H, W, method=method, task=task, K=K, input_depth=8, lr=1e-3, seed=seed, dropout_p=p or 0.3, net_kwargs=small, autotune=False)
    net = O.make_net(H, W, input_depth=8, n_out=n_out, drop_down=p, drop_up=p, **small)
    img = O.phantom(H, W, seed)
    theta = np.arange(0, 180., 4., dtype=np.float32)
    tgt = np.ascontiguousarray(img[::4, ::4]) if task == "sr" else O.radon_fwd(img, theta)
    eng.set_target(dev(tgt))
    mu, bnp = host(eng.mu).copy(), host(eng.bn).copy()
    eng.grad_only(step=2, perturb=True, with_kl=False)
    z0 = host(eng.z0)
    z = z0 + 0.1 * O.normal_fill(seed, 1, 0, 0, 2, z0.size).reshape(z0.shape)
    r = O.sibling_grad(net, mu, bnp, z, tgt, loss=loss, seed=seed, step=2, K=K, want_out=True, factor=4 if task == "sr" else 1, theta_deg=theta)
    assert relerr(host(eng.out), r["out"]) < 2e-5
    assert abs(eng.losses()[0] - r["loss"]) < 5e-5 * abs(r["loss"])
    assert relerr(host(eng.dmu), r["dmu"]) < 3e-4 and relerr(host(eng.dbn), r["dbn"]) < 3e-4
