"""GPU parity of everything around the ELBO iteration that writes a number into save.npz or decides whether an update happens:
the runners' per-iteration bookkeeping (mfvi_bookkeep / _inpainting / ring_stats / post_step / decimate, driven through the
product's own _Book / _BookInp classes) against the reference-generated golden (tests/golden/bookkeeping.npz) and the oracle,
the CT runners' NaN guard, and the drop-in gaussian_nll / gaussian_nll_inpainting autograd functions."""
import os
import types

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
    M_._lib.lib()
    return M_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _case(g, task):
    from test_oracle_golden import _bookkeeping_case
    return _bookkeeping_case(g, task)


def _stub_engine(H, W, C, n):
    """What _Book / _BookInp read from an engine: the output buffer and its geometry."""
    return types.SimpleNamespace(H=H, W=W, out=torch.zeros((n, C, H, W), device="cuda"), chunk=n)


@pytest.mark.parametrize("task", ["den", "sr", "ct"])
def test_bookkeeping_against_reference_golden(M, golden_dir, task):
    """K = 1: the reference's loop (bayesian_optimization.py:1374-1416 den, :2190-2236 sr incl. the low-resolution column, :584-626 ct):
    all 8 stored numbers per iteration, the EMA, and the ring-buffer snapshots (unbiased variance over ALL 25 slots, zeros included)."""
    from mfvi_dip_mia_amd.runner import _Book
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    bk, raws = _case(g, task)
    H, W, C = [int(v) for v in g[task + "_shape"]]
    eng = _stub_engine(H, W, C, 1)
    book = _Book(eng, len(raws), bk.gt, bk.noisy if task == "den" else None, task=task, factor=4)
    snaps = {}
    for i, raw in enumerate(raws):
        eng.out.copy_(dev(raw))
        book.iteration(eng, i, 1)
        if i in (10, 27):
            snaps[i] = book.snapshot()
    mse_noisy, mse_gt, psnrs, ssims = book.results()
    Mg = g[task + "_metrics"]
    assert relerr(mse_noisy, Mg[:, 0]) < 2e-5 and relerr(mse_gt, Mg[:, 1]) < 2e-5
    assert np.abs(psnrs - Mg[:, 2:5]).max() < 2e-4            # dB
    assert np.abs(ssims - Mg[:, 5:8]).max() < 2e-5
    assert relerr(host(book.ema), g[task + "_ema"]) < 2e-6
    for i, (var, ale, recon) in snaps.items():
        assert relerr(var, g["%s_var%d" % (task, i)]) < 2e-5
        assert relerr(recon, g["%s_recon%d" % (task, i)]) < 2e-6
        if task != "ct":
            assert relerr(ale, g["%s_ale%d" % (task, i)]) < 2e-6


@pytest.mark.parametrize("mc", [1, 3])
def test_inpainting_bookkeeping_against_reference_golden(M, golden_dir, mc):
    """bayesian_optimization.py:3039-3090: sigmoid colour channels, masked PSNR / SSIM, 3-channel epistemic ring."""
    from mfvi_dip_mia_amd.runner import _BookInp
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    task = "inp%d" % mc
    bk, raws = _case(g, task)
    H, W, _ = [int(v) for v in g["inp_shape"]]
    eng = _stub_engine(H, W, 4, 1)
    book = _BookInp(eng, len(raws), bk.gt, bk.mask)
    snaps = {}
    for i, raw in enumerate(raws):
        eng.out.copy_(dev(raw))
        book.iteration(eng, i, 1)
        if i in (10, 27):
            snaps[i] = book.snapshot()
    mse_c, mse_gt, psnrs, ssims = book.results()
    Mg = g[task + "_metrics"]
    assert relerr(mse_c, Mg[:, 0]) < 2e-5 and relerr(mse_gt, Mg[:, 1]) < 2e-5
    assert np.abs(psnrs - Mg[:, 2:5]).max() < 2e-4 and np.abs(ssims - Mg[:, 5:8]).max() < 2e-5
    assert relerr(host(book.ema), g[task + "_ema"]) < 2e-6
    for i, (var, ale, recon) in snaps.items():
        assert relerr(var, g["%s_var%d" % (task, i)]) < 2e-5 and relerr(ale, g["%s_ale%d" % (task, i)]) < 2e-6
        assert relerr(recon, g["%s_recon%d" % (task, i)]) < 2e-6


@pytest.mark.parametrize("task", ["den", "sr", "inp"])
def test_bookkeeping_k_samples_against_oracle(M, task):
    """K = 3 MC samples per iteration (the build's generalisation: `out` := sample mean of [out_k[:1], exp(-out_k[1:])]) against the
    oracle's Bookkeeper, which the K = 1 golden pins."""
    from mfvi_dip_mia_amd.runner import _Book, _BookInp
    K, n_it = 3, 6
    if task == "inp":
        H, W, C, seed = 24, 32, 4, 71
        img = np.stack([O.phantom(H, W, seed + c) for c in range(3)])
        mask = (O.uniform_fill(seed, 3, 0, 0, H * W).reshape(1, H, W) > 0.3).astype(np.float32)
        bk = O.Bookkeeper("inp", H, W, img, mask=mask)
        eng = _stub_engine(H, W, C, K); book = _BookInp(eng, n_it, img, mask)
    else:
        H, W, C, seed = 32, 48, 2, 72
        img = O.phantom(H, W, seed); noisy = O.noisy(img, 0.1, seed)
        bk = O.Bookkeeper(task, H, W, img, noisy=noisy)
        eng = _stub_engine(H, W, C, K); book = _Book(eng, n_it, img, noisy if task == "den" else None, task=task, factor=4)
    rows = []
    for i in range(n_it):
        raw = np.stack([O.bookkeeping_raw(task, seed, i, img, C, k=k) for k in range(K)])
        eng.out.copy_(dev(raw)); book.iteration(eng, i, K)
        rows.append(bk.step(raw))
    rows = np.array(rows)
    a, b, psnrs, ssims = book.results()
    assert relerr(a, rows[:, 0]) < 2e-5 and relerr(b, rows[:, 1]) < 2e-5
    assert np.abs(psnrs - rows[:, 2:5]).max() < 2e-4 and np.abs(ssims - rows[:, 5:8]).max() < 2e-5
    var, ale, recon = book.snapshot(); ovar, oale, orecon = bk.snapshot()
    assert relerr(var, ovar) < 2e-5 and relerr(ale, oale) < 2e-6 and relerr(recon, orecon) < 2e-6


def test_post_step_and_decimate(M):
    """mfvi_post_step: the reference's in-place `out[:, 1:] = exp(-out[:, 1:])` + EMA over sample 0 (bayesian_optimization.py:1374-1381);
    mfvi_decimate: x[::f, ::f]."""
    L = M._lib
    n, C, H, W = 2, 2, 20, 28
    raw = np.stack([O.bookkeeping_raw("den", 5, 0, O.phantom(H, W, 5), C, k=k) for k in range(n)])
    out = dev(raw); ema = torch.zeros((C, H, W), device="cuda")
    L.check(L.lib().mfvi_post_step(L.ptr(out), n, C, H, W, L.ptr(ema), 0.99, 1, L.stream_ptr()))
    ref = raw.copy(); ref[:, 1] = np.exp(-ref[:, 1])
    assert relerr(host(out), ref) < 2e-6 and relerr(host(ema), ref[0]) < 2e-6
    raw2 = np.stack([O.bookkeeping_raw("den", 5, 1, O.phantom(H, W, 5), C, k=k) for k in range(n)])
    out2 = dev(raw2)
    L.check(L.lib().mfvi_post_step(L.ptr(out2), n, C, H, W, L.ptr(ema), 0.99, 0, L.stream_ptr()))
    ref2 = raw2.copy(); ref2[:, 1] = np.exp(-ref2[:, 1])
    assert relerr(host(ema), ref[0] * np.float32(0.99) + ref2[0] * np.float32(0.01)) < 2e-6
    x = dev(O.normal_fill(9, 2, 0, 0, 0, 24 * 36).reshape(24, 36)); d = torch.empty((6, 9), device="cuda")
    L.check(L.lib().mfvi_decimate(L.ptr(x), 24, 36, 4, L.ptr(d), L.stream_ptr()))
    assert np.array_equal(host(d), host(x)[::4, ::4])
    assert L.lib().mfvi_decimate(L.ptr(x), 2, 36, 4, L.ptr(d), L.stream_ptr()) == -1


def test_ct_nan_guard_skips_the_update_on_the_device(M):
    """`if not torch.isnan(loss): optimizer.step()` (bayesian_optimization.py:581-582): a poisoned sinogram leaves parameters, both Adam
    moments and the count of applied steps untouched; the next, healthy iteration is Adam's FIRST update (bias corrections of t = 1),
    bit-identical to an engine that only ever ran that iteration."""
    from mfvi_dip_mia_amd.engine import ElboEngine
    S = 32
    kw = dict(task="ct", K=2, input_depth=8, temp=2.2e-10, sigma=1.7e-7, lr=1e-3, seed=3,
              net_kwargs=dict(nd=(8, 16), nu=(8, 16), ns=(4, 4)), autotune=False)
    img = O.phantom(S, S, 3)
    theta = np.arange(0, 180., 4., dtype=np.float32)
    sino = O.radon_fwd(img, theta)
    bad = sino.copy(); bad[7, 11] = np.nan
    a = ElboEngine(S, S, **kw)
    p0 = a.params.clone()
    a.set_target(torch.from_numpy(bad)); a.step()
    nll, kl, loss = a.losses()
    assert np.isnan(nll) and np.isfinite(kl)
    assert torch.equal(a.params, p0) and float(a.m.abs().max()) == 0.0 and float(a.v.abs().max()) == 0.0 and int(a.t_applied) == 0
    a.set_target(torch.from_numpy(sino)); a.step()
    assert int(a.t_applied) == 1 and np.isfinite(a.losses()[2]) and not torch.equal(a.params, p0)
    b = ElboEngine(S, S, **kw)
    b.set_target(torch.from_numpy(sino)); b.t = 1; b.step()          # iteration index 1 (its eps / input noise), first applied update
    assert torch.equal(a.params, b.params) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
    # the guarded update == the plain one when nothing is NaN (same arithmetic, device-side bias corrections)
    c = ElboEngine(S, S, **dict(kw, task="den", temp=5.7e-7, sigma=1.5e-5))
    d = ElboEngine(S, S, **dict(kw, task="den", temp=5.7e-7, sigma=1.5e-5))
    tgt = torch.from_numpy(O.noisy(img, 0.1, 3))
    c.set_target(tgt); d.set_target(tgt)
    L = M._lib
    for it in range(3):
        c.step()
        d.grad_only(d.t, with_kl=False); d.t += 1
        L.check(L.lib().mfvi_elbo_update_guarded(L.ptr(d.params), L.ptr(d.grads), L.ptr(d.m), L.ptr(d.v), d.n_vi, d.n_bn, 0.0, d.prior_sigma, d.temp,
                                                 d.lr, 0.9, 0.999, 1e-8, L.ptr(d.t_applied), L.ptr(d.acc), None, L.ptr(d.acc[1:]), L.ptr(d.upd_scratch),
                                                 L.stream_ptr()))
    assert int(d.t_applied) == 3
    assert float((c.params - d.params).abs().max()) < 1e-6 and abs(float(c.acc[1]) - float(d.acc[1])) < 1e-9 * abs(float(c.acc[1]))


def test_sibling_ct_nan_guard(M):
    """The DIP / MC-dropout / SGLD CT runners carry the same guard (bayesian_optimization.py:380, 792, 994): AdamW blocks untouched,
    counter not advanced."""
    from mfvi_dip_mia_amd.engine import SiblingEngine
    S = 32
    img = O.phantom(S, S, 4); theta = np.arange(0, 180., 4., dtype=np.float32); sino = O.radon_fwd(img, theta)
    bad = sino.copy(); bad[0, 0] = np.nan
    e = SiblingEngine(S, S, method="dip", task="ct", K=1, input_depth=8, lr=1e-3, seed=4, net_kwargs=dict(nd=(8, 16), nu=(8, 16), ns=(4, 4)), autotune=False)
    p0 = e.params.clone()
    e.set_target(torch.from_numpy(bad)); e.step()
    assert torch.equal(e.params, p0) and int(e.t_applied) == 0 and float(e.m.abs().max()) == 0.0
    e.set_target(torch.from_numpy(sino)); e.step(); e.step()
    assert int(e.t_applied) == 2 and not torch.equal(e.params, p0) and bool(torch.isfinite(e.params).all())


def test_dropin_gaussian_nll_functions(M, golden_dir):
    """gaussian_nll / gaussian_nll_inpainting of the package are autograd Functions over the HIP kernels (utils/bayesian_utils.py:29-39):
    value and gradients against the reference goldens (micro.npz, inpainting.npz: generated by the reference's own functions)."""
    g = np.load(os.path.join(golden_dir, "micro.npz"))
    o = O.normal_fill(8, 2, 0, 0, 0, 2 * 32 * 32).reshape(1, 2, 32, 32).copy()
    o[0, 1, 0, :4] = [25.0, -25.0, 20.0, -20.0]
    t = O.uniform_fill(8, 1, 0, 0, 32 * 32).reshape(1, 1, 32, 32)
    ot = dev(o).requires_grad_(True)
    nll = M.gaussian_nll(ot[:, :1], ot[:, 1:], dev(t))
    assert nll.dim() == 0 and nll.dtype == torch.float32
    (3.0 * nll).backward()                                     # the upstream gradient is applied on the device
    assert abs(float(nll) - float(g["nll"])) < 2e-6 * abs(float(g["nll"]))
    assert relerr(host(ot.grad)[0], 3.0 * g["nll_dout"]) < 5e-6
    s = M.gaussian_nll(ot[:, :1], ot[:, 1:], dev(t), reduction='sum')
    assert abs(float(s) - float(g["nll"]) * 1024) < 1e-5 * abs(float(g["nll"]) * 1024)
    gi = np.load(os.path.join(golden_dir, "inpainting.npz"))
    H, W = 12, 20
    o = (2.0 * O.normal_fill(311, 2, 0, 0, 0, 4 * H * W)).reshape(1, 4, H, W).copy(); o[0, 3, 0, :4] = [25.0, -30.0, 19.9, 0.0]
    tgt = O.uniform_fill(311, 1, 0, 0, 3 * H * W).reshape(1, 3, H, W)
    for mc in (1, 3):
        mask = (O.uniform_fill(311, 2 + mc, 0, 0, mc * H * W).reshape(1, mc, H, W) > 0.3).astype(np.float32)
        ot = dev(o).requires_grad_(True)
        nll = M.gaussian_nll_inpainting(ot[:, :3].sigmoid(), ot[:, 3:], dev(tgt), dev(mask))      # the runner's call (bayesian_optimization.py:3033-3036)
        nll.backward()
        assert abs(float(nll) - float(gi["nll_mask%d" % mc])) < 5e-6 * abs(float(gi["nll_mask%d" % mc]))
        assert relerr(host(ot.grad)[0], gi["nll_mask%d_dout" % mc]) < 1e-5
    with pytest.raises(ValueError):
        M.gaussian_nll(ot[:, :1], ot[:, 3:], dev(tgt[:, :1]), reduction='none')
    import inspect
    from mfvi_dip_mia_amd import bayes
    src = inspect.getsource(bayes.gaussian_nll) + inspect.getsource(bayes._GaussianNLL)
    assert "torch.exp" not in src and "torch.clamp" not in src       # the arithmetic is in libmfvi_hip, not ATen


def test_bookkeeping_beside_the_backward_pass_equals_the_serial_order(M):
    """The runners start the per-iteration bookkeeping behind the forward pass on a second stream (`eng.step(after_forward=book.hook(...))`),
    beside the backward pass: same kernels, same inputs — every stored metric and the snapshot equal those of bookkeeping after the step
    (to the last bits only: two training runs differ there through the summation order of the BatchNorm statistics' atomics)."""
    from mfvi_dip_mia_amd.engine import ElboEngine
    from mfvi_dip_mia_amd.runner import _Book
    H = W = 64; K, n_it, seed = 2, 7, 11
    img = O.phantom(H, W, seed); noisy = O.noisy(img, 0.1, seed)
    res = []
    for mode in ("serial", "beside"):
        eng = ElboEngine(H, W, task="den", K=K, temp=5.66e-7, sigma=1.46e-5, lr=1e-3, seed=seed, autotune=False)
        eng.set_target(torch.from_numpy(noisy))
        book = _Book(eng, n_it, img, noisy, task="den")
        for i in range(n_it):
            if mode == "serial":
                eng.step(); book.iteration(eng, i, eng.chunk)
            else:
                eng.step(after_forward=book.hook(eng, i, eng.chunk))
        res.append(book.results() + book.snapshot())
    for a, b in zip(*res):
        assert relerr(np.asarray(a, np.float64), np.asarray(b, np.float64)) < 1e-5
