"""Fit QUALITY over hundreds of iterations: the HIP paths against curves recorded from the reference's own loop
(oracle/make_golden.py --fitcurve: MeanFieldVI + skip + gaussian_nll + temp * net.kl() + torch.optim.AdamW, K = 1, eps and the input
perturbation injected from the RNG spec; bayesian_optimization.py:1356-1406, metrics as eval_denoising.ipynb:94-122 reads them).

The golden holds the reference run in float32 AND in float64.  The two decorrelate (a 26-layer net with train-mode BatchNorm over as few
as 2x2 pixels, LeakyReLU kinks and Adam's sign-like first steps is chaotic): at 64x64 they agree to 0.03 in the ELBO for ~100 iterations and
wander up to 0.7 apart afterwards, the smoothed PSNR stays within 0.26 dB; at 128x128 the first 100 iterations agree to 0.03 / 0.04 dB.
So the assertions are: (a) while the reference agrees with itself, the HIP run agrees with it about as well (3x its own band);
(b) over the whole run the smoothed PSNR — what the paper reports — stays within the reference's own float32-vs-float64 spread + 0.15 dB;
(c) storing mu / rho in bfloat16 with stochastic rounding (no float32 master copy: an extension the reference does not have) costs
less than 0.3 dB of smoothed PSNR."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

DEN = dict(temp=5.656911698337764e-07, sigma=1.4616642493692077e-05, lr=1e-3)


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available(), "these tests need the GPU"
    M_._lib.lib()
    return M_


def psnr(a, b):
    return float(10 * torch.log10(1 / torch.mean((a - b) ** 2)))


def engine_curve(M, size, steps, every, seed, **kw):
    """The fused runner on the golden's inputs: same initial parameters, z0, per-step perturbation and eps (all from the RNG spec)."""
    # autotune=False: the heuristic tilings, the same on every box — the per-box choice of the autotuner changes summation orders, and over
    # hundreds of iterations this fit amplifies rounding differences to tenths of a dB (as the reference's own float32 / float64 runs show)
    eng = M.engine.ElboEngine(size, size, task="den", K=1, input_depth=16, seed=seed, autotune=False, **DEN, **kw)
    gt = torch.from_numpy(O.phantom(size, size, seed)).cuda()
    eng.set_target(torch.from_numpy(O.noisy(O.phantom(size, size, seed), 0.1, seed)))
    rows, avg = [], None
    for it in range(steps):
        eng.step()
        o = eng.out[0, 0]
        avg = o.clone() if avg is None else avg * 0.99 + o * 0.01
        if it % every == 0 or it == steps - 1:
            nll, kl, loss = eng.losses()
            rows.append((it, loss, nll, kl, psnr(gt, o.clip(0, 1)), psnr(gt, avg.clip(0, 1))))
    return np.array(rows)


def dropin_curve(M, size, steps, every, seed, flat=False):
    """INTEGRATION.md's loop on the drop-in classes (get_net + MeanFieldVI + gaussian_nll + torch.optim.AdamW), fed the golden's inputs."""
    dev = torch.device("cuda")
    net = M.get_net(16, 'skip', 'reflection', 'bilinear', n_channels=2, skip_n33d=[16, 32, 64, 128, 128], skip_n33u=[16, 32, 64, 128, 128],
                    skip_n11=4, num_scales=5)
    net = M.MeanFieldVI(net, prior={'mu': 0.0, 'sigma': float(np.sqrt(DEN["temp"]) * DEN["sigma"])}, replace_layers='all', device=dev,
                        reparam='', seed=seed, flat_parameters=flat, autotune=False)      # (heuristic tilings as in engine_curve: the autotuner's per-box choices change summation orders, and this fit amplifies them)
    onet = O.make_net(size, size)
    mu, rho, bnp = O.init_params(onet, seed)
    with torch.no_grad():
        net._flat.copy_(torch.from_numpy(np.concatenate([mu, rho, bnp])).to(dev))
    opt = torch.optim.AdamW(net.parameters(), lr=DEN["lr"], weight_decay=0)
    gt = torch.from_numpy(O.phantom(size, size, seed)).to(dev)
    tgt = torch.from_numpy(O.noisy(O.phantom(size, size, seed), 0.1, seed)).to(dev)[None, None]
    z0 = (0.1 * O.uniform_fill(seed, 0, 0, 0, 16 * size * size)).reshape(1, 16, size, size)
    rows, avg = [], None
    for it in range(steps):
        z = torch.from_numpy(z0 + 0.1 * O.normal_fill(seed, 1, 0, 0, it, z0.size).reshape(z0.shape)).to(dev)
        opt.zero_grad()
        out = net(z)
        nll = M.gaussian_nll(out[:, :1], out[:, 1:], tgt); kl = net.kl()
        loss = nll + DEN["temp"] * kl
        loss.backward(); opt.step()
        o = out.detach()[0, 0]
        avg = o.clone() if avg is None else avg * 0.99 + o * 0.01
        if it % every == 0 or it == steps - 1:
            rows.append((it, float(loss), float(nll), float(kl), psnr(gt, o.clip(0, 1)), psnr(gt, avg.clip(0, 1))))
    return np.array(rows)


def check_against_reference(c, g, agree_until, name):
    f32, f64 = g["curve"], g["curve_f64"]
    assert np.array_equal(c[:, 0], f64[:, 0])
    w = f64[:, 0] <= agree_until                                    # the stretch over which the reference agrees with itself
    for col, what, floor in ((1, "elbo", 5e-3), (3, "kl", 0.0), (5, "psnr_gt_sm", 0.02)):
        band = np.abs(f32[w, col] - f64[w, col]).max()
        dev = np.abs(c[w, col] - f64[w, col]).max()
        tol = 3 * band + floor + (1e-4 * np.abs(f64[w, col]).max() if col == 3 else 0.0)
        print("%s %-10s first %d its: |hip - ref64| max %.4g, reference's own f32-f64 band %.4g" % (name, what, agree_until, dev, band))
        assert dev <= tol, (name, what, dev, tol)
    # whole run: the smoothed PSNR (the quantity the paper reports), against the spread of the reference itself
    spread = np.abs(f32[:, 5] - f64[:, 5]).max()
    dev = np.abs(c[:, 5] - f64[:, 5]).max()
    print("%s psnr_gt_sm whole run: |hip - ref64| max %.3f dB (final %.3f vs %.3f / %.3f), reference spread %.3f dB" % (name, dev, c[-1, 5], f64[-1, 5], f32[-1, 5], spread))
    assert dev <= spread + 0.15, (name, dev, spread)
    assert abs(c[-1, 5] - f64[-1, 5]) <= max(2 * abs(f32[-1, 5] - f64[-1, 5]), 0.1) + 0.15       # (another summation order in one kernel moved this by 0.19 dB at 64^2: 15.39 -> 15.57)
    # and the fit did move: smoothed PSNR rose by what the reference's rose
    assert c[-1, 5] - c[0, 5] > 0.8 * (f64[-1, 5] - f64[0, 5])


@pytest.mark.parametrize("name,agree_until", [("fitcurve_den_64", 100), ("fitcurve_den_128", 99)])
def test_engine_fit_curve_against_reference(M, golden_dir, name, agree_until):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    c = engine_curve(M, int(g["size"]), int(g["steps"]), int(g["every"]), int(g["seed"]))
    check_against_reference(c, g, agree_until, "engine " + name)


@pytest.mark.parametrize("flat", [False, True])
def test_dropin_fit_curve_against_reference(M, golden_dir, flat):
    """flat=True: MeanFieldVI(flat_parameters=True) — torch.optim.AdamW over the one flat Parameter; same trajectory."""
    g = np.load(os.path.join(golden_dir, "fitcurve_den_128.npz"))
    c = dropin_curve(M, int(g["size"]), int(g["steps"]), int(g["every"]), int(g["seed"]), flat=flat)
    check_against_reference(c, g, 99, "drop-in%s fitcurve_den_128" % (" (flat)" if flat else ""))


def test_bf16_storage_reaches_the_float32_fit(M, golden_dir):
    """mu / rho stored in bfloat16, Adam on the float32 value the word denotes, stochastic rounding of the result, no master copy
    (mfvi_elbo_update_bf16): 300 iterations at 128x128 land where the float32 engine lands.  (At 64x64 the deepest maps are 2x2 / 4x4:
    below the MFMA kernels' minimum width, and bf16 storage exists on the MFMA path only.)"""
    size, steps, every, seed = 128, 300, 10, 1
    f = engine_curve(M, size, steps, every, seed)
    b = engine_curve(M, size, steps, every, seed, param_dtype="bf16")
    tail = f[:, 0] >= steps - 100
    d_sm = abs(b[tail, 5].mean() - f[tail, 5].mean()); d_elbo = abs(b[tail, 1].mean() - f[tail, 1].mean())
    print("bf16 vs f32 engine, last 100 iterations: smoothed PSNR %.3f vs %.3f dB, ELBO %.3f vs %.3f" % (b[tail, 5].mean(), f[tail, 5].mean(), b[tail, 1].mean(), f[tail, 1].mean()))
    assert d_sm < 0.3 and d_elbo < 0.5
    assert b[-1, 5] - b[0, 5] > 0.8 * (f[-1, 5] - f[0, 5])
