"""GPU tests of the fused engine on the three tasks (den / SR / CT) against the oracle, and of the runner's artefacts."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available()
    M_._lib.lib()
    return M_


SMALL = dict(nd=(8, 16), nu=(8, 16), ns=(4, 4))


@pytest.mark.parametrize("task,temp,sigma", [("den", 5.656911698337764e-07, 1.4616642493692077e-05), ("sr", 4.381719802264805e-07, 4.9e-08),
                                             ("ct", 2.2e-10, 1.7e-7)])
def test_engine_steps_match_oracle(M, task, temp, sigma):
    """Three fused ELBO iterations (perturbation, K=2 forwards, loss, backward, KL, Adam) vs the oracle doing the same."""
    H = W = 32; K, seed, lr = 2, 4, 1e-3
    n_out = 1 if task == "ct" else 2
    eng = M.engine.ElboEngine(H, W, task=task, K=K, input_depth=8, temp=temp, sigma=sigma, lr=lr, seed=seed, net_kwargs=SMALL)
    onet = O.make_net(H, W, input_depth=8, n_out=n_out, **SMALL)
    img = O.phantom(H, W, seed)
    theta = None
    if task == "den":
        tgt = O.noisy(img, 0.1, seed)
    elif task == "sr":
        tgt = np.ascontiguousarray(img[::4, ::4])
    else:
        theta = np.arange(0, 180., 4., dtype=np.float32); tgt = O.radon_fwd(img, theta)
    eng.set_target(torch.from_numpy(tgt))
    n = eng.n_vi
    p = eng.params.cpu().numpy().copy(); z0 = eng.z0.cpu().numpy().copy()
    # engine init follows the RNG spec (domain INIT / UNIFORM): the oracle regenerates the same numbers
    mu0, rho0, bn0 = O.init_params(onet, seed)
    # (bit-equal eps; the affine map a + b*eps is one fma on the device, two roundings in numpy)
    assert np.abs(p[:n] - mu0).max() < 1e-7 and np.abs(p[n:2 * n] - rho0).max() < 5e-7 and np.array_equal(p[2 * n:], bn0)
    assert np.abs(z0.ravel() - 0.1 * O.uniform_fill(seed, 0, 0, 0, z0.size)).max() < 1e-8
    m = np.zeros_like(p); v = np.zeros_like(p)
    tid = {"den": 0, "sr": 1, "ct": 2}[task]
    for it in range(3):
        eng.step()
        nll, kl, loss = eng.losses()
        z = z0 + 0.1 * O.normal_fill(seed, 1, 0, 0, it, z0.size).reshape(z0.shape)
        r = O.elbo_grad(onet, p[:n], p[n:2 * n], p[2 * n:], z, tgt, task=tid, factor=4, theta_deg=theta, seed=seed, step=it, K=K,
                        temp=temp, prior_sigma=eng.prior_sigma)
        assert abs(loss - r["loss"]) < 2e-4 * max(abs(r["loss"]), 1e-3), (task, it, loss, r["loss"])
        O.adam(p, np.concatenate([r["dmu"], r["drho"], r["dbn"]]), m, v, lr, it + 1)
        d = np.abs(eng.params.cpu().numpy() - p)
        assert d.max() < 2.5e-3 * (it + 1) and d.mean() < 5e-5 * (it + 1), (task, it, d.max(), d.mean())
        p = eng.params.cpu().numpy().copy()       # re-anchor: Adam amplifies rounding noise of near-zero gradients


def test_runner_artifacts(M, tmp_path):
    r = M.runner.run_den_mfvi(img="phantom", imsize=(64, 64), num_iter=24, lr=1e-3, temp=5.656911698337764e-07, sigma=1.4616642493692077e-05,
                              input_depth=16, seed=1, show_every=5, save=True, save_path=str(tmp_path), K=2)
    z = np.load(os.path.join(r["run_dir"], "save.npz"), allow_pickle=True)
    # schema read by eval_denoising.ipynb:79-82 (.flat[0]['mfvi'])
    for key, shape in [("mse_noisy", (25,)), ("mse_gt", (25,)), ("psnrs", (25, 3)), ("ssims", (25, 3)), ("recons", (6, 1, 64, 64)),
                       ("uncerts", (6, 1, 64, 64)), ("uncerts_ale", (6, 1, 64, 64))]:
        assert z[key].flat[0]["mfvi"].shape == shape, key
    assert z["img_gt"].shape == (1, 64, 64) and z["img_noisy"].shape == (1, 64, 64)          # get_image() arrays: (1, H, W)
    assert os.path.exists(os.path.join(r["run_dir"], "locals.txt"))
    ps, ss = z["psnrs"].flat[0]["mfvi"], z["ssims"].flat[0]["mfvi"]
    assert np.isfinite(ps).all() and np.isfinite(ss).all()
    # metrics kernels vs the oracle on the stored snapshot of the smoothed reconstruction (iteration 20 = snapshot 4)
    rec = z["recons"].flat[0]["mfvi"][4, 0].astype(np.float32)
    assert abs(O.psnr(z["img_gt"][0], rec) - ps[20, 2]) < 1e-3
    assert abs(O.ssim(z["img_gt"][0], rec) - ss[20, 2]) < 1e-4
    assert abs(float(np.mean((rec - z["img_gt"][0]) ** 2)) - 10 ** (-ps[20, 2] / 10)) < 1e-6
    # the fit moves: PSNR of the smoothed output improves over the first 25 iterations
    assert ps[-1, 2] > ps[0, 2]
    r2 = M.runner.run_ct_mfvi(img="phantom", imsize=(32, 32), num_iter=4, lr=1e-3, temp=2.2e-10, sigma=1.7e-7, input_depth=8, seed=1, show_every=2,
                              save=True, save_path=str(tmp_path), K=1, net_kwargs=SMALL)
    z2 = np.load(os.path.join(r2["run_dir"], "save.npz"), allow_pickle=True)
    assert z2["img_radon"].shape == (1, 1, 45, 32) and np.isfinite(z2["psnrs"].flat[0]["mfvi"]).all()
    r3 = M.runner.run_sr_mfvi(img="phantom", imsize=(64, 64), num_iter=4, lr=1e-3, temp=4.4e-7, sigma=4.9e-8, input_depth=8, seed=2, show_every=2,
                              save=True, save_path=str(tmp_path), K=2, net_kwargs=SMALL)
    assert np.isfinite(r3["psnrs"]).all()
    z3 = np.load(os.path.join(r3["run_dir"], "save.npz"), allow_pickle=True)           # the SR runner's keys (bayesian_optimization.py:2258-2260)
    assert set(z3.files) == {"img_hr", "img_lr", "mse_noisy", "mse_gt", "recons", "uncerts", "uncerts_ale", "psnrs", "ssims"}
    assert z3["img_hr"].shape == (1, 64, 64) and z3["img_lr"].shape == (16, 16)
    # column 0 is the low-resolution metric (psnr_lr / mse of the projection), not a copy of the ground-truth column
    assert not np.allclose(r3["psnrs"][:, 0], r3["psnrs"][:, 1]) and not np.allclose(r3["mse_noisy"], r3["mse_gt"])


def test_inpainting_runner_artifacts(M, tmp_path):
    """run_inp_mfvi (bayesian_optimization.py:2892-3114): save.npz carries the reference's keys for this task; the masked PSNR of the
    smoothed reconstruction improves over the run; device bookkeeping == numpy on the final state."""
    from mfvi_dip_mia_amd.runner import run_inp_mfvi
    r = run_inp_mfvi(img="phantom", imsize=(192, 192), num_iter=60, lr=2e-3, temp=1e-12, sigma=6.5e-4, input_depth=16, seed=2, show_every=20,
                     plot=True, save=True, save_path=str(tmp_path), K=1)
    z = np.load(os.path.join(r["run_dir"], "save.npz"), allow_pickle=True)
    assert set(z.files) == {"img_inpainting", "img_mask", "mse_corrupted", "mse_gt", "recons", "uncerts", "uncerts_ale", "psnrs", "ssims"}
    assert z["img_inpainting"].shape == (3, 192, 192) and z["img_mask"].shape == (1, 192, 192) and 0.05 < 1 - z["img_mask"].mean() < 0.4
    ps = z["psnrs"].item()["mfvi"]; ss = z["ssims"].item()["mfvi"]
    assert ps.shape == (61, 3) and ss.shape == (61, 3) and np.isfinite(ps).all() and np.isfinite(ss).all()
    assert ps[-1, 2] > ps[5, 2]                                      # the EMA reconstruction approaches the known pixels
    assert z["recons"].item()["mfvi"].shape == (4, 3, 192, 192)
    assert os.path.exists(os.path.join(r["run_dir"], "locals.txt")) and os.path.exists(os.path.join(r["run_dir"], "out_avg.png"))
    # metric kernels vs numpy on the last recorded reconstruction
    rec = z["recons"].item()["mfvi"][-1]; img = z["img_inpainting"]; mk = z["img_mask"]
    psnr_np = 10 * np.log10(1.0 / np.mean((img * mk - rec * mk) ** 2))
    assert abs(psnr_np - ps[-1, 2]) < 1e-3 * abs(psnr_np)


@pytest.mark.parametrize("method", ["dip", "mcd", "sgld"])
def test_sibling_runner_artifacts(M, tmp_path, method):
    """run_den_{dip,mcd,sgld} (bayesian_optimization.py:1064-1237, 1447-1860): save.npz keyed by the method name like the reference's
    MSE_CORRUPTED['dip'] ... dicts (DIP leaves the uncertainty dicts empty); the fit moves; CLI candidates come from the method's bo_params."""
    fn = getattr(M.runner, "run_den_" + method)
    r = fn(img="phantom", imsize=(64, 64), num_iter=24, lr=1e-3, input_depth=16, seed=1, show_every=5, save=True, save_path=str(tmp_path), K=1)
    z = np.load(os.path.join(r["run_dir"], "save.npz"), allow_pickle=True)
    for key, shape in [("mse_noisy", (25,)), ("mse_gt", (25,)), ("psnrs", (25, 3)), ("ssims", (25, 3)), ("recons", (6, 1, 64, 64))]:
        assert z[key].flat[0][method].shape == shape, key
    if method == "dip":
        assert z["uncerts"].flat[0] == {} and z["uncerts_ale"].flat[0] == {}
    else:
        assert z["uncerts"].flat[0][method].shape == (6, 1, 64, 64) and z["uncerts_ale"].flat[0][method].shape == (6, 1, 64, 64)
    ps = z["psnrs"].flat[0][method]
    assert np.isfinite(ps).all() and ps[-1, 2] > ps[0, 2]
    assert r["engine"].method == method and float(r["engine"].rho.abs().max()) == 0.0
    if method == "sgld":
        assert abs(r["engine"].lr - 1e-3 * 0.996 ** 25) < 1e-12          # ExponentialLR stepped once per iteration
    cands, rp = M.runner.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", method + "_den.json"), method)
    assert list(cands[0]) == list(M.runner.BO_KEYS[method]) and rp["input_depth"] == 16


def test_sibling_runners_other_tasks(M, tmp_path):
    r = M.runner.run_ct_mcd(img="phantom", imsize=(32, 32), num_iter=3, lr=1e-3, input_depth=8, seed=1, show_every=2, save=False, net_kwargs=SMALL,
                            dropout_p=0.1, weight_decay=1e-6)
    assert np.isfinite(r["psnrs"]).all()
    r = M.runner.run_sr_sgld(img="phantom", imsize=(64, 64), num_iter=3, lr=1e-3, input_depth=8, seed=1, show_every=2, save=False, net_kwargs=SMALL)
    assert np.isfinite(r["psnrs"]).all()
    r = M.runner.run_sr_dip(img="phantom", imsize=(64, 64), num_iter=3, lr=1e-3, input_depth=8, seed=1, show_every=2, save=False, net_kwargs=SMALL)
    assert np.isfinite(r["psnrs"]).all()
    r = M.runner.run_inp_mcd(img="phantom", imsize=(192, 192), num_iter=3, input_depth=8, seed=1, show_every=2, save=True, save_path=str(tmp_path),
                             dropout_p=0.1)
    z = np.load(os.path.join(r["run_dir"], "save.npz"), allow_pickle=True)
    assert z["psnrs"].item()["mcd"].shape == (4, 3) and np.isfinite(z["psnrs"].item()["mcd"]).all()
    r = M.runner.run_inp_dip(img="phantom", imsize=(192, 192), num_iter=40, input_depth=8, seed=1, show_every=20, save=True, save_path=str(tmp_path))
    z = np.load(os.path.join(r["run_dir"], "save.npz"), allow_pickle=True)
    ps = z["psnrs"].item()["dip"]
    assert ps.shape == (41, 3) and np.isfinite(ps).all() and ps[-1, 2] > ps[3, 2] and z["uncerts"].item() == {}
    assert r["engine"].weight_decay == 0.0


@pytest.mark.gpu
def test_cli_gp_outer_loop_two_rounds(M, tmp_path, capsys):
    """`--bo-rounds 2` of the runner CLI: round 0 fits the config's candidates, the GP (bo.py, parity unpinned) proposes new
    (temp, sigma) pairs inside the config's log bounds, round 1 fits those through the same HIP path — every fit returns a finite PSNR."""
    import json
    cfg = {"bo_params": {"temp": {"logbounds": [-8.0, -5.0], "candidates": [5.66e-7, 5e-6]}, "sigma": {"logbounds": [-6.0, -4.0], "candidates": [1.46e-5]}},
           "run_params": {"img": "phantom", "num_iter": 12, "lr": 1e-3, "seed": 1, "p_sigma": 0.1, "input_depth": 16, "show_every": 4, "plot": False,
                          "save": False, "save_path": str(tmp_path)}}
    path = os.path.join(str(tmp_path), "bo_den.json")
    json.dump(cfg, open(path, "w"))
    X, Y, nxt = M.runner.main(["--task", "denoising", "--bayes", "mfvi", "--config", path, "--imsize", "64", "--k", "2", "--bo-rounds", "2"])
    assert len(X) == len(Y) >= 3 and all(np.isfinite(y) for y in Y)          # 2 fits in round 0, >= 1 proposal in round 1
    for t, s in list(X[2:]) + list(nxt):
        assert 1e-8 * (1 - 1e-9) <= t <= 1e-5 * (1 + 1e-9) and 1e-6 * (1 - 1e-9) <= s <= 1e-4 * (1 + 1e-9)
    assert "psnr" in capsys.readouterr().out
