"""CPU tests of the host-side mirror of the reference interface: the skip() builder's module names, the layer-program
compiler's parameter layout, and the K-sharded all-reduce path on two gloo ranks."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_skip_builder_matches_reference_state_dict_keys(golden_dir):
    """Module names are part of the drop-in contract (checkpoints, notebooks): compare with the key list dumped from the
    reference's MeanFieldVI(skip(...)) (tests/golden/full_den_64_k1.npz)."""
    import mfvi_dip_mia_amd as M
    g = np.load(os.path.join(golden_dir, "full_den_64_k1.npz"))
    ref = [str(k) for k in g["state_dict_keys"]]
    net = M.get_net(16, 'skip', 'reflection', skip_n33d=[16, 32, 64, 128, 128], skip_n33u=[16, 32, 64, 128, 128], skip_n11=4,
                    num_scales=5, n_channels=2, upsample_mode='bilinear')
    mine = []
    for k in net.state_dict().keys():
        if k.endswith(".weight") and "Conv2d" in k:
            mine += ["net." + k[:-6] + s for s in ("W_mu", "W_rho")]
        elif k.endswith(".bias") and "Conv2d" in k:
            mine += ["net." + k[:-4] + s for s in ("bias_mu", "bias_rho")]
        else:
            mine.append("net." + k)
    # the reference registers W_mu, W_rho, bias_mu, bias_rho per layer: same grouping
    assert mine == ref
    # layer names in execution order
    ref_layers = [str(k) for k in g["layer_names"]]
    my_layers = ["net." + n for n, m in net.named_modules() if isinstance(m, torch.nn.Conv2d)]
    assert my_layers == ref_layers


def test_plain_torch_net_agrees_with_oracle():
    """The torch module tree itself (not the product path) is a faithful skip(): eval-free forward vs the oracle."""
    import mfvi_dip_mia_amd as M
    kw = dict(H=32, W=32, input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4))
    onet = O.make_net(**kw)
    mu, rho, bnp = O.init_params(onet, 5)
    conv, bn, n_vi, n_bnp = O.net_table(onet)
    net = M.skip(8, 2, num_channels_down=[8, 16, 16], num_channels_up=[8, 16, 16], num_channels_skip=[4, 4, 4],
                 upsample_mode='bilinear', need_sigmoid=False, pad='reflection', dropout_mode_down='None', dropout_mode_up='None')
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(convs) == len(conv) and len(bns) == len(bn)
    with torch.no_grad():
        for m, r in zip(convs, conv):
            cin, cout, k, s, wo, bo = [int(v) for v in r]
            assert (m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0]) == (cin, cout, k, s)
            m.weight.copy_(torch.from_numpy(mu[wo:wo + cout * cin * k * k].reshape(cout, cin, k, k))); m.bias.copy_(torch.from_numpy(mu[bo:bo + cout]))
    z = (0.1 * O.uniform_fill(5, 0, 0, 0, 8 * 32 * 32)).reshape(1, 8, 32, 32)
    out = net(torch.from_numpy(z)).detach().numpy()[0]
    ref, tape = O.net_forward(onet, mu, rho, bnp, z[0], 5, 0, 0, sample_weights=False)
    tape.free()
    assert np.abs(out - ref).max() < 2e-5 * np.abs(ref).max()


def test_program_layout_and_unsupported_options():
    import mfvi_dip_mia_amd as M
    P, zin, zout, names = M.skip_program(64, 64)
    conv, bn, n_vi, n_bnp = O.net_table(O.make_net(64, 64))
    assert (P.n_vi, P.n_bn) == (n_vi, n_bnp) == (1035446, 3984)
    assert [(l["w_off"], l["b_off"]) for l in P.layers] == [(int(c[4]), int(c[5])) for c in conv]
    Q = M.Program(); tq = Q.tensor(4, 8, 8)
    with pytest.raises(ValueError):
        Q.set_bn(tq, act=True, slope=1.5)            # LeakyReLU is formed as max(v, slope * v): slopes outside [0, 1] are refused
    with pytest.raises(NotImplementedError):
        M.skip(16, 2, need_sigmoid=True, pad='reflection', upsample_mode='bilinear')
    with pytest.raises(NotImplementedError):
        M.skip(16, 2, need_sigmoid=False, pad='zero', upsample_mode='bilinear')
    with pytest.raises(NotImplementedError):
        M.skip(16, 2, need_sigmoid=False, pad='reflection', upsample_mode='bilinear', dropout_mode_down='1d')
    with pytest.raises(NotImplementedError):
        M.get_net(16, 'UNet', 'reflection', 'bilinear')
    # the MC-dropout runners' net (bayesian_optimization.py:1526-1549): Dropout2d behind every deeper / up convolution, named like the reference
    mcd = M.get_net(16, 'skip', 'reflection', 'bilinear', n_channels=2, skip_n33d=[16, 32], skip_n33u=[16, 32], skip_n11=4, num_scales=2,
                    dropout_mode_down='2d', dropout_p_down=0.3, dropout_mode_up='2d', dropout_p_up=0.3)
    names = [n for n, m in mcd.named_modules() if isinstance(m, torch.nn.Dropout2d)]
    assert len(names) == 8 and names[0].endswith("Sequential_deeper_1.Dropout2d_deeper_1") and all(m.p == 0.3 for m in mcd.modules() if isinstance(m, torch.nn.Dropout2d))
    assert not any(isinstance(m, torch.nn.Dropout2d) for m in M.get_net(16, 'skip', 'reflection', 'bilinear', num_scales=2, skip_n33d=16, skip_n33u=16).modules())
    assert M.sharding.shard_samples(16, 3, 4) == (12, 4)
    with pytest.raises(ValueError):
        M.sharding.shard_samples(10, 0, 4)


def _rank_main(rank, world, port, tmp):
    """One rank of the K-sharded ELBO step on CPU (gloo): the oracle stands in for the HIP kernels; what is under test is
    the decomposition — global sample keying, ONE all-reduce of the flat buffer, redundant KL, identical Adam on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import mfvi_dip_mia_amd as M
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    O.set_threads(2)
    kw = dict(H=16, W=16, input_depth=8, n_out=2, nd=(8, 8), nu=(8, 8), ns=(4, 4))
    net = O.make_net(**kw)
    K, seed, temp, ps = 4, 11, 5.6e-7, 1e-6
    mu, rho, bnp = O.init_params(net, seed)
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, 8 * 16 * 16)).reshape(8, 16, 16)
    tgt = O.noisy(O.phantom(16, 16, seed), 0.1, seed)
    k0, kl_ = M.sharding.shard_samples(K, rank, world)
    r = O.elbo_grad(net, mu, rho, bnp, z, tgt, seed=seed, step=2, k0=k0, K=kl_, K_total=K, temp=temp, prior_sigma=ps, with_kl=False)
    flat = torch.from_numpy(np.concatenate([r["dmu"], r["drho"], r["dbn"], np.array([r["nll"]], np.float32)]))
    M.sharding.allreduce_sum_(flat)
    n = mu.size
    kl, dmu, drho = O.kl(mu, rho, ps, scale=temp, want_grad=True)
    g = flat.numpy().copy(); g[:n] += dmu; g[n:2 * n] += drho
    p = np.concatenate([mu, rho, bnp]); m = np.zeros_like(p); v = np.zeros_like(p)
    O.adam(p, g[:p.size], m, v, 1e-3, 1)
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), g=g, p=p)
    dist.destroy_process_group()


def test_k_sharding_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["g"], b["g"]) and np.array_equal(a["p"], b["p"])        # every rank holds the same update
    # and it equals the single-process K=4 result (rank-count invariance of the global sample keying)
    kw = dict(H=16, W=16, input_depth=8, n_out=2, nd=(8, 8), nu=(8, 8), ns=(4, 4))
    net = O.make_net(**kw)
    mu, rho, bnp = O.init_params(net, 11)
    z = (0.1 * O.uniform_fill(11, 0, 0, 0, 8 * 16 * 16)).reshape(8, 16, 16)
    tgt = O.noisy(O.phantom(16, 16, 11), 0.1, 11)
    r = O.elbo_grad(net, mu, rho, bnp, z, tgt, seed=11, step=2, K=4, temp=5.6e-7, prior_sigma=1e-6)
    full = np.concatenate([r["dmu"], r["drho"], r["dbn"]])
    assert np.abs(a["g"][:full.size] - full).max() < 1e-5 * np.abs(full).max()
    assert abs(a["g"][full.size] - r["nll"]) < 1e-5 * abs(r["nll"])


def test_inpainting_skip_builder_matches_reference_state_dict_keys(golden_dir):
    """skip() with the inpainting runner's options (no skip branches, need1x1_up=False: bayesian_optimization.py:2970-2998) gets
    the reference's module names too, including its 'Sequential_up_n' / 'Sequential_up_n_1' collision suffixes."""
    from mfvi_dip_mia_amd.nets import skip
    g = np.load(os.path.join(golden_dir, "inpainting.npz"))
    ref = [str(k) for k in g["net_keys"]]
    net = skip(8, num_output_channels=4, pad='reflection', num_channels_down=[8, 16, 16], num_channels_up=[8, 16, 16],
               num_channels_skip=[0, 0, 0], filter_size_down=5, filter_size_up=3, filter_skip_size=1, need1x1_up=False,
               upsample_mode='nearest', need_sigmoid=False)
    mine = []
    for k in net.state_dict().keys():
        if k.endswith(".weight") and "Conv2d" in k:
            mine += ["net." + k[:-6] + s for s in ("W_mu", "W_rho")]
        elif k.endswith(".bias") and "Conv2d" in k:
            mine += ["net." + k[:-4] + s for s in ("bias_mu", "bias_rho")]
        else:
            mine.append("net." + k)
    assert mine == ref


def test_independent_fit_fanout_with_stub_worker():
    """mfvi_dip_mia_amd.fanout (the reference's candidate fan-out, bayesian_optimization.py:3760-3781 / eval_result.py:27-53): jobs dealt
    round-robin over the devices, ONE fresh (spawned) process per device, (candidate, psnr) gathered through a queue, NaN and crashed
    fits dropped, order restored."""
    from mfvi_dip_mia_amd import fanout
    assert fanout.assign(7, ["a", "b", "c"]) == {"a": [0, 3, 6], "b": [1, 4], "c": [2, 5]}
    cands = [dict(temp=float(t), sigma=0.5) for t in range(1, 8)]
    jobs = [dict(c, img=im) for im in ("a", "b") for c in cands]          # 14 independent fits = 2 images x 7 candidates
    devices = ["cpu:0", "cpu:1", "cpu:2"]
    results, dropped = fanout.run_jobs(jobs, devices, "fanout_stub:fit", dict(scale=2.0))
    assert [i for i, _, _ in results] == [i for i in range(14) if jobs[i]["temp"] not in (3.0, 5.0)]
    for i, job, psnr in results:
        assert psnr == 2.0 * (10.0 * job["temp"] + 0.5) + (100.0 if job["img"] == "b" else 0.0)
    assert sorted(i for i, _, _ in dropped) == [2, 4, 9, 11]
    assert {why for _, job, why in dropped if job["temp"] == 3.0} == {"nan"} and all("boom" in why for _, job, why in dropped if job["temp"] == 5.0)
    place = fanout.run_jobs.last_placement
    assert all(place[i][0] == devices[i % 3] for i in range(14))           # zip(jobs, itertools.cycle(devices))
    pids = {}
    for i, (dev, pid) in place.items():
        pids.setdefault(dev, set()).add(pid)
    assert all(len(v) == 1 for v in pids.values()) and len({next(iter(v)) for v in pids.values()}) == 3     # one process per device
    assert os.getpid() not in {next(iter(v)) for v in pids.values()}


def test_fanout_survives_a_worker_that_dies_hard():
    """A worker process that exits without running its `finally` (os._exit here; a GPU fault, SIGSEGV or the OOM killer in production)
    sends no sentinel: run_jobs must return — the jobs that worker had not reported are dropped with its exit code, the jobs it had
    finished and those of the other workers are kept (the reference joins its children: bayesian_optimization.py:3764-3781).  A worker
    whose set-up fails (a device that is not on the box) reports the reason for each of its jobs."""
    import time
    from mfvi_dip_mia_amd import fanout
    # device cpu:0 gets jobs 0, 2, 4 (temps 1, 9, 2): it finishes job 0, dies inside job 2 and never reaches job 4; cpu:1 gets 1, 3, 5
    jobs = [dict(temp=t, sigma=0.5) for t in (1.0, 4.0, 9.0, 6.0, 2.0, 8.0)]
    t0 = time.time()
    results, dropped = fanout.run_jobs(jobs, ["cpu:0", "cpu:1"], "fanout_stub:fit", dict(scale=1.0), poll_seconds=0.2)
    assert time.time() - t0 < 60
    assert [i for i, _, _ in results] == [0, 1, 3, 5]
    assert sorted(i for i, _, _ in dropped) == [2, 4]
    assert all("died" in why and "exit code 3" in why for _, _, why in dropped)
    # set-up failure: torch.cuda.set_device on a box without that GPU (or without any): reported per job, nothing hangs
    results, dropped = fanout.run_jobs(jobs[:2], ["cuda:63"], "fanout_stub:fit", dict(scale=1.0), poll_seconds=0.2)
    assert results == [] and len(dropped) == 2 and all("worker setup failed" in why for _, _, why in dropped)


def test_runner_config_devices_and_jobs(tmp_path):
    """load_config keeps the reference's run_params.devices (eval_result.py:21-22) for the fan-out; the CLI's job list is images x candidates."""
    import json
    from mfvi_dip_mia_amd import runner
    cfg = dict(bo_params=dict(temp=dict(candidates=[1e-6, 2e-6]), sigma=dict(candidates=[1e-5])),
               run_params=dict(img="phantom", num_iter=3, devices=["cuda:0", "cuda:1"], bo_results_path="x", lr=1e-3))
    p = tmp_path / "c.json"; p.write_text(json.dumps(cfg))
    cands, rp, devices = runner.load_config(str(p), "mfvi", with_devices=True)
    assert devices == ["cuda:0", "cuda:1"] and "devices" not in rp and "bo_results_path" not in rp
    assert cands == [dict(temp=1e-6, sigma=1e-5), dict(temp=2e-6, sigma=1e-5)]
    assert runner.load_config(str(p), "mfvi") == (cands, rp)


@pytest.mark.parametrize("task,method", [("den", "mfvi"), ("sr", "mcd"), ("ct", "mfvi"), ("inp", "sgld"), ("den", "dip")])
def test_artifacts_open_in_the_reference_notebooks(tmp_path, task, method):
    """save.npz as the runner writes it (mfvi_dip_mia_amd.artifacts, pure host code) replayed through the reads of
    eval_denoising.ipynb:79-82,339-342,363-364 / eval_sr.ipynb / eval_ct.ipynb / eval_inp.ipynb: np.load(allow_pickle=True), the
    `.flat[0][method]` dicts, the image keys of the task and the shape arithmetic of the error / uncertainty cells; plus the PNG set
    of plot=True (bayesian_optimization.py:201-258, :1418-1422)."""
    from mfvi_dip_mia_amd import artifacts as A
    rng = np.random.default_rng(0)
    H, W, n_it, show = 32, 48, 301, 10
    n_snap = n_it // show + 1
    C = 3 if task == "inp" else 1
    head = {"den": (rng.random((1, H, W)), rng.random((1, H, W))), "sr": (rng.random((1, H, W)), rng.random((H // 4, W // 4))),
            "ct": (rng.random((1, 1, H, W)), rng.random((1, 1, 45, W))), "inp": (rng.random((3, H, W)), (rng.random((1, H, W)) > 0.2).astype(np.float32))}[task]
    mse_c, mse_g = rng.random(n_it) * 0.02, rng.random(n_it) * 0.01
    psnrs, ssims = 20 + rng.random((n_it, 3)), rng.random((n_it, 3))
    recons, unc, ale = rng.random((n_snap, C, H, W)), rng.random((n_snap, C, H, W)) * 1e-3, rng.random((n_snap, 1, H, W)) * 1e-2
    run_dir = str(tmp_path)
    A.write_locals(run_dir, task=task, method=method, num_iter=n_it - 1)
    A.save_npz(run_dir, task, method, head, mse_c, mse_g, recons, unc, ale, psnrs, ssims)
    z = np.load(os.path.join(run_dir, "save.npz"), allow_pickle=True)
    assert set(z.files) == set(A.HEAD_KEYS[task]) | {A.MSE_KEY[task], "mse_gt", "recons", "uncerts", "uncerts_ale", "psnrs", "ssims"}
    r = A.read_like_notebooks(os.path.join(run_dir, "save.npz"), task, method)
    assert r["psnr_curve"].shape == (4, 3) and r["final_recon"].shape == (H, W)
    if method != "dip":
        assert r["errvar"].shape[-2:] == (H, W) and r["uncerts"].shape[-2:] == (H, W)
    with open(os.path.join(run_dir, "locals.txt"), "a") as f:
        A.plot_results({method: mse_c}, {method: mse_g}, {method: psnrs}, {method: ssims}, run_dir, f)
    A.snapshot_pngs(run_dir, method, 200, mse_c, mse_g, psnrs, recons[-1], None if method == "dip" else unc[-1], None if method == "dip" else ale[-1])
    if task == "sr":
        A.sr_input_png(run_dir, head[0], head[1], 4)
    want = {"mse_noisy.png", "mse_gt.png", "psnrs.png", "ssims.png", "out_avg.png", "loss_%s.png" % method} | (set() if method == "dip" else {"out_var.png", "out_ale.png"})
    want |= {"input.png"} if task == "sr" else set()
    assert want <= set(os.listdir(run_dir))
    txt = open(os.path.join(run_dir, "locals.txt")).read()
    assert "%s PSNR_max" % method in txt and "%s SSIM_max" % method in txt and "num_iter = 300" in txt


def test_gp_outer_loop_restatement():
    """bo.py restates the reference's GP outer loop (bayesian_optimization.py:3545-3880) without gpytorch / skimage — parity unpinned, so
    the checks are the model's own properties: initial hyper-parameters as gpytorch sets them, the posterior interpolating its training
    points, EI >= 0 and ~0 where the mean is low and certain, the peak finder's exclusion rules, candidates inside [0, 1]^2, the box
    normalisation, and the loop improving a synthetic objective through the candidate -> evaluate -> refit cycle."""
    import math
    import torch
    from mfvi_dip_mia_amd import bo as B
    torch.manual_seed(0)
    X = torch.rand(12, 2, dtype=torch.float64)
    f = lambda x: 20.0 + 5.0 * torch.exp(-((x - torch.tensor([0.7, 0.3])) ** 2).sum(-1) / 0.05)
    Y = f(X)
    gp0 = B.ExactGP(X, Y)
    assert abs(gp0.lengthscale.item() - 0.3) < 1e-12 and abs(gp0.outputscale.item() - math.log(2)) < 1e-12
    assert abs(gp0.noise.item() - (math.log(2) + 1e-4)) < 1e-12 and gp0.raw_mean.item() == 0.0
    l0 = gp0.neg_mll().item()
    gp = B.train_gp(X, Y, iter_max=1200)
    assert gp.neg_mll().item() < l0                                   # Adam on the marginal likelihood made progress
    mu, var = gp.posterior(X)
    assert (mu - Y).abs().max().item() < 0.1 and (var >= -1e-9).all() and gp.noise.item() < 0.01     # noise-free data: the fit interpolates
    g = torch.linspace(0, 1, 100, dtype=torch.float64)
    G1, G2 = torch.meshgrid(g, g, indexing="ij")
    grid = torch.stack([G1.reshape(-1), G2.reshape(-1)]).transpose(1, 0)
    ei = B.expected_improvement(gp, grid, X)
    assert ei.shape == (10000, 1) and (ei >= 0).all() and ei.max().item() > 0
    ucb = B.upper_confidence_bound(gp, grid)
    assert ucb.shape == (10000,)
    # peak finder: two separated bumps and one inside the excluded border
    img = np.zeros((40, 40)); img[10, 12] = 5.0; img[30, 25] = 3.0; img[2, 2] = 9.0; img[11, 13] = 4.0; img[20, 20] = 0.2
    pk = B.peak_local_max(img, min_distance=5, threshold_rel=0.1, num_peaks=4)
    assert [tuple(p) for p in pk] == [(10, 12), (30, 25)]
    cands, imp, acq = B.find_candidates(gp, grid, X)
    assert acq.shape == (100, 100) and 1 <= len(cands) <= 4 and len(imp) == len(cands)
    for c in cands:
        assert c.shape == (1, 2) and (c >= 0).all() and (c <= 1).all()
    Xn = B.normalize_X(torch.tensor([[1e-3, 1e-5]], dtype=torch.float64), [-4.0, -2.0], [-6.0, -4.0])
    assert torch.allclose(Xn, torch.tensor([[(-3.0 + 4.0) / 2.0, (-5.0 + 6.0) / 2.0]], dtype=torch.float64))       # log10 first (bayesian_optimization.py:3688)
    assert torch.allclose(B.unnormalize_X(Xn, [-4.0, -2.0], [-6.0, -4.0]), torch.tensor([[1e-3, 1e-5]], dtype=torch.float64))
    # the loop on a synthetic objective; one candidate fails (NaN) in every round and must be dropped
    calls = []

    def evaluate(cl):
        calls.append(list(cl))
        out = []
        for i, (a, b) in enumerate(cl):
            y = 25.0 - 40.0 * ((math.log10(a) + 3.0) ** 2 + (math.log10(b) + 5.0) ** 2) if i != 1 else float("nan")
            out.append(((a, b), y))
        return out
    bo_params = {"temp": {"logbounds": [-4.0, -2.0], "candidates": [1e-4, 1e-2]}, "sigma": {"logbounds": [-6.0, -4.0], "candidates": [1e-6, 1e-4]}}
    Xh, Yh, nxt = B.bo(bo_params, evaluate, n_rounds=2, gp_iters=150, verbose=False)
    assert len(calls) == 2 and len(calls[0]) == 4 and len(Xh) == len(Yh) == 3 + len(calls[1]) - (1 if len(calls[1]) > 1 else 0)
    assert all(not math.isnan(y) for y in Yh) and len(nxt) >= 1
    for a, b in nxt:                                                  # proposals stay inside the log-bounds box
        assert 1e-4 * (1 - 1e-9) <= a <= 1e-2 * (1 + 1e-9) and 1e-6 * (1 - 1e-9) <= b <= 1e-4 * (1 + 1e-9)
