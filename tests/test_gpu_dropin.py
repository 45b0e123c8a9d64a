"""GPU tests of the drop-in surface: written the way the reference's runner uses BayTorch
(bayesian_optimization.py:1327-1372): get_net -> MeanFieldVI -> net(x) -> gaussian_nll + temp*net.kl() -> backward -> AdamW."""
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
    assert torch.cuda.is_available()
    M_._lib.lib()
    return M_


def _load_flat(net, mu, rho, bnp):
    n = net.n_vi
    with torch.no_grad():
        net._flat[:n].copy_(torch.from_numpy(mu)); net._flat[n:2 * n].copy_(torch.from_numpy(rho)); net._flat[2 * n:].copy_(torch.from_numpy(bnp))


def test_state_dict_keys_match_reference(M, golden_dir):
    g = np.load(os.path.join(golden_dir, "full_den_64_k1.npz"))
    net = M.get_net(16, 'skip', 'reflection', skip_n33d=[16, 32, 64, 128, 128], skip_n33u=[16, 32, 64, 128, 128], skip_n11=4,
                    num_scales=5, n_channels=2, upsample_mode='bilinear')
    net = M.MeanFieldVI(net, prior={'mu': 0.0, 'sigma': 1e-6}, replace_layers='all', device=torch.device('cuda'), reparam='')
    assert list(net.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    assert net.n_vi == 1035446 and net.n_bn == 3984
    assert sum(p.numel() for p in net.parameters()) == 2074876
    assert all(p.is_leaf and p.is_cuda for p in net.parameters())
    # initial KL of the den config ~1.29e7 and per-layer _kl adds up
    net2 = M.MeanFieldVI(M.get_net(16, 'skip', 'reflection', skip_n33d=[16, 32, 64, 128, 128], skip_n33u=[16, 32, 64, 128, 128], skip_n11=4,
                                   num_scales=5, n_channels=2, upsample_mode='bilinear'),
                         prior={'mu': 0.0, 'sigma': np.sqrt(5.656911698337764e-07) * 1.4616642493692077e-05}, device=torch.device('cuda'), reparam='')
    kl = net2.kl()
    assert kl.shape == (1,) and kl.dtype == torch.float32 and 1.2e7 < float(kl) < 1.4e7
    per = sum(float(m._kl) for m in net2.modules() if hasattr(m, '_kl'))
    assert abs(per - float(kl)) < 1e-5 * float(kl)


@pytest.mark.parametrize("H,W", [(32, 32), (36, 44)])      # 36 x 44: 9 -> 5 rows, 11 -> 6 columns at the third scale, Concat crops (models/common.py:31-41)
def test_reference_style_training_step(M, H, W):
    kw = dict(H=H, W=W, input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4))
    onet = O.make_net(**kw)
    seed, temp, sig = 5, 5.656911698337764e-07, 1.4616642493692077e-05
    mu, rho, bnp = O.init_params(onet, seed)
    device = torch.device('cuda')
    net = M.get_net(8, 'skip', 'reflection', skip_n33d=[8, 16, 16], skip_n33u=[8, 16, 16], skip_n11=4, num_scales=3, n_channels=2,
                    upsample_mode='bilinear')
    prior = {'mu': 0.0, 'sigma': np.sqrt(temp) * sig}
    net = M.MeanFieldVI(net, prior=prior, replace_layers='all', device=device, reparam='', seed=seed)
    _load_flat(net, mu, rho, bnp)
    z = (0.1 * O.uniform_fill(seed, 0, 0, 0, 8 * H * W)).reshape(1, 8, H, W)
    tgt = O.noisy(O.phantom(H, W, seed), 0.1, seed)
    net_input = torch.from_numpy(z).to(device); img = torch.from_numpy(tgt)[None, None].to(device)
    optimizer = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0)
    optimizer.zero_grad()
    out = net(net_input)
    assert out.shape == (1, 2, H, W) and out.requires_grad
    nll = M.gaussian_nll(out[:, :1], out[:, 1:], img)
    kl = net.kl()
    loss = nll + temp * kl
    loss.backward()
    ps = float(np.float32(np.sqrt(temp) * sig + 1e-6))
    r = O.elbo_grad(onet, mu, rho, bnp, z[0], tgt, seed=seed, step=0, K=1, temp=temp, prior_sigma=ps, want_out=True)
    assert relerr(out.detach().cpu().numpy(), r["out"]) < 1e-4
    assert abs(float(loss) - r["loss"]) < 1e-4 * abs(r["loss"])
    n = net.n_vi
    gmu = np.zeros(n, np.float32); grho = np.zeros(n, np.float32)
    for m in net._vi:
        nw = m.W_mu.numel()
        gmu[m._w_off:m._w_off + nw] = m.W_mu.grad.cpu().numpy().ravel(); grho[m._w_off:m._w_off + nw] = m.W_rho.grad.cpu().numpy().ravel()
        gmu[m._b_off:m._b_off + m.out_channels] = m.bias_mu.grad.cpu().numpy(); grho[m._b_off:m._b_off + m.out_channels] = m.bias_rho.grad.cpu().numpy()
    assert relerr(gmu, r["dmu"]) < 3e-4 and relerr(grho, r["drho"]) < 3e-4
    optimizer.step()
    p = np.concatenate([mu, rho, bnp]); g = np.concatenate([r["dmu"], r["drho"], r["dbn"]])
    O.adam(p, g, np.zeros_like(p), np.zeros_like(p), 1e-3, 1)
    assert net._views_intact()
    assert np.abs(net._flat.cpu().numpy() - p).max() < 2.1e-3 and np.abs(net._flat.cpu().numpy() - p).mean() < 2e-5
    # second call draws fresh eps (step 1), like randn_like per call in the reference
    out2 = net(net_input)
    ref2, tape = O.net_forward(onet, *[a.cpu().numpy() for a in (net._flat[:n], net._flat[n:2 * n], net._flat[2 * n:])], z[0], seed, 1, 0)
    tape.free()
    assert relerr(out2.detach().cpu().numpy()[0], ref2) < 1e-4
    # RTLayer eval branch: w = mu
    net.set_sampling(False)
    out3 = net(net_input)
    ref3, tape = O.net_forward(onet, *[a.cpu().numpy() for a in (net._flat[:n], net._flat[n:2 * n], net._flat[2 * n:])], z[0], seed, 0, 0, sample_weights=False)
    tape.free()
    assert relerr(out3.detach().cpu().numpy()[0], ref3) < 1e-4


def test_k_samples_in_one_call_and_errors(M):
    device = torch.device('cuda')
    mk = lambda: M.get_net(8, 'skip', 'reflection', skip_n33d=[8, 16], skip_n33u=[8, 16], skip_n11=4, num_scales=2, n_channels=2, upsample_mode='bilinear')
    net = M.MeanFieldVI(mk(), prior={'mu': 0.0, 'sigma': 0.05}, device=device, reparam='', seed=3, n_samples=4)
    x = torch.rand(1, 8, 16, 16, device=device) * 0.1
    out = net(x)
    assert out.shape == (4, 2, 16, 16)
    assert float((out[0] - out[1]).abs().max()) > 0            # every sample has its own eps
    out.mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    o1 = net(x); o2 = net(x)
    with pytest.raises(RuntimeError, match="no longer the latest"):
        o1.sum().backward()
    with pytest.raises(NotImplementedError):
        M.MeanFieldVI(mk(), device=torch.device('cpu'), reparam='')
    with pytest.raises(NotImplementedError):
        M.MeanFieldVI(mk(), prior={'mu': 0, 'sigma': 0.1, 'pi': 0.5}, device=device, reparam='')


def test_batchnorm_running_statistics_and_eval_mode(M):
    """nn.BatchNorm2d as the reference's nets carry it (models/common.py:96-97): training forwards update running_mean / running_var with
    momentum 0.1 and the unbiased variance, one update per batch-1 forward (n_samples of them per call), num_batches_tracked counts them,
    and .eval() normalises with those statistics and w = mu — all compared with the same net executed by plain torch modules."""
    import copy
    import torch.nn.functional as F
    device = torch.device('cuda')
    torch.manual_seed(0)
    mk = lambda: M.get_net(8, 'skip', 'reflection', skip_n33d=[8, 16], skip_n33u=[8, 16], skip_n11=4, num_scales=2, n_channels=2, upsample_mode='bilinear')
    plain = mk().to(device)                                              # torch executes this copy (deterministic weights = the mu of the wrapped one)
    net = M.MeanFieldVI(copy.deepcopy(plain), prior={'mu': 0.0, 'sigma': 0.05}, device=device, reparam='', seed=3, n_samples=2)
    net.set_sampling(False)                                              # w = mu: comparable with the plain net
    convs_p = [m for m in plain.modules() if isinstance(m, torch.nn.Conv2d)]
    convs_w = [m for m in net.modules() if hasattr(m, "W_mu")]
    with torch.no_grad():
        for a, b in zip(convs_p, convs_w):
            a.weight.copy_(b.W_mu); a.bias.copy_(b.bias_mu)
    x = torch.rand(1, 8, 32, 32, device=device)
    for _ in range(3):
        out = net(x)                                                     # 2 forwards per call
        for _ in range(2):
            ref = plain(x)
    assert float((out[0] - ref[0]).abs().max()) < 1e-4 * float(ref.abs().max())
    bn_p = [m for m in plain.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    bn_w = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for a, b in zip(bn_p, bn_w):
        assert int(b.num_batches_tracked) == 6 == int(a.num_batches_tracked)
        assert float((a.running_mean - b.running_mean).abs().max()) < 1e-5 * max(1.0, float(a.running_mean.abs().max()))
        assert float((a.running_var - b.running_var).abs().max()) < 1e-4 * float(a.running_var.abs().max())
    sd = net.state_dict()
    assert all(k in sd for k in ("net.1.0.1.1.running_mean",)) or any(k.endswith("running_var") for k in sd)
    plain.eval(); net.eval()
    with torch.no_grad():
        o_eval = net(x); r_eval = plain(x)
    assert float((o_eval[0] - r_eval[0]).abs().max()) < 1e-4 * float(r_eval.abs().max())
    assert int(bn_w[0].num_batches_tracked) == 6                         # eval forwards leave the statistics alone
    y_eval = net(x)                                                      # plain `net.eval(); y = net(x)` outside no_grad: forward-only use works
    assert torch.equal(y_eval.detach(), o_eval)
    with pytest.raises(NotImplementedError, match="eval mode"):          # ... and the backward nobody built is refused where it is asked for
        y_eval.sum().backward()
    net.train()
    assert net(x).shape == (2, 2, 32, 32)


@pytest.mark.parametrize("method", ["dip", "mcd", "sgld"])
def test_fusednet_plain_nets_follow_the_reference_loops(M, golden_dir, method):
    """The reference's calling sequence for its non-Bayesian runs (bayesian_optimization.py:1140-1181 DIP, 1526-1581 MC dropout,
    1739-1786 SGLD) on the drop-in classes: get_net(...) [with the '2d' dropout options] -> FusedNet -> torch AdamW / ExponentialLR,
    add_noise by re-assigning .data.  Checked against the golden run of the reference's own modules (tests/golden/siblings.npz)."""
    import torch.nn.functional as F
    g = np.load(os.path.join(golden_dir, "siblings.npz"))
    H, W, seed, lr, steps = int(g["H"]), int(g["W"]), int(g["seed"]), float(g["lr"]), int(g["steps"])
    p, wd, gamma = float(g[method + "_p"]), float(g[method + "_wd"]), float(g[method + "_gamma"])
    mode = '2d' if p else 'None'
    net = M.get_net(8, 'skip', 'reflection', 'bilinear', n_channels=2, skip_n33d=[8, 16, 16], skip_n33u=[8, 16, 16], skip_n11=4, num_scales=3,
                    dropout_mode_down=mode, dropout_p_down=p, dropout_mode_up=mode, dropout_p_up=p)
    net = M.FusedNet(net, device=torch.device('cuda'), seed=seed)
    onet = O.make_net(H, W, input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4))
    conv, bn, n_vi, n_bnp = O.net_table(onet)
    mu, _, bnp = O.init_params(onet, seed)
    gg = O.normal_fill(seed, 2, 7, 0, 0, n_bnp)
    for c, off in bn:
        bnp[off:off + c] = 1.0 + 0.1 * gg[off:off + c]; bnp[off + c:off + 2 * c] = 0.1 * gg[off + c:off + 2 * c]
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    with torch.no_grad():
        for m, r in zip(convs, conv):
            cin, cout, k, s_, wo, bo = [int(v) for v in r]
            m.weight.copy_(torch.from_numpy(mu[wo:wo + cout * cin * k * k].reshape(cout, cin, k, k))); m.bias.copy_(torch.from_numpy(mu[bo:bo + cout]))
        for m, (c, off) in zip(bns, bn):
            c, off = int(c), int(off)
            m.weight.copy_(torch.from_numpy(bnp[off:off + c])); m.bias.copy_(torch.from_numpy(bnp[off + c:off + 2 * c]))
    assert [k for k in net.state_dict() if "Conv2d" in k][:2] == ["net.Concat_up_5.0.Sequential_skip_1.Conv2d_skip_1.weight",
                                                                  "net.Concat_up_5.0.Sequential_skip_1.Conv2d_skip_1.bias"]
    z0 = (0.1 * O.uniform_fill(seed, 0, 0, 0, 8 * H * W)).reshape(1, 8, H, W)
    tgt = torch.from_numpy(O.noisy(O.phantom(H, W, seed), 0.1, seed))[None, None].cuda()
    opt = torch.optim.AdamW(net.parameters(), lr=lr, weight_decay=wd)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma) if method == "sgld" else None
    losses = []
    for it in range(steps):
        opt.zero_grad()
        if method == "sgld":                      # add_noise(net, 2, LR) (bayesian_optimization.py:166-170) with the spec's noise
            for lid, m in enumerate(convs):
                nz = torch.from_numpy(O.normal_fill(seed, 4, lid, 0, it, m.weight.numel()).reshape(tuple(m.weight.shape))).cuda()
                m.weight.data = m.weight.data + nz * 2 * lr
        z = torch.from_numpy(z0 + 0.1 * O.normal_fill(seed, 1, 0, 0, it, z0.size).reshape(z0.shape)).cuda()
        out = net(z)
        loss = M.gaussian_nll(out[:, :1], out[:, 1:], tgt) if method == "mcd" else F.mse_loss(out[:, :1], tgt)
        loss.backward()
        if it == 0:
            assert relerr(out.detach().cpu().numpy()[0], g[method + "_out0"]) < 1e-4
            gmu = np.zeros(n_vi, np.float32)
            for m, r in zip(convs, conv):
                cin, cout, k, s_, wo, bo = [int(v) for v in r]
                gmu[wo:wo + cout * cin * k * k] = m.weight.grad.cpu().numpy().ravel(); gmu[bo:bo + cout] = m.bias.grad.cpu().numpy()
            assert relerr(gmu, g[method + "_dmu0"]) < 2e-4
        opt.step()
        if sched is not None and sched.get_last_lr()[0] > 1e-8:
            sched.step()
        losses.append(float(loss))
    gl = g[method + "_loss"]
    assert np.abs(np.array(losses) - gl).max() < 2e-4 * np.abs(gl).max(), (losses, gl)
    fmu = np.zeros(n_vi, np.float32)
    for m, r in zip(convs, conv):
        cin, cout, k, s_, wo, bo = [int(v) for v in r]
        fmu[wo:wo + cout * cin * k * k] = m.weight.detach().cpu().numpy().ravel(); fmu[bo:bo + cout] = m.bias.detach().cpu().numpy()
    assert np.abs(fmu - g[method + "_mu"]).max() < 2 * steps * lr and np.abs(fmu - g[method + "_mu"]).mean() < 3e-6
    if method == "mcd":                           # nn.Dropout2d is the identity in eval mode (BatchNorm stays in train mode, as everywhere here)
        for d in net._drops:
            d.eval()
        a = net(z).detach(); b = net(z).detach()
        assert torch.equal(a, b)


@pytest.mark.parametrize("mode,H,W", [("nearest", 36, 44), ("bilinear", 36, 44), ("nearest", 35, 45), ("bilinear", 35, 45), ("nearest", 33, 47)])
def test_fusednet_odd_sizes_against_plain_torch(M, mode, H, W):
    """Sizes that are not multiples of 2^scales: `Concat` centre-crops the up-sampled branch (models/common.py:31-41).  The same module tree
    runs as plain PyTorch on the CPU in float64 (get_net builds ordinary nn.Modules; its Concat crops as the reference's does) and through
    FusedNet on the HIP kernels: forward and every convolution's weight gradient agree — an oracle-independent check of the cropped
    up-sampling and its adjoint, in both up-sampling modes, including odd widths (generic fp32 kernels).
    Gradients: LeakyReLU's kink makes any float32 run (PyTorch's own included: 4.6e-2 from its float64 run at seed 3, 35 x 45, where the HIP
    result is within 2e-6) jump when one BatchNorm output lies within rounding of zero, and the deepest BatchNorm here normalises over 5 x 6
    pixels; hence float64 as the arbiter, a tight relative L2 bound and a loose max-norm one (scripts/dev/odd_probe.py sweeps seeds)."""
    torch.manual_seed(4)
    mk = lambda: M.get_net(8, 'skip', 'reflection', mode, n_channels=2, skip_n33d=[8, 16, 16], skip_n33u=[8, 16, 16], skip_n11=4, num_scales=3)
    ref32 = mk()
    ref = mk().double(); ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in ref32.state_dict().items()})
    base = mk(); base.load_state_dict(ref32.state_dict())
    hip = M.FusedNet(base, device=torch.device('cuda'), seed=1)
    x = torch.rand(1, 8, H, W) * 0.1
    o_ref = ref(x.double())
    o_hip = hip(x.cuda())
    assert tuple(o_hip.shape) == tuple(o_ref.shape) == (1, 2, H, W)
    assert relerr(o_hip.detach().cpu().numpy(), o_ref.detach().numpy()) < 2e-5
    g = torch.randn(o_ref.shape)
    o_ref.backward(g.double()); o_hip.backward(g.cuda())
    cr = [m for m in ref.modules() if isinstance(m, torch.nn.Conv2d)]
    ch = [m for m in hip.modules() if isinstance(m, torch.nn.Conv2d)]
    assert len(cr) == len(ch) > 0
    for a, b in zip(cr, ch):
        ga, gb = a.weight.grad.numpy(), b.weight.grad.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(gb - ga) < 5e-3 * np.linalg.norm(ga) and relerr(gb, ga) < 5e-2, (mode, tuple(a.weight.shape))
