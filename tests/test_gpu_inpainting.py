"""The inpainting MFVI variant (bayesian_optimization.py:2892-3114): skip() without skip branches, 5x5
down filters, nearest upsampling, no 1x1 up convs, 4 output channels, sigmoid + masked Gaussian NLL.  The three new ops are
checked against the oracle; the whole net against a float64 PyTorch interpreter of the same layer program (autograd through
w = mu + softplus(rho) * eps with the oracle's eps, train-mode BatchNorm, LeakyReLU, reflection pad, F.interpolate)."""
import numpy as np
import pytest

from conftest import note_margin as _note

from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available()
    M_._lib.lib()
    return M_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    _v = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    _note(_v, 'relerr')
    return _v


def torch_program(P, zin, out_id, mu, rho, bn, z, seed, step, sample):
    """float64 PyTorch interpreter of a layer program (one MC sample); mu/rho/bn/z are leaf tensors."""
    vals = {zin: z}

    def view(t):
        x, d = vals[t], P.tensors[t]
        if d["has_bn"]:
            c, off = d["C"], d["bn_off"]
            x = F.batch_norm(x[None], None, None, bn[off:off + c], bn[off + c:off + 2 * c], True, 0.0, d["eps"])[0]
        if d["has_act"]:
            x = F.leaky_relu(x, d["slope"])
        return x

    for op in P.ops:
        if op["type"] == 1:
            lay = P.layers[op["layer_id"]]
            cin, cout, k = lay["cin"], lay["cout"], lay["k"]
            nw = cout * cin * k * k
            ew = torch.from_numpy(O.eps(seed, step, sample, op["layer_id"], 0, nw).astype(np.float64))
            eb = torch.from_numpy(O.eps(seed, step, sample, op["layer_id"], 1, cout).astype(np.float64))
            w = (mu[lay["w_off"]:lay["w_off"] + nw] + F.softplus(rho[lay["w_off"]:lay["w_off"] + nw]) * ew).reshape(cout, cin, k, k)
            b = mu[lay["b_off"]:lay["b_off"] + cout] + F.softplus(rho[lay["b_off"]:lay["b_off"] + cout]) * eb
            x = view(op["in0"])[None]
            if k > 1:
                x = F.pad(x, (k // 2,) * 4, mode="reflect")
            vals[op["out"]] = F.conv2d(x, w, b, stride=lay["stride"])[0]
        else:
            mode = "nearest" if op["up_mode"] == 1 else "bilinear"
            up = F.interpolate(view(op["in1"])[None], scale_factor=2, mode=mode, **({} if mode == "nearest" else {"align_corners": False}))[0]
            vals[op["out"]] = up if op["in0"] < 0 else torch.cat([view(op["in0"]), up], 0)
    return vals[out_id]


def torch_inp_nll(out4, target, mask):
    mu = torch.sigmoid(out4[:3]); s = torch.clamp(out4[3:], -20, 20)
    return ((torch.exp(s) * (target - mu) ** 2 - s) * mask).mean()


def test_masked_sigmoid_nll_and_nearest_upsample_vs_oracle(M):
    L = M._lib
    rng = np.random.default_rng(3)
    n, H, W = 3, 12, 20
    out = (2.0 * rng.standard_normal((n, 4, H, W))).astype(np.float32); out[0, 3, 0, :4] = [25.0, -30.0, 19.9, 0.0]      # clamp branches
    tgt = rng.random((3, H, W)).astype(np.float32)
    for mc in (1, 3):
        mask = (rng.random((mc, H, W)) > 0.3).astype(np.float32)
        dout = torch.empty((n, 4, H, W), device="cuda"); acc = torch.zeros(1, dtype=torch.float64, device="cuda")
        d_out, d_tgt, d_mask = dev(out), dev(tgt), dev(mask)          # keep the device buffers alive across the asynchronous launch
        L.check(L.lib().mfvi_gaussian_nll_inpainting(L.ptr(d_out), L.ptr(d_tgt), L.ptr(d_mask), mc, n, H, W, 0.5, L.ptr(dout), L.ptr(acc),
                                                     L.stream_ptr()))
        ref = [O.gaussian_nll_inp(out[i], tgt, mask, scale=0.5, want_grad=True) for i in range(n)]
        assert abs(float(acc) - sum(r[0] for r in ref)) < 1e-5 * abs(sum(r[0] for r in ref))
        for i in range(n):
            assert relerr(dout[i].cpu().numpy(), ref[i][1]) < 5e-6
    # nearest upsample (+ the BatchNorm that follows it) through a plan: conv -> BN/act -> Upsample(nearest) -> BN -> conv
    P = M.Program()
    zin = P.tensor(4, 6, 10)
    a = P.tensor(8, 6, 10); P.conv(zin, a, 3, 1); P.set_bn(a, act=True)
    u = P.tensor(8, 12, 20); P.concat_up(None, a, u, "nearest"); P.set_bn(u, act=False)
    o = P.tensor(3, 12, 20); P.conv(u, o, 3, 1)
    plan = P.compile(zin, o, 2)
    assert relerr(O.upsample2_nearest_bwd(O.upsample2_nearest_fwd(out[0, :, :6, :10])), 4 * out[0, :, :6, :10]) < 1e-6
    _check_program(M, P, plan, zin, o, seed=11, n=2, loss=lambda y: (y ** 2).sum() * 0.01)


def _check_program(M, P, plan, zin, out_id, seed, n, loss, tol=5e-4):
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi)
    g = O.normal_fill(seed, 2, 7, 0, 0, max(P.n_bn, 1))
    bn = np.zeros(max(P.n_bn, 1), np.float32)
    for b in P.bns:
        c, off = b["C"], b["off"]
        bn[off:off + c] = 1.0 + 0.1 * g[off:off + c]; bn[off + c:off + 2 * c] = 0.1 * g[off + c:off + 2 * c]
    t0 = P.tensors[zin]
    z = O.normal_fill(seed, 2, 2, 0, 0, t0["C"] * t0["H"] * t0["W"]).reshape(t0["C"], t0["H"], t0["W"])
    d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bn), dev(z)
    y = plan.forward(d_mu, d_rho, d_bn, d_z, seed, 4, 0, n)
    tm, tr, tb, tz = (torch.tensor(v.astype(np.float64), requires_grad=True) for v in (mu, rho, bn, z))
    outs = [torch_program(P, zin, out_id, tm, tr, tb, tz, seed, 4, k) for k in range(n)]
    for k in range(n):
        assert relerr(y[k].cpu().numpy(), outs[k].detach().numpy()) < 5e-5, ("forward", k)
    yt = y.detach().clone().requires_grad_(True)
    loss(yt).backward()
    sum(loss(o[None])[()] if False else loss(o[None]) for o in outs).backward()
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
    dz = torch.empty((n,) + z.shape, device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, 4, 0, n, yt.grad.contiguous(), dmu, drho, dbn, dz=dz)
    assert relerr(dmu.cpu().numpy(), tm.grad.numpy()) < tol
    assert relerr(drho.cpu().numpy(), tr.grad.numpy()) < tol
    if P.n_bn:
        assert relerr(dbn.cpu().numpy(), tb.grad.numpy()) < tol
    assert relerr(dz.sum(0).cpu().numpy(), tz.grad.numpy()) < tol


def test_inpainting_net_vs_float64_torch(M):
    H = W = 24
    P, zin, out, _ = M.program.skip_program(H, W, input_depth=8, n_out=4, nd=(8, 16, 16), nu=(8, 16, 16), ns=(0, 0, 0), fd=5, fu=3,
                                            need1x1_up=False, upsample_mode="nearest")
    assert [l["k"] for l in P.layers] == [5, 5, 5, 5, 5, 5, 3, 3, 3, 1] and all(o["in0"] < 0 for o in P.ops if o["type"] == 2)
    plan = P.compile(zin, out, 2)
    rng = np.random.default_rng(5)
    tgt = torch.tensor(rng.random((3, H, W))); mask = torch.tensor((rng.random((1, H, W)) > 0.25).astype(np.float64))
    _check_program(M, P, plan, zin, out, seed=21, n=2,
                   loss=lambda y: sum(torch_inp_nll(y[i].double().cpu() if y.is_cuda else y[i], tgt, mask) for i in range(y.shape[0])).to(y.device) * 0.5)


def test_inpainting_engine_runs_and_descends(M):
    H = W = 192                      # 6 stride-2 scales: the deepest 5x5 conv (reflection pad 2) needs a 3x3 map
    eng = M.engine.ElboEngine(H, W, task="inp", K=2, input_depth=16, temp=1e-7, sigma=1e-5, lr=1e-2, seed=2)
    assert eng.out.shape[1] == 4 and len(eng.prog.layers) == 19
    rng = np.random.default_rng(1)
    img = np.stack([O.phantom(H, W, s) for s in (1, 2, 3)]).astype(np.float32)
    mask = (rng.random((1, H, W)) > 0.3).astype(np.float32)
    eng.set_target(torch.from_numpy(img), torch.from_numpy(mask))
    losses = []
    for _ in range(30):
        eng.step(); losses.append(eng.losses()[0])
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])


def test_inpainting_net_against_reference_golden(M, golden_dir):
    """The reference's own skip() built with the inpainting runner's options (3 scales, 24x24) wrapped in MeanFieldVI, with
    the runner's sigmoid + gaussian_nll_inpainting: output, loss and every gradient (oracle/make_golden.py::golden_inpainting)."""
    import os
    g = np.load(os.path.join(golden_dir, "inpainting.npz"), allow_pickle=False)
    H = W = 24; seed = 21
    P, zin, zout, _ = M.program.skip_program(H, W, input_depth=8, n_out=4, nd=(8, 16, 16), nu=(8, 16, 16), ns=(0, 0, 0), fd=5, fu=3,
                                             need1x1_up=False, upsample_mode="nearest")
    plan = P.compile(zin, zout, 1)
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi)
    gg = O.normal_fill(seed, 2, 7, 0, 0, P.n_bn); bn = np.zeros(P.n_bn, np.float32)
    for b in P.bns:
        c, off = b["C"], b["off"]
        bn[off:off + c] = 1.0 + 0.1 * gg[off:off + c]; bn[off + c:off + 2 * c] = 0.1 * gg[off + c:off + 2 * c]
    z = O.normal_fill(seed, 2, 2, 0, 0, 8 * H * W).reshape(8, H, W)
    tgt = O.uniform_fill(312, 1, 0, 0, 3 * H * W).reshape(3, H, W)
    mask = (O.uniform_fill(312, 2, 0, 0, H * W).reshape(1, H, W) > 0.25).astype(np.float32)
    d_mu, d_rho, d_bn, d_z, d_t, d_m = dev(mu), dev(rho), dev(bn), dev(z), dev(tgt), dev(mask)
    out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, 4, 0, 1)
    assert relerr(out[0].cpu().numpy(), g["net_out"]) < 1e-4
    L = M._lib
    dout = torch.empty_like(out); acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    L.check(L.lib().mfvi_gaussian_nll_inpainting(L.ptr(out), L.ptr(d_t), L.ptr(d_m), 1, 1, H, W, 1.0, L.ptr(dout), L.ptr(acc), L.stream_ptr()))
    assert abs(float(acc) - float(g["net_nll"])) < 1e-4 * max(abs(float(g["net_nll"])), 1e-2)
    dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn); dz = torch.empty((1, 8, H, W), device="cuda")
    plan.backward(d_mu, d_rho, d_bn, d_z, seed, 4, 0, 1, dout, dmu, drho, dbn, dz=dz)
    for got, key in ((dmu, "net_dmu"), (drho, "net_drho"), (dbn, "net_dbn"), (dz[0], "net_dz")):
        assert relerr(got.cpu().numpy(), g[key]) < 2e-3, key              # fp32 reference, LeakyReLU kinks over 3x3 maps


def test_inpainting_dropin_wrapper_matches_reference_golden(M, golden_dir):
    """The reference's calling sequence for inpainting (bayesian_optimization.py:2970-3040) on the drop-in classes: skip(...) with
    the runner's options -> MeanFieldVI -> out[:, :3].sigmoid() + masked NLL -> backward, against the reference's own run."""
    import os
    from mfvi_dip_mia_amd.nets import skip
    g = np.load(os.path.join(golden_dir, "inpainting.npz"), allow_pickle=False)
    H = W = 24; seed = 21
    net = skip(8, num_output_channels=4, pad='reflection', num_channels_down=[8, 16, 16], num_channels_up=[8, 16, 16],
               num_channels_skip=[0, 0, 0], filter_size_down=5, filter_size_up=3, filter_skip_size=1, need1x1_up=False,
               upsample_mode='nearest', dropout_mode_down='None', dropout_mode_up='None', dropout_mode_skip='None', dropout_mode_output='None',
               need_sigmoid=False)
    net = M.MeanFieldVI(net, prior={'mu': 0.0, 'sigma': 0.1}, replace_layers='all', device=torch.device('cuda'), reparam='', seed=seed)
    assert list(net.state_dict().keys()) == [str(k) for k in g["net_keys"]]
    P, _, _, _ = M.program.skip_program(H, W, input_depth=8, n_out=4, nd=(8, 16, 16), nu=(8, 16, 16), ns=(0, 0, 0), fd=5, fu=3,
                                        need1x1_up=False, upsample_mode="nearest")
    mu = 0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi); rho = -3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi)
    gg = O.normal_fill(seed, 2, 7, 0, 0, P.n_bn); bn = np.zeros(P.n_bn, np.float32)
    for b in P.bns:
        c, off = b["C"], b["off"]
        bn[off:off + c] = 1.0 + 0.1 * gg[off:off + c]; bn[off + c:off + 2 * c] = 0.1 * gg[off + c:off + 2 * c]
    with torch.no_grad():                       # the flat buffer IS the parameters (views): same layout as the layer program
        net._flat[:P.n_vi].copy_(dev(mu)); net._flat[P.n_vi:2 * P.n_vi].copy_(dev(rho)); net._flat[2 * P.n_vi:].copy_(dev(bn))
    net._step = 4                               # the golden drew eps at step 4, sample 0
    z = dev(O.normal_fill(seed, 2, 2, 0, 0, 8 * H * W).reshape(1, 8, H, W)).requires_grad_(True)
    tgt = dev(O.uniform_fill(312, 1, 0, 0, 3 * H * W).reshape(1, 3, H, W))
    mask = dev((O.uniform_fill(312, 2, 0, 0, H * W).reshape(1, 1, H, W) > 0.25).astype(np.float32))
    out = net(z)
    assert relerr(out[0].detach().cpu().numpy(), g["net_out"]) < 1e-4
    s = torch.clamp(out[:, 3:], -20, 20)
    nll = ((torch.exp(s) * (tgt - out[:, :3].sigmoid()) ** 2 - s) * mask).mean()
    assert abs(float(nll) - float(g["net_nll"])) < 1e-4 * max(abs(float(g["net_nll"])), 1e-2)
    nll.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert relerr(z.grad[0].cpu().numpy(), g["net_dz"]) < 2e-3
    assert torch.isfinite(grads).all()
