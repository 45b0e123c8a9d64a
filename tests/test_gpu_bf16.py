"""bfloat16 storage of mu / rho (BASELINE configs[4]: "bf16 mu/rho with fp32 KL accumulate").  The reference keeps float32 Parameters
(BayTorch/modules/module.py:45-62), so parity is defined on the VALUES: the reference's own MeanFieldVI loaded with bf16-rounded
parameters (tests/golden/full_den_128_k1_bf16.npz) must agree with the HIP path that reads the bf16 blocks directly; the update rule
(stochastic rounding, no float32 master) is the build's own and is checked against its restatement in the oracle."""
import os

import numpy as np
import pytest

from conftest import note_margin as _note

from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

DEN = dict(temp=5.656911698337764e-07, sigma=1.4616642493692077e-05, lr=1e-3)


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available()
    M_._lib.lib()
    return M_


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    _v = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    _note(_v, 'relerr')
    return _v


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dev16(a):
    """float32 array of bf16-representable values -> torch.bfloat16 device tensor (exact)."""
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda().bfloat16()
    assert torch.equal(t.float().cpu(), torch.from_numpy(np.ascontiguousarray(a, np.float32)))
    return t


def test_bf16_plan_is_bit_identical_to_f32_plan_on_the_same_values(M):
    """Same numbers, two storages: every output and gradient is bit-identical (the draw forms w = mu + softplus(rho) * eps in float32
    either way); also the w = mu branch and a net with generic-kernel layers (5x5, Cin % 4 != 0: expanded copies)."""
    from test_oracle_golden import _golden_params
    for kw, S_, seed in ((dict(input_depth=8, n_out=2, nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4)), 32, 5),
                         (dict(input_depth=6, n_out=4, nd=(8, 12), nu=(8, 12), ns=(0, 0), fd=5, need1x1_up=False, upsample_mode="nearest"), 24, 6)):
        P, zin, zout, _ = M.skip_program(S_, S_, **kw)
        n = 3
        pf = P.compile(zin, zout, n); pb = P.compile(zin, zout, n, param_dtype="bf16")
        mu = O.bf16_round(0.1 * O.normal_fill(seed, 2, 0, 0, 0, P.n_vi)); rho = O.bf16_round(-3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, P.n_vi))
        bn = np.concatenate([np.r_[1 + 0.1 * O.normal_fill(seed, 2, 7 + i, 0, 0, b["C"]), 0.1 * O.normal_fill(seed, 2, 40 + i, 0, 0, b["C"])] for i, b in enumerate(P.bns)]).astype(np.float32)
        z = dev(O.normal_fill(seed, 2, 3, 0, 0, kw["input_depth"] * S_ * S_).reshape(kw["input_depth"], S_, S_))
        d_mu, d_rho, d_bn = dev(mu), dev(rho), dev(bn)
        pad = (P.n_vi + 7) // 8 * 8
        b16 = torch.zeros(2 * pad, dtype=torch.bfloat16, device="cuda")
        b_mu, b_rho = b16[:P.n_vi], b16[pad:pad + P.n_vi]
        b_mu.copy_(dev16(mu)); b_rho.copy_(dev16(rho))
        for sample in (True, False):
            of = pf.forward(d_mu, d_rho, d_bn, z, 9, 2, 1, n, sample); ob = pb.forward(b_mu, b_rho, d_bn, z, 9, 2, 1, n, sample)
            assert torch.equal(of, ob), (kw, sample)
            dout = dev(O.normal_fill(seed, 2, 4, 0, 0, of.numel()).reshape(tuple(of.shape)))
            gf = [torch.zeros(P.n_vi, device="cuda"), torch.zeros(P.n_vi, device="cuda"), torch.zeros(P.n_bn, device="cuda")]
            gb = [torch.zeros_like(t) for t in gf]
            pf.backward(d_mu, d_rho, d_bn, z, 9, 2, 1, n, dout, *gf, sample); pb.backward(b_mu, b_rho, d_bn, z, 9, 2, 1, n, dout, *gb, sample)
            for a, b in zip(gf, gb):
                if "fd" in kw:      # the generic fp32 kernels accumulate dW with float atomics: equal up to their summation order
                    assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (kw, sample)
                else:
                    assert torch.equal(a, b), (kw, sample)
        with pytest.raises(TypeError):
            pb.forward(d_mu, d_rho, d_bn, z, 9, 2, 1, n)


def test_bf16_net_against_reference_golden(M, golden_dir):
    """The reference's MeanFieldVI + skip() with bf16-rounded mu / rho (oracle/make_golden.py --fullsize) vs the plan that reads bf16:
    image, NLL, KL, ELBO at 1e-4; gradients as in test_full_net_against_reference_golden."""
    from test_oracle_golden import _golden_params
    g = np.load(os.path.join(golden_dir, "full_den_128_k1_bf16.npz"))
    size = 128
    net = O.make_net(size, size)
    seed, step = int(g["seed"]), int(g["step"])
    mu, rho, bnp = _golden_params(net, seed)
    mu, rho = O.bf16_round(mu), O.bf16_round(rho)
    P, zin, out_id, _ = M.skip_program(size, size)
    plan = P.compile(zin, out_id, max_samples=1, param_dtype="bf16")
    z = dev((0.1 * O.uniform_fill(seed, 0, 0, 0, 16 * size * size)).reshape(16, size, size))
    tgt = dev(O.noisy(O.phantom(size, size, seed), 0.1, seed))
    b_mu, b_rho, d_bn = dev16(mu), dev16(rho), dev(bnp)
    out = plan.forward(b_mu, b_rho, d_bn, z, seed, step, 0, 1)
    assert relerr(out.cpu().numpy(), g["out"]) < 1e-4 and relerr(out.cpu().numpy(), g["out_f64"]) < 1e-4
    L = M._lib; lib = L.lib()
    nll = torch.zeros(1, dtype=torch.float64, device="cuda"); klv = torch.zeros(1, dtype=torch.float64, device="cuda")
    dout = torch.empty_like(out)
    L.check(lib.mfvi_gaussian_nll(L.ptr(out), L.ptr(tgt), 1, size, size, 1, 1.0, L.ptr(dout), L.ptr(nll), L.stream_ptr()))
    f_mu, f_rho = b_mu.float(), b_rho.float()
    L.check(lib.mfvi_kl(L.ptr(f_mu), L.ptr(f_rho), P.n_vi, 0.0, float(g["prior_sigma"]), L.ptr(klv), L.stream_ptr()))
    temp = float(g["temp"])
    assert abs(float(nll) - float(g["nll"])) < 1e-4 * abs(float(g["nll"]))
    assert abs(float(klv) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    assert abs(float(nll) + temp * float(klv) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    dmu = torch.zeros(P.n_vi, device="cuda"); drho = torch.zeros(P.n_vi, device="cuda"); dbn = torch.zeros(P.n_bn, device="cuda")
    plan.backward(b_mu, b_rho, d_bn, z, seed, step, 0, 1, dout, dmu, drho, dbn)
    L.check(lib.mfvi_kl_backward(L.ptr(f_mu), L.ptr(f_rho), P.n_vi, 0.0, float(g["prior_sigma"]), temp, L.ptr(dmu), L.ptr(drho), L.stream_ptr()))
    st = max(1, P.n_vi // 4096)
    gms, grs = dmu.cpu().numpy()[::st][:4096], drho.cpu().numpy()[::st][:4096]
    rel2 = lambda a, b: float(np.linalg.norm(np.float64(a) - b) / np.linalg.norm(b))
    assert rel2(gms, g["dmu_s_f64"]) < 2e-3 and rel2(grs, g["drho_s_f64"]) < 2e-3 and rel2(dbn.cpu().numpy(), g["dbn_f64"]) < 4e-3
    assert relerr(gms, g["dmu_s_f64"]) < 5e-2 and relerr(grs, g["drho_s_f64"]) < 5e-2


def test_bf16_update_against_oracle(M):
    """mfvi_elbo_update_bf16 = KL (fp64 sum of fp32 terms) + KL gradient + Adam in fp32 + stochastic rounding from RNG domain 6.
    Against the oracle: KL and moments to 1e-6; the rounded parameters agree except where the fp32 update differs in its last
    bits (a 1-ulp fp32 difference flips the rounding with probability 2^-16): a handful of one-bf16-ulp differences."""
    L = M._lib; lib = L.lib()
    n_vi, n_bn, seed = 100004, 96, 17
    mu = O.bf16_round(0.1 * O.normal_fill(seed, 2, 0, 0, 0, n_vi)); rho = O.bf16_round(-3 + 0.1 * O.normal_fill(seed, 2, 1, 0, 0, n_vi))
    bn = (1 + 0.1 * O.normal_fill(seed, 2, 2, 0, 0, n_bn)).astype(np.float32)
    pad = (n_vi + 7) // 8 * 8
    b16 = torch.zeros(2 * pad, dtype=torch.bfloat16, device="cuda"); b_mu, b_rho = b16[:n_vi], b16[pad:pad + n_vi]
    b_mu.copy_(dev16(mu)); b_rho.copy_(dev16(rho)); d_bn = dev(bn)
    m = torch.zeros(2 * n_vi + n_bn, device="cuda"); v = torch.zeros_like(m)
    scratch = torch.zeros(lib.mfvi_elbo_update_scratch_bytes(), dtype=torch.uint8, device="cuda")
    kl = torch.zeros(1, dtype=torch.float64, device="cuda")
    prior_sigma, temp, lr = 0.05, 3e-4, 1e-3
    om_bits, or_bits = O.bf16_bits(mu), O.bf16_bits(rho)
    om = np.zeros(2 * n_vi + n_bn, np.float32); ov = np.zeros_like(om); obn = bn.copy()
    moved = 0
    for t in range(1, 4):
        g = (1e-3 * O.normal_fill(seed, 2, 10 + t, 0, 0, 2 * n_vi + n_bn)).astype(np.float32)
        d_g = dev(g)
        L.check(lib.mfvi_elbo_update_bf16(L.ptr(b_mu), L.ptr(b_rho), L.ptr(d_bn), L.ptr(d_g), L.ptr(m), L.ptr(v), n_vi, n_bn, 0.0, prior_sigma, temp, lr,
                                          0.9, 0.999, 1e-8, t, seed, L.ptr(kl), L.ptr(scratch), L.stream_ptr()))
        fmu, frho = O.bf16_from_bits(om_bits), O.bf16_from_bits(or_bits)
        okl, dkm, dkr = O.kl(fmu, frho, prior_sigma, scale=temp, want_grad=True)
        gg = g.copy(); gg[:n_vi] += dkm; gg[n_vi:2 * n_vi] += dkr
        assert abs(float(kl) - okl) < 1e-6 * abs(okl)                     # fp32 terms summed in fp64 vs the oracle's all-double sum
        assert relerr(d_g.cpu().numpy(), gg) < 2e-6                    # grads += temp * dKL, written back
        before = om_bits.copy()
        O.adam_bf16_sr(om_bits, np.ascontiguousarray(gg[:n_vi]), om[:n_vi], ov[:n_vi], lr, t, seed, 0)
        O.adam_bf16_sr(or_bits, np.ascontiguousarray(gg[n_vi:2 * n_vi]), om[n_vi:2 * n_vi], ov[n_vi:2 * n_vi], lr, t, seed, 1)
        O.adam(obn, np.ascontiguousarray(gg[2 * n_vi:]), om[2 * n_vi:], ov[2 * n_vi:], lr, t)
        moved += int((before != om_bits).sum())
        for dev_t, bits in ((b_mu, om_bits), (b_rho, or_bits)):
            got = dev_t.view(torch.int16).cpu().numpy().view(np.uint16)
            diff = np.abs(got.astype(np.int64) - bits.astype(np.int64))
            assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (t, diff.max(), (diff != 0).mean())
            bits[:] = got                                             # re-anchor on the device's rounding decisions
        assert relerr(m.cpu().numpy(), om) < 1e-6 and relerr(v.cpu().numpy(), ov) < 1e-6 and relerr(d_bn.cpu().numpy(), obn) < 1e-6
    assert moved > 0.3 * n_vi      # lr = 1e-3 against ulp_bf16(0.1) = 4.9e-4: the parameters do move


def test_bf16_stochastic_rounding_is_unbiased(M):
    """A constant gradient far below one bf16 ulp still moves rho in expectation: after 200 updates the mean drift of 64k copies of
    rho = -3 equals the float32 Adam drift (200 * lr) within 2 %, where round-to-nearest would not move at all."""
    L = M._lib; lib = L.lib()
    n = 1 << 16
    b16 = torch.zeros(2 * n, dtype=torch.bfloat16, device="cuda"); b_mu, b_rho = b16[:n], b16[n:]
    b_rho.fill_(-3.0); b_mu.fill_(0.125)
    m = torch.zeros(2 * n, device="cuda"); v = torch.zeros_like(m); bn = torch.zeros(1, device="cuda")
    scratch = torch.zeros(lib.mfvi_elbo_update_scratch_bytes(), dtype=torch.uint8, device="cuda"); kl = torch.zeros(1, dtype=torch.float64, device="cuda")
    for t in range(1, 201):
        g = torch.full((2 * n,), 1e-2, device="cuda")
        L.check(lib.mfvi_elbo_update_bf16(L.ptr(b_mu), L.ptr(b_rho), L.ptr(bn), L.ptr(g), L.ptr(m), L.ptr(v), n, 0, 0.0, 0.05, 0.0, 1e-3, 0.9, 0.999, 1e-8, t, 5,
                                          L.ptr(kl), L.ptr(scratch), L.stream_ptr()))
    drift_rho = float((b_rho.float() + 3.0).mean()); drift_mu = float((b_mu.float() - 0.125).mean())
    assert abs(drift_rho + 0.2) < 0.004 and abs(drift_mu + 0.2) < 0.004, (drift_rho, drift_mu)


def test_cfg5_den_512_k64_bf16(M):
    """configs[4] as stated: one 512x512 denoising fit with K = 64 MC samples (4 launches of 16) and bf16 mu / rho.  The gradient equals
    the float32 engine's on the same (bf16-representable) values bit for bit, chunking does not change it, and the fit descends."""
    S2, K2 = 512, 64
    tgt = torch.from_numpy(O.noisy(O.phantom(S2, S2, 1), 0.1, 1))
    def eng(spl, dtype):
        e = M.engine.ElboEngine(S2, S2, task="den", K=K2, input_depth=16, seed=1, samples_per_launch=spl, autotune=False, param_dtype=dtype, **DEN)
        e.set_target(tgt); return e
    a = eng(16, "bf16"); a.grad_only(step=2)
    ga = a.grads[:a.n_params].clone(); la = a.losses()
    f = eng(16, "f32")
    mu32, rho32, _ = a.params_f32()
    f.mu.copy_(mu32); f.rho.copy_(rho32)
    f.grad_only(step=2)
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    # same arithmetic on the same values; only the fp64 atomics of the BN / loss sums may land in a different order
    assert rel(f.grads[:a.n_params], ga) < 1e-6 and abs(f.losses()[0] - la[0]) < 1e-10 * abs(la[0]) and abs(f.losses()[1] - la[1]) < 1e-10 * abs(la[1])
    del f
    b = eng(8, "bf16"); b.grad_only(step=2)
    assert rel(b.grads[:a.n_params], ga) < 2e-4 and abs(b.losses()[0] - la[0]) < 1e-5 * abs(la[0]) + 1e-7
    del b
    losses = []
    for _ in range(6):
        a.step(); losses.append(a.losses()[2])
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert a.mu.dtype == torch.bfloat16 and a.rho.dtype == torch.bfloat16 and a.params is None
