"""Size-independent properties of the hot path at BASELINE.json's bench configuration (256x256, K=16 MC samples, the
26-layer den net): the oracle is too slow there, so these check invariants the algorithm must satisfy at any size."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

S, K, SEED = 256, 16, 1
DEN = dict(temp=5.656911698337764e-07, sigma=1.4616642493692077e-05, lr=1e-3)


@pytest.fixture(scope="module")
def M():
    import mfvi_dip_mia_amd as M_
    assert torch.cuda.is_available()
    M_._lib.lib()
    return M_


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _engine(M, **kw):
    eng = M.engine.ElboEngine(S, S, task="den", K=K, input_depth=16, seed=SEED, **DEN, **kw)
    eng.set_target(torch.from_numpy(O.noisy(O.phantom(S, S, SEED), 0.1, SEED)))
    return eng


def test_chunking_and_sharding_do_not_change_the_gradient(M):
    """eps is keyed by the GLOBAL sample index: one launch of 16 samples, 4 launches of 4, and two 'ranks' of 8 whose
    gradient buffers are summed (what the all-reduce does) give the same ELBO gradient and loss."""
    one = _engine(M)
    one.grad_only(step=3)
    g_ref = one.grads[:one.n_params].clone(); nll_ref = one.losses()[0]
    chunked = _engine(M, samples_per_launch=4, autotune=False)
    chunked.grad_only(step=3)
    assert rel(chunked.grads[:one.n_params], g_ref) < 2e-4 and abs(chunked.losses()[0] - nll_ref) < 1e-5 * abs(nll_ref) + 1e-7
    acc = torch.zeros_like(g_ref); nll = 0.0
    for r in range(2):
        e = _engine(M, rank=r, world_size=2, autotune=False)
        e.world = 1                     # no process group here: emulate the all-reduce by summing the two local buffers
        e.grad_only(step=3, with_kl=False)
        acc += e.grads[:one.n_params]; nll += float(e.acc[0])
    one.grad_only(step=3, with_kl=False)
    assert rel(acc, one.grads[:one.n_params]) < 2e-4 and abs(nll / K - nll_ref) < 1e-5 * abs(nll_ref) + 1e-7


def test_backward_is_linear_in_the_output_gradient(M):
    eng = _engine(M, autotune=False)
    args = (eng.mu, eng.rho, eng.bn, eng.z0, SEED, 0, 0, K)
    out = eng.plan.forward(*args, True, eng.out)
    torch.manual_seed(0)
    d1 = torch.randn_like(out); d2 = torch.randn_like(out)
    def bwd(d):
        g = torch.zeros(eng.n_params, device="cuda")
        eng.plan.backward(*args, d.contiguous(), g[:eng.n_vi], g[eng.n_vi:2 * eng.n_vi], g[2 * eng.n_vi:], True)
        return g
    g1, g2, g12 = bwd(d1), bwd(d2), bwd(d1 + 2.0 * d2)
    assert rel(g12, g1 + 2.0 * g2) < 5e-4


def test_forward_is_reproducible_and_eval_mode_is_sample_free(M):
    eng = _engine(M, autotune=False)
    a = eng.plan.forward(eng.mu, eng.rho, eng.bn, eng.z0, SEED, 5, 0, K).clone()
    b = eng.plan.forward(eng.mu, eng.rho, eng.bn, eng.z0, SEED, 5, 0, K)
    assert rel(a, b) < 1e-5                                            # only the fp64 BN-statistic atomics may reorder
    c = eng.plan.forward(eng.mu, eng.rho, eng.bn, eng.z0, SEED, 5, 8, 8)      # samples 8..15 alone == the same samples inside the batch
    assert rel(c[:8], a[8:]) < 1e-5
    e = eng.plan.forward(eng.mu, eng.rho, eng.bn, eng.z0, SEED, 5, 0, 2, sample_weights=False)
    assert rel(e[0], e[1]) < 1e-5                                      # w = mu: every "sample" is the same deterministic pass
    assert rel(a[0], a[1]) > 1e-3                                      # sampled passes differ


def test_elbo_decreases_at_bench_size(M):
    eng = _engine(M)
    losses = []
    for _ in range(40):
        eng.step(); losses.append(eng.losses()[2])
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])
