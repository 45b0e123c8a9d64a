"""Device-resident iteration: the RNG step counter and Adam's update count on the device (ElboEngine.enable_device_step),
and one captured HIP graph of an iteration replayed for every step (ElboEngine.enable_graph).  The loop being matched is the reference's
K = 1 loop, bayesian_optimization.py:1360-1372; what is asserted is that NOTHING changes: parameters and losses bit-identical to the
host-driven engine after the same number of iterations."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _same_losses(a, b):
    """the loss accumulators are float64 sums by atomic adds (order not fixed from run to run): equal to the last few bits, not bit for bit"""
    return all(abs(x - y) <= 1e-12 * max(abs(y), 1e-30) for x, y in zip(a, b))


def _engine(task, K, S=64, **kw):
    from mfvi_dip_mia_amd.engine import ElboEngine
    eng = ElboEngine(S, S, task=task, K=K, input_depth=8, temp=5.7e-7, sigma=1.5e-5, lr=1e-3, seed=11,
                     net_kwargs=dict(nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4)), autotune=False, **kw)
    img = O.phantom(S, S, 11)
    if task == "ct":
        tgt = O.radon_fwd(img, np.arange(0, 180., 4., dtype=np.float32))
    elif task == "sr":
        tgt = np.ascontiguousarray(img[::4, ::4])
    else:
        tgt = O.noisy(img, 0.1, 11)
    eng.set_target(torch.from_numpy(tgt))
    return eng


@pytest.mark.parametrize("task,K", [("den", 1), ("den", 3), ("sr", 2), ("ct", 1)])
def test_device_step_and_graph_replay_are_bit_identical(task, K):
    n = 9
    ref = _engine(task, K)
    for _ in range(n):
        ref.step()
    ref_losses = ref.losses()
    # counters on the device, still launched kernel by kernel
    dev = _engine(task, K).enable_device_step()
    for _ in range(n):
        dev.step()
    assert int(dev.step_dev) == n and int(dev.t_applied) == n
    assert torch.equal(dev.params, ref.params), "device-resident counters changed the trajectory"
    assert _same_losses(dev.losses(), ref_losses)
    # one captured graph, replayed (3 un-captured warm-up iterations + 6 replays)
    gr = _engine(task, K).enable_graph(warmup=3)
    assert gr._graph is not None and gr.t == 3
    for _ in range(n - 3):
        gr.step()
    torch.cuda.synchronize()
    assert int(gr.step_dev) == n
    assert torch.equal(gr.params, ref.params), "graph replay changed the trajectory"
    assert _same_losses(gr.losses(), ref_losses)
    assert torch.equal(gr.m, ref.m) and torch.equal(gr.v, ref.v)


def test_graph_engine_keeps_the_eager_interface():
    """after_forward hooks (the runners' bookkeeping) still work on a graph engine: such an iteration runs kernel by kernel."""
    eng = _engine("den", 2).enable_graph(warmup=2)
    seen = []
    eng.step(after_forward=lambda n_: seen.append(n_))
    eng.step()
    torch.cuda.synchronize()
    assert seen == [2] and eng.t == 4 and int(eng.step_dev) == 4
    ref = _engine("den", 2)
    for _ in range(4):
        ref.step()
    assert torch.equal(eng.params, ref.params)
