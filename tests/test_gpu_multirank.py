"""The multi-rank branch of ElboEngine itself: K MC samples sharded over 2 ranks, the single all-reduce of the flat gradient
buffer with the NLL scalar riding along, losses() under world > 1, identical update on every rank.  The ranks are fresh child processes
(spawned before any GPU call) that share the one GPU of the test box through the gloo backend; the result must equal the single-rank
engine with K_total samples (eps is keyed by the GLOBAL sample index)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(tmp_path, task, K, steps, world=2, overlap=0, backend="gloo"):
    out = str(tmp_path / ("ranks_%s_%d_%s.npz" % (task, overlap, backend)))
    port = str(29500 + (os.getpid() % 2000) + 2 * overlap)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rank_worker.py"), str(r), str(world), port, out, task, str(K), str(steps),
                               str(overlap), backend],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


@pytest.mark.parametrize("task", ["den", "ct"])
def test_engine_two_ranks_equal_single_rank(tmp_path, task):
    import mfvi_dip_mia_amd as M
    from mfvi_dip_mia_amd.engine import ElboEngine
    from oracle import oracle as O
    M._lib.lib()
    K, steps, S = 4, 3, 64
    z = _run_ranks(tmp_path, task, K, steps)
    assert z["identical"].all() and int(z["k_local"]) == 2 and int(z["k0"]) == 0
    eng = ElboEngine(S, S, task=task, K=K, input_depth=8, temp=5.7e-7, sigma=1.5e-5, lr=1e-3, seed=7,
                     net_kwargs=dict(nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4)), autotune=False)
    img = O.phantom(S, S, 7)
    tgt = O.radon_fwd(img, np.arange(0, 180., 4., dtype=np.float32)) if task == "ct" else O.noisy(img, 0.1, 7)
    eng.set_target(torch.from_numpy(tgt))
    losses = []
    for _ in range(steps):
        eng.step(); losses.append(eng.losses())
    p1 = eng.params.cpu().numpy()
    # same samples, same update; the gradient sum is formed in a different order (two partial sums + all-reduce), and the NLL rides the
    # all-reduce as a float.  Adam turns the rounding noise of a near-zero gradient element into a step of up to lr, so: mean tight,
    # max bounded by steps * lr (the same reading as tests/test_gpu_runner.py::test_engine_steps_match_oracle)
    d = np.abs(z["params"] - p1)
    assert d.max() < 1.5e-3 * steps and d.mean() < 1e-5, (d.max(), d.mean())
    assert np.allclose(z["losses"], np.array(losses), rtol=2e-6)
    if task == "ct":
        assert int(z["t_applied"]) == steps            # the NaN guard read the all-reduced scalar and let every update through


@pytest.mark.parametrize("task", ["den", "ct"])
def test_overlapped_exchange_is_bit_identical(tmp_path, task):
    """ElboEngine.set_allreduce_overlap: the plan reduces the weight gradients of the deep / up-path ops on a second
    stream in the middle of the backward pass, their all-reduce runs there, the head of the layout follows in a packed all-reduce.  Same
    sums, same order: the parameters after 3 steps are bit-identical to the single all-reduce after the pass, on both ranks."""
    K, steps = 4, 3
    a = _run_ranks(tmp_path, task, K, steps, overlap=0)
    b = _run_ranks(tmp_path, task, K, steps, overlap=1)
    assert a["identical"].all() and b["identical"].all()
    assert int(a["split_op"]) == -1 and int(b["split_op"]) > 0 and 0 < int(b["split_off"]) <= int(b["n_vi"]) // 2
    assert np.array_equal(a["params"], b["params"])
    assert np.array_equal(a["losses"], b["losses"])


def test_overlapped_exchange_schedule_on_rccl(tmp_path):
    """The same two schedules with real RCCL collectives (one-rank group: the collectives run, nothing is summed): the stream schedule —
    early gradient reduction on the exchange stream, RCCL's own stream behind it, the join before the update — leaves the update unchanged."""
    K, steps = 2, 3
    a = _run_ranks(tmp_path, "den", K, steps, world=1, overlap=0, backend="nccl")
    b = _run_ranks(tmp_path, "den", K, steps, world=1, overlap=1, backend="nccl")
    assert int(b["split_op"]) > 0
    assert np.array_equal(a["params"], b["params"])
    assert np.array_equal(a["losses"], b["losses"])
