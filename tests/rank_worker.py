"""One rank of the K-sharded ELBO iteration (launched by tests/test_gpu_multirank.py as a fresh process, before any GPU call):
ElboEngine(rank, world) with the gloo backend on ONE GPU (RCCL refuses two ranks on one device), `steps` iterations, then rank 0 writes
the final parameters / losses.  usage: rank_worker.py rank world port out.npz task K steps [overlap [backend]]
overlap = 1: the exchange starts under the tail of the backward pass (ElboEngine.set_allreduce_overlap).  backend = nccl with world = 1: the
same schedule with real RCCL collectives on a one-rank group (the exchange is forced; nothing is summed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out, task, K, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    overlap = len(sys.argv) > 8 and sys.argv[8] == "1"
    backend = sys.argv[9] if len(sys.argv) > 9 else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world)
    from mfvi_dip_mia_amd.engine import ElboEngine
    from oracle import oracle as O
    S = 64
    eng = ElboEngine(S, S, task=task, K=K, input_depth=8, temp=5.7e-7, sigma=1.5e-5, lr=1e-3, seed=7, rank=rank, world_size=world,
                     net_kwargs=dict(nd=(8, 16, 16), nu=(8, 16, 16), ns=(4, 4, 4)), autotune=False)
    if world == 1:
        eng._force_exchange = True
    eng.set_allreduce_overlap(overlap, tail_fraction=0.5)
    img = O.phantom(S, S, 7)
    if task == "ct":
        tgt = O.radon_fwd(img, np.arange(0, 180., 4., dtype=np.float32))
    else:
        tgt = O.noisy(img, 0.1, 7)
    eng.set_target(torch.from_numpy(tgt))
    losses = []
    for _ in range(steps):
        eng.step()
        losses.append(eng.losses())
    torch.cuda.synchronize()
    ref = eng.params.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(ref, eng.params))            # every rank applies the identical update
    flags = [None] * world
    dist.all_gather_object(flags, same)
    if rank == 0:
        np.savez(out, params=eng.params.cpu().numpy(), losses=np.array(losses), identical=np.array(flags), k_local=eng.K_local, k0=eng.k0,
                 t_applied=int(eng.t_applied), split_op=-1 if eng._ov is None else eng._ov["op"], split_off=-1 if eng._ov is None else eng._ov["off"],
                 n_vi=eng.n_vi)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
