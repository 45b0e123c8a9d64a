"""Multi-GPU decomposition of one ELBO iteration: the K Monte-Carlo samples are sharded over the
ranks of one node (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  eps is keyed by the GLOBAL sample index, so the summed gradient does not depend on the rank count;
the only exchange per iteration is ONE all-reduce of the flat gradient buffer (with the NLL scalar riding along).
The KL term and its gradient are deterministic and computed redundantly on every rank.  Independent fits (one image
per GPU) need no exchange at all."""


def shard_samples(K, rank, world_size):
    """-> (k0, K_local): the contiguous block of global sample indices evaluated by `rank`."""
    if K % world_size:
        raise ValueError("K=%d must be divisible by the number of ranks %d" % (K, world_size))
    k_local = K // world_size
    return rank * k_local, k_local


def allreduce_sum_(flat, group=None, force=False):
    """In-place sum over ranks of the flat [d mu | d rho | d BN | scalars] buffer (or of a slice of it: engine.set_allreduce_overlap).
    force: issue the collective on a one-rank group too (tests of the stream schedule on a one-GPU box)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat
