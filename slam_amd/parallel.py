"""Multi-GPU layer: chains are sharded over ranks (one process per GPU), nothing
else is.  The only exchange is what the reference's multi_mcmc does serially
(src/slam/multi_mcmc.hpp:105-139): per-chain summaries are all-gathered every
report interval, the best chain is the arg-max of the log likelihoods (the FIRST
maximum in global chain order, matching the strict `>` at :115), and its state
is broadcast from its owner only when a result is asked for.

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.  With a world size of 1 no collective is issued.
"""
import numpy as np
import torch
import torch.distributed as dist


def _alone(world):
    """True when there is nobody to exchange with.  A process group of ONE rank that has been set up
    anyway (bench.py --collectives-with-one-rank: the only way to run the RCCL calls on a one-GPU
    machine) still goes through every collective."""
    return world == 1 and not dist.is_initialized()


def _host_staged(group=None):
    """gloo moves host memory only; device tensors are staged through the host for it (the CPU
    tests, and rehearsals of the multi-rank path on one GPU).  RCCL takes device memory as is."""
    return dist.is_initialized() and dist.get_backend(group) == "gloo"


def shard_range(chains_per_rank, rank):
    """Global chain ids owned by `rank`: a contiguous block, GPU g owns [g*C, (g+1)*C)."""
    return rank * chains_per_rank, (rank + 1) * chains_per_rank


def shard_seeds(all_seeds, chains_per_rank, rank):
    """multi_mcmc draws one seed per chain from a single generator, in chain order
    (multi_mcmc.hpp:160-166); every rank takes its block of that one sequence."""
    lo, hi = shard_range(chains_per_rank, rank)
    return np.ascontiguousarray(all_seeds[lo:hi])


def gather_summaries(local, world, group=None):
    """local: [C, 2] float64 tensor of mcmcgpu_chain_summary records (column 0 = log
    likelihood).  Returns the [world*C, 2] tensor of all ranks, in global chain order."""
    if _alone(world):
        return local
    if local.is_cuda and _host_staged(group):
        return gather_summaries(local.cpu(), world, group).to(local.device)
    if local.dim() == 3:  # [intervals, C, 2] from mcmcgpu_run_reported -> [intervals, world*C, 2]
        n, C, w = local.shape
        out = torch.empty((world, n, C, w), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(-1, w), local.contiguous().view(-1, w), group=group)
        return out.permute(1, 0, 2, 3).reshape(n, world * C, w)
    out = torch.empty((local.shape[0] * world, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def best_chain(all_summaries, chains_per_rank):
    """(log likelihood, global chain id, owner rank, index on the owner) of the best chain;
    ties go to the lowest global chain id."""
    lls = all_summaries[:, 0]
    best = torch.max(lls)
    gid = int(torch.nonzero(lls == best)[0].item())
    return float(best), gid, gid // chains_per_rank, gid % chains_per_rank


def broadcast_from_owner(buf, owner, world, group=None):
    """Broadcast of the winner's packed state (relative poses + landmark estimates)."""
    if not _alone(world):
        if buf.is_cuda and _host_staged(group):
            host = buf.cpu()
            dist.broadcast(host, src=owner, group=group)
            buf.copy_(host)
        else:
            dist.broadcast(buf, src=owner, group=group)
    return buf


def all_reduce(t, op, world, group=None):
    """In-place all-reduce of a tensor on any backend."""
    if not _alone(world):
        if t.is_cuda and _host_staged(group):
            host = t.cpu()
            dist.all_reduce(host, op=op, group=group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=op, group=group)
    return t


def counts_of(all_summaries):
    """(updates, accepts) per chain, decoded from the second 8 bytes of each record."""
    raw = all_summaries[:, 1].contiguous().cpu().numpy().view(np.uint32).reshape(-1, 2)
    return raw[:, 0].copy(), raw[:, 1].copy()


def average_over_ranks(g, chains_per_rank, rank, world, device, group=None):
    """multi_mcmc::get_average (src/slam/multi_mcmc.hpp:71-73) over the chains of all ranks.  The
    reference's accumulator keeps the first chain's vectors as its base and sums the wrapped
    differences of the others (src/slam/average_slam_result.hpp:27-54): rank 0 broadcasts the base
    (global chain 0), every rank sums over its own chains on its GPU, one all-reduce (SUM) of
    3T + 2F doubles joins them.  Returns (accumulated poses [T,3], landmarks [F,2]) on `device`;
    `g` must run on torch's current stream."""
    T, F, _ = g.dims()
    n = 3 * T + 2 * F
    base = torch.zeros(n, dtype=torch.float64, device=device)
    total = torch.zeros(n, dtype=torch.float64, device=device)
    if rank == 0:
        g.average_base(0, base.data_ptr())
    broadcast_from_owner(base, 0, world, group)
    g.average_sum(base.data_ptr(), 0 if rank == 0 else -1, total.data_ptr())
    all_reduce(total, dist.ReduceOp.SUM, world, group)
    g.average_finish(base.data_ptr(), total.data_ptr(), chains_per_rank * world)
    return base[:3 * T].view(T, 3), base[3 * T:].view(F, 2)
