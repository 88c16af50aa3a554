"""Multi-GPU sharding of sample batches: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The projector path shards embarrassingly: sample i of a global batch is a pure function of
(problem, seed, i) — the counter-based sampler uses the GLOBAL index — so rank r of W projects the
contiguous index range [r*B/W, (r+1)*B/W) with no data-path exchange.  The only collective is the
one the north star names: an all-gather that returns the VALID projected states to the host tree
(which lives in one process per rank here; every rank ends up with all valid states, in global
sample order).  Payload is compacted first: ~22 % of the states are valid, so the gather moves
~6.5 MB per rank instead of 29.4 MB at 262144 samples per GPU.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of `total` samples owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_valid(q_valid, count, group=None):
    """All-gather the first `count` rows of every rank's `q_valid` (padded (cap,14) tensor).

    Returns (states, counts): states is (sum(counts), 14) in rank order — i.e. global sample order
    for contiguous shards — and counts the per-rank row counts (python ints).  Works on CUDA
    tensors over RCCL and on CPU tensors over gloo (used by the CPU tests).
    Two collectives: counts (8 B per rank), then rows padded to the largest count.
    """
    world = dist.get_world_size(group)
    cnt = count.reshape(1).to(torch.int64)
    counts = torch.empty(world, dtype=torch.int64, device=cnt.device)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts_host = [int(v) for v in counts.cpu().tolist()]
    m = max(counts_host) if counts_host else 0
    if m == 0:
        return q_valid.new_empty((0, q_valid.shape[1])), counts_host
    if q_valid.shape[0] < m:
        raise ValueError("q_valid has %d rows, another rank holds %d" % (q_valid.shape[0], m))
    send = q_valid[:m].contiguous()
    recv = torch.empty((world, m, q_valid.shape[1]), dtype=q_valid.dtype, device=q_valid.device)
    dist.all_gather_into_tensor(recv.view(world * m, q_valid.shape[1]), send, group=group)
    parts = [recv[r, : counts_host[r]] for r in range(world)]
    return torch.cat(parts, dim=0), counts_host


def sample_project_sharded(constraint, seed, total, group=None, want_iters=False):
    """Global batch of `total` sampleUniform projections across the ranks of `group`.

    Each rank projects its contiguous shard on its own GPU, compacts its valid states and joins the
    all-gather.  Returns (all_valid_states, counts, local) where local = (q, ok, iters) of this rank.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(total, rank, world)
    q, ok, it, _ = constraint.sample_project_batch(seed, lo, hi - lo, want_iters=want_iters)
    q_valid, cnt = constraint.compact_valid(q, ok)
    states, counts = gather_valid(q_valid, cnt, group)
    return states, counts, (q, ok, it)
