"""Multi-GPU sharding of sample batches: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The projector path shards embarrassingly: sample i of a global batch is a pure function of
(problem, seed, i) — the counter-based sampler uses the GLOBAL index — so rank r of W projects the
contiguous index range [r*B/W, (r+1)*B/W) with no data-path exchange.  The only collective is the
one the north star names: an all-gather that returns the VALID projected states to the host tree
(which lives in one process per rank here; every rank ends up with all valid states, in global
sample order).

ONE collective per step, asynchronous, and no host synchronisation inside the step: every rank sends a fixed-capacity
block — row 0 carries its count of valid states (a uint64 in the first 8 bytes), rows 1.. the
compacted states — so nothing has to be known on the host before the all-gather is enqueued, and
the gather runs on the stream behind the projector kernels.  The capacity is a fraction of the
shard (about 22 % of uniform samples are valid; default capacity 50 %, i.e. 14.7 MB instead of
29.4 MB per rank at 262144 samples per GPU); a count above the capacity is visible to every
consumer in row 0 and `unpack` refuses to return a cut list.
"""
import torch
import torch.distributed as dist

DIM = 14


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of `total` samples owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ValidGather:
    """Preallocated send / receive blocks of the all-gather of valid states, `depth` of them used in turn.

        vg = ValidGather(capacity, device)
        constraint.compact_valid(q, ok, out=vg.rows, cnt=vg.count)   # compaction writes straight into the send block
        vg.launch()                                                  # one all_gather, asynchronous, no host sync
        ...                                                          # next batch: vg.rows / vg.count are the NEXT block
        states, counts = vg.unpack()                                 # when the host tree consumes the last launch

    The collective is issued asynchronously (over RCCL it runs on the process group's own stream behind the work already
    queued on the current stream), so the projector kernels of the next batch do not wait for it; a block is handed out
    again only after the collective that last used it has been waited for.  Works on CUDA tensors over RCCL and on CPU
    tensors over gloo (the CPU tests)."""

    def __init__(self, capacity, device, group=None, depth=2):
        self.group = group
        self.world = dist.get_world_size(group)
        self.capacity = int(capacity)
        self.depth = int(depth)
        self._send = [torch.zeros((self.capacity + 1, DIM), dtype=torch.float64, device=device) for _ in range(self.depth)]
        self._recv = [torch.zeros((self.world, self.capacity + 1, DIM), dtype=torch.float64, device=device) for _ in range(self.depth)]
        self._work = [None] * self.depth
        self._k = 0       # block the next compaction writes into
        self._last = 0    # block of the most recent launch

    def _acquire(self):
        i = self._k % self.depth
        if self._work[i] is not None:  # the collective that last read this block must be through with it
            self._work[i].wait()
            self._work[i] = None
        return i

    @property
    def send(self):
        return self._send[self._acquire()]

    @property
    def rows(self):
        """(capacity, 14): where the compacted valid states of the next launch go"""
        return self.send[1:]

    @property
    def count(self):
        """uint64 count of the next launch, in the first 8 bytes of row 0 of its send block"""
        return self.send.view(torch.int64)[0, :1]

    @property
    def recv(self):
        self.wait()
        return self._recv[self._last]

    def launch(self):
        i = self._acquire()
        self._work[i] = dist.all_gather_into_tensor(self._recv[i].view(self.world * (self.capacity + 1), DIM), self._send[i],
                                                    group=self.group, async_op=True)
        self._last = i
        self._k += 1

    def wait(self):
        """every launched collective has completed (over RCCL: the current stream waits; no host synchronisation)"""
        for i in range(self.depth):
            if self._work[i] is not None:
                self._work[i].wait()
                self._work[i] = None

    def counts(self):
        """per-rank counts of valid states of the last launch (host synchronisation)"""
        return [int(v) for v in self.recv.view(torch.int64)[:, 0, 0].cpu().tolist()]

    def unpack(self):
        """(states (sum(counts),14) in rank order = global sample order for contiguous shards, counts) of the last launch"""
        counts = self.counts()
        if max(counts, default=0) > self.capacity:
            raise OverflowError("a rank holds %d valid states, the gather blocks hold %d: repeat with a larger capacity"
                                % (max(counts), self.capacity))
        recv = self.recv
        parts = [recv[r, 1: 1 + counts[r]] for r in range(self.world)]
        return torch.cat(parts, dim=0), counts


def gather_valid(q_valid, count, group=None):
    """Convenience form: all-gather the first `count` rows of every rank's `q_valid` ((cap,14), cap >= count; every
    rank must pass the same cap).  One collective; the host synchronises once, at the end, to cut the padding.
    Returns (states, counts)."""
    vg = ValidGather(q_valid.shape[0], q_valid.device, group, depth=1)
    vg.rows.copy_(q_valid)
    vg.count.copy_(count.reshape(1).to(torch.int64))
    vg.launch()
    return vg.unpack()


def sample_project_sharded(constraint, seed, total, group=None, want_iters=False, capacity_fraction=0.5):
    """Global batch of `total` sampleUniform projections across the ranks of `group`.

    Each rank projects its contiguous shard on its own GPU, compacts its valid states into the gather block and joins
    the all-gather.  Returns (all_valid_states, counts, local) where local = (q, ok, iters) of this rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(total, rank, world)
    biggest = -(-int(total) // world)  # every rank allocates the same block
    q, ok, it, _ = constraint.sample_project_batch(seed, lo, hi - lo, want_iters=want_iters)
    vg = ValidGather(max(1, int(biggest * capacity_fraction)), q.device, group)
    constraint.compact_valid(q, ok, out=vg.rows, cnt=vg.count)
    vg.launch()
    try:
        states, counts = vg.unpack()
    except OverflowError:  # an unusually valid shard: full-capacity blocks cannot overflow
        vg = ValidGather(biggest, q.device, group)
        constraint.compact_valid(q, ok, out=vg.rows, cnt=vg.count)
        vg.launch()
        states, counts = vg.unpack()
    return states, counts, (q, ok, it)
