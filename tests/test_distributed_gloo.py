"""The N>1 path on CPU: two processes, gloo backend — shard arithmetic and the all-gather that
returns valid states in global sample order (the compute itself needs a GPU and is covered by the
-m gpu tests; sharding never changes arithmetic because samples depend on the global index only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from closed_chain_motion_planner_amd.distributed import ValidGather, gather_valid, shard_range

    lo, hi = shard_range(total, rank, world)
    cap = -(-total // world)  # every rank brings a block of the same capacity: the biggest shard
    # synthetic "projected" shard: row i carries its global index; every third sample is valid
    rows = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1).repeat(1, 14)
    ok = (torch.arange(lo, hi) % 3 == 0)
    valid = rows[ok]
    padded = torch.zeros((cap, 14), dtype=torch.float64)
    padded[: valid.shape[0]] = valid
    states, counts = gather_valid(padded, torch.tensor(valid.shape[0]))
    # an empty rank must work too
    states0, counts0 = gather_valid(torch.zeros((4, 14), dtype=torch.float64), torch.tensor(0 if rank else 2))
    # the steady-state form: preallocated blocks, one collective, count in row 0, nothing read on the host until
    # unpack; a count above the capacity is reported by unpack instead of returning a cut list
    vg = ValidGather(2, "cpu")
    vg.rows.copy_(torch.full((2, 14), float(rank)))
    vg.count.fill_(2 if rank == 0 else 1)
    vg.launch()
    st2, c2 = vg.unpack()
    # two launches in flight use the two blocks in turn; unpack returns the last one
    for k in range(3):
        vg.rows.copy_(torch.full((2, 14), float(10 * k + rank)))
        vg.count.fill_(1)
        vg.launch()
    st3, c3 = vg.unpack()
    assert st3[:, 0].tolist() == [20.0, 21.0] and c3 == [1, 1]
    vg.count.fill_(5 if rank == 1 else 1)
    vg.launch()
    try:
        vg.unpack()
        overflow = False
    except OverflowError:
        overflow = True
    if rank == 0:
        q.put((states.numpy(), counts, states0.shape[0], counts0, st2[:, 0].tolist(), c2, overflow))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 257])
def test_gather_valid_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    states, counts, n0, counts0, st2, c2, overflow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = np.array([i for i in range(total) if i % 3 == 0], dtype=np.float64)
    assert states.shape == (len(exp), 14) and np.array_equal(states[:, 0], exp)  # global sample order
    assert sum(counts) == len(exp)
    assert n0 == 2 and counts0 == [2, 0]
    assert st2 == [0.0, 0.0, 1.0] and c2 == [2, 1] and overflow is True


def test_shard_range_partitions_exactly():
    from closed_chain_motion_planner_amd.distributed import shard_range

    for total in (0, 1, 7, 8, 2097152, 262145):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert shard_range(2097152, 3, 8) == (3 * 262144, 4 * 262144)
