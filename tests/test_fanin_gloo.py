"""N > 1 path on CPU: world-size-2 gloo processes exercise the channel plan and the fan-in combiner host logic
(the kernels themselves need a GPU; the collective semantics do not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnuradio4_amd import fanin


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = fanin.channel_plan(8, world)
        g = torch.Generator().manual_seed(1234)
        allch = torch.rand((8, frames, n), generator=g, dtype=torch.float32)  # same on every rank
        mine = fanin.local_sum([allch[c] for c in plan[rank]])
        shard, work = fanin.fan_in_sum(mine)
        assert work is None
        lo, hi = fanin.shard_frames(frames, world, rank)
        want = allch.sum(dim=0)[lo:hi]
        shard2, _ = fanin.fan_in_sum(mine, algo="all_to_all")  # shards sent peer to peer, folded in rank order: the same sums
        assert float((shard2 - want).abs().max()) <= 1e-5 * float(want.abs().max()) and shard2.shape == shard.shape
        q.put((rank, float((shard - want).abs().max()), float(want.abs().max()), plan[rank], (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_fanin_world2_gloo():
    world, frames, n = 2, 6, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert [r[4] for r in res] == [(0, 3), (3, 6)]
    for _, err, scale, _, _ in res:
        assert err <= 1e-6 * scale  # float sums are order dependent: tolerance, not bit-exact


def test_channel_plan_and_shards():
    assert fanin.channel_plan(8, 1) == [list(range(8))]
    assert fanin.channel_plan(8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert fanin.channel_plan(8, 8) == [[c] for c in range(8)]
    assert fanin.shard_frames(8192, 8, 3) == (3072, 4096)
    with pytest.raises(ValueError):
        fanin.shard_frames(10, 4, 0)
    with pytest.raises(ValueError):
        fanin.channel_plan(0, 2)
    a = [torch.full((2, 4), float(i)) for i in range(3)]
    assert torch.equal(fanin.local_sum(a), torch.full((2, 4), 3.0))
