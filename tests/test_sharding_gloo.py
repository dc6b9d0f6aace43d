"""CPU, world_size 2 over gloo: the image-sharded inference wrapper returns exactly the unsharded result on every
rank (even, uneven and smaller-than-world batches).  The forward here is a stand-in function -- the collective
plumbing is what is under test; the HIP model itself is covered by the -m gpu suite."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hip_ext.parallel import shard_range, sharded_forward


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_depth(x, mask, obs):
    # per-image function (no cross-image coupling), shaped like the model output [B,1,H,W]
    # purely elementwise so that slicing the batch cannot change a single bit
    return (x[:, :1] * 3 + x[:, 1:2] - x[:, 2:3]) * 0.5 + 0.5 * mask - 0.25 * obs


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for B in (4, 5, 1):
            g = torch.Generator().manual_seed(B)
            x = torch.rand(B, 3, 28, 42, generator=g)
            mask = (torch.rand(B, 1, 28, 42, generator=g) > 0.5).float() * 2 - 1
            obs = torch.rand(B, 1, 28, 42, generator=g) * 2 - 1
            full = sharded_forward(_fake_depth, [x, mask, obs])
            ok &= torch.equal(full, _fake_depth(x, mask, obs))
            local = sharded_forward(_fake_depth, [x, mask, obs], gather=False)
            lo, hi = shard_range(B, rank, world)
            ok &= (local is None) if hi == lo else torch.equal(local, _fake_depth(x, mask, obs)[lo:hi])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for total in (0, 1, 5, 32, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_sharded_forward_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
