"""CPU, world_size 2 over gloo: the image-sharded inference path (hip_ext.parallel.DepthGather / sharded_forward -- the same
objects bench.py --gpus N and the dataset runner drive over RCCL) returns exactly the unsharded result on every rank: even,
uneven and smaller-than-world batches, repeated asynchronous steps.  The forward here is a stand-in function -- the collective
plumbing is what is under test; the HIP model itself is covered by the -m gpu suite."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hip_ext.parallel import DepthGather, shard_range, sharded_forward


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
        # the gather object bench.py and the dataset runner use: per-rank shards produced locally, asynchronous start / finish,
        # reused across steps, uneven shards (5 items over 2 ranks) and an empty shard (1 item over 2 ranks)
        for B in (4, 5, 1):
            g = DepthGather(B, (1, 6, 7), torch.float32, "cpu")
            for step in range(3):
                full = torch.arange(B * 42, dtype=torch.float32).reshape(B, 1, 6, 7) + 1000 * step
                lo, hi = shard_range(B, rank, world)
                g.start(full[lo:hi] if hi > lo else None)
                ok &= torch.equal(g.finish(), full)
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


def _runner_worker(rank, world, port, q, tree, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "tests"))
        import test_dataset_runner_cpu as T
        from src.scripts import amodal_dav2_inference as R
        ids = ["101", "102", "103", "104", "105"]
        res = R.run(T._FakeModel(), ids, tree["occ"], tree["whole"], tree["obs"], outdir, tree["gt"], batch_size=2, device="cpu",
                    evaluate=T._oracle_evaluate)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_dataset_runner_shards_samples_over_ranks(tmp_path):
    """src/scripts/amodal_dav2_inference.run under a world-2 gloo group: every sample's PNG is written exactly once and the
    all-reduced metrics equal the single-process result."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_dataset_runner_cpu as T
    from src.scripts import amodal_dav2_inference as R
    ids = ["101", "102", "103", "104", "105"]
    d = T._make_tree(tmp_path, ids, np.random.default_rng(3))
    tree = {k: str(v) for k, v in d.items()}
    single = R.run(T._FakeModel(), ids, tree["occ"], tree["whole"], tree["obs"], str(tmp_path / "single"), tree["gt"], batch_size=2,
                   device="cpu", evaluate=T._oracle_evaluate)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, q, tree, str(tmp_path / "sharded"))) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert sorted(os.listdir(tmp_path / "sharded" / "amodal_depth")) == [f"{i}_depth.png" for i in ids]
    for r in (0, 1):
        assert res[r].keys() == single.keys()
        for k in single:
            assert abs(res[r][k] - single[k]) <= 1e-9 * max(1.0, abs(single[k])), (k, res[r][k], single[k])
