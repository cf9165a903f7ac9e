"""N > 1 path on CPU: world_size-2 gloo run of the frame sharding + the one collective of the batched mode
(the all-gather of fixed-stride per-frame keypoint records)."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from orb_slam2_detailed_comments_amd import sharding


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 64, 65):
        for world in (1, 2, 3, 4, 8):
            blocks = [sharding.shard_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_resharding_around_dead_ranks_covers_the_batch_in_order():
    """a missing GPU takes no frames; the live ranks' blocks, in rank order, are the whole batch in frame order (the path is
    stateless per frame, so re-sharding is only a new plan)"""
    import itertools
    for n in (0, 5, 64, 65, 1024):
        for world in (2, 3, 8):
            for k in range(0, world):
                for dead in itertools.islice(itertools.combinations(range(world), k), 12):
                    blocks = sharding.shard_ranges_live(n, world, dead)
                    assert len(blocks) == world and all(blocks[r][0] == blocks[r][1] for r in dead)
                    live = [blocks[r] for r in range(world) if r not in dead]
                    assert live[0][0] == 0 and live[-1][1] == n and all(live[i][1] == live[i + 1][0] for i in range(len(live) - 1))
                    sizes = [e - b for b, e in live]
                    assert max(sizes) - min(sizes) <= 1
                    if not dead:
                        assert blocks == [sharding.shard_range(n, world, r) for r in range(world)]
    import pytest
    with pytest.raises(ValueError):
        sharding.shard_ranges_live(8, 2, dead=(0, 1))


def _fake_records(rank, frames, cap):
    rng = np.random.default_rng(1000 + rank)
    counts = torch.from_numpy(rng.integers(0, cap + 1, frames).astype(np.int32))
    kps = torch.from_numpy(rng.integers(0, 256, (frames, cap * 28)).astype(np.uint8))
    desc = torch.from_numpy(rng.integers(0, 256, (frames, cap * 32)).astype(np.uint8))
    return counts, kps, desc


def _worker(rank, world, port, frames, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for mode in ("all_gather", "gather"):
            g = sharding.RecordGatherer(frames, cap, torch.device("cpu"), mode=mode)
            for it, async_op in enumerate((False, True, True)):
                counts, kps, desc = _fake_records(rank + 10 * it, frames, cap)
                buf = g.gather(counts, kps, desc, async_op=async_op)
                g.wait_all()
                if mode == "gather" and rank != 0:
                    ok &= buf.shape[0] == 0             # only the root receives
                    continue
                c, k, d = sharding.unpack_records(buf, cap)
                for r in range(world):
                    ec, ek, ed = _fake_records(r + 10 * it, frames, cap)
                    sl = slice(r * frames, (r + 1) * frames)
                    ok &= bool(torch.equal(c[sl], ec) and torch.equal(k[sl], ek) and torch.equal(d[sl], ed))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_record_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_pack_unpack_roundtrip_single_process():
    counts, kps, desc = _fake_records(0, 5, 8)
    buf = sharding.pack_records(counts, kps, desc)
    assert buf.shape == (5, sharding.record_bytes(8))
    c, k, d = sharding.unpack_records(buf, 8)
    assert torch.equal(c, counts) and torch.equal(k, kps) and torch.equal(d, desc)
