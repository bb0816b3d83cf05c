"""N>1 path on CPU: image sharding + the one gather of detection records, world_size 2, gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_partitions_everything():
    from bayes_od_rc_amd.distributed import shard_range
    for total in (0, 1, 7, 16, 37):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake(rank, b=3, k=5, c=8):
    rng = np.random.default_rng(100 + rank)
    num = torch.tensor([k, 0, 2][:b], dtype=torch.int32)
    return (num, torch.from_numpy(rng.random((b, k, c)).astype(np.float32)),
            torch.from_numpy(rng.random((b, k, 4)).astype(np.float32)),
            torch.from_numpy(rng.random((b, k, 16)).astype(np.float32)),
            torch.from_numpy(rng.random((b, k, c)).astype(np.float32)))


def test_pack_unpack_round_trip():
    from bayes_od_rc_amd import distributed as bd
    num, scores, means, covs, counts = _fake(0)
    rec = bd.pack_records(num, scores, means, covs, counts)
    assert rec.shape == (3, 5, bd.record_width(8))
    dets = bd.unpack_records(rec, 8)
    assert [d[0].shape[0] for d in dets] == [5, 0, 2]
    assert np.array_equal(dets[2][1], means[2, :2].numpy())
    assert np.array_equal(dets[0][2], covs[0].numpy().reshape(5, 4, 4))
    assert np.array_equal(dets[0][3], counts[0].numpy())


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayes_od_rc_amd import distributed as bd
    rec = bd.pack_records(*_fake(rank))
    out = bd.gather_records(rec, dst=0)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_world2_gloo():
    from bayes_od_rc_amd import distributed as bd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (2, 3, 5, bd.record_width(8))
    for r in range(2):
        assert np.array_equal(got[r], bd.pack_records(*_fake(r)).numpy())


# ---- second mode: MC-sample sharding (one all-gather of the raw head outputs) -------------------------------
def test_sample_shard_split():
    from bayes_od_rc_amd.distributed import sample_shard
    assert [sample_shard(30, 2, r) for r in range(2)] == [(0, 15), (15, 15)]
    assert [sample_shard(8, 8, r) for r in range(8)] == [(r, 1) for r in range(8)]
    with pytest.raises(ValueError):
        sample_shard(10, 4, 0)


def _sample_block(b, n_total, a, c):
    """value encodes (image, sample, anchor, channel) so a misplaced slice is visible"""
    i = np.arange(b)[:, None, None, None] * 1e6 + np.arange(n_total)[None, :, None, None] * 1e3
    return (i + np.arange(a)[None, None, :, None] * 10 + np.arange(c)[None, None, None, :]).astype(np.float32)


def _sample_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayes_od_rc_amd import distributed as bd
    out = {}
    for b in (1, 2):
        full_ref = _sample_block(b, 6, 7, 4)
        base, n = bd.sample_shard(6, world, rank)
        local = torch.from_numpy(full_ref[:, base:base + n].copy())
        full = torch.zeros(b, 6, 7, 4)
        bd.all_gather_samples(local, full)
        out[b] = np.array_equal(full.numpy(), full_ref)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_samples_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == {0: {1: True, 2: True}, 1: {1: True, 2: True}}


def _mean_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayes_od_rc_amd import distributed as bd
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    bd.all_reduce_mean_(g)
    q.put((rank, g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_mean_world2_gloo():
    """The data-parallel training step's single collective: in-place mean of the contiguous gradient arena."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mean_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = np.arange(1000, dtype=np.float32) * 1.5
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
