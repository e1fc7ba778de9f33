"""CPU: the N > 1 path with a world_size-2 gloo group as the fake cluster -- image sharding +
the single dense all_gather return the same rows as one rank."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_scores(ids, n_t=3):
    i = torch.tensor(ids, dtype=torch.float32)
    t = torch.arange(n_t, dtype=torch.float32)
    return torch.stack([i[:, None] * 0.5 + t[None, :], (i[:, None] + 1) * (t[None, :] + 2)], dim=2)


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddpm_ood_amd.data import partition
    from ddpm_ood_amd.trainer import gather_scores

    from ddpm_ood_amd.trainer import rows_from_scores

    ids = partition(n_images, rank, world)
    calls = []
    for name in ("all_gather", "all_gather_into_tensor", "all_gather_object", "all_reduce", "broadcast", "gather"):
        orig = getattr(dist, name)
        setattr(dist, name, lambda *a, _o=orig, _n=name, **k: (calls.append(_n), _o(*a, **k))[1])
    gids, gsc, counts = gather_scores(torch.tensor(ids, dtype=torch.int32), _fake_scores(ids), -(-n_images // world))
    assert calls == ["all_gather_into_tensor"], calls  # ONE collective: the static shard capacity needs no size exchange
    assert counts == [len(partition(n_images, r, world)) for r in range(world)]
    rows = rows_from_scores(gids.tolist(), gsc.numpy(), counts, [10, 50, 90], {i: f"img_{i}.npy" for i in range(99)},
                            2, "in")
    q.put((rank, gids.tolist(), gsc.numpy(), rows))  # plain data: a torch tensor on the queue needs this process alive at get()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [7, 8, 1])
def test_two_rank_gather_equals_one_rank(n_images):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _fake_scores(list(range(n_images)))
    for rank, ids, sc, rows in got:  # every rank holds every row (all_gather_object semantics of the reference)
        assert len(rows) == 3 * n_images and {r["filename"] for r in rows} == {f"img_{i}" for i in range(n_images)}
        one = _fake_scores(list(range(n_images)))
        for r in rows:  # the (image, t) -> score association survives sharding + gather
            i, j = int(r["filename"][4:]), [10, 50, 90].index(r["t"])
            assert r["perceptual_difference"] == float(one[i, j, 0]) and r["mse"] == float(one[i, j, 1])
        if n_images == 7:  # rank-major, then per batch (2), per t, per image -- rank 0 owns images 0 2 4 6
            assert [r["filename"] for r in rows[:6]] == ["img_0", "img_2"] * 3
            assert [r["t"] for r in rows[:6]] == [10, 10, 50, 50, 90, 90]
        assert sorted(ids) == list(range(n_images))  # ragged shards: no padding ids, no duplicates (Q6)
        order = torch.tensor(ids).argsort()
        assert torch.equal(torch.from_numpy(sc)[order], ref)
        assert ids == [i for r in range(2) for i in range(r, n_images, 2)]  # rank-major order


def _bench_worker(rank, world, port, q):
    """bench.py's strong-scaling bookkeeping on a gloo group: a fixed 11-image set over 2 ranks, scores through
    the same gather_scores / rows_from_scores as the trainer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from ddpm_ood_amd.data import get_data_loader
    from ddpm_ood_amd.trainer import gather_scores, rows_from_scores

    n_images, per_rank = bench.shard_sizes("strong", world, 4, 11)
    loader = get_data_loader(f"synthetic:blobs:n={n_images}:size=8:seed=0", batch_size=4, is_grayscale=True, rank=rank,
                             world=world)
    assert len(loader.names) == per_rank[rank]
    ids = [i for b in loader for i in b["index"]]
    gids, gsc, counts = gather_scores(torch.tensor(ids, dtype=torch.int32), _fake_scores(ids), -(-n_images // world))
    rows = rows_from_scores(gids.tolist(), gsc.numpy(), counts, [10, 50, 90], dict(enumerate(loader.all_names)), 4,
                            "val")
    q.put((rank, n_images, per_rank, len(rows), sorted({r["filename"] for r in rows})))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_strong_scaling_bookkeeping_two_ranks():
    import bench

    assert bench.shard_sizes("weak", 4, 256, 2048) == (1024, [256] * 4)
    assert bench.shard_sizes("strong", 8, 256, 2048) == (2048, [256] * 8)
    assert bench.shard_sizes("strong", 1, 256, 2048) == (2048, [2048])
    assert bench.shard_sizes("strong", 3, 256, 10) == (10, [4, 3, 3])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_images, per_rank, n_rows, names in got:
        assert (n_images, per_rank) == (11, [6, 5])
        assert n_rows == 11 * 3 and len(names) == 11  # the whole fixed set comes back on every rank, once


def test_single_process_gather_is_identity():
    from ddpm_ood_amd.trainer import gather_scores

    ids = torch.arange(4, dtype=torch.int32)
    a, b, c = gather_scores(ids, _fake_scores([0, 1, 2, 3]))
    assert a is ids and torch.equal(b, _fake_scores([0, 1, 2, 3])) and c == [4]
