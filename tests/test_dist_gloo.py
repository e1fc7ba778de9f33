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
    gids, gsc, counts = gather_scores(torch.tensor(ids, dtype=torch.int32), _fake_scores(ids))
    rows = rows_from_scores(gids.tolist(), gsc.numpy(), counts, [10, 50, 90], {i: f"img_{i}.npy" for i in range(99)},
                            2, "in")
    q.put((rank, gids.tolist(), gsc, rows))
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
        assert torch.equal(sc[order], ref)
        assert ids == [i for r in range(2) for i in range(r, n_images, 2)]  # rank-major order


def test_single_process_gather_is_identity():
    from ddpm_ood_amd.trainer import gather_scores

    ids = torch.arange(4, dtype=torch.int32)
    a, b, c = gather_scores(ids, _fake_scores([0, 1, 2, 3]))
    assert a is ids and torch.equal(b, _fake_scores([0, 1, 2, 3])) and c == [4]
