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

    ids = partition(n_images, rank, world)
    gids, gsc = gather_scores(torch.tensor(ids, dtype=torch.int32), _fake_scores(ids))
    q.put((rank, gids.tolist(), gsc))
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
    for rank, ids, sc in got:  # every rank holds every row (all_gather_object semantics of the reference)
        assert sorted(ids) == list(range(n_images))  # ragged shards: no padding ids, no duplicates (Q6)
        order = torch.tensor(ids).argsort()
        assert torch.equal(sc[order], ref)
        assert ids == [i for r in range(2) for i in range(r, n_images, 2)]  # rank-major order


def test_single_process_gather_is_identity():
    from ddpm_ood_amd.trainer import gather_scores

    ids = torch.arange(4, dtype=torch.int32)
    a, b = gather_scores(ids, _fake_scores([0, 1, 2, 3]))
    assert a is ids and torch.equal(b, _fake_scores([0, 1, 2, 3]))
