"""-m gpu: the RCCL code path of the score gather on ONE device (the 8-GPU node is the driver's): a 1-rank
``backend="nccl"`` group -- nccl IS RCCL on ROCm -- through ``gather_scores`` with device tensors, i.e. the same
``all_gather_into_tensor`` the N-rank run issues (reference: all_gather_object at
/root/reference/src/trainers/reconstruct.py:238-242).  Runs in a subprocess with a timeout so that a wedged
communicator cannot hang the suite."""

import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ddpm_ood_amd.trainer import gather_scores, rows_from_scores
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
calls = []
orig = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
dev = torch.device("cuda:0")
ids = torch.tensor([0, 2, 4, 6, 8], dtype=torch.int32, device=dev)
scores = torch.arange(5 * 3 * 2, dtype=torch.float32, device=dev).reshape(5, 3, 2) / 7
for n_max in (5, 8):   # exact fit and a padded shard
    gids, gsc, counts = gather_scores(ids, scores, n_max)
    assert gids.is_cuda and gsc.is_cuda and counts == [5]
    assert torch.equal(gids, ids) and torch.equal(gsc, scores)
assert len(calls) == 2
rows = rows_from_scores(gids.cpu().tolist(), gsc.cpu().numpy(), counts, [10, 50, 90], {}, 2, "in")
assert len(rows) == 15 and rows[0]["filename"] == "0" and rows[0]["mse"] == float(scores[0, 0, 1])
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_single_rank_gather(device):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _SCRIPT, str(ROOT)], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


# ---- two ranks on ONE GPU: the N > 1 product path end to end ---------------------------------------------------
# RCCL refuses two ranks on one device, so the ranks rendezvous over gloo (DDPM_DIST_BACKEND) and both compute on
# cuda:0 (DDPM_DIST_SHARED_DEVICE): everything except the transport of the one collective is the N-rank code path --
# partition, per-rank batches, static-capacity payload, rank-major row order, rank-0-only CSV.

def _launch_ranks(world, argv, tmp_path, extra_env=None, timeout=900):
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), DDPM_DIST_BACKEND="gloo", DDPM_DIST_SHARED_DEVICE="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, (o[-1500:], e[-3000:])
    return outs


def test_two_ranks_on_one_gpu_write_the_same_scores_as_one_rank(device, tmp_path):
    """reconstruct.py as 1 rank and as 2 ranks (reference: torchrun + partition_dataset,
    /root/reference/src/trainers/base.py:22-33, src/data/get_train_and_val_dataloader.py:21-31; gather at
    src/trainers/reconstruct.py:238-248).  8 / 7 / 8 images in batches of 4: each rank's batches have the size of the
    1-rank run's, so the per-image arithmetic is the same launch geometry and the scores must be IDENTICAL; only the
    row order differs (rank-major, as the reference's all_gather_object list)."""
    import pandas as pd
    from ddpm_ood_amd import synthetic

    def run(root, world):
        model = "fashionmnist_dist"
        synthetic.write_checkpoint(root / model, "small", 1, seed=1)
        argv = [str(ROOT / "reconstruct.py"), "--output_dir", str(root), "--model_name", model, "--is_grayscale", "1",
                "--validation_ids", "synthetic:blobs:n=8:seed=10", "--in_ids", "synthetic:blobs:n=7:seed=11",
                "--out_ids", "synthetic:noise:n=8:seed=12:name=MNIST",
                "--beta_schedule", "scaled_linear_beta", "--beta_start", "0.0015", "--beta_end", "0.0195",
                "--batch_size", "4", "--inference_skip_factor", "32"]
        if world == 1:
            out = subprocess.run([sys.executable, *argv], capture_output=True, text=True, timeout=900)
            assert out.returncode == 0, out.stderr[-3000:]
        else:
            _launch_ranks(world, argv, root)
        return {n: pd.read_csv(root / model / "ood" / f"results_{n}.csv", index_col=0) for n in ("val", "in", "MNIST")}

    (tmp_path / "w1").mkdir()
    (tmp_path / "w2").mkdir()
    one, two = run(tmp_path / "w1", 1), run(tmp_path / "w2", 2)
    for n in one:
        a, b = one[n], two[n]
        assert len(a) == len(b) and len(b) == len(b.drop_duplicates(["filename", "t"]))  # no padding duplicates (Q6)
        # rank-major order: rank 0 holds images 0, 2, 4, 6 -> its first batch comes first
        names = list(dict.fromkeys(a["filename"]))  # the 1-rank file lists the images in id order within a batch
        assert list(b["filename"][:4]) == [names[i] for i in (0, 2, 4, 6)]
        key = ["filename", "t"]
        a, b = a.sort_values(key).reset_index(drop=True), b.sort_values(key).reset_index(drop=True)
        assert a.to_csv(index=False) == b.to_csv(index=False)  # byte for byte once the rows are in the same order


_TRAIN_SCRIPT = r"""
import argparse, hashlib, os, sys, torch
sys.path.insert(0, sys.argv[1])
from ddpm_ood_amd.train import DDPMTrainer
a = argparse.Namespace(
    seed=2, output_dir=sys.argv[2], model_name="ddp_train", training_ids="synthetic:blobs:n=9:seed=1",
    validation_ids="synthetic:blobs:n=4:seed=10", spatial_dimension=2, image_size=None, image_roi=None, latent_pad=None,
    vqvae_checkpoint=None, prediction_type="epsilon", model_type="small", beta_schedule="scaled_linear_beta",
    beta_start=0.0015, beta_end=0.0195, b_scale=1.0, snr_shift=1, simplex_noise=0, batch_size=4, n_epochs=1, eval_freq=1,
    augmentation=1, num_workers=0, cache_data=1, checkpoint_every=0, ddpm_checkpoint_epoch=None, is_grayscale=1,
    quick_test=0)
tr = DDPMTrainer(a)
h = hashlib.sha256()
for p in tr.model.parameters():
    h.update(p.detach().cpu().numpy().tobytes())
start = h.hexdigest()
steps = []
orig = tr._sync_grads
tr._sync_grads = lambda: (steps.append(1), orig())[1]
tr.train_epoch(0)
h = hashlib.sha256()
for p in tr.model.parameters():
    h.update(p.detach().cpu().numpy().tobytes())
sys.__stdout__.write(f"RANK{tr.rank} start={start} end={h.hexdigest()} steps={len(steps)} n_local={len(tr.train_loader.names)}\n")
sys.__stdout__.flush()
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_training_starts_and_stays_in_step(device, tmp_path):
    """Row f-3 under 2 ranks (reference: DistributedDataParallel's parameter broadcast, base.py:160-163): both ranks
    start from rank 0's parameters, run the same number of steps although the shards are 5 and 4 images long (batches
    of 4: two steps each, the short shard wraps), and hold identical parameters after the epoch."""
    import re

    outs = _launch_ranks(2, ["-c", _TRAIN_SCRIPT, str(ROOT), str(tmp_path)], tmp_path)
    recs = {}
    for _, o, _e in outs:
        m = re.search(r"RANK(\d) start=(\w+) end=(\w+) steps=(\d+) n_local=(\d+)", o)
        assert m, o[-2000:]
        recs[int(m.group(1))] = m.groups()[1:]
    assert recs[0][0] == recs[1][0]            # same initial parameters
    assert recs[0][1] == recs[1][1] != recs[0][0]  # same parameters after one epoch, and they moved
    assert recs[0][2] == recs[1][2] == "2" and {recs[0][3], recs[1][3]} == {"5", "4"}


def test_bench_self_launches_eight_ranks_on_one_gpu(device):
    """The driver's 8-GPU command line -- `python bench.py --gpus 8` -- with all eight ranks on the one GPU of this box (gloo
    transport, shared device): self-launch, rendezvous on 127.0.0.1, eight strong-scaling shards of 8 images, barriers, MAX over
    ranks, one collective per step, ONE JSON line that lists eight ranks and 64 x 25 rows; and every rank's start-up (weights
    packed, workspace allocated) is logged BEFORE the timed region starts although --warmup is 0.  What an 8-GPU node adds is
    the RCCL transport under the same calls (reference launch model /root/reference/src/trainers/base.py:22-33, gather
    /root/reference/src/trainers/reconstruct.py:238-248)."""
    import json

    env = dict(os.environ, DDPM_DIST_BACKEND="gloo", DDPM_DIST_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--images", "64",
                          "--batch", "8", "--no-cpu-baseline", "--dataset-images", "96"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_world_size"] == 8 and d["scaling"] == "strong" and d["dist_backend"] == "gloo"
    assert d["config"]["images_per_step"] == 64 and d["config"]["reconstructions_per_step"] == 64 * 25
    assert d["config"]["unet_forwards_per_image"] == 1250
    assert [x.split(":")[0] for x in d["devices"]] == [f"rank {r}" for r in range(8)]
    assert d["value"] > 0 and d["numeric_guard"] == {"batches_rerun_fp32": 0, "batches_nonfinite": 0}
    # every rank's own timing is on the line, and the dataset-scale pass (here 96 images: 12 per rank) beside the 64-image one
    assert [p["rank"] for p in d["per_rank"]] == list(range(8)) and all(p["images"] == 8 and p["timed_s"] > 0 for p in d["per_rank"])
    assert all(len(p["gather_ms_per_step"]) == 1 and p["startup_s"] > 0 for p in d["per_rank"])
    assert d["dataset_scale"]["images"] == 96 and d["dataset_scale"]["reconstructions"] == 96 * 25 and d["value_dataset_scale"] > 0
    err = out.stderr
    assert 0 <= err.find("start-up done") < err.find("timed region done"), err[-2000:]


def test_bench_self_launches_two_ranks_on_one_gpu(device):
    """`python bench.py --gpus 2` as the driver runs it (no launcher): bench.py starts its own ranks through
    torch.distributed.run, shards the fixed image set (strong scaling is the default for N > 1), brackets the timed region
    with barriers, takes the MAX over ranks and rank 0 prints ONE JSON line.  Two ranks share the GPU here (gloo transport,
    DESIGN 4.1 hooks); on a multi-GPU node the same command runs over RCCL."""
    import json

    env = dict(os.environ, DDPM_DIST_BACKEND="gloo", DDPM_DIST_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "32",
                          "--images", "64", "--no-dataset-scale"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["scaling"] == "strong" and d["dist_backend"] == "gloo"
    assert d["config"]["images_per_step"] == 64 and d["config"]["reconstructions_per_step"] == 64 * 25
    assert len(d["devices"]) == 2 and d["devices"][0].startswith("rank 0") and d["devices"][1].startswith("rank 1")
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
