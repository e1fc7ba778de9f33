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
