"""Write tests/golden/wino44h_digests.json (run on the GPU box while the kernel under test is the one to pin):
    python tools/r06/make_w44_digests.py > gpurun_out/wino44h_digests.json
Digests of conv_wino44r.hip's outputs (+ GroupNorm statistics) over every item shape of tests/test_gpu_wino44h.py.  First written
in round 6 from the kernel source that round 5's suite had held bit-identical to its LDS-fed predecessor (unchanged since;
profiles/r06_w44r_ir_route_sets3_bit_identity.log is this round's run of that comparison), when the predecessor was retired."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import pytest  # noqa: E402
import torch  # noqa: E402

import test_gpu_wino44h as tw  # noqa: E402

dev = torch.device("cuda:0")
mp = pytest.MonkeyPatch()
from ddpm_ood_amd import _lib  # noqa: E402


class MP:  # monkeypatch with the conftest's reload-on-change behaviour
    def setenv(self, k, v):
        mp.setenv(k, v)
        _lib.reload_env()

    def delenv(self, k, raising=False):
        mp.delenv(k, raising=raising)
        _lib.reload_env()


out = {"2d": {}, "3d": {}}
for case in tw.CASES + tw.XITEM_CASES + tw.SPLIT_CASES:
    _, y1, st1, _ = tw._run_pinned_2d(dev, case, MP())
    out["2d"][repr(tuple(case))] = tw._digest(y1, st1)
for case in tw.CASES_3D:
    y1, u1, _ = tw._run_pinned_3d(dev, case, MP())
    out["3d"][repr(tuple(case))] = tw._digest(y1, u1)
mp.undo()
print(json.dumps(out, indent=0))
