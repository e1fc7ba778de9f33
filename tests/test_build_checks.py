"""CPU: build-time property of the inline-asm MFMA kernels -- no compiler-generated access to an accumulator register.

conv_wino44r.hip addresses eight accumulator tiles by NAME (a[0:127]) inside its asm statements, so the compiler must
never allocate an AGPR for anything of its own, and nothing may spill inside the MFMA loops (hipcc does not know that
an asm MFMA writes its destination asynchronously: a spill store behind it saves stale values -- the bug that produced
run-to-run different 1e-3 errors on the 16x16 variant before the tiles were pinned).  tools/check_acc_spills.py compiles
the file for gfx950 (cross-compilation: no GPU needed) and inspects the assembly."""

import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


import pytest


@pytest.mark.parametrize("name", ["conv_wino44r"])
def test_wino44h_accumulators_are_never_touched_by_compiler_code(name):
    """The split-f16 F(4x4) kernel (conv_wino44r.hip; its LDS-fed predecessor in conv_wino44h.hip was retired in round 6, that file
    now holds host code only) pins its tiles by name."""
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "check_acc_spills.py"),
                          str(ROOT / "ddpm_ood_amd" / "csrc" / f"{name}.hip"), "-fno-slp-vectorize"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    assert out.stdout.count(": OK") == 24, out.stdout  # 2 (affine) x 4 (shapes) x 2 (residual) + 4 three-dimensional instantiations


def test_older_mfma_kernels_do_not_spill_accumulators_inside_their_loops():
    """The kernels whose accumulators are C++ variables bound to asm MFMAs ("+a" / "+v"): the same stale-spill hazard exists
    wherever hipcc decides to move an accumulator inside an MFMA loop.  --loops-only: compiler-generated AGPR traffic or
    scratch accesses inside loops of depth >= 2 fail; epilogue reads are expected."""
    from concurrent.futures import ThreadPoolExecutor

    def check(name):
        return name, subprocess.run([sys.executable, str(ROOT / "tools" / "check_acc_spills.py"),
                                     str(ROOT / "ddpm_ood_amd" / "csrc" / f"{name}.hip"), "--loops-only"],
                                    capture_output=True, text=True, timeout=1200)

    with ThreadPoolExecutor(3) as pool:
        for name, out in pool.map(check, ["conv_wino44", "conv_wino", "attention"]):
            assert out.returncode == 0, (name, out.stdout[-2000:])
