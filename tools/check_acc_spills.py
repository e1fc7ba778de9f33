"""Build-time check for the inline-asm MFMA kernels: an accumulator tile must never be spilled or copied by compiler-
generated code inside the MFMA loops.  hipcc does not know that an `asm volatile("v_mfma...")` statement writes its
destination registers asynchronously over the following passes, so a spill store placed right behind it saves STALE
values (found on the 16x16 variant of conv_wino44h.hip: registers 0..3 of one tile wrong by 1e-3, run to run different).

    python tools/check_acc_spills.py ddpm_ood_amd/csrc/conv_wino44h.hip [extra hipcc flags]

Compiles the file to assembly and fails, for every kernel that contains MFMAs, if
  * any compiler-generated instruction (outside the ;;#ASMSTART ... ;;#ASMEND brackets of inline asm) names an AGPR anywhere in the
    kernel -- the kernels that address their accumulators by name (conv_wino44h.hip) rely on the compiler never using one;
  * any scratch access happens inside a loop of depth >= 2 (the MFMA loops).
"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flags = [a for a in sys.argv[2:] if a != "--loops-only"]
# --loops-only: for kernels whose accumulators are C++ variables ("+a" / "+v" asm operands): compiler-generated AGPR traffic is
# expected in their epilogues; what must not happen is a spill or copy of an accumulator INSIDE the MFMA loops (depth >= 2)
LOOPS_ONLY = "--loops-only" in sys.argv
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                    "--cuda-device-only", "-S", src, "-o", f.name, *flags], check=True, stderr=subprocess.DEVNULL)
    lines = open(f.name).read().split("\n")
bad = 0
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    end = next(j for j in range(i, len(lines)) if ".end_amdhsa_kernel" in lines[j] or j == len(lines) - 1)
    body = lines[i:end]
    i = end + 1
    if not any("v_mfma" in l for l in body):
        continue
    depth = 0
    in_app = False
    hits = []
    for k, l in enumerate(body):
        mm = re.match(r"\.LBB\d+_\d+:\s*;.*Depth=(\d)", l)
        if mm:
            depth = int(mm.group(1))
        elif l.startswith(".LBB"):
            depth = 0
        if "#ASMSTART" in l:
            in_app = True
        elif "#ASMEND" in l:
            in_app = False
        code = l.split(";")[0]
        if not in_app and re.search(r"\ba\[?\d", code) and (depth >= 2 or not LOOPS_ONLY):
            hits.append((k, "AGPR in compiler code: " + l.strip()[:70]))
        if depth >= 2 and "scratch_" in code:
            hits.append((k, "scratch in an MFMA loop: " + l.strip()[:70]))
    short = re.sub(r"^_ZN4ddpm\d+", "", name)[:60]
    print(f"{short}: {'OK' if not hits else f'{len(hits)} accumulator / scratch accesses inside the MFMA loops'}")
    for h in hits[:4]:
        print("   ", h)
    bad += bool(hits)
sys.exit(1 if bad else 0)
