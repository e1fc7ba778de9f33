"""Per kernel of a gfx950 assembly file (clang -S): compiler-generated AGPR accesses and scratch accesses by loop depth.
    python tools/asm_spill_report.py file.s
Used on conv_wino44r.hip compiled through LLVM IR with "amdgpu-agpr-alloc"="0,0" (tools/build_w44r_ir.sh): the accumulator
tiles live in a[0:127] BY NAME, so any compiler-generated AGPR access is a bug and every scratch access inside the chunk loop
(depth >= 2) is a real cost."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
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
    depth, in_app = 0, False
    agpr = [0, 0, 0, 0]
    scr = [0, 0, 0, 0]
    for l in body:
        mm = re.match(r"\.LBB\d+_\d+:\s*;.*Depth=(\d)", l)
        if mm:
            depth = min(int(mm.group(1)), 3)
        elif l.startswith(".LBB"):
            depth = 0
        if "#ASMSTART" in l:
            in_app = True
        elif "#ASMEND" in l:
            in_app = False
        code = l.split(";")[0]
        if not in_app and re.search(r"\ba\[?\d", code):
            agpr[depth] += 1
        if "scratch_" in code:
            scr[depth] += 1
    sc = next((l.split(":")[1].strip() for l in lines[end:end + 80] if "ScratchSize" in l), "?")
    short = re.sub(r"^_ZN4ddpm\d+conv_wino44r_kernelI", "", name)[:40]
    print(f"{short:42s} scratch bytes {sc:>5s}  compiler AGPR accesses by depth {agpr}  scratch instructions by depth {scr}")
