"""Print the ISA of one kernel around its MFMA loop (development tool).
    python tools/kasm.py <file.s> <kernel-name-substring> [context-before] [lines]"""
import sys
path, sub = sys.argv[1], sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
body = []
for l in lines[start:]:
    t = l.strip()
    if t and not t.startswith(";"):
        body.append(l)
    if "s_endpgm" in l:
        break
mf = [i for i, l in enumerate(body) if "\tv_mfma" in l]
print(f"{len(body)} instrs, {len(mf)} mfma, first at {mf[0]}")
for l in body[max(0, mf[0] - before): mf[0] - before + count]:
    print(l[:92])
