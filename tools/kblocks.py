"""Basic-block statistics of one kernel in a hipcc -S listing:  python tools/kblocks.py listing.s <mangled-name substring> [min lines]"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r"\n(_Z\w*" + re.escape(key) + r"\w*):[^\n]*\n", s)
st = m.start()
en = s.index("s_endpgm", st)
body = s[st:en]
open("/tmp/k.s", "w").write(body)
blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
print(m.group(1), "blocks", len(blocks), "lines", body.count("\n"))
for b in blocks:
    n = b.count("\n")
    if n > minl:
        name = b.split("\n")[0][:14]
        cnt = lambda pat: len(re.findall(pat, b))  # noqa: E731
        print(f"{name:14s} lines {n:5d} mfma {cnt('v_mfma'):3d} dsr {cnt('ds_read'):3d} dsw {cnt('ds_write'):3d} bufld {cnt('buffer_load'):3d} "
              f"rdlane {cnt('v_readlane'):3d} wrlane {cnt('v_writelane'):3d} bar {cnt('s_barrier'):2d} wait {cnt('s_waitcnt'):3d} "
              f"exp {cnt('v_exp'):2d} accrd {cnt('v_accvgpr_read'):3d} st {cnt('global_store|buffer_store'):3d} salu {cnt(chr(10) + chr(9) + 's_'):4d}")
