"""Groups the rows of `DDPM_PROF_SHAPES=1 python tools/train_step_bench.py <B> 3 native` into the classes DESIGN.md 3.14 quotes:
    python tools/train_step_classes.py profiles/r06_train_native_b256_shapes_v5.txt"""
import collections
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m:
        rows.append((m.group(1), int(m.group(2)), float(m.group(3))))


def cls(n):
    if n.startswith("conv3x3_wino|") or n.startswith("conv3x3_wino_up"):
        return "input gradients on F(2x2)"
    if n.startswith(("train_conv3x3_wgrad", "train_wgrad_reduce", "train_conv_wgrad")):
        return "3x3 weight gradients"
    if n.startswith(("conv3x3_wino44h", "conv1x1", "conv3x3_s", "linear", "conv3x3_small")):
        return "forward convolutions / Linear (+ 1x1 input gradients)"
    if n.startswith("train_gemm"):
        return "strided GEMM (attention, 1x1 / Linear gradients)"
    if n.startswith("train_gn"):
        return "GroupNorm forward + backward"
    return "reductions / copies / element-wise"


tot = sum(r[2] for r in rows)
agg, cnt = collections.Counter(), collections.Counter()
for n, k, ms in rows:
    agg[cls(n)] += ms
    cnt[cls(n)] += k
for k, v in agg.most_common():
    print(f"{k:58s} {cnt[k]:4d} launches {v:8.3f} ms {100 * v / tot:5.1f} %")
print(f"sum of the library's per-launch timers: {tot:.3f} ms")
