"""Per-kernel resource summary of one HIP source: python tools/kres.py ddpm_ood_amd/csrc/conv_wino44r.hip [extra hipcc flags]
(SGPRs / VGPRs / AGPRs / scratch bytes / spilled VGPRs from -Rpass-analysis=kernel-resource-usage; CPU only)."""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
       "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\S+)", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k == "VGPRs Spill":
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ddpm::", "")
        print(f"{name:60s} sgpr {cur.get('TotalSGPRs'):>4} vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} "
              f"scratch {cur.get('ScratchSize [bytes/lane]'):>4} vspill {cur.get('VGPRs Spill'):>3} sspill {cur.get('SGPRs Spill'):>3}")
