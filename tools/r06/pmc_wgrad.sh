# rocprofv3 counter passes over tools/wgrad_ab.py (the split-f16 weight gradient on the step's shapes, batch 256)
o=$GRAFT_REPO_ROOT/gpurun_out/r06_pmc_wgrad
PMC_CMD="python tools/wgrad_ab.py 256" bash tools/pmc_collect.sh $o > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_pmc_wgrad"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(root + "/*/pmc_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wgrad_f16x3" not in k and "absmax" not in k and "wgrad_reduce" not in k:
            continue
        k = k.split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"], f)
        if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
            seen.add(key)
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:16.0f}")
PY
