"""Aggregate rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE / SQ / GRBM runs of
tools/microbench.py) into per-kernel HBM traffic and MFMA utilisation.

    python tools/pmc_summary.py gpurun_out/pmc profiles/r01_pmc_per_kernel.csv profiles/pmc_traffic.json [--merge] [--batch N]

--batch N: the passes ran tools/microbench.py --batch N; the per-kernel figures go under "by_batch"[N] of the JSON (what
bench.py quotes as roofline.traffic when N is the timed batch) instead of replacing the top-level (batch 256) ones.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB; on gfx950
FETCH_SIZE reports half of a wide coalesced read, so reads are 2 x FETCH_SIZE x 1024 (an upper
estimate for this kernel's 4-byte input loads).  Only launches of the steady-state forwards are used.
"""
import json
import sys

import pandas as pd


def load(root, d):
    df = pd.read_csv(f"{root}/{d}/pmc_counter_collection.csv")
    df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    return df[df.Kernel_Name.str.contains("ddpm")]


def short(n):
    n = n.replace("void ddpm::", "").replace("ddpm::", "")
    return n.split("(")[0]


def main(root, out_csv, out_json):
    key = ["Kernel_Name"]
    f, w = load(root, "FETCH_SIZE"), load(root, "WRITE_SIZE")
    s, g = load(root, "SQ_WAVE_CYCLES"), load(root, "GRBM_GUI_ACTIVE")
    m = f.groupby(key).agg(launches=("Counter_Value", "size"), fetch_kb=("Counter_Value", "mean"),
                           dur_us=("dur_us", "mean")).reset_index()
    m = m.merge(w.groupby(key).agg(write_kb=("Counter_Value", "mean")).reset_index(), on=key)
    for df in (s, g):
        pv = df.pivot_table(index=key + ["Dispatch_Id"], columns="Counter_Name", values="Counter_Value").reset_index()
        m = m.merge(pv.groupby(key).mean(numeric_only=True).reset_index().drop(columns=["Dispatch_Id"]), on=key)
    m["kernel"] = m.Kernel_Name.map(short)
    m["hbm_read_MB_per_launch"] = m.fetch_kb * 2 * 1024 / 1e6
    m["hbm_write_MB_per_launch"] = m.write_kb * 1024 / 1e6
    m["hbm_MB_per_launch"] = m.hbm_read_MB_per_launch + m.hbm_write_MB_per_launch
    m["clock_GHz"] = m.GRBM_GUI_ACTIVE / 8 / (m.dur_us * 1e3)          # GRBM counter summed over 8 XCDs
    m["mfma_util"] = m.SQ_VALU_MFMA_BUSY_CYCLES / (1024 * m.dur_us * 1e3 * m.clock_GHz)  # 1024 SIMDs
    m["l2_hit"] = m.TCC_HIT_sum / (m.TCC_HIT_sum + m.TCC_MISS_sum)
    cols = ["kernel", "launches", "dur_us", "hbm_read_MB_per_launch", "hbm_write_MB_per_launch", "hbm_MB_per_launch",
            "clock_GHz", "mfma_util", "l2_hit", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
            "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU"]
    m = m.sort_values("dur_us", ascending=False)[cols]
    m.to_csv(out_csv, index=False, float_format="%.4g")
    print(m.head(12).to_string())
    out = json.load(open(out_json)) if "--merge" in sys.argv else {}
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else None
    top = out
    if batch is not None:
        out = top.setdefault("by_batch", {}).setdefault(str(batch), {})
    out["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/microbench.py, B=256 (2-D kernels) and "
                     "tools/vqvae_bench.py (3-D kernels); reads = 2 x FETCH_SIZE KB (gfx950 correction), launch-weighted mean "
                     "over the kernel's launches")
    # profiler key of bench.py's roofline object -> the kernel instantiations behind it
    # (a regular expression on the instantiation name; the last template argument of the Winograd kernels is the 3-D form)
    for key, pattern in (("conv3x3_wino44h_gn_silu", r"conv_wino44[hr]_kernel<true"),  # (either form of the split-f16 F(4x4) kernel: same profiler key)
                         ("conv3x3_wino_up", r"conv_wino_up_kernel"),
                         ("conv3x3_wino44_gn_silu", r"conv_wino44_kernel<true, \d, \d, (?:true|false), false>"),
                         ("conv3x3_wino_gn_silu", r"conv_wino_kernel<true, \d, (?:true|false), false>"),
                         ("conv3x3_mfma_gn_silu", r"conv_mfma_kernel<9, 1, true, 128"),
                         ("conv3d_wino44", r"conv_wino44_kernel<false, \d, \d, (?:true|false), true>"),
                         ("conv3d_wino", r"conv_wino_kernel<false, \d, (?:true|false), true>"),
                         ("conv1x1_dma", r"conv1x1_dma_kernel"),
                         ("attention", r"attention_kernel"),
                         ("lpips_conv_mfma", r"lpips_conv_mfma_kernel")):
        sel = m[m.kernel.notna() & m.kernel.str.contains(pattern, na=False)]  # (rows without a kernel name: copies, fills)
        if len(sel):
            wgt = sel.launches / sel.launches.sum()
            out[key + "_bytes_per_launch"] = float((sel.hbm_MB_per_launch * wgt).sum() * 1e6)
            out[key + "_mfma_util"] = float((sel.mfma_util * sel.dur_us * sel.launches).sum() / (sel.dur_us * sel.launches).sum())
            out[key + "_avg_launch_us"] = float((sel.dur_us * wgt).sum())
    if batch is not None:
        out["source"] = out["source"].replace("B=256", f"B={batch}")
    json.dump(top, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
