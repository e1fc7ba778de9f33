"""Summarise a rocprofv3 rocpd (.db) kernel trace as CSV (rocprofv3 --kernel-trace --stats writes
the sqlite form by default): one row per kernel with calls, total / average / min / max duration.

    python tools/rocpd_summary.py gpurun_out/prof_kt/r01_results.db > profiles/r01_kernel_stats.csv
"""

import csv
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
         "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name "
         "order by sum(duration) desc")
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR",
                "SGPR", "LDS"])
    for r in rows[:top]:
        w.writerow([r[0][:160], r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100.0 * r[2] / total, 3), *r[6:]])
    w.writerow(["TOTAL (all kernels)", sum(r[1] for r in rows), int(total), "", "", "", 100.0, "", "", "", ""])


if __name__ == "__main__":
    main(sys.argv[1])
