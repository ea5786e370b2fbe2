"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this turns it into the two CSVs the profiles/ summaries use:

    python tools/rocpd_to_csv.py stats    gpurun_out/prof/b_results.db > profiles/rNN/kernel_stats.csv
    python tools/rocpd_to_csv.py counters gpurun_out/pmc_X/p_results.db > gpurun_out/pmc_X/p_counter_collection.csv
"""
import csv
import math
import sqlite3
import sys
from collections import defaultdict


def stats(db):
    rows = defaultdict(list)
    for name, dur in sqlite3.connect(db).execute("select name, duration from kernels"):
        rows[name].append(dur)
    total = sum(sum(v) for v in rows.values())
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        sd = math.sqrt(sum((x - avg) ** 2 for x in v) / max(1, len(v) - 1))
        w.writerow([name, len(v), sum(v), round(avg, 3), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 3)])


def counters(db):
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
    for row in sqlite3.connect(db).execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id"):
        w.writerow(row)


if __name__ == "__main__":
    {"stats": stats, "counters": counters}[sys.argv[1]](sys.argv[2])
