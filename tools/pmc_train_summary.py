"""HBM-side traffic per training step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass) over
tools/train_only.py, per kernel, with the corrections of /opt/skills/guides/MI355X_MICROARCH.md (KiB units; FETCH_SIZE x 2 on gfx950
for 128-byte requests; Infinity-Cache hits included: an upper bound on HBM bytes).
python tools/pmc_train_summary.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <steps in the run>"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            tot[name] += float(row["Counter_Value"])
            calls[name] += 1
    return tot, calls


d, steps = sys.argv[1], int(sys.argv[2])
fetch, calls = per_kernel(f"{d}/pmc_FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE")
write, _ = per_kernel(f"{d}/pmc_WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/train_only.py 128 N 0 0 (eager steps, Basic, B=128)",
       "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950; Infinity-Cache hits included", "steps_in_run": steps, "kernels": {}}
tot = 0.0
for k in sorted(fetch, key=lambda k: -(fetch[k] * 2 + write.get(k, 0.0))):
    rd, wr = fetch[k] * 1024 * 2 / steps, write.get(k, 0.0) * 1024 / steps
    tot += rd + wr
    if rd + wr > 1e6:
        out["kernels"][k] = {"launches_per_step": calls[k] / steps, "read_bytes_per_step": rd, "write_bytes_per_step": wr}
out["bytes_per_step"] = tot
print(json.dumps(out, indent=1))
