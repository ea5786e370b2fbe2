"""Summarise rocprofv3 PMC passes (one counter per pass: FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM-side traffic
per predict, following /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
  * FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch;
  * on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes -> doubled for wide (16 B/lane) reads;
  * Infinity-Cache hits are counted too (the counters sit on the L2's fabric side), so this is an upper bound on
    what reaches HBM.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv \
                                gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv  > profiles/rNN/traffic.json
"""
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


def main():
    fetch, fcalls = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    # number of predicts in the run = launches of the encoder kernel
    enc = next(k for k in fcalls if k.startswith("mmt_encoder_kernel") or k.startswith("mmt_encoder_split_kernel"))
    n_pred = fcalls[enc]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied as 64 B); Infinity-Cache hits included",
           "predicts_in_run": n_pred, "kernels": {}}
    dec_bytes = 0.0
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        if not k.startswith("mmt_"):
            continue
        rd = fetch[k] * 1024 * 2 / n_pred
        wr = write.get(k, 0.0) * 1024 / n_pred
        out["kernels"][k] = {"launches_per_predict": fcalls[k] / n_pred, "read_bytes_per_predict": rd,
                             "write_bytes_per_predict": wr}
        if k.startswith("mmt_dec_") or k.startswith("mmt_decoder_kernel"):
            dec_bytes += rd + wr
    out["decoder_bytes_per_predict"] = dec_bytes
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
