"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE pass (rocpd
database) per kernel: MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs; 32 per v_mfma_f32_16x16x4_f32,
checked against the CNN kernel's known MFMA count) / (cycles x 1024), cycles = GRBM_GUI_ACTIVE / 8 (the counter comes back summed
over the 8 XCDs: 1.30 M for a 61 us launch at 2.4 GHz), and the fp32 MFMA operations issued (MOPS counter x 512, padding rows and
columns included) against the kernel's duration and the 157.3 TFLOP/s dense fp32 MFMA peak.

    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma_map_cnn/p_results.db > profiles/rNN/mfma_map_cnn.json
"""
import collections
import json
import sqlite3
import sys

N_SIMD = 256 * 4


def main(db):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for k, cn, v, d, di in c.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection"):
        k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][cn].append(v)
        dur[k][di] = d
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE "
                     "(one pass; counters serialise the kernels)", "kernels": {}}
    for k, cs in agg.items():
        if not k.startswith("mmt_"):
            continue
        mean = {cn: sum(v) / len(v) for cn, v in cs.items()}
        d_ns = sum(dur[k].values()) / max(1, len(dur[k]))
        busy, gui, mops = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mean.get("GRBM_GUI_ACTIVE", 0.0), mean.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        out["kernels"][k] = {"launches": len(dur[k]), "avg_duration_us": d_ns / 1e3, "counters_per_launch": mean,
                             "mfma_busy_share": busy / (gui / 8 * N_SIMD) if gui else None,
                             "mfma_f32_tflops_issued": mops * 512 / d_ns / 1e3 if d_ns else None,
                             "mfma_f32_issued_frac_of_peak": mops * 512 / d_ns / 1e3 / 157.3 if d_ns else None}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
