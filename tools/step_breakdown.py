"""Break one eager training step of a rocprofv3 kernel trace (tools/profile_r03.sh: kt_train/k_kernel_trace.csv) into its parts:
encoder + prologue, autoregressive decode, all-steps pass (if any), loss, reverse-time sweep, everything behind the sweep.
python tools/step_breakdown.py gpurun_out/prof_r03/kt_train/k_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]


idx = [i for i, r in enumerate(rows) if "history_prepare" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
phase, tot, cnt, det = "encoder+prologue", collections.OrderedDict(), collections.Counter(), collections.defaultdict(collections.Counter)
for r in seg:
    n, d = short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "dec_pipe4" in n:
        phase = "ar_decode"
    elif phase == "ar_decode" and not any(k in n for k in ("dec_phase_kernel", "finalize", "dec_pipe4")) and "copy" not in n.lower():
        phase = "all_steps"
    if "l2_loss" in n:
        phase = "loss"
    if "bwd_pipe_fill" in n:
        phase = "sweep"
    elif phase == "sweep" and "bwd_pipe" not in n:
        phase = "after_sweep"
    tot[phase] = tot.get(phase, 0) + d
    cnt[phase] += 1
    det[phase][n] += d
for k, v in tot.items():
    print(f"{k:18s} {v:8.1f} us  {cnt[k]:3d} kernels")
    for n, t in det[k].most_common(6):
        print(f"      {t:7.1f} us  {n}")
print(f"sum of kernel durations {sum(tot.values()):.1f} us")
