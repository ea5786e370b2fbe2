"""profiles/rNN/traffic_b128_basic.json from the two PMC passes of tools/profile_r03.sh, stamped with the hash of the kernel sources
(bench.py quotes it only while that hash matches the tree it runs from) and the decoder form that ran."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r03"
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f"{d}/pmc_FETCH_SIZE/p_counter_collection.csv",
                               f"{d}/pmc_WRITE_SIZE/p_counter_collection.csv"])
j = json.loads(out)
j["kernel_source_hash"] = kernel_source_hash()
forms = [k for k in j["kernels"] if k.startswith("mmt_dec_pipe4") or k.startswith("mmt_dec_fused4") or k.startswith("mmt_decoder_kernel")]
j["decoder_form"] = forms[0].split("<")[0] if forms else "mmt_dec_phase"
j["workload"] = "tools/decode_only.py 128 auto 25 (Basic predict, batch 128, 8-step history, 12-step horizon)"
print(json.dumps(j, indent=1))
