"""Turns the two rocprofv3 --pmc summaries (FETCH_SIZE, WRITE_SIZE; tools/pmc_summary.py output) of the decode GEMV into the JSON
record bench.py reads for roofline.traffic.  The record carries the sha256 of the kernel source it was measured on, so bench.py can
refuse a stale file.
usage: make_pmc_json.py <raw.txt> <out.json> <algorithmic_bytes_per_launch>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw, out, alg = sys.argv[1], sys.argv[2], float(sys.argv[3])
txt = open(raw).read()
vals = {m.group(1): (int(m.group(2)), float(m.group(3))) for m in re.finditer(r"(\w+): dispatches=(\d+) sum=\S+ per_dispatch=(\S+)", txt)}
fetch_kb, write_kb = vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]
# rocprofv3 reports both in KB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced stream at 64 B: x2 (MI355X_MICROARCH.md, HBM section)
hbm = fetch_kb * 1024.0 * 2.0 + write_kb * 1024.0
sha = hashlib.sha256(open(os.path.join(ROOT, "usdm_amd", "csrc", "llm_k.hip"), "rb").read()).hexdigest()
rec = {"kernel": "gemv_kernel", "dispatches": vals["FETCH_SIZE"][0],
       "source": f"profiles/{os.path.basename(out).replace('.json', '.txt')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, no tracing; "
                 "FETCH x2 gfx950 correction)",
       "fetch_size_kb_per_launch_as_counted": fetch_kb, "write_size_kb_per_launch": write_kb,
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "llm_k_hip_sha256": sha}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
