"""Summarise gpurun_out/dr_<rows>/dr_kernel_stats.csv (tools/decode_rows_prof.sh) into gpurun_out/r6_decode_steps.json (copied to profiles/): the kernels of an
isolated merged decode step at several row counts."""
import csv
import json
import sys

out = {"what": "rocprofv3 --kernel-trace --stats of tools/decode_rows_time.py <rows> (23 isolated steps of rv_llm_decode_rows at Vicuna-7B shapes, 32 layers)",
       "rows": {}}
for R in sys.argv[1:]:
    rows = list(csv.DictReader(open(f"gpurun_out/dr_{R}/dr_kernel_stats.csv")))
    ks = []
    for r in rows:
        n = r["Name"].replace("void (anonymous namespace)::", "")
        if any(t in n for t in ("rows_kernel", "gemv_stream", "attn_kernel", "rmsnorm", "sample", "splice", "rope_table")):
            ks.append({"kernel": n.split("(")[0], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1)})
    per_layer = sum(k["avg_us"] * k["calls"] for k in ks if k["calls"] >= 700) / 736.0
    out["rows"][R] = {"kernels": ks[:8], "projection_and_attention_us_per_layer": round(per_layer, 1)}
import os, re
ms = {}
if os.path.exists("gpurun_out/r6_decode_ms.log"):
    for line in open("gpurun_out/r6_decode_ms.log"):
        m = re.match(r"(fp8 )?rows\s+(\d+): ([\d.]+) ms/step", line)
        if m:
            ms[m.group(2) + ("f8" if m.group(1) else "")] = float(m.group(3))
out["ms_per_step"] = ms
json.dump(out, open("gpurun_out/r6_decode_steps.json", "w"), indent=1)
print(json.dumps({k: v["projection_and_attention_us_per_layer"] for k, v in out["rows"].items()}))
