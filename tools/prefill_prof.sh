#!/bin/bash
# per-kernel averages of the batched prefill pass: bash tools/prefill_prof.sh [G] [option=value ...]   (GPU box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
rm -rf gpurun_out/pf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -o pf -- python3 tools/prefill_prof.py "$@" > gpurun_out/pf.log 2>&1
find gpurun_out/pf -name '*trace.csv' -delete
tail -1 gpurun_out/pf.log
python3 - "$@" <<'PY'
import csv, glob, json, sys
f = glob.glob("gpurun_out/pf/**/pf_kernel_stats.csv", recursive=True)[0]
G = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
rows_ = G * (32 + 7 * 139)
out = {"what": f"one batched prefill pass of the headline ({G} prefills = {rows_} rows, 8 Vicuna-7B blocks, 6 passes) under rocprofv3 --kernel-trace --stats: "
               "per-kernel averages at a FIXED row count (tools/prefill_prof.sh)", "rows": rows_, "options": sys.argv[2:], "kernels": []}
flops = {"1, 2, 0": 2.0 * rows_ * 22016 * 4096, "0, 0, 1": 2.0 * rows_ * 12288 * 4096}      # gate/up, fused QKV (o + down share a kernel name)
for r in list(csv.DictReader(open(f)))[:12]:
    n = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0][:60]
    avg = float(r["AverageNs"]) / 1e3
    print(f"{n:60s} calls {int(r['Calls']):5d}  avg {avg:8.1f} us  {float(r['Percentage']):5.1f} %")
    if n.startswith(("gemm_pp", "attn_kernel", "rmsnorm")):
        e = {"kernel": n, "calls": int(r["Calls"]), "avg_us": round(avg, 1)}
        for key, fl in flops.items():
            if n.startswith("gemm_pp") and f"<{key}" in n:
                e["tflops"] = round(fl / avg / 1e6, 1)
        if n.startswith("gemm_pp") and "<0, 0, 0" in n:      # o (K = 4096) and down (K = 11008) launches average together
            e["tflops"] = round((2.0 * rows_ * 4096 * (4096 + 11008) / 2) / avg / 1e6, 1)
        out["kernels"].append(e)
json.dump(out, open("gpurun_out/r6_prefill_pass.json", "w"), indent=1)
PY
