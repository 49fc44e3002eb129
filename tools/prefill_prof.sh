#!/bin/bash
# per-kernel averages of the batched prefill pass: bash tools/prefill_prof.sh [G] [option=value ...]   (GPU box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
rm -rf gpurun_out/pf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -o pf -- python3 tools/prefill_prof.py "$@" > gpurun_out/pf.log 2>&1
find gpurun_out/pf -name '*trace.csv' -delete
tail -1 gpurun_out/pf.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pf/**/pf_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    n = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0][:60]
    print(f"{n:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
