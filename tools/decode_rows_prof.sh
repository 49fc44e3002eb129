#!/bin/bash
# per-kernel averages of isolated merged decode steps: bash tools/decode_rows_prof.sh 14 28
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
for R in "$@"; do
  rm -rf gpurun_out/dr_$R
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dr_$R -o dr -- python3 tools/decode_rows_time.py $R > /dev/null 2>&1
  find gpurun_out/dr_$R -name '*trace.csv' -delete
  python3 - "$R" <<'PY'
import csv, sys
R = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/dr_{R}/dr_kernel_stats.csv")))
print("rows", R)
for r in rows[:5]:
    n = r["Name"].replace("void (anonymous namespace)::", "")[:40]
    print(f"  {n:42s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
done
