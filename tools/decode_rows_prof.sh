#!/bin/bash
# per-kernel averages of isolated merged decode steps: bash tools/decode_rows_prof.sh 14 28
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
# a trailing "fp8" argument profiles the steps on the FP8 weight copies (outputs dr_<rows>f8)
FP8=""; SUF=""
for a in "$@"; do if [ "$a" = "fp8" ]; then FP8="--fp8"; SUF="f8"; fi; done
for R in "$@"; do
  [ "$R" = "fp8" ] && continue
  R2=$R$SUF
  rm -rf gpurun_out/dr_$R2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dr_$R2 -o dr -- python3 tools/decode_rows_time.py $R $FP8 > /dev/null 2>&1
  R=$R2
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
