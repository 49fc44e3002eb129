#!/bin/bash
# per-kernel totals of one bench workload: bash tools/workload_prof.sh <workload> [bench flags ...]   (GPU box) -> gpurun_out/wl_<workload>.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
W=$1; shift
rm -rf gpurun_out/wl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wl -o wl -- python3 bench.py --workload $W --steps 64 --warmup 32 --no-cpu-baseline --no-extras "$@" > gpurun_out/wl.log 2>&1
find gpurun_out/wl -name '*trace.csv' -delete
grep -o '"value": [0-9.]*, "unit": "[a-z/]*"' gpurun_out/wl.log | head -1
python3 - $W <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/wl/**/wl_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
out = [f"total kernel time {tot / 1e6:.1f} ms (96 windows incl. warm-up)"]
for r in rows[:22]:
    n = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0][:56]
    out.append(f"{n:58s} calls {int(r['Calls']):6d}  total {int(r['TotalDurationNs']) / 1e6:8.2f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
open(f"gpurun_out/wl_{sys.argv[1]}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
