cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2m_stats -o bench -- python3 bench.py --steps 8 --warmup 2 --settle 0 --merge-decode 1 --streams 4 --no-cpu-baseline --no-extras > gpurun_out/r2m_stats.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2m_stats/bench_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:24]:
    print(r['Name'][:86].ljust(86), r['Calls'].rjust(6), ("%.1f"%(float(r['TotalDurationNs'])/1e6)).rjust(8),"ms", ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(8),"us")
PY
tail -1 gpurun_out/r2m_stats.log | cut -c1-120
