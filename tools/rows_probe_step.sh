#!/bin/bash
# In-step cost of the wide decode kernel's epilogue parts: per-kernel averages of isolated merged decode steps (rocprofv3 kernel trace)
# for the regular library and for probe builds (tools/rows_probe.sh build <p...>: RS_PROBE bits 16 / 32 / 64 / 128, see gemv_finish.h).
#   gpurun -- 'bash tools/rows_probe_step.sh 140 16 32 64 128 240'
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$1; shift
run() {  # $1 = tag, lib through REVISION_HIP_LIB
  rm -rf gpurun_out/ps_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_$1 -o ps -- python3 tools/decode_rows_time.py $R > gpurun_out/ps_$1.log 2>&1
  find gpurun_out/ps_$1 -name '*trace.csv' -delete
  python3 - "$1" <<'PY'
import csv, sys
t = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/ps_{t}/ps_kernel_stats.csv")))
out = []
for r in rows[:6]:
    n = r["Name"].replace("void (anonymous namespace)::", "")
    n = n[:n.index("(")] if "(" in n else n
    out.append(f"{n[:28]} {float(r['AverageNs'])/1e3:6.1f}")
print(f"probe {t:>4s}: " + " | ".join(out))
PY
  grep "ms/step" gpurun_out/ps_$1.log | tail -1
}
unset REVISION_HIP_LIB
run reg
for p in "$@"; do
  export REVISION_HIP_LIB=$PWD/revisionllm_amd/librevision_hip_p$p.so
  run $p
done
