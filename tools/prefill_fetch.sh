#!/bin/bash
# FETCH_SIZE (fabric reads, L2 misses) per launch of the batched prefill pass's GEMMs: bash tools/prefill_fetch.sh [G] [option=value ...]   (GPU box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
rm -rf gpurun_out/pff
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pff -o pf -- python3 tools/prefill_prof.py "$@" > gpurun_out/pff.log 2>&1
python3 - "$@" <<'PY'
import csv, glob, collections, sys
sys.path.insert(0, "tools")
from pmc_summary import short
f = glob.glob("gpurun_out/pff/**/pf_counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(f"{k:40s} launches {len(v):4d}  fetch per launch {2 * sum(v) / len(v) * 1024 / 1e6:9.1f} MB   (2 x FETCH_SIZE KiB, gfx950 correction)  options {sys.argv[1:]}")
PY
find gpurun_out/pff -name '*.csv' -size +2M -delete
