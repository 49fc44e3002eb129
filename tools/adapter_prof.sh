#!/bin/bash
# per-kernel profile of the adapter alone: bash tools/adapter_prof.sh <tag> <WxT[xQ]> [option=value ...]   (GPU box) -> gpurun_out/<tag>_kernel_stats.csv + .txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp PYTHONPATH=.
TAG=$1; GEOM=$2; shift; shift
rm -rf gpurun_out/ap
ADAPTER_GEOM=$GEOM timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ap -o ap -- python3 tools/adapter_prof.py 30 f16 "$@" > gpurun_out/${TAG}.log 2>&1
tail -1 gpurun_out/${TAG}.log
f=$(find gpurun_out/ap -name 'ap_kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY' | tee gpurun_out/${TAG}.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms over 35 adapter calls = {tot / 35e6:.3f} ms per call")
for r in rows[:24]:
    n = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0][:60]
    print(f"{n:62s} calls {int(r['Calls']):5d}  per call {int(r['TotalDurationNs']) / 35e3:8.1f} us  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
# the launches of the LAST adapter call, in order (which of the equally named GEMM launches costs what)
python3 - "$(find gpurun_out/ap -name 'ap_kernel_trace.csv' | head -1)" > gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0],
       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
first = [i for i, k in enumerate(ks) if k[0].startswith("invert_mask_kernel") or k[0].startswith("sine_pos_kernel")]
start = first[-1] if first else max(0, len(ks) - 70)
t0 = ks[start][3]
for k in ks[start:]:
    print(f"{(k[3] - t0) / 1e3:9.1f} us  {k[0][:58]:60s} {k[1]:8.1f} us  {k[2]:6d} workgroups")
print(f"wall of the call {(ks[-1][4] - t0) / 1e3:.1f} us, kernel time {sum(k[1] for k in ks[start:]):.1f} us")
PY
