# SQ / GRBM / TCC counters of the kernels of isolated 70-row merged decode steps (bf16 and FP8 weights): what bounds rows_kernel?
# gpurun -- 'bash tools/pmc_rows.sh'   -> gpurun_out/r3rows{,f8}_pmc_mfma.json, gpurun_out/r3rows*_tcc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for v in "" "--fp8"; do
  s=r3rows${v:+f8}
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${s}_sq -o dr -- python3 tools/decode_rows_time.py 70 $v > gpurun_out/${s}_sq.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/${s}_grbm -o dr -- python3 tools/decode_rows_time.py 70 $v > gpurun_out/${s}_grbm.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/${s}_tcc -o dr -- python3 tools/decode_rows_time.py 70 $v > gpurun_out/${s}_tcc.log 2>&1
  python3 - $s <<'PY'
import sys, glob, csv, json, collections
sys.path.insert(0, "tools")
import pmc_summary
s = sys.argv[1]
pmc_summary.sq_summary(f"gpurun_out/{s}_sq", f"gpurun_out/{s}_grbm", f"gpurun_out/{s}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"gpurun_out/{s}_tcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[(pmc_summary.short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (k, g), c in acc.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    row = {"kernel": k, "grid_threads": g, "launches": max(len(v) for v in c.values()), **{n: round(v, 1) for n, v in a.items()}}
    if a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0) > 0:
        row["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
    rows.append(row)
rows.sort(key=lambda r: -r.get("TCC_REQ_sum", 0) * r["launches"])
json.dump({"note": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum of tools/decode_rows_time.py 70 (isolated merged decode steps); averages per launch", "kernels": rows[:12]},
          open(f"gpurun_out/{s}_tcc.json", "w"), indent=1)
for r in rows[:6]:
    print(r)
PY
  for d in ${s}_sq ${s}_grbm ${s}_tcc; do rm -rf gpurun_out/$d; done
done
