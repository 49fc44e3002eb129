# Counters of the kernels of isolated 140-row merged decode steps: where do the waves of rows_kernel<9,...> spend their cycles?
# gpurun -- 'bash tools/pmc_rows140.sh'        -> the SQ pass only (profiles/r5_rows140_sq.json is its summary)
# gpurun -- 'bash tools/pmc_rows140.sh all'    -> also the TCP / TCC / TA groups.  CAUTION: on this pool every one of those five passes HUNG rocprofv3 until its
#   timeout killed it (round 5: 25 GPU-minutes gone for nothing) - they now get 60 s each; do not raise that before one of them has been seen to finish.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 -L > gpurun_out/r5rows140_counters_all.txt 2>&1
grep -o -E "\b(TCP|TCC|TA|TD|SQ)_[A-Z0-9_]+(_sum)?\b" gpurun_out/r5rows140_counters_all.txt | sort -u > gpurun_out/r5rows140_counters.txt
have() { grep -q -x "$1" gpurun_out/r5rows140_counters.txt; }
pass() {  # $1 = tag, rest = wanted counters (those the box lacks are dropped)
  tag=$1; shift; sel=""
  for c in "$@"; do if have $c; then sel="$sel $c"; fi; done
  [ -z "$sel" ] && return
  echo "pass $tag:$sel"
  timeout ${PASS_TIMEOUT:-60} rocprofv3 --pmc $sel --kernel-trace --output-format csv -d gpurun_out/r5rows140_$tag -o dr -- python3 tools/decode_rows_time.py 140 --share > gpurun_out/r5rows140_$tag.log 2>&1
}
PASS_TIMEOUT=240 pass e SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAVES
if [ "$1" = all ]; then
pass a TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass b TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum
pass c TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass d TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
pass f TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
fi
python3 - <<'PY'
import glob, csv, json, collections, sys
sys.path.insert(0, "tools")
import pmc_summary
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r5rows140_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[pmc_summary.short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    if "rows_kernel" not in k and "attn_kernel" not in k and "gemv" not in k:
        continue
    row = {"kernel": k, "launches": max(len(v) for v in c.values())}
    row.update({n: round(sum(v) / len(v), 1) for n, v in sorted(c.items())})
    rows.append(row)
json.dump({"note": "rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over tools/decode_rows_time.py 140 --share: averages per launch of the decode kernels", "kernels": rows}, open("gpurun_out/r5rows140_mem.json", "w"), indent=1)
for r in rows:
    print(json.dumps(r)[:1500])
PY
for t in a b c d e f; do rm -rf gpurun_out/r5rows140_$t; done
