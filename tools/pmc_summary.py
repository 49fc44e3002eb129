"""Summarise rocprofv3 outputs into profiles/: per-kernel stats and the PMC-derived HBM traffic.

Inputs (written on the GPU box under gpurun_out/ by the commands in profiles/README.md):
  <stats_dir>/bench_kernel_stats.csv           rocprofv3 --kernel-trace --stats -- python bench.py ...
  <fetch_dir>/bench_counter_collection.csv     rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py ...
  <write_dir>/bench_counter_collection.csv     rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py ...
Counters are collected in separate passes.  Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced read
stream, so traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (WRITE_SIZE is uncalibrated: it is small here).
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(gem[mv]_\w+<[^>]*>|\w+_kernel(?:_\w+)?(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name[:60]


def pmc(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[(short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    return d


def sq_summary(sq_dir, grbm_dir, out_prefix, n_xcd=8, simds=1024):
    """SQ counters of every kernel (averages per launch) + derived figures.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES /
    SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs;
    GRBM_GUI_ACTIVE counts GPU cycles summed over the XCDs.  mfma_util = MFMA busy cycles / (kernel cycles x SIMDs)."""
    import glob
    names = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
             "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (sq_dir, grbm_dir):
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] in names:
                    acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for (k, grid), c in acc.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        row = {"kernel": k, "grid_threads": grid, "launches": max(len(v) for v in c.values()), **{n: round(v, 1) for n, v in a.items()}}
        if "GRBM_GUI_ACTIVE" in a and "SQ_VALU_MFMA_BUSY_CYCLES" in a and a["GRBM_GUI_ACTIVE"] > 0:
            cyc = a["GRBM_GUI_ACTIVE"] / n_xcd
            row["kernel_cycles"] = round(cyc, 1)
            row["mfma_util"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * simds), 4)
        if a.get("SQ_WAVE_CYCLES", 0) > 0:
            for n, key in (("SQ_WAIT_ANY", "frac_waves_parked_at_waitcnt_or_barrier"), ("SQ_WAIT_INST_ANY", "frac_waves_issue_stalled"),
                           ("SQ_ACTIVE_INST_ANY", "frac_waves_issuing")):
                if n in a:
                    row[key] = round(a[n] / a["SQ_WAVE_CYCLES"], 4)
        rows.append(row)
    rows.sort(key=lambda r: -r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * r["launches"])
    pp = [r for r in rows if r["kernel"].startswith("gemm_pp_sk") and "mfma_util" in r]
    summary = None
    if pp:
        busy = sum(r["SQ_VALU_MFMA_BUSY_CYCLES"] * r["launches"] for r in pp)
        cyc = sum(r["kernel_cycles"] * r["launches"] for r in pp)
        summary = {"kernels": "gemm_pp_sk<...> (the prefill projections of the recursion: passes of up to 8 x 1005 rows)", "mfma_util": round(busy / (cyc * simds), 4),
                   "lds_bank_conflict_cycles": sum(r.get("SQ_LDS_BANK_CONFLICT", 0) for r in pp),
                   "by_kernel": {r["kernel"]: {"mfma_util": r["mfma_util"], "launches": r["launches"], "parked": r.get("frac_waves_parked_at_waitcnt_or_barrier"),
                                               "issue_stalled": r.get("frac_waves_issue_stalled")} for r in pp}}
    json.dump({"note": "rocprofv3 --pmc, two passes (SQ set / GRBM set) of `bench.py --steps 16 --warmup 0 --settle 0 --no-cpu-baseline --no-extras` (tools/profile_r6.sh); "
                       "averages per launch; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
               "summary": summary, "kernels": rows[:30]}, open(out_prefix + "_pmc_mfma.json", "w"), indent=1)
    print("wrote", out_prefix + "_pmc_mfma.json", summary)


def main(stats_dir, fetch_dir, write_dir, out_prefix):
    f = pmc(f"{fetch_dir}/bench_counter_collection.csv", "FETCH_SIZE")
    w = pmc(f"{write_dir}/bench_counter_collection.csv", "WRITE_SIZE")
    rows = []
    for k in sorted(f, key=lambda k: -sum(f[k])):
        fv, wv = f[k], w.get(k, [0.0])
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        rows.append({"kernel": k[0], "grid_threads": k[1], "workgroup": k[2], "launches": len(fv), "fetch_size_kib_avg": fa,
                     "write_size_kib_avg": wa, "traffic_bytes_per_launch": (2 * fa + wa) * 1024})
    json.dump({"note": "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction per MI355X_MICROARCH.md; separate --pmc passes",
               "kernels": rows[:40]}, open(out_prefix + "_pmc_traffic.json", "w"), indent=1)
    with open(out_prefix + "_kernel_stats.csv", "w") as o:
        for line in open(f"{stats_dir}/bench_kernel_stats.csv"):
            o.write(line)
    print("wrote", out_prefix + "_pmc_traffic.json", out_prefix + "_kernel_stats.csv")


if __name__ == "__main__":
    main(*sys.argv[1:5])
    if len(sys.argv) >= 7:
        sq_summary(sys.argv[5], sys.argv[6], sys.argv[4])
