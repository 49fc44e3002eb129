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
    m = re.search(r"(gem[mv]_\w+<[^>]*>|\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name[:60]


def pmc(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[(short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    return d


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
