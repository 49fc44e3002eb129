"""Where the wall time of the headline goes that no kernel accounts for: reads a rocprofv3 --kernel-trace CSV of bench.py and, inside the
steady part of the run, sweeps the dispatch intervals of ALL streams: how long the device ran a big kernel of each class (prefill GEMM /
merged decode step / adapter), only small ones (copies, norms, sampling), or nothing at all, and which dispatches the idle gaps sit between.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_tl -o bench -- python3 bench.py --steps 32 --warmup 16 ...
    python3 tools/timeline_gaps.py gpurun_out/r4_tl gpurun_out/r4_timeline.json
"""
import csv
import glob
import json
import sys
from collections import Counter, defaultdict

CLASSES = (("prefill_gemm", ("gemm_pp",)), ("decode_proj", ("rows_kernel", "gemv_stream")), ("decode_attn", ("attn_kernel<128, true",)),
           ("prefill_attn", ("attn_kernel_pair", "attn_kernel<128, false")), ("adapter", ("gemm_arows", "gemm_tile", "attn_kernel<96", "attn_kernel<64", "layernorm", "transpose_v")),
           ("copy", ("copyBuffer", "copy_kernel", "direct_copy", "fillBuffer", "FillFunctor")))


def klass(name):
    for k, pats in CLASSES:
        if any(p in name for p in pats):
            return k
    return "other"


def short(name):
    n = name.split("(anonymous namespace)::")[-1]
    return n.split("(")[0][:48]


def main(src, dst, lo=0.45, hi=0.95):
    f = sorted(glob.glob(src + "/**/*kernel_trace.csv", recursive=True))[0]
    ev = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    w0, w1 = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
    ev = [e for e in ev if e[1] > w0 and e[0] < w1]
    # sweep: at every boundary the set of running classes
    pts = []
    for i, (s, e, n, q) in enumerate(ev):
        pts.append((max(s, w0), 1, i))
        pts.append((min(e, w1), 0, i))
    pts.sort()
    running = Counter()
    state_time = defaultdict(float)
    prev = w0
    last_end = (None, None)
    gaps = []
    gap_open = None
    for t, kind, i in pts:
        if t > prev:
            big = [k for k in ("prefill_gemm", "decode_proj", "decode_attn", "prefill_attn", "adapter") if running[k] > 0]
            if big:
                state_time["+".join(big)] += t - prev
            elif sum(running.values()) > 0:
                state_time["small_only"] += t - prev
            else:
                state_time["idle"] += t - prev
        k = klass(ev[i][2])
        if kind == 1:
            if sum(running.values()) == 0 and gap_open is not None and t > gap_open[0]:
                gaps.append((t - gap_open[0], gap_open[1], short(ev[i][2])))
            running[k] += 1
            gap_open = None
        else:
            running[k] -= 1
            if sum(running.values()) == 0:
                gap_open = (t, short(ev[i][2]))
        prev = t
    span = w1 - w0
    by_pair = defaultdict(lambda: [0, 0.0])
    for g, a, b in gaps:
        by_pair[(a, b)][0] += 1
        by_pair[(a, b)][1] += g
    out = {"what": "device timeline of bench.py (all streams) inside the steady part of a rocprofv3 --kernel-trace run: share of the wall time by what was running",
           "window_ms": span / 1e6, "dispatches": len(ev),
           "share": {k: round(v / span, 4) for k, v in sorted(state_time.items(), key=lambda kv: -kv[1])},
           "idle_gaps": {"count": len(gaps), "total_ms": sum(g[0] for g in gaps) / 1e6,
                         "by_neighbours": [{"after": a, "before": b, "count": c, "total_us": round(t / 1e3, 1), "avg_us": round(t / c / 1e3, 2)}
                                           for (a, b), (c, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]]},
           "exclusive_kernel_ms": {}}
    tot = defaultdict(float)
    for s, e, n, q in ev:
        tot[klass(n)] += min(e, w1) - max(s, w0)
    out["exclusive_kernel_ms"] = {k: round(v / 1e6, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["share"]), out["idle_gaps"]["total_ms"], "ms idle of", span / 1e6)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
