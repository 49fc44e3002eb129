"""GPU busy fraction of a rocprofv3 kernel trace: union of the kernel intervals inside the window [lo, hi] (fractions of the span covered
by the pipeline's decode kernel `rows_kernel`, i.e. of the recursion steps) - is the pipeline device-bound or does the device wait for the host?
python tools/gpu_busy.py <kernel_trace.csv> [lo hi]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
flo, fhi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.2, 0.8)
dec = [r for r in rows if "rows_kernel" in r[2]]
t0, t1 = dec[0][0], dec[-1][1]
lo, hi = t0 + (t1 - t0) * flo, t0 + (t1 - t0) * fhi
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(r[1] for r in sel) - sel[0][0]
gaps.sort()
print(f"kernels {len(sel)}  span {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms = {busy / span:.3f}  idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms, "
      f"> 20 us: {sum(1 for g in gaps if g > 20000)} ({sum(g for g in gaps if g > 20000) / 1e6:.2f} ms), largest {gaps[-5:] if gaps else []}")
