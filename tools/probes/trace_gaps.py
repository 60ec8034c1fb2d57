"""Idle-time analysis of a rocprofv3 kernel trace (csv): per frame-sized window, wall time vs the union of kernel intervals,
and the largest gaps with the kernels on either side.  usage: python tools/probes/trace_gaps.py <kernel_trace.csv> [n_gaps [last_ms]]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void prv2::", "")))
rows.sort()
if len(sys.argv) > 3:  # only the last <ms> milliseconds of the trace (steady-state frames)
    tend = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= tend - int(float(sys.argv[3]) * 1e6)]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"{len(rows)} kernels, span {(t1 - t0) / 1e6:.1f} ms")
# union of intervals + gaps
gaps = []
cur_end = rows[0][1]
busy = rows[0][1] - rows[0][0]
last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, last_name, n))
        busy += e - s
        cur_end = e
        last_name = n
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
        last_name = n
print(f"busy (union) {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
big = [g for g in gaps if g[0] > 20000]
print(f"gaps > 20 us: {len(big)}, total {sum(g[0] for g in big) / 1e6:.2f} ms; gaps <= 20 us: total {sum(g[0] for g in gaps if g[0] <= 20000) / 1e6:.2f} ms")
for g in sorted(gaps, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"  {g[0] / 1e3:9.1f} us at {g[1] / 1e6:9.2f} ms   after {g[2][:50]:50s} before {g[3][:50]}")
