#!/usr/bin/env python3
"""Print the kernel timeline of one MSM call (the middle one) from a rocprofv3 kernel_trace.csv: start, duration, and how
much of each kernel ran while another kernel of the same call was running.  usage: trace_timeline.py TRACE.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_digits" in r["Kernel_Name"] or "k_hist" in r["Kernel_Name"]]
k = len(idx) // 2
lo = idx[k] - 1 if idx[k] > 0 and "k_prepare_points" in rows[idx[k] - 1]["Kernel_Name"] else idx[k]
hi = idx[k + 1] - 1 if k + 1 < len(idx) else len(rows)
call = rows[lo:hi]
t0 = int(call[0]["Start_Timestamp"])
for r in call:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ov = 0
    for q in call:
        if q is r:
            continue
        qs, qe = int(q["Start_Timestamp"]), int(q["End_Timestamp"])
        ov += max(0, min(e, qe) - max(s, qs))
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  overlapped {ov / 1e3:7.1f}  queue {r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'].split('(')[0][:44]}")
print(f"total {(max(int(r['End_Timestamp']) for r in call) - t0) / 1e3:.1f} us")
