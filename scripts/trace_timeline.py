"""Print the kernel timeline (start offsets, durations, gaps) of the last scan in a rocprofv3
kernel trace csv.  usage: trace_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of a sweep kernel starts the last scan
last = max(i for i, nme in enumerate(names) if "k_sweep<" in nme or "k_sweep_multi" in nme)
t0 = int(rows[last]["Start_Timestamp"])
prev_end = None
for r in rows[max(0, last - 2):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  dur %8.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:90]))
    prev_end = e
