"""Kernel timeline of the last evaluation in a rocprofv3 --kernel-trace csv: start offset, duration, gap to the previous
kernel's end [us], name.    usage: python3 tools/trace_last_eval.py <kernel_trace.csv> [first-kernel-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "cov_upper"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:90]))
    busy += e - s
    prev_end = max(prev_end, e)
print("kernels %d  span %.1f us  busy %.1f us" % (len(rows) - lo, (prev_end - t0) / 1e3, busy / 1e3))
