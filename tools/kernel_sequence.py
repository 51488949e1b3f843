"""Ordered launch sequence of the last training step of a rocprofv3 kernel trace (short names, duration in us)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
lo, hi = adam[-2] + 1, adam[-1] + 1
prev_end = None
for r in rows[lo:hi]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    print("%8.1f us  gap %6.1f  grid %7d  %s" % ((e - s) / 1e3, gap, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), name))
