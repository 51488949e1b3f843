"""Unique (kernel, LDS, VGPR, AGPR, workgroup, grid) rows of a rocprofv3 kernel trace with total time."""
import csv, sys, collections
t = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Workgroup_Size_X"],
         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_X"]))
    t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
print("%-70s %7s %5s %5s %5s %7s %6s %9s" % ("kernel", "LDS", "VGPR", "AGPR", "wg", "grid", "calls", "total ms"))
for k, v in t.most_common(60):
    print("%-70s %7s %5s %5s %5s %7d %6d %9.3f" % (k + (n[k], v / 1e6)))
