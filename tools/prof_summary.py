"""Summarise a rocprofv3 kernel_stats.csv per bench step: python tools/prof_summary.py file.csv nsteps_total"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step %.2f" % (tot / n / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    print("%-72s %6.1f calls/step %8.3f ms/step  avg %9.1f us" % (name, int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6,
                                                                 float(r["AverageNs"]) / 1e3))
