"""Average every PMC counter per launch for each kernel from a rocprofv3 --pmc counter_collection.csv.
usage: pmc_counters.py <csv> [kernel-substring]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    a = acc[n][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    if pat not in k: continue
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"    {c:32s} launches {n:5d}  avg {s/n:16.1f}")
