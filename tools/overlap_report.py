"""Timeline of one training step from a rocprofv3 --kernel-trace csv: per queue busy time, how much of the step two
kernels overlap, and which kernels run longer than their shortest instance (sharing the GPU).
python tools/overlap_report.py kernel_trace.csv [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last-but-`steps` Adam launch ... use the last occurrence window of 'adam'
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
lo, hi = adam[-2] + 1, adam[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print("step wall %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"]))
for q, ks in byq.items():
    print("queue %s: %d launches, busy %.3f ms, first %.3f last %.3f" % (q, len(ks), sum(e - s for s, e, _ in ks) / 1e6,
                                                                      ks[0][0] / 1e6, ks[-1][1] / 1e6))
# coverage: time with >=1 and >=2 kernels in flight
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]) - t0, 1)); ev.append((int(r["End_Timestamp"]) - t0, -1))
ev.sort()
depth = 0; last = 0; cov = collections.Counter()
for t, d in ev:
    cov[min(depth, 3)] += t - last; last = t; depth += d
print("time with 0 / 1 / 2 / 3+ kernels in flight: " + " / ".join("%.3f" % (cov[i] / 1e6) for i in range(4)) + " ms")
# idle gaps > 20 us
gaps = []
depth = 0; last = 0
for t, d in ev:
    if depth == 0 and t - last > 20000: gaps.append((last, t))
    last = t; depth += d
print("idle gaps > 20 us: %d, total %.3f ms" % (len(gaps), sum(b - a for a, b in gaps) / 1e6))
# stretch of each kernel name vs its minimum over the whole trace
mn = {}
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = (r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else "")
    mn[k] = min(mn.get(k, d), d)
stretch = collections.Counter(); base = collections.Counter()
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = (r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else "")
    stretch[k[0][:60]] += d - mn[k]; base[k[0][:60]] += mn[k]
print("sum of durations %.3f ms; sum of per-kernel minima %.3f ms" % (sum(stretch.values()) / 1e6 + sum(base.values()) / 1e6,
                                                                   sum(base.values()) / 1e6))
for k, v in stretch.most_common(12):
    print("  +%.3f ms over %.3f  %s" % (v / 1e6, base[k] / 1e6, k))
