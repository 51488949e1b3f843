"""Matrix-core utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).
    python tools/mfma_util.py <counter_collection.csv> [top N]
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the launch's cycles in which a
SIMD's matrix pipe is busy, averaged over the chip's 1024 SIMDs (rocprofv3 sums the SQ counters over all SIMDs and
GRBM_GUI_ACTIVE over the 8 XCDs: MI355X_MICROARCH.md "DVFS give-back").  clock = GRBM_GUI_ACTIVE / 8 / duration.
wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / barriers), stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import collections, csv, sys
acc = collections.defaultdict(lambda: {"n": collections.Counter(), "v": collections.Counter(), "ns": 0.0, "disp": set()})
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("bpbf16::", "")
    a = acc[name]
    a["v"][r["Counter_Name"]] += float(r["Counter_Value"]); a["n"][r["Counter_Name"]] += 1
    if r["Dispatch_Id"] not in a["disp"]:
        a["disp"].add(r["Dispatch_Id"])
        a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
for name, a in acc.items():
    n = len(a["disp"])
    avg = lambda c: a["v"][c] / max(a["n"][c], 1)
    g = avg("GRBM_GUI_ACTIVE")
    if g <= 0:
        continue
    dur = a["ns"] / n
    wc = max(avg("SQ_WAVE_CYCLES"), 1.0)
    rows.append((a["ns"], name, n, dur / 1e3, avg("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * g / 8.0), g / 8.0 / dur,
                 avg("SQ_WAIT_ANY") / wc, avg("SQ_WAIT_INST_ANY") / wc, avg("SQ_ACTIVE_INST_ANY") / wc))
rows.sort(reverse=True)
print("%-78s %6s %10s %9s %6s %6s %6s %6s" % ("kernel (under the profiler: counters serialise the launches)", "calls", "avg us",
                                          "mfma_util", "GHz", "wait", "stall", "issue"))
for _, name, n, us, mu, ghz, w, s, act in rows[:top]:
    print("%-78s %6d %10.1f %9.3f %6.2f %6.2f %6.2f %6.2f" % (name[:78], n, us, mu, ghz, w, s, act))
