"""Average FETCH_SIZE / WRITE_SIZE (KiB) per launch of each kernel from rocprofv3 --pmc CSVs.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads: MI355X_MICROARCH.md, HBM section)."""
import csv, sys, collections
def load(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        acc[n][0] += float(r["Counter_Value"]); acc[n][1] += 1
    return acc
f, w = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in f:
    fa = f[k][0] / f[k][1]; wa = w[k][0] / w[k][1] if k in w else 0.0
    rows.append(((2 * fa + wa) * 1024 * f[k][1], k, f[k][1], fa, wa))
for tot, k, n, fa, wa in sorted(rows, reverse=True)[:14]:
    print(f"{k:60s} launches {n:4d}  FETCH {fa/1024:9.1f} MiB  WRITE {wa/1024:9.1f} MiB  HBM/launch {(2*fa+wa)/1024:9.1f} MiB")
