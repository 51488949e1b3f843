"""profiles/pmc_traffic.json from the two rocprofv3 --pmc passes of tools/profile_round.sh.
usage: pmc_to_json.py <FETCH_SIZE counter csv> <WRITE_SIZE counter csv> <out.json> [<bf16 FETCH csv> <bf16 WRITE csv>]
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read, MI355X_MICROARCH.md "HBM").  Kernel names are mapped to the labels bench.py prints; the file is stamped
with bench.source_hash() so that bench.py quotes a figure only for the kernel sources it was measured on."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def load(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        acc[n][0] += float(r["Counter_Value"]); acc[n][1] += 1
    return acc


def label(name):
    """rocprof kernel name -> bench.py label (the fp32 kernels whose template arguments identify them)."""
    m = re.match(r"igemm_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        cc, nt, wn, slots, nw = map(int, m.groups())
        return "%s<%d,%d,%d,4>" % ("igemm_dma8_kernel" if nw == 8 else "igemm_dma_kernel", cc, nt, wn)
    m = re.match(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return "igemm_kernel<%s,%s,%s,%s>" % m.groups()
    m = re.match(r"igemm_dmaf_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return "igemm_dmaf_kernel<%s,%s,1,4>" % m.groups()[:2]
    # bf16 step: the two trunk kernels are the only users of their template instances
    if name.startswith("wgrad_bf16_kernel<3, 3, 1, 2, 2, 2, 2, 4, true, true>"):
        return "wgrad_bf16_kernel[C128->128 k3s1]"
    if name.startswith("bpbf16::igemm_bf16_kernel<32, 4, 2, 8, 6, true, true, 8>"):
        return "igemm_bf16_kernel[C128->128 k3s1]"
    return name.split("(")[0]


out = {}
pairs = [(sys.argv[1], sys.argv[2])] + ([(sys.argv[4], sys.argv[5])] if len(sys.argv) > 5 else [])
for fp, wp in pairs:
    f, w = load(fp), load(wp)
    for k in f:
        fa = f[k][0] / f[k][1]
        wa = w[k][0] / w[k][1] if k in w else 0.0
        out.setdefault(label(k), int((2 * fa + wa) * 1024))
k = "igemm_bf16_kernel[C128->128 k3s1]"
if k in out:                       # (forward and data gradient are launches of the same instance)
    out[k[:-1] + " fwd]"] = out[k[:-1] + " dgrad]"] = out.pop(k)
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh, serial "
                     "schedule), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged per launch (gfx950 FETCH_SIZE correction)",
           "source_hash": bench.source_hash(), "hbm_bytes_per_launch": dict(sorted(out.items(), key=lambda kv: -kv[1]))},
          open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels, stamp", bench.source_hash())
