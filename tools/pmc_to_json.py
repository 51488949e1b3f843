"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes of tools/profile_round.sh.
usage: pmc_to_json.py <out.json> f32:<FETCH csv>:<WRITE csv> [bf16:<FETCH csv>:<WRITE csv>] [paint_f32:...] ...

HBM bytes of a dispatch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read, MI355X_MICROARCH.md "HBM"; separate passes: the TCC block has 4 counter slots).  Per leg (dtype) the file
holds
  * kernels: one entry per (kernel instance, grid size, workgroup size) -- NOT per kernel name: one instance serves layers of
    different sizes, and fp32 / bf16 steps share streaming kernels -- with launches per step and bytes per launch;
  * by_label: the same bytes under the labels bench.py prints for its roofline (launch-weighted over the shapes a label covers);
  * step_bytes: the sum over ONE training step (the dispatches between two consecutive optimizer launches of the steady
    state), which bench.py divides by the algorithmic bytes of the step (`roofline.whole_step.traffic_ratio`).
The file is stamped with bench.source_hash() so that bench.py quotes a figure only for the kernel sources it was measured on."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

STEP_MARK = ("adam_dev_kernel", "adam_kernel")          # one launch per training step (FlatAdam)


def clean(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def load(path):
    """[(dispatch id, kernel, grid, workgroup, value)] in dispatch order."""
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Dispatch_Id"]), clean(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]),
                     float(r["Counter_Value"])))
    rows.sort()
    return rows


def label(name):
    """rocprof kernel instance -> the label bench.py prints (None: no roofline label for it)."""
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
    if name.startswith("ws3_f32_kernel"):
        return "ws3_f32_kernel[C128->128 k3s1]"
    if name.startswith("ws4_bf16_kernel"):
        return "ws4_bf16_kernel"
    if name.startswith("wst_bf16_kernel"):
        return "ws5_bf16_kernel"
    if name.startswith("ws3_bf16_kernel"):
        return "ws3_bf16_kernel[C128->128 k3s1]"
    if name.startswith("wgrad_bf16_kernel<3, 3, 1, 2, 2, 2, 2, 4"):
        return "wgrad_bf16_kernel[C128->128 k3s1]"
    return re.sub(r"<.*", "", name)          # (streaming passes etc.: bench.py names the family)


def one_step(rows):
    """Dispatches of the LAST complete training step: between the last two optimizer launches."""
    marks = [i for i, r in enumerate(rows) if r[1].startswith(STEP_MARK)]
    if len(marks) < 2:
        return rows, 1
    return rows[marks[-2] + 1:marks[-1] + 1], 1


def leg(fetch_csv, write_csv):
    f, w = load(fetch_csv), load(write_csv)
    fs, _ = one_step(f)
    ws_, _ = one_step(w)
    acc = collections.OrderedDict()
    for rows, col in ((fs, 0), (ws_, 1)):
        for _, name, grid, wg, v in rows:
            e = acc.setdefault((name, grid, wg), [0.0, 0.0, 0, 0])
            e[col] += v
            e[2 + col] += 1
    kernels, by_label, step = [], collections.defaultdict(lambda: [0.0, 0]), 0.0
    for (name, grid, wg), (fsum, wsum, nf, nw) in acc.items():
        n = max(nf, nw)
        tot = (2 * fsum + wsum) * 1024
        step += tot
        kernels.append({"kernel": name, "grid": grid, "workgroup": wg, "launches_per_step": n,
                        "hbm_bytes_per_launch": int(tot / max(n, 1)), "fetch_KiB_per_launch": round(fsum / max(nf, 1), 1),
                        "write_KiB_per_launch": round(wsum / max(nw, 1), 1)})
        lb = label(name)
        by_label[lb][0] += tot
        by_label[lb][1] += n
    kernels.sort(key=lambda k: -k["hbm_bytes_per_launch"] * k["launches_per_step"])
    return {"step_bytes": int(step), "dispatches_per_step": sum(k["launches_per_step"] for k in kernels), "kernels": kernels,
            "by_label": {k: {"hbm_bytes_per_launch": int(v[0] / v[1]), "launches_per_step": v[1]} for k, v in
                         sorted(by_label.items(), key=lambda kv: -kv[1][0])}}


if __name__ == "__main__":
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh, serial schedule); "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch (gfx950 FETCH_SIZE correction), one training step "
                     "= the dispatches between the last two optimizer launches; entries per (kernel instance, grid, workgroup)",
           "source_hash": bench.source_hash(), "legs": {}}
    for spec in sys.argv[2:]:
        name, fp, wp = spec.split(":")
        out["legs"][name] = leg(fp, wp)
        print(name, "step bytes %.2f GB" % (out["legs"][name]["step_bytes"] / 1e9), out["legs"][name]["dispatches_per_step"], "dispatches")
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print("wrote", sys.argv[1], "stamp", bench.source_hash())
