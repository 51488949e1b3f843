"""Where does paint_stream lose time against the resident graph?  Replays the two slot graphs of CVAE.paint_graph with
parts of the pipeline switched off:  python tools/paint_probe.py [f32|bf16] [batch] [nbatch]
  graph      back-to-back replays of slot 0 (the 'resident' figure of bench.py)
  alt        alternating slot 0 / slot 1 replays, nothing else
  up         + the uploads on their stream (events as in paint_stream), no downloads
  down       + the downloads, no uploads
  both       uploads and downloads (= paint_stream without its host-side work)
  stream     CVAEPainter.paint_stream itself
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import contextlib
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.painter import CVAEPainter
from baryon_painter_amd.utils.datasets import SyntheticTileDataset

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = "cuda:0"
torch.manual_seed(1)
with contextlib.redirect_stdout(sys.stderr):
    model = CVAE(A.fiducial_architecture(512), dev, dtype=dtype)
model.train(False)
g = model.paint_graph(B)
H = W = 512
hin = [torch.rand((B, 1, H, W)).pin_memory() for _ in range(2)]
hout = [torch.empty((B, 1, H, W)).pin_memory() for _ in range(2)]
hblk = [torch.zeros(g["block_bytes"], dtype=torch.uint8).pin_memory() for _ in range(2)]
for s, sl in enumerate(g["slots"]):
    sl["raw"].copy_(hin[s]); sl["xf_in"].fill_(1.0); sl["xf_out"].fill_(1.0)
main = torch.cuda.current_stream()
prio = int(os.environ.get("COPY_PRIO", "0"))
up, down = torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)


def run(mode):
    ev = [{k: torch.cuda.Event() for k in ("up", "done", "down")} for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(NB):
        s = 0 if mode == "graph" else b % 2
        sl, e = g["slots"][s], ev[s]
        if mode in ("up", "both"):
            up.wait_event(e["done"])
            with torch.cuda.stream(up):
                sl["raw"].copy_(hin[s], non_blocking=True)
                e["up"].record(up)
            main.wait_event(e["up"])
        if mode in ("down", "both"):
            main.wait_event(e["down"])
        sl["graph"].replay()
        e["done"].record(main)
        if mode in ("down", "both"):
            down.wait_event(e["done"])
            with torch.cuda.stream(down):
                hout[s].copy_(sl["out"], non_blocking=True)
                e["down"].record(down)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / NB
    print(f"{mode:8s} {dt * 1e3:8.3f} ms/batch  {B / dt:9.1f} tiles/s", flush=True)


for mode in ("graph", "alt", "up", "down", "both"):
    run(mode); run(mode)

# bare copies: PCIe rates with nothing else running
for name, fn in (("H2D", lambda: g["slots"][0]["raw"].copy_(hin[0], non_blocking=True)),
                 ("D2H", lambda: hout[0].copy_(g["slots"][0]["out"], non_blocking=True))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"{name} {B} MiB: {dt * 1e3:.3f} ms = {B * 2 ** 20 / dt / 1e9:.1f} GB/s", flush=True)

ds = SyntheticTileDataset(n_sample=8, tile_size=512, seed=3)
pt = CVAEPainter.__new__(CVAEPainter)
pt.model, pt.compute_device, pt.sync = model, dev, None
pt.input_field, pt.label_fields = ds.input_field, ds.label_fields
pt.transform, pt.inverse_transform = ds.transform, ds.inverse_transform
n = B * NB
raw = np.stack([ds.raw_fields(i)[0] for i in range(8)])
tin = torch.from_numpy(np.tile(raw, (n // 8, 1, 1))).pin_memory()
tout = torch.empty((n, H, W)).pin_memory()
zs = np.zeros(n)
with torch.no_grad():
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pt.paint_stream(tin, zs, batch_size=B, out=tout)
        dt = (time.perf_counter() - t0) / NB
        print(f"stream   {dt * 1e3:8.3f} ms/batch  {B / dt:9.1f} tiles/s", flush=True)

# ---- timeline of the 'both' pipeline: when do the copies actually run relative to the graphs?
T = lambda: torch.cuda.Event(enable_timing=True)
ev = [{k: torch.cuda.Event() for k in ("up", "done", "down")} for _ in range(2)]
marks = []
torch.cuda.synchronize()
t_ref = T(); t_ref.record(main)
for b in range(8):
    s = b % 2
    sl, e = g["slots"][s], ev[s]
    m = {k: T() for k in ("u0", "u1", "g0", "g1", "d0", "d1")}
    up.wait_event(e["done"])
    with torch.cuda.stream(up):
        m["u0"].record(up)
        sl["raw"].copy_(hin[s], non_blocking=True)
        m["u1"].record(up)
        e["up"].record(up)
    main.wait_event(e["up"]); main.wait_event(e["down"])
    m["g0"].record(main)
    sl["graph"].replay()
    m["g1"].record(main)
    e["done"].record(main)
    down.wait_event(e["done"])
    with torch.cuda.stream(down):
        m["d0"].record(down)
        hout[s].copy_(sl["out"], non_blocking=True)
        m["d1"].record(down)
        e["down"].record(down)
    marks.append(m)
torch.cuda.synchronize()
for b, m in enumerate(marks):
    print("batch %d: " % b + "  ".join("%s %.2f" % (k, t_ref.elapsed_time(m[k])) for k in ("u0", "u1", "g0", "g1", "d0", "d1")))

# ---- is graph.replay() blocking the host?  and the pipeline with the NEXT batch's upload enqueued before the replay
torch.cuda.synchronize()
t0 = time.perf_counter(); g["slots"][0]["graph"].replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host time inside replay(): %.3f ms; until the GPU is done: %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))


def run_ahead():
    ev = [{k: torch.cuda.Event() for k in ("up", "done", "down")} for _ in range(2)]

    def upload(b):
        s = b % 2
        up.wait_event(ev[s]["done"])
        with torch.cuda.stream(up):
            g["slots"][s]["raw"].copy_(hin[s], non_blocking=True)
            ev[s]["up"].record(up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    upload(0)
    for b in range(NB):
        s = b % 2
        sl, e = g["slots"][s], ev[s]
        if b + 1 < NB:
            upload(b + 1)
        main.wait_event(e["up"]); main.wait_event(e["down"])
        sl["graph"].replay()
        e["done"].record(main)
        down.wait_event(e["done"])
        with torch.cuda.stream(down):
            hout[s].copy_(sl["out"], non_blocking=True)
            e["down"].record(down)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / NB
    print(f"ahead    {dt * 1e3:8.3f} ms/batch  {B / dt:9.1f} tiles/s", flush=True)


run_ahead(); run_ahead()
