"""Per-layer timing table of the CGAN iteration (serial schedule): python tools/cgan_layers.py [batch] [tile]"""
import contextlib, os, sys
os.environ["BP_SIDE_WGRAD"] = "0"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from baryon_painter_amd.models.cgan import CGAN
from baryon_painter_amd.utils import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model = CGAN(tile_size=tile, device=dev)
nb = min(n, 8)
x, y, z = syn.synthetic_batch(nb, tile, tile, seed=1234)
reps = (n + nb - 1) // nb
x = torch.from_numpy(np.tanh(3 * np.tile(x, (reps, 1, 1, 1))[:n] - 0.5).astype(np.float32)).to(dev)
y = torch.from_numpy(np.tile(y, (reps, 1, 1, 1))[:n]).to(dev)
z = torch.from_numpy(np.tile(z, reps)[:n]).to(dev)
og = torch.optim.Adam(model.g_parameters(), lr=5e-5, betas=(0.5, 0.999))
od = torch.optim.Adam(model.d_parameters(), lr=5e-5, betas=(0.5, 0.999))
for _ in range(2):
    model.train_step(x, y, z, og, od)
plan = model._plan(n)
plan.prof = []
model.train_step(x, y, z, og, od)
torch.cuda.synchronize()
lay = {}
for e0, e1, unit, kind, ns in plan.prof:
    fl = 2.0 * unit.macs(kind) if kind in ("forward", "backward_data", "backward_weight") else 0.0
    d = lay.setdefault((unit.name, kind), [0.0, 0, fl, bench.kernel_name(kind, unit, model._lib)])
    d[0] += e0.elapsed_time(e1); d[1] += 1
tot = 0.0
for (name, kind), (ms, cnt, fl, kn) in sorted(lay.items(), key=lambda kv: -kv[1][0]):
    tot += ms
    print(f"{name:36s} {kind:16s} x{cnt} {ms:8.3f} ms  {fl * cnt / ms / 1e9 if ms else 0:7.2f} TF/s  {kn}")
print("total", tot)
