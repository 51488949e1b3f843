"""Wall time of the phases of the OVERLAPPED training step (rocprofv3 serialises kernels and cannot show this):
timed events on the main stream at the phase boundaries of cvae._Plan (BP_PHASE_EVENTS=1), averaged over steps.
  python tools/phase_times.py [f32|bf16] [steps]
A phase's time is main-stream time: kernels of the weight-gradient / branch streams that run beside it slow it down,
and the time until they are joined shows up in the 'join' rows."""
import os, sys, contextlib, collections
os.environ["BP_PHASE_EVENTS"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.optim import FlatAdam
from baryon_painter_amd.utils import synthetic as syn

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev, n, tile = "cuda:0", 64, 512
torch.manual_seed(1234)
with contextlib.redirect_stdout(sys.stderr):
    model = CVAE(A.fiducial_architecture(tile), dev, dtype=dtype)
opt = FlatAdam(model, lr=1e-3)
x, y, aux = syn.synthetic_batch(8, tile, tile, seed=1234)
x = torch.from_numpy(np.tile(x, (8, 1, 1, 1))).to(dev)
y = torch.from_numpy(np.tile(y, (8, 1, 1, 1))).to(dev)
aux = torch.from_numpy(np.tile(aux, 8)).to(dev)


def step():
    elbo = model(x, y, aux)
    opt.zero_grad()
    (-elbo).backward()
    opt.step()


for _ in range(3):
    step()
plan = model._last
torch.cuda.synchronize()
plan.marks.clear()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    step()
t1.record()
torch.cuda.synchronize()
marks = plan.marks
per = len(marks) // steps
tot = collections.OrderedDict()
for s in range(steps):
    ms = marks[s * per:(s + 1) * per]
    nxt = marks[(s + 1) * per][1] if s + 1 < steps else None
    for i, (name, ev) in enumerate(ms):
        if i + 1 < len(ms):
            key, dt = ms[i + 1][0], ev.elapsed_time(ms[i + 1][1])
        elif nxt is not None:
            key, dt = "optimizer + next step's launch gap", ev.elapsed_time(nxt)
        else:
            continue
        tot.setdefault(key, []).append(dt)
print(f"{dtype}: {t0.elapsed_time(t1) / steps:.3f} ms per step over {steps} steps")
for k, v in tot.items():
    print(f"  {k:40s} {np.mean(v):8.3f} ms")
