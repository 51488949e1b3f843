"""Debug aid: number of small collectives per training step under 2 ranks (run under torch.distributed.run)."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from baryon_painter_amd.dist import Sync
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
sync = Sync()
m = CVAE(A.fiducial_architecture(64), "cuda:0", sync=sync); m.train(True)
x, y, aux = [torch.from_numpy(t) for t in syn.synthetic_batch(2, 64, 64, seed=sync.rank)]
e = m(x, y, aux); (-e).backward()
n0 = sync.n_small
e = m(x, y, aux); (-e).backward()
if sync.rank == 0:
    print("BP_LEVEL_SYNC=%s: small collectives per step: %d" % (os.environ.get("BP_LEVEL_SYNC", "1"), sync.n_small - n0))
dist.destroy_process_group()
