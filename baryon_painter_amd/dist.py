"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device (scripts/CVAE_single_scale.py:23).  Sharding the minibatch
changes nothing arithmetically if (a) batch-norm uses GLOBAL batch statistics -- forward
{sum x, sum x^2} and backward {sum g, sum g*x} are all-reduced per layer (tiny messages) -- and
(b) parameter gradients are averaged across ranks (each rank normalises its loss by its local
batch, cvae.py:129,144).  Gradients travel as ONE flat buffer (all parameters are views of a
single allocation, see CVAE._flatten_parameters): 6.65 MB per step for the fiducial network.
"""
import torch
import torch.distributed as dist


class Sync:
    """Collectives the launch plan needs.  ``sync_bn=False`` keeps batch-norm statistics local
    (throughput mode: different arithmetic from the single-device reference)."""

    def __init__(self, group=None, sync_bn=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sync_bn = sync_bn
        self.n_small = 0

    def all_reduce_sum(self, t):
        """Batch-norm statistics (float64 vector of 2*C entries)."""
        if self.sync_bn and self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.n_small += 1

    def all_reduce_mean(self, flat):
        """The flat gradient buffer."""
        if self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / self.world_size)


class LocalBNSync(Sync):
    """Gradient averaging only."""

    def __init__(self, group=None):
        super().__init__(group, sync_bn=False)

    @property
    def bn_world(self):
        return 1


def shard_indices(permutation, rank, world_size, batch_size):
    """Slice a GLOBAL sample permutation (the reference's DataLoader(shuffle=True) order,
    painter.py:87-91) into this rank's share of every global batch of ``batch_size*world_size``
    samples: rank r takes positions [r*batch_size, (r+1)*batch_size) of each global batch, so
    the union over ranks reproduces the single-device batches exactly."""
    gb = batch_size * world_size
    out = []
    for start in range(0, len(permutation) - gb + 1, gb):
        out.append(list(permutation[start + rank * batch_size:start + (rank + 1) * batch_size]))
    return out
