"""Fused Adam over the model's single flat parameter buffer.

Same arithmetic as ``torch.optim.Adam`` (the reference's optimiser, painter.py:93,228: betas
(0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) in ONE kernel launch (``bp_adam_step``) over all
1.66 M parameters instead of one multi-tensor sweep per state tensor.  It is a
``torch.optim.Optimizer`` so learning-rate schedulers (``LambdaLR`` in the training script) drive
it through ``param_groups[0]["lr"]`` as usual.
"""
import ctypes as C
import math

import torch

from . import _lib as L


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not hasattr(model, "_flat_params"):
            raise TypeError("FlatAdam needs a baryon_painter_amd CVAE (flat parameter storage)")
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self.model = model
        self._lib = L.load()
        self.exp_avg = torch.zeros_like(model._flat_params)
        self.exp_avg_sq = torch.zeros_like(model._flat_params)
        self.n_steps = 0
        self._storage = model._flat_params.data_ptr()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self.model
        if m._flat_params.data_ptr() != self._storage:
            raise RuntimeError("the model's parameter storage moved (e.g. .to()); create a new optimiser")
        for p, gv in zip(m._params, m._grad_views):
            if p.grad is None:
                raise RuntimeError("FlatAdam.step() before backward(): a parameter has no gradient")
            if p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)                    # a foreign .grad tensor: bring it into the flat buffer
        g = self.param_groups[0]
        self.n_steps += 1
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self._lib.bp_adam_step(L.ptr(m._flat_params), L.ptr(m._flat_grads), L.ptr(self.exp_avg),
                                       L.ptr(self.exp_avg_sq), m._flat_params.numel(), float(g["lr"]),
                                       float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self.n_steps,
                                       st), "adam step")
        m._bump_param_versions()                    # the convolution units re-pack their weights
        return loss

    # ---- the same update as a launch that can live in a captured hipGraph (CVAE.make_graphed_train_step)
    def _hyper_dev_buffer(self):
        if not hasattr(self, "_hyper_dev"):
            self._hyper_dev = torch.zeros(6, dtype=torch.float32, device=self.model._flat_params.device)
        return self._hyper_dev

    def upload_hyper(self, step):
        """{lr, beta1, beta2, eps, 1-beta1^step, sqrt(1-beta2^step)} of update number ``step`` -> device (the
        same double-precision bias corrections as bp_adam_step, so graph replay and eager steps agree bit for bit).
        The source is a fresh pageable tensor: the runtime stages it before returning, so a host that runs ahead of
        the GPU cannot overwrite the scalars of a step that has not executed yet."""
        g = self.param_groups[0]
        # bp_adam_step receives beta1/beta2 as C floats and forms the corrections from those in double
        b1 = float(torch.tensor(g["betas"][0], dtype=torch.float32))
        b2 = float(torch.tensor(g["betas"][1], dtype=torch.float32))
        vals = [float(g["lr"]), b1, b2, float(g["eps"]), 1.0 - math.pow(b1, step), math.sqrt(1.0 - math.pow(b2, step))]
        self._hyper_dev_buffer().copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=True)

    @torch.no_grad()
    def device_step(self):
        """Launch the update with the scalars last uploaded by ``upload_hyper`` (capturable)."""
        m = self.model
        dev = self._hyper_dev_buffer()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self._lib.bp_adam_step_dev(L.ptr(m._flat_params), L.ptr(m._flat_grads), L.ptr(self.exp_avg),
                                           L.ptr(self.exp_avg_sq), m._flat_params.numel(), L.ptr(dev), st),
                "adam step (device scalars)")
