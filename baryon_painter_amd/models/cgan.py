"""Conditional GAN (pix2pix-style) on the same HIP kernels as the CVAE.

The reference documents this model (trained_models/README.md:95-144, the pickled generator
structure, ``GAN_Painter(...).paint(input, z, transform, inverse_transform)`` at
scripts/create_lightcone.py:47-54) but its code lives in an un-vendored external module, so there
is nothing to be bit-compatible with: parity is **unpinned**  and is checked against
this repository's own torch restatement (oracle/cgan_torch.py) only.

What is fixed by the reference's tables and what is a choice made here:
  * generator / discriminator layer tables, LeakyReLU(0.2), spectral norm on every discriminator
    layer, Adam(lr 5e-5, betas (0.5, 0.999)), lr x0.85 every 1568 iterations, batch 6, lambda 2.5,
    redshift conditioning f(z) = z - 1 as a constant input plane            -- from the reference;
  * losses: BCE on the discriminator's sigmoid output (README's loss levels D~ln2, G_adv~ln2/2 fit
    0.5*(BCE(real,1)+BCE(fake,0)) and 0.5*BCE(fake,1)); the "perceptual" term is NOT defined in the
    reference -- an L1 distance between generated and true field is used            -- chosen here.
"""
import ctypes as C
import os

import torch

from .. import _lib as L
from .arch import cgan_discriminator_architecture, cgan_generator_architecture
from .graph import SNConv2d, Slot, build_holders, compile_sequential, _stream


class _GanPlan:
    def __init__(self, model, n):
        self.model, self.lib, self.device, self.impl, self.sync = model, model._lib, model.device, L.IMPL_AUTO, model.sync
        self.n, self.ws_bytes, self.ws, self.prof = n, 0, None, None
        H, W = model.tile_size, model.tile_size
        dev = self.device
        self.y2 = Slot.new(n, H, W, 2, dev)
        gu, gs, tr = compile_sequential(self, "generator.", model.g_arch, model.generator, self.y2,
                                        need_input_grad=False)
        if [t[0] for t in tr] != ["tanh"]:
            raise NotImplementedError("the generator must end in conv + tanh")
        if gs.shape() != (n, H, W, 1) or gs.pw is not None:
            raise ValueError(f"generator output {gs.shape()}")
        self.g_units, self.g_raw = gu, gs
        # discriminator input: [dm, z-1, pressure] for the real (first n) and the fake (last n) half
        self.d_in = Slot.new(2 * n, H, W, 3, dev, cstride=4)
        # (the discriminator step needs no d(loss)/d(input): its first layer computes no data gradient)
        du, ds, tr = compile_sequential(self, "discriminator.", model.d_arch, model.discriminator, self.d_in,
                                        need_input_grad=False)
        if [t[0] for t in tr] != ["sigmoid"]:
            raise NotImplementedError("the discriminator must end in conv + sigmoid")
        self.d_units, self.d_raw = du, ds
        if ds.c != 1 or ds.pw is not None:
            raise ValueError("discriminator head must be a single raw channel")
        # The generator step goes through the discriminator for d(loss)/d(fake) only.  The discriminator has no batch-norm,
        # so D(fake) does not depend on the real half and the real half's gradients are exactly zero: a second set of units
        # over the FAKE half alone (same parameter holders, n images instead of 2n) does half the work of the full pass,
        # and its first layer's data gradient is restricted to the generated channel.
        self.d_in_fake = Slot(self.d_in.buf[n:], n, H, W, 3, 0)
        du2, ds2, _ = compile_sequential(self, "discriminator.", model.d_arch, model.discriminator, self.d_in_fake,
                                         need_input_grad=True)
        du2[0].restrict_dgrad(2, 3)
        self.d_units_fake, self.d_raw_fake = du2, ds2
        self.x_nchw = torch.zeros((n, 1, H, W), device=dev)
        self.sums = torch.zeros(4, device=dev, dtype=torch.float64)
        self.need_ws(256 * 8)
        for u in reversed(du):
            u.prepare_backward()
        for u in reversed(du2):
            u.prepare_backward()
        for u in reversed(gu):
            u.prepare_backward()
        self.ws = torch.zeros(max(self.ws_bytes, 256) // 8 + 32, device=dev, dtype=torch.float64)
        self.ws_bytes = self.ws.numel() * 8
        # weight gradients on a second stream beside the data-gradient chain (graph.ConvUnit.conv_backward)
        self.side = None
        if os.environ.get("BP_SIDE_WGRAD", "1") != "0":
            self.side = torch.cuda.Stream(device=dev)
            self.ws2 = torch.zeros_like(self.ws)
        n_, h_, w_ = self.d_in.n, H, W
        di = self.d_in
        self.v_real_cond = L.View(di.buf.data_ptr(), n, h_, w_, 2, di.cstride, 0)
        self.v_fake_cond = L.View(di.buf[n:].data_ptr(), n, h_, w_, 2, di.cstride, 0)
        self.v_real_x = L.View(di.buf.data_ptr(), n, h_, w_, 1, di.cstride, 2)
        self.v_fake_x = L.View(di.buf[n:].data_ptr(), n, h_, w_, 1, di.cstride, 2)
        self.d_in_fake.ensure_grad()
        self.v_dfake = L.View(self.d_in_fake.grad_buf.data_ptr(), n, h_, w_, 1, di.cstride, 2)
        self.cnt_d = float(n * ds.h * ds.w)           # discriminator outputs per half
        self.cnt_px = float(n * H * W)

    def need_ws(self, nbytes):
        self.ws_bytes = max(self.ws_bytes, int(nbytes))

    def impl_of(self, kind, unit=None):
        return self.impl

    def prof_begin(self):
        """HIP events around every launch when ``self.prof`` is a list (bench / tools), as in cvae._Plan."""
        if self.prof is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def prof_end(self, e0, unit, kind, nstreams=1):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.prof.append((e0, e1, unit, kind, nstreams))

    # ---- forward pieces
    def generate(self, y, zc, training):
        lib, st = self.lib, _stream()
        self._keep = (y.contiguous(), zc.reshape(self.n, 1).contiguous())
        L.check(lib.bp_nchw_to_view(L.ptr(self._keep[0]), 1, L.ptr(self._keep[1]), 1, C.byref(self.y2.view), st),
                "generator input")
        for u in self.g_units:
            u.forward(training)
        # fake = tanh(g_raw), written into the pressure channel of the fake half of the D input
        L.check(lib.bp_unary_forward(C.byref(self.g_raw.view), None, 1, C.byref(self.v_fake_x), st), "tanh")

    def load_real(self, x):
        lib, st = self.lib, _stream()
        self.x_nchw.copy_(x)
        y, zc = self._keep
        for v in (self.v_real_cond, self.v_fake_cond):
            L.check(lib.bp_nchw_to_view(L.ptr(y), 1, L.ptr(zc), 1, C.byref(v), st), "condition planes")
        L.check(lib.bp_nchw_to_view(L.ptr(self.x_nchw), 1, None, 0, C.byref(self.v_real_x), st), "real field")

    def discriminate(self, training, fake_only=False):
        for h in self.model.sn_layers:
            h.refresh(training)
        for u in (self.d_units_fake if fake_only else self.d_units):
            u.forward(training)

    def d_losses(self):
        """sums[0] = sum BCE(real, 1), sums[1] = sum BCE(fake, 0), sums[2] = sum BCE(fake, 1)."""
        lib, st, n, r = self.lib, _stream(), self.n, self.d_raw
        for k, (n0, n1, t) in enumerate(((0, n, 1.0), (n, 2 * n, 0.0), (n, 2 * n, 1.0))):
            L.check(lib.bp_bce_logits(C.byref(r.view), n0, n1, t, L.ptr(self.sums[k:]), L.ptr(self.ws), self.ws_bytes,
                                      st), "bce")

    def g_adv_loss(self):
        """sums[2] = sum BCE(D(fake), 1) from the fake-half units (generator step)."""
        L.check(self.lib.bp_bce_logits(C.byref(self.d_raw_fake.view), 0, self.n, 1.0, L.ptr(self.sums[2:]), L.ptr(self.ws),
                                       self.ws_bytes, _stream()), "bce")

    def backward_d_fake(self, grads, scale):
        """d(sum BCE(D(fake), 1) * scale)/d(fake) through the fake-half units (no parameter gradients: skip_wgrad)."""
        r = self.d_raw_fake
        L.check(self.lib.bp_bce_logits_grad(C.byref(r.view), 0, self.n, 1.0, scale, C.byref(r.grad), _stream()), "bce grad")
        for u in reversed(self.d_units_fake):
            u.backward(grads)
        self._join_side()

    def backward_d(self, grads, seed_real, seed_fake_target, fake_scale):
        """Seed d(loss)/d(logits): real half scale*(sigmoid-1) or 0; fake half with the given target."""
        lib, st, n, r = self.lib, _stream(), self.n, self.d_raw
        L.check(lib.bp_bce_logits_grad(C.byref(r.view), 0, n, 1.0, seed_real, C.byref(r.grad), st), "bce grad")
        L.check(lib.bp_bce_logits_grad(C.byref(r.view), n, 2 * n, seed_fake_target, fake_scale, C.byref(r.grad), st),
                "bce grad")
        for u in reversed(self.d_units):
            u.backward(grads)
        self._join_side()

    def backward_g(self, grads, l1_scale):
        lib, st = self.lib, _stream()
        L.check(lib.bp_tanh_l1_backward(C.byref(self.v_fake_x), L.ptr(self.x_nchw), C.byref(self.v_dfake), l1_scale,
                                        C.byref(self.g_raw.grad), st), "generator head backward")
        for u in reversed(self.g_units):
            u.backward(grads)
        self._join_side()

    def _join_side(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)     # every weight gradient is written


class CGAN(torch.nn.Module):
    """Generator + discriminator with their alternating training step."""

    def __init__(self, tile_size=512, device="cuda:0", n_res=9, lambda_perceptual=2.5, g_arch=None, d_arch=None,
                 sync=None):
        """``sync`` (baryon_painter_amd.dist.Sync): data parallel, one process per GPU -- the generator's batch-norm
        statistics become those of the global batch, the discriminator's and the generator's gradients are averaged
        over ranks as ONE flat buffer each, right after their backward pass (the spectral-norm power iteration is
        parameter-side arithmetic and identical on every rank)."""
        super().__init__()
        self.sync = sync
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("baryon_painter_amd.CGAN runs on an AMD GPU only; there is no CPU implementation.")
        self._lib = L.load()
        self.tile_size = tile_size
        self.lambda_perceptual = lambda_perceptual
        self.g_arch = g_arch or cgan_generator_architecture(n_res)
        self.d_arch = d_arch or cgan_discriminator_architecture()
        self.generator = build_holders(self.g_arch)
        self.discriminator = build_holders(self.d_arch)
        self._init_weights()
        self.to(self.device)
        self.sn_layers = [m for m in self.discriminator if isinstance(m, SNConv2d)]
        self._plans = {}
        self._grads = {}
        self._flat = {}
        for name, net in (("d", self.discriminator), ("g", self.generator)):
            ps = list(net.parameters())
            flat = torch.zeros(sum(p.numel() for p in ps), device=self.device)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self._flat[name] = flat
        for h in self.sn_layers:
            self._grads[id(h.weight)] = h.weight_grad
            h._grad_view = h.weight_orig.grad
        self._grad_pairs = [(p, p.grad) for p in self.parameters()]
        for p, gv in self._grad_pairs:
            self._grads[id(p)] = gv
        self.last = {}

    def _attach_grads(self):
        """``p.grad`` = the views of the flat gradient buffers the kernels write (and data parallelism averages) --
        re-attached at the start of every step: ``optimizer.zero_grad()`` / ``model.zero_grad()`` set them to None by
        default, and a gradient that lived outside the flat buffer would silently not be averaged across ranks."""
        for p, gv in self._grad_pairs:
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv

    def _init_weights(self):
        """Kaiming-normal except the generator's last layer: Xavier with gain 0.25 (README.md:102)."""
        convs = [m for m in list(self.generator.modules()) + list(self.discriminator.modules())
                 if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
        last_g = [m for m in self.generator.modules() if isinstance(m, torch.nn.Conv2d)][-1]
        for m in convs:
            w = m.weight_orig if isinstance(m, SNConv2d) else m.weight
            if m is last_g:
                torch.nn.init.xavier_normal_(w, gain=0.25)
            else:
                torch.nn.init.kaiming_normal_(w, a=0.2)
            if m.bias is not None:
                torch.nn.init.zeros_(m.bias)

    def _plan(self, n):
        if n not in self._plans:
            self._plans[n] = _GanPlan(self, n)
        return self._plans[n]

    def g_parameters(self):
        return list(self.generator.parameters())

    def d_parameters(self):
        return list(self.discriminator.parameters())

    @staticmethod
    def z_transform(z):
        return z - 1.0                                  # README.md:99

    def _inputs(self, y, z):
        y = torch.as_tensor(y, device=self.device, dtype=torch.float32)
        if y.dim() != 4 or tuple(y.shape[1:]) != (1, self.tile_size, self.tile_size):
            raise ValueError(f"y has shape {tuple(y.shape)}, model expects (N, 1, {self.tile_size}, {self.tile_size})")
        z = torch.as_tensor(z, device=self.device, dtype=torch.float32).reshape(-1)
        if z.numel() == 1 and y.shape[0] > 1:
            z = z.expand(y.shape[0])
        if z.shape[0] != y.shape[0]:
            raise ValueError("one redshift per sample")
        return y, self.z_transform(z)

    def generate(self, y, z):
        """G(dm, z) -> pressure in the network's (tanh) domain, (N,1,H,W)."""
        with torch.no_grad():
            y, zc = self._inputs(y, z)
            plan = self._plan(y.shape[0])
            plan.generate(y, zc, self.training)
            out = torch.empty((y.shape[0], 1, self.tile_size, self.tile_size), device=self.device)
            L.check(self._lib.bp_view_to_nchw(C.byref(plan.v_fake_x), None, 0, L.ptr(out), _stream()), "fake layout")
            return out

    def train_step(self, x, y, z, opt_g, opt_d, capture=None):
        """One alternating iteration: D on (real, G(y).detach()), then G through the updated D.
        Returns the loss terms as device scalars (dict)."""
        with torch.no_grad():
            x = torch.as_tensor(x, device=self.device, dtype=torch.float32)
            y, zc = self._inputs(y, z)
            n = y.shape[0]
            plan = self._plan(n)
            self._attach_grads()
            plan.generate(y, zc, self.training)
            plan.load_real(x)
            # ---- discriminator step
            plan.discriminate(self.training)
            plan.d_losses()
            plan.backward_d(self._grads, 0.5 / plan.cnt_d, 0.0, 0.5 / plan.cnt_d)
            for h in self.sn_layers:
                h.finish_backward()
            loss_d = 0.5 * (plan.sums[0] + plan.sums[1]) / plan.cnt_d
            if self.sync is not None:
                self.sync.all_reduce_mean(self._flat["d"])
            if capture is not None:        # tests: gradients of the discriminator step
                capture["d"] = {k: p.grad.clone() for k, p in self.discriminator.named_parameters()}
            opt_d.step()
            # ---- generator step (same fake, updated discriminator)
            plan.discriminate(self.training, fake_only=True)
            plan.g_adv_loss()
            lib, st = self._lib, _stream()
            L.check(lib.bp_l1_sum(C.byref(plan.v_fake_x), L.ptr(plan.x_nchw), L.ptr(plan.sums[3:]), L.ptr(plan.ws),
                                  plan.ws_bytes, st), "l1")
            plan.skip_wgrad = True          # through D only for d(loss)/d(fake): its parameters do not step here
            try:
                plan.backward_d_fake(self._grads, 0.5 / plan.cnt_d)
            finally:
                plan.skip_wgrad = False
            plan.backward_g(self._grads, self.lambda_perceptual / plan.cnt_px)
            loss_g_adv = 0.5 * plan.sums[2] / plan.cnt_d
            loss_g_perc = plan.sums[3] / plan.cnt_px
            if self.sync is not None:
                self.sync.all_reduce_mean(self._flat["g"])
            if capture is not None:
                capture["g"] = {k: p.grad.clone() for k, p in self.generator.named_parameters()}
            opt_g.step()
            self.last = {"D": loss_d, "G_adv": loss_g_adv, "G_perceptual": loss_g_perc}
            return self.last
