"""MI355X-native CVAE with the reference's interface.

Mirrors ``baryon_painter.models.cvae.CVAE`` (/root/reference/baryon_painter/models/cvae.py:8-199):
same constructor, attributes (``z_mu``, ``z_log_var``, ``KL_term``, ``log_likelihood``, ``ELBO``,
``x_mu``, ``alpha_var``, ``beta_KL``, ``dim_*``, ``architecture``), methods and ``state_dict``
keys -- but ``forward``/``sample_P`` run a launch plan of hand-written gfx950 kernels
(``csrc/``) instead of ``torch.nn`` modules, and the backward pass is scheduled explicitly
(one ``torch.autograd.Function`` for the whole network).

There is no CPU implementation: constructing the model on a non-GPU device or without the
built library raises.
"""
import ctypes as C
import math
import os

import torch

from .. import _lib as L
from .arch import conv_block, conv_down, conv_up, res_block  # noqa: F401  (reference: ``from .utils import *``)
from .graph import (PW, ConvUnit, PackBatch, Slot, build_holders, compile_sequential, probe_output, _stream)

pi = math.pi


class _Plan:
    """All device buffers and the launch schedule for one (batch, mode) signature."""

    def __init__(self, model, n, with_grad, with_q=True):
        self.model = model
        self.lib = model._lib
        self.device = model.device
        self.impl = model.impl
        self.bf16 = model.dtype == "bf16"
        self.sync = model.sync
        self.n = n
        self.with_grad = with_grad
        self.with_q = with_q
        self.ws_bytes = 0
        self.ws = None
        a = model.architecture
        dev = self.device
        Lr = model.L
        cy, H, W = model.dim_y
        cx = model.dim_x[0]
        if tuple(model.dim_x[1:]) != (H, W):
            raise NotImplementedError("dim_x and dim_y must share the spatial size")
        caux = model.n_aux if model.use_aux_label else 0
        zc, zh, zw = model.dim_z
        self.caux = caux

        # ---- inputs: y (+aux planes) as NHWC; x is read as NCHW by the loss and as NHWC by Q
        self.y2 = Slot.new(n, H, W, cy + caux, dev)
        self.x_in = Slot.new(n, H, W, cx, dev)

        # ---- recognition network Q (cvae.py:68-80): cat[q_x_in(x), q_y_in(y)] -> q_out
        self.q_units, self.q_head = [], None
        if with_q:
            cqx, hq, wq = probe_output(a["q_x_in"], cx, H, W)
            cqy, hq2, wq2 = probe_output(a["q_y_in"], cy + caux, H, W)
            if (hq, wq) != (hq2, wq2):
                raise ValueError("q_x_in and q_y_in outputs differ in spatial size")
            cat = Slot.new(n, hq, wq, cqx + cqy, dev)
            cat.pw = PW.identity(cqx + cqy, dev)
            sx0 = cat.sub(0, cqx, pw=cat.pw.slice(0, cqx))
            sy0 = cat.sub(cqx, cqx + cqy, pw=cat.pw.slice(cqx, cqx + cqy))
            ux, sx, tr = compile_sequential(self, "q_x_in.", a["q_x_in"], model.q_x_in, self.x_in,
                                            out_slot=sx0, out_pw=sx0.pw, need_input_grad=False)
            self._no_trailing(tr, "q_x_in")
            uy, sy, tr = compile_sequential(self, "q_y_in.", a["q_y_in"], model.q_y_in, self.y2,
                                            out_slot=sy0, out_pw=sy0.pw, need_input_grad=False)
            self._no_trailing(tr, "q_y_in")
            if sx is not sx0 or sy is not sy0:
                raise NotImplementedError("q_x_in / q_y_in must end in a convolution")
            uo, so, tr = compile_sequential(self, "q_out.", a["q_x_y_out"], model.q_out, cat)
            self._latent_trailing(tr, "q_x_y_out")
            self.q_units = [ux, uy, uo]
            self.q_head = so
            if so.shape() != (n, zh, zw, 2 * zc):
                raise ValueError(f"q_x_y_out produces {so.shape()}, dim_z needs {(n, zh, zw, 2 * zc)}")

        # ---- prior network (cvae.py:82-95)
        self.p_units, self.p_head = [], None
        if model.prior_network is not None:
            up, sp, tr = compile_sequential(self, "prior_network.", a["prior_z_y"], model.prior_network,
                                            self.y2, need_input_grad=False)
            self._latent_trailing(tr, "prior_z_y")
            self.p_units, self.p_head = up, sp
            if sp.shape() != (n, zh, zw, 2 * zc):
                raise ValueError(f"prior_z_y produces {sp.shape()}, dim_z needs {(n, zh, zw, 2 * zc)}")

        # ---- latent
        self.lat = L.Latent(n, Lr, zc, zh, zw, float(model.min_z_var))
        self.z = Slot.new(n * Lr, zh, zw, zc, dev)
        self.stats4 = torch.zeros((4, n, zc, zh, zw), device=dev)
        self.kl_sum = torch.zeros(1, device=dev, dtype=torch.float64)
        self.eps = None
        self.need_ws(256 * 8)

        # ---- generator P (cvae.py:103-120)
        nL = n * Lr
        # p_z_in output and p_y_in(y) are concatenated: [h_z, h_y.repeat(L)]
        c_hz, hz_h, hz_w = probe_output(a["p_z_in"], zc, zh, zw)
        if (hz_h, hz_w) != (H, W):
            raise ValueError(f"p_z_in produces {hz_h}x{hz_w}, dim_y is {H}x{W}")
        if model.p_y_in is not None:
            raise NotImplementedError("p_y_in networks are not supported by the HIP path yet "
                                      "(the reference configurations use p_y_in=None)")
        c_hy = cy + caux
        ccat = c_hz + c_hy
        self.p_in = Slot.new(nL, H, W, ccat, dev, cstride=((ccat + 3) // 4) * 4)
        self.p_in.pw = PW.identity(ccat, dev)
        hz_slot = self.p_in.sub(0, c_hz, pw=self.p_in.pw.slice(0, c_hz),
                                dense_grad=os.environ.get("BP_DENSE_ZGRAD", "1") != "0")
        self.hy_slot = self.p_in.sub(c_hz, ccat)
        uz, sz, tr = compile_sequential(self, "p_z_in.", a["p_z_in"], model.p_z_in, self.z,
                                        out_slot=hz_slot, out_pw=hz_slot.pw)
        self._no_trailing(tr, "p_z_in")
        if sz is not hz_slot:
            raise NotImplementedError("p_z_in must end in a convolution")
        ub, sb, tr = compile_sequential(self, "p_y_z_in.", a["p_y_z_in"], model.p_y_z_in, self.p_in)
        self._no_trailing(tr, "p_y_z_in")
        if ub and isinstance(ub[0], ConvUnit):
            ub[0].restrict_dgrad(0, c_hz, target=hz_slot)   # y and the aux label are data: only h_z carries a gradient
        self.g_units = [uz, ub]
        self.h = sb
        um, sm, tr = compile_sequential(self, "p_mu_out.", a["p_y_z_out"][0], model.p_mu_out, sb)
        self.mu_softplus = self._head_trailing(tr, "p_y_z_out[0]")
        self.mu_units, self.mu_head = um, sm
        self.var_units, self.var_head = [], None
        if model.predict_var:
            uv, sv, tr = compile_sequential(self, "p_var_out.", a["p_y_z_out"][1], model.p_var_out, sb)
            if self._head_trailing(tr, "p_y_z_out[1]"):
                raise NotImplementedError("softplus on the variance head")
            self.var_units, self.var_head = uv, sv
        if sm.shape() != (nL, H, W, cx):
            raise ValueError(f"p_y_z_out produces {sm.shape()}, dim_x needs {(nL, H, W, cx)}")
        for s in (sm, self.var_head):
            if s is not None and s.pw is not None:
                raise NotImplementedError("a head must end in conv (+softplus), without batch-norm/ReLU")

        # ---- loss
        self.ll = L.Loglik(n, Lr, cx, H, W, 1 if self.mu_softplus else 0, 1 if model.predict_var else 0,
                           1.0, 1.0, float(model.likelihood_scaling))
        # (one channel: NCHW and NHWC are the same bytes -- the loss reads x from the recognition network's input slot)
        self.x_nchw = self.x_in.buf.view(n, cx, H, W) if cx == 1 and self.x_in.cstride == 1 \
            else torch.zeros((n, cx, H, W), device=dev)
        self.x_mu = torch.zeros((nL, cx, H, W), device=dev)
        self.x_log_var = torch.zeros((nL, cx, H, W), device=dev) if model.predict_var else None
        self.stats = torch.zeros(2 + 3 * cx, device=dev)
        self.need_ws(self.lib.bp_loglik_workspace(C.byref(self.ll)))

        if with_grad:
            self._prepare_backward()
        self.ws = torch.zeros(max(self.ws_bytes, 256) // 8 + 32, device=dev, dtype=torch.float64)
        self.ws_bytes = self.ws.numel() * 8
        # Weight gradients on a second stream: they depend only on a layer's d_raw and its input, not on the
        # chain act-backward -> data gradient -> act-backward of the layers below, so their MFMA work can run
        # beside that chain's HBM-bound passes.  They get their own workspace.
        # Split-K reductions of the weight gradients: recorded during the backward pass and launched together at its
        # end (bp_wgrad_defer_*: two launches instead of ~30 latency-bound ones).  Such a layer keeps its partial sums
        # until the flush, in a workspace of its own.  fp32 layers only: the 128 ... 512-way splits of the bf16 kernels
        # are cheaper reduced at once, out of L2 (measured: deferring them costs the bf16 step 0.15 ms).
        self.marks = [] if os.environ.get("BP_PHASE_EVENTS") == "1" else None
        self.deferring = False
        self.defer_reduce = with_grad and os.environ.get("BP_DEFER_REDUCE", "1") != "0"
        if self.defer_reduce:
            for u in self._flat([u for us in self.q_units for u in us] + list(self.p_units)
                                + [u for us in self.g_units for u in us] + list(self.mu_units) + list(self.var_units)):
                # (BP_DEFER_MAX_MB: layers with larger partial sums reduce at once.  Measured: deferring ALL fp32 layers,
                #  the trunk's 16 MB and the encoder layer's 25 MB of partial sums included, is best -- 41.3 vs 41.6 ms
                #  with a 4 MB limit -- although the batched launch then ends the backward pass with 50 us of reads)
                if isinstance(u, ConvUnit) and 0 < getattr(u, "_wgrad_ws_bytes", 0) <= DEFER_MAX_BYTES and not u.bf16:
                    u.ws_own = torch.empty(u._wgrad_ws_bytes // 8 + 32, device=dev, dtype=torch.float64)
        self.side = self._side_stream = None
        if with_grad and os.environ.get("BP_SIDE_WGRAD", "1") != "0":
            self.side = self._side_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("BP_SIDE_PRIORITY", "0")))
            self.ws2 = torch.zeros_like(self.ws)
        # q_x_in, q_y_in and the prior network are independent chains of small kernels (none fills the GPU):
        # q_y_in and the prior run on their own streams, each with its own reduction workspace
        all_units = self._flat([u for us in self.q_units for u in us] + list(self.p_units)
                               + [u for us in self.g_units for u in us] + list(self.mu_units) + list(self.var_units))
        for u in self._flat([u for us in self.g_units for u in us] + list(self.mu_units) + list(self.var_units)):
            u.pack_late = True              # (graph.PackBatch: nothing in front of run_generator reads their weights)
        self.pack_batch = PackBatch(self, [u for u in all_units if isinstance(u, ConvUnit)])
        # Under data parallelism the recognition branches and the prior network advance level by level and share ONE
        # all-reduce of batch-norm sums per level (forward and backward): 14 fewer latency-bound collectives per step.
        # Where the statistics travel over peer memory (dist.Sync.fused: exchanged inside the kernels that finalize them, or
        # by the one-kernel all-reduce -- both on whatever stream the layer runs on, ordered by the ring of slots and not by
        # a communicator) the branches keep their own streams instead, as on a single device.
        sync_bn = model.sync is not None and model.sync.sync_bn
        peer_stats = sync_bn and model.sync.fused(dev)
        self.levels = None
        if sync_bn and not peer_stats and self.q_units and self.p_units and os.environ.get("BP_LEVEL_SYNC", "1") != "0":
            self.levels = self._build_levels()
        self.branch = self._branch_streams = None
        # (not with global batch-norm statistics through a process group: every batch-norm layer then all-reduces its
        #  sums, and collectives of one communicator must not be in flight on several streams at once; with local
        #  statistics -- throughput mode -- the only collective is the gradient all-reduce after the joins)
        if os.environ.get("BP_BRANCH_STREAMS", "1") != "0" and self.q_units and self.p_units \
                and (not sync_bn or peer_stats):
            self.branch = self._branch_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            self.ws_b, self.ws_c = torch.zeros_like(self.ws), torch.zeros_like(self.ws)
            for u in self._flat(self.q_units[1]):
                u.ws_name = "ws_b"
            for u in self._flat(self.p_units):
                u.ws_name = "ws_c"

        # The weight gradients of the q_y_in / prior branches get streams of their own: nine small launches that would
        # otherwise queue behind each other on the one weight-gradient stream after the main stream has finished.  Also
        # under data parallelism (weight gradients issue no collectives), where the branches themselves share the
        # main stream.
        if self.side is not None and self.q_units and self.p_units and os.environ.get("BP_BRANCH_SIDE", "1") != "0":
            self.side_branch = {k: (torch.cuda.Stream(device=dev), torch.zeros_like(self.ws)) for k in ("b", "c")}
            for u in self._flat(self.q_units[1]):
                u.side_name = "b"
            for u in self._flat(self.p_units):
                u.side_name = "c"

    # ---- bf16 policy (dtype="bf16", BASELINE.json configs[3]).  The generator trunk p_y_z_in and the first two layers
    # of each head run on the bf16 matrix-core kernels; every p_y_z_in activation / gradient and the heads' 8-channel
    # slot (one 16-byte bf16 vector per pixel at 512^2: 0.27 GB instead of 0.54 GB per pass) are stored as bf16: 96 %
    # of the activation bytes of a step (SURVEY.md 8a totals).  The recognition / prior networks, p_z_in and the
    # 1-channel tails of the heads stay fp32.  Parameters, gradients of parameters, batch-norm statistics and every
    # loss reduction are fp32 / fp64 in both modes.  BP_BF16_HEAD_TAIL=0: the round-3 policy (8-channel slot in fp32,
    # second head layer on the vector ALUs).
    _HEAD_TAIL = os.environ.get("BP_BF16_HEAD_TAIL", "1") != "0"

    def bf16_unit(self, name):
        return self.bf16 and (name.startswith("p_y_z_in.") or name in ("p_mu_out.0", "p_var_out.0")
                              or (self._HEAD_TAIL and name in ("p_mu_out.2", "p_var_out.2")))

    def bf16_out(self, name):
        return self.bf16 and (name.startswith("p_y_z_in.") or (self._HEAD_TAIL and name in ("p_mu_out.0", "p_var_out.0")))

    # ---- helpers
    @staticmethod
    def _flat(units):
        out = []
        for u in units:
            out += u.body if hasattr(u, "body") else [u]
        return out

    def need_ws(self, nbytes):
        self.ws_bytes = max(self.ws_bytes, int(nbytes))

    side_branch = None

    def side_of(self, unit):
        """(stream, workspace) for a unit's weight gradient, or (None, None): serial schedule."""
        if self.side is None:
            return None, None
        if self.side_branch is not None and getattr(unit, "side_name", None) in self.side_branch:
            return self.side_branch[unit.side_name]
        return self.side, self.ws2

    def impl_of(self, kind, unit=None):
        """Kernel family per operation; BP_IMPL_FWD / BP_IMPL_DGRAD / BP_IMPL_WGRAD (auto|direct|mfma)
        override the model's choice for debugging (BP_IMPL_UNITS=a,b restricts the override to units)."""
        v = os.environ.get("BP_IMPL_" + kind.upper())
        only = os.environ.get("BP_IMPL_UNITS")
        if only and unit is not None and unit not in only.split(","):
            v = None
        return self.impl if v is None else {"auto": L.IMPL_AUTO, "direct": L.IMPL_DIRECT, "mfma": L.IMPL_MFMA}[v]

    # per-launch HIP-event timing of the convolution kernels (bench.py roofline); off by default
    prof = None

    def prof_begin(self):
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

    @staticmethod
    def _no_trailing(tr, where):
        if tr:
            raise NotImplementedError(f"{where}: trailing layers {tr} are not supported by the HIP path")

    @staticmethod
    def _latent_trailing(tr, where):
        for name, _ in tr:
            if name != "unflatten":
                raise NotImplementedError(f"{where}: trailing layer '{name}' is not supported by the HIP path")

    @staticmethod
    def _head_trailing(tr, where):
        sp = False
        for name, _ in tr:
            if name == "softplus":
                sp = True
            else:
                raise NotImplementedError(f"{where}: trailing layer '{name}' is not supported by the HIP path")
        return sp

    def _prepare_backward(self):
        # consumers claim gradient views in reverse topological order
        for us in (self.var_units, self.mu_units):
            for u in reversed(us):
                u.prepare_backward()
        for us in reversed(self.g_units):
            for u in reversed(us):
                u.prepare_backward()
        self.z.ensure_grad()
        for u in reversed(self.p_units):
            u.prepare_backward()
        for us in reversed(self.q_units):
            for u in reversed(us):
                u.prepare_backward()
        self.seed = torch.ones(1, device=self.device)

    # ---- execution
    def _build_levels(self):
        """[[units of one depth of q_x_in / q_y_in / prior], ..., [q_out units + what is left of the prior]] with the
        units' ``sums`` re-seated as views of one buffer per level."""
        ux, uy, uo = self.q_units
        pu = list(self.p_units)
        depth = max(len(ux), len(uy))
        levels = [[c[i] for c in (ux, uy, pu) if i < len(c)] for i in range(depth)]
        rest_p = pu[depth:]
        for i in range(max(len(uo), len(rest_p))):
            levels.append([c[i] for c in (uo, rest_p) if i < len(c)])
        out = []
        for lvl in levels:
            if not all(isinstance(u, ConvUnit) and u.act != "prelu" for u in lvl):
                return None                     # residual blocks / PReLU in a branch: keep per-layer collectives
            buf = torch.zeros(sum(u.sums.numel() for u in lvl), device=self.device, dtype=torch.float64)
            off = 0
            for u in lvl:
                n = u.sums.numel()
                u.sums = buf[off:off + n]
                off += n
            out.append((lvl, buf))
        return out

    def _run_levels(self, levels, start):
        """Drive the units of each level in lock step: first halves, one collective, second halves."""
        for lvl, buf in levels:
            gens = [start(u) for u in lvl]
            live = []
            for g in gens:
                try:
                    next(g)
                    live.append(g)
                except StopIteration:
                    pass
            if live:
                self.model.sync.all_reduce_sum(buf)      # every unit's sums (views of buf) at once
            for g in live:
                for _ in g:
                    raise RuntimeError("a unit asked for a second synchronisation point")

    def pack_all(self):
        """Re-pack every layer's weights (one launch) if any parameter changed since the last pack."""
        us = self.pack_batch.units
        if all(u._packed_version == (u.holder.weight._version, u.holder.weight.data_ptr(),
                                      getattr(self.model, "_param_epoch", 0)) for u in us):
            return
        self.pack_batch.run()

    def load_inputs(self, y, aux, x=None):
        self.pack_all()
        lib, st = self.lib, _stream()
        m = self.model
        cy = m.dim_y[0]
        y = y.contiguous()
        auxp = None
        if self.caux:
            aux = aux.reshape(self.n, self.caux).to(torch.float32).contiguous()
            auxp = L.ptr(aux)
        L.check(lib.bp_nchw_to_view(L.ptr(y), cy, auxp, self.caux, C.byref(self.y2.view), st), "merge_aux_label")

        def generator_copy(stream):
            for l in range(m.L):
                v = L.View(self.p_in.buf[l * self.n:].data_ptr(), self.n, self.hy_slot.h, self.hy_slot.w,
                           self.hy_slot.c, self.hy_slot.cstride, self.hy_slot.coff)
                L.check(lib.bp_nchw_to_view(L.ptr(y), cy, auxp, self.caux, C.byref(v), stream), "merge_aux_label (P)")
        # the generator's copy of y is not read before run_generator: in a training plan it goes to the weight-
        # gradient stream (idle during the forward pass), beside the recognition / prior networks
        self._gen_inputs = None
        if self.side is not None and not torch.cuda.is_current_stream_capturing():
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side):
                generator_copy(_stream())
                self._gen_inputs = torch.cuda.Event()
                self._gen_inputs.record(self.side)
        else:
            generator_copy(st)
        self._keep = (y, aux)
        if x is not None:
            if not self.with_q:
                raise RuntimeError("this plan was built without the recognition network")
            self.x_nchw.copy_(x)
            if self.x_nchw.data_ptr() != self.x_in.buf.data_ptr():
                L.check(lib.bp_nchw_to_view(L.ptr(self.x_nchw), m.dim_x[0], None, 0, C.byref(self.x_in.view), st),
                        "x layout")

    def run_prior(self, training):
        for u in self.p_units:
            u.forward(training)

    def run_latent(self, eps, use_q):
        lib, st = self.lib, _stream()
        self.eps = eps.to(torch.float32).contiguous()
        q = self.q_head if use_q else self.p_head
        if q is None:
            raise NotImplementedError("sampling from the standard-normal prior (no prior_z_y) is not implemented")
        p = self.p_head
        L.check(lib.bp_latent_forward(C.byref(self.lat), C.byref(q.view), q.pw_struct(),
                                      None if p is None else C.byref(p.view),
                                      None if p is None else p.pw_struct(), L.ptr(self.eps),
                                      L.ptr(self.stats4), C.byref(self.z.view), L.ptr(self.kl_sum),
                                      L.ptr(self.ws), self.ws_bytes, st), "latent forward")

    def run_generator(self, training):
        for i, us in enumerate(self.g_units):
            if i == 0 and getattr(self, "_gen_inputs", None) is not None:
                torch.cuda.current_stream(self.device).wait_event(self._gen_inputs)    # load_inputs: y for the generator
                self._gen_inputs = None
            if i == 0 and getattr(self, "_own_packed", None) is not None:
                torch.cuda.current_stream(self.device).wait_event(self._own_packed)    # graph.PackBatch: late packs
                self._own_packed = None
            for u in us:
                u.forward(training)
        for u in self.mu_units:
            u.forward(training)
        for u in self.var_units:
            u.forward(training)

    def mark(self, name):
        """Phase mark on the main stream (BP_PHASE_EVENTS=1; tools/phase_times.py): a timed event in ``self.marks``."""
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.marks.append((name, ev))

    def forward_train(self, x, y, aux, eps, training=True):
        m = self.model
        self.mark("step")
        self.load_inputs(y, aux, x)
        self.mark("inputs+packs")
        if self.levels is not None and training:
            self._run_levels(self.levels, lambda u: u.forward_steps(training))
        elif self.branch is None:
            for us in self.q_units:
                for u in us:
                    u.forward(training)
            self.run_prior(training)
        else:
            main = torch.cuda.current_stream(self.device)
            sy, sp = self.branch
            ux, uy, uo = self.q_units
            sy.wait_stream(main); sp.wait_stream(main)              # fork: the inputs are loaded
            with torch.cuda.stream(sy):
                for u in uy:
                    u.forward(training)
            with torch.cuda.stream(sp):
                self.run_prior(training)
            for u in ux:
                u.forward(training)
            main.wait_stream(sy)                                    # q_out reads [q_x_in | q_y_in]
            for u in uo:
                u.forward(training)
            main.wait_stream(sp)                                    # the KL term reads the prior
        self.mark("encoders fwd")
        self.run_latent(eps, use_q=True)
        self.mark("latent fwd")
        self.run_generator(training)
        self.mark("generator fwd")
        self.ll.alpha_var = float(m.alpha_var)
        self.ll.beta_kl = float(m.beta_KL)
        lib, st = self.lib, _stream()
        L.check(lib.bp_loglik_forward(C.byref(self.ll), L.ptr(self.x_nchw), C.byref(self.mu_head.view),
                                      None if self.var_head is None else C.byref(self.var_head.view),
                                      L.ptr(self.kl_sum), L.ptr(self.x_mu), L.ptr(self.x_log_var),
                                      L.ptr(self.stats), L.ptr(self.ws), self.ws_bytes, st), "log-likelihood")
        self.mark("loss fwd")

    def backward(self, seed, grads):
        """d(seed*ELBO)/d(parameters) into ``grads`` (id(param) -> tensor)."""
        if self.defer_reduce and not getattr(self, "skip_wgrad", False):
            L.check(self.lib.bp_wgrad_defer_begin(), "defer weight-gradient reductions")
            self.deferring = True
        try:
            self._backward(seed, grads)
        finally:
            if self.deferring:          # (an exception above: drop what was recorded, stop deferring)
                self.deferring = False
                self.lib.bp_wgrad_defer_flush(-1, None)

    def _join_branch_sides(self):
        if self.side is not None and self.side_branch is not None:
            for st, _ in self.side_branch.values():
                self.side.wait_stream(st)

    def _flush_reductions(self, end):
        """Launch the recorded split-K reductions behind the weight-gradient kernels (their stream)."""
        if end:
            self._join_branch_sides()
        if not self.deferring:
            return
        if self.side is not None:
            with torch.cuda.stream(self.side):
                L.check(self.lib.bp_wgrad_defer_flush(1 if end else 0, _stream()), "weight-gradient reductions")
        else:
            L.check(self.lib.bp_wgrad_defer_flush(1 if end else 0, _stream()), "weight-gradient reductions")
        if end:
            self.deferring = False

    def _backward(self, seed, grads):
        lib, st = self.lib, _stream()
        self.seed.copy_(seed.reshape(1))
        self.mu_head.ensure_grad()
        if self.var_head is not None:
            self.var_head.ensure_grad()
        L.check(lib.bp_loglik_backward(C.byref(self.ll), L.ptr(self.x_nchw), C.byref(self.mu_head.view),
                                       None if self.var_head is None else C.byref(self.var_head.view),
                                       L.ptr(self.seed), C.byref(self.mu_head.grad),
                                       None if self.var_head is None else C.byref(self.var_head.grad), st),
                "log-likelihood backward")
        self.mark("loss bwd")
        for us in (self.var_units, self.mu_units):
            for u in reversed(us):
                u.backward(grads)
        self.mark("heads bwd")
        for u in reversed(self.g_units[1]):              # generator trunk p_y_z_in
            u.backward(grads)
        self.mark("trunk bwd")
        early = self._reduce_trunk_gradients()
        fine = self.marks is not None and os.environ.get("BP_PHASE_EVENTS_FINE") == "1"     # (tools/phase_times.py)
        for u in reversed(self.g_units[0]):              # p_z_in
            u.backward(grads)
            if fine:
                self.mark(f"  bwd {u.name}")
        L.check(lib.bp_latent_backward(C.byref(self.lat), C.byref(self.z.grad), L.ptr(self.stats4),
                                       L.ptr(self.eps), L.ptr(self.seed), float(self.model.beta_KL),
                                       C.byref(self.q_head.grad),
                                       None if self.p_head is None else C.byref(self.p_head.grad), st),
                "latent backward")
        if self.levels is not None:
            self._run_levels(list(reversed(self.levels)), lambda u: u.backward_steps(grads))
        elif self.branch is None:
            for u in reversed(self.p_units):
                u.backward(grads)
            for us in reversed(self.q_units):
                for u in reversed(us):
                    u.backward(grads)
        else:
            main = torch.cuda.current_stream(self.device)
            sy, sp = self.branch
            ux, uy, uo = self.q_units
            if fine:
                self.mark("  latent bwd")
            sp.wait_stream(main)                                    # fork: d(loss)/d(prior head) is written
            with torch.cuda.stream(sp):
                for u in reversed(self.p_units):
                    u.backward(grads)
            for u in reversed(uo):
                u.backward(grads)
                if fine:
                    self.mark(f"  bwd {u.name}")
            sy.wait_stream(main)                                    # fork: d/d[q_x_in | q_y_in] is written
            with torch.cuda.stream(sy):
                for u in reversed(uy):
                    u.backward(grads)
            for u in reversed(ux):
                u.backward(grads)
                if fine:
                    self.mark(f"  bwd {u.name}")
            main.wait_stream(sy)
            if fine:
                self.mark("  join q_y_in")
            main.wait_stream(sp)
        self.mark("p_z_in + latent + encoders bwd")
        self._flush_reductions(end=True)
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)    # join: every weight gradient is written
        self.mark("join weight gradients")
        sync, flat = self.model.sync, self.model._flat_grads
        if sync is not None:
            if early is None:
                sync.all_reduce_mean(flat)
            else:                                        # what the early all-reduce did not cover: 3 % of the buffer
                for a, b in ((0, early[0]), (early[1], flat.numel())):
                    if b > a:
                        sync.all_reduce_mean(flat[a:b])

    def _reduce_trunk_gradients(self):
        """Data parallel: average the gradients of the generator trunk and heads over ranks NOW, on the weight-
        gradient stream (right behind the trunk's last weight gradient), while the main stream still runs the
        backward pass of p_z_in, the latent heads, the prior and the recognition networks.  Uses the gradient
        communicator of ``dist.Sync`` (the batch-norm statistics keep flowing on the main stream's own).
        Returns the slice it covered, or None."""
        sync, sl = self.model.sync, self.model._early_slice
        if sync is None or not getattr(sync, "active", sync.world_size > 1) or not sync.overlap or sl is None or self.side is None \
                or os.environ.get("BP_EARLY_ALLREDUCE", "0") != "1":
            return None
        main = torch.cuda.current_stream(self.device)
        self._flush_reductions(end=False)    # the trunk's weight gradients must be complete
        self.side.wait_stream(main)          # batch-norm / PReLU parameter gradients are written on the main stream
        with torch.cuda.stream(self.side):
            sync.all_reduce_mean(self.model._flat_grads[sl[0]:sl[1]])
        return sl


DEFER_MAX_BYTES = int(os.environ.get("BP_DEFER_MAX_MB", "1048576")) << 20


class _ELBOFunction(torch.autograd.Function):
    """The whole network as one autograd node: forward launches the plan, backward launches
    the hand-scheduled gradient plan and hands the parameter gradients back to autograd."""

    @staticmethod
    def forward(ctx, model, plan, x, y, aux, eps, *params):
        plan.forward_train(x, y, aux, eps, training=model.training)
        ctx.model, ctx.plan, ctx.was_training = model, plan, model.training
        return plan.stats[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        model, plan = ctx.model, ctx.plan
        if not plan.with_grad:
            raise RuntimeError("this forward was run without gradient buffers")
        if not ctx.was_training:
            raise RuntimeError("backward through an eval-mode forward (batch-norm on running statistics) is not "
                               "implemented; the reference never does it (painter.py:85,372)")
        plan.backward(grad_out.detach().to(torch.float32), model._grad_views_by_id)     # (reduces over ranks too)
        # Parameter gradients live in ONE flat buffer; after optimizer.zero_grad() (grad = None) the
        # views are attached directly instead of letting autograd clone 92 tensors per step.
        out = []
        for p, gv in zip(model._params, model._grad_views):
            if p.grad is None:
                p.grad = gv
                out.append(None)
            elif p.grad.data_ptr() == gv.data_ptr():
                out.append(None)          # already attached: the plan overwrote it in place
            else:
                out.append(gv)            # a foreign .grad tensor: let autograd accumulate
        return (None, None, None, None, None, None) + tuple(out)


class CVAE(torch.nn.Module):
    """Drop-in for ``baryon_painter.models.cvae.CVAE`` (cvae.py:8-61)."""

    def __init__(self, architecture, device="cuda:0", impl="auto", sync=None, dtype="f32"):
        """``dtype``: "f32" = the reference's arithmetic (parity mode); "bf16" = bf16 activations / gradients with
        fp32 accumulation, fp32 master weights and fp32/fp64 statistics in the generator trunk (throughput mode,
        BASELINE.json configs[3]; see ``_Plan.bf16_unit``)."""
        super().__init__()
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("baryon_painter_amd.CVAE runs on an AMD GPU only (device='cuda:N'); "
                               "there is no CPU implementation of the hot path.")
        self._lib = L.load()
        self.impl = {"auto": L.IMPL_AUTO, "direct": L.IMPL_DIRECT, "mfma": L.IMPL_MFMA}[impl]
        self.sync = sync

        print("CVAE with {} architecture.".format(architecture["type"]))
        self.architecture = architecture
        self.dim_x = tuple(architecture["dim_x"])
        self.dim_y = tuple(architecture["dim_y"])
        self.dim_z = tuple(architecture["dim_z"])
        self.L = architecture["L"] if "L" in architecture else 1
        self.n_x_features = architecture["n_x_features"]

        if architecture["type"] == "Type-1":
            self.q_x_in = build_holders(architecture["q_x_in"])
            self.q_y_in = build_holders(architecture["q_y_in"])
            self.q_out = build_holders(architecture["q_x_y_out"])
            self.p_y_in = build_holders(architecture["p_y_in"])
            self.p_z_in = build_holders(architecture["p_z_in"])
            self.p_y_z_in = build_holders(architecture["p_y_z_in"])
            self.p_mu_out = build_holders(architecture["p_y_z_out"][0])
            if len(architecture["p_y_z_out"]) > 1:
                self.predict_var = True
                self.p_var_out = build_holders(architecture["p_y_z_out"][1])
                self.x_var_init_std = architecture.get("x_var_init_std", 0.01)

                def init_weight(m):        # cvae.py:37-40: every module with a .weight
                    if hasattr(m, "weight") and isinstance(m.weight, torch.Tensor):
                        torch.nn.init.normal_(m.weight, std=self.x_var_init_std)
                self.p_var_out.apply(init_weight)
                self.min_x_var = architecture.get("min_x_var", 1e-7)
            else:
                self.predict_var = False
                self.p_var_out = None
            self.use_aux_label = architecture["aux_label"]
            if "prior_z_y" in architecture:
                self.prior_network = build_holders(architecture["prior_z_y"])
            else:
                self.prior_network = None
        else:
            raise NotImplementedError("Architecture {} not supported yet!".format(architecture["type"]))

        self.min_z_var = architecture.get("min_z_var", 1e-7)
        self.likelihood_scaling = architecture.get("likelihood_scaling", 1.0)
        self.alpha_var = 1.0
        self.beta_KL = 1.0
        self.n_aux = 1
        self._plans = {}
        self._graphs = {}
        self._eps_override = None
        self.to(self.device)
        self._flatten_parameters()

    # ---- parameter storage: one flat buffer (single all-reduce / fused Adam), torch views on it
    def _flatten_parameters(self):
        params = list(self.parameters())
        n = sum(p.numel() for p in params)
        self._flat_params = torch.zeros(n, device=self.device)
        self._flat_grads = torch.zeros(n, device=self.device)
        off = 0
        self._grad_views, self._grad_views_by_id = [], {}
        # slice of the flat buffers that holds the generator trunk and its heads (97 % of the parameters): their
        # gradients are complete long before the backward pass ends (dist.Sync: early all-reduce on the side stream)
        names = {id(p): n for n, p in self.named_parameters()}
        early = [i for i, p in enumerate(params) if names[id(p)].startswith(("p_y_z_in.", "p_mu_out.", "p_var_out."))]
        self._early_slice = None
        if early and early == list(range(early[0], early[-1] + 1)):
            e0 = sum(p.numel() for p in params[:early[0]])
            self._early_slice = (e0, e0 + sum(params[i].numel() for i in early))
        for p in params:
            k = p.numel()
            self._flat_params[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self._flat_params[off:off + k].view_as(p)
            gv = self._flat_grads[off:off + k].view_as(p)
            self._grad_views.append(gv)
            self._grad_views_by_id[id(p)] = gv
            off += k
        self._params = params

    def overlap_weight_gradients(self, enabled):
        """Run the weight gradients (and the independent q_y_in / prior branches) on their own streams beside the
        main chain (default), or everything serially on the main stream (``False``: every kernel has the GPU to
        itself, e.g. to time kernels)."""
        for plan in self._plans.values():
            plan.side = plan._side_stream if enabled else None
            plan.branch = plan._branch_streams if enabled else None

    def _bump_param_versions(self):
        """Called after an out-of-band in-place update of the flat buffer (FlatAdam): convolution units
        re-pack their weights when this counter changes."""
        self._param_epoch = getattr(self, "_param_epoch", 0) + 1

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if hasattr(self, "_flat_params"):
            self._flatten_parameters()
            self._plans = {}
            self._graphs = {}
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        return out

    def _plan(self, n, with_grad, with_q=True):
        """Plans are cached per (batch, gradients, recognition net); a richer plan serves a
        poorer request."""
        want_g, want_q = bool(with_grad), bool(with_q)
        for (pn, g, q), plan in self._plans.items():
            if pn == n and g >= want_g and q >= want_q:
                return plan
        plan = _Plan(self, n, want_g, want_q)
        self._plans[(n, want_g, want_q)] = plan
        return plan

    def cuda(self, device=None):
        return self

    # ---- reference API --------------------------------------------------------------------
    def _draw_eps(self, n):
        if self._eps_override is not None:
            e = torch.as_tensor(self._eps_override, device=self.device, dtype=torch.float32)
            if tuple(e.shape) != (self.L, n, *self.dim_z):
                raise ValueError(f"eps override has shape {tuple(e.shape)}, expected {(self.L, n, *self.dim_z)}")
            return e
        return torch.randn(size=(self.L, n, *self.dim_z), device=self.device)     # cvae.py:64

    def _check_inputs(self, x, y):
        if y.dim() != 4 or tuple(y.shape[1:]) != self.dim_y:
            raise ValueError(f"y has shape {tuple(y.shape)}, model expects (N, {self.dim_y})")
        if x is not None and (x.dim() != 4 or tuple(x.shape[1:]) != self.dim_x or x.shape[0] != y.shape[0]):
            raise ValueError(f"x has shape {tuple(x.shape)}, model expects (N, {self.dim_x})")

    def _aux(self, aux_label, n):
        if not self.use_aux_label:
            return None
        if aux_label is None:
            raise ValueError("this architecture needs an aux_label (redshift)")
        a = torch.as_tensor(aux_label, device=self.device, dtype=torch.float32)
        if a.dim() <= 1:
            a = a.reshape(-1, 1)
        if a.shape[0] != n:
            raise ValueError("aux_label batch size needs to match that of y")        # utils.py:175-176
        if a.shape[1] != self.n_aux:
            raise NotImplementedError("one scalar aux label per sample")
        return a

    def forward(self, x, y, aux_label=None):
        """ELBO of the batch (cvae.py:122-147); supports ``.backward()``."""
        x = torch.as_tensor(x, device=self.device, dtype=torch.float32)
        y = torch.as_tensor(y, device=self.device, dtype=torch.float32)
        self._check_inputs(x, y)
        n = y.shape[0]
        aux = self._aux(aux_label, n)
        need_grad = torch.is_grad_enabled()
        plan = self._plan(n, need_grad)
        eps = self._draw_eps(n)
        elbo = _ELBOFunction.apply(self, plan, x, y, aux, eps, *self._params)
        self._last = plan
        cx = self.dim_x[0]
        s = plan.stats
        self.ELBO = elbo
        self.KL_term = s[1]
        self.log_likelihood = s[2:2 + cx]
        if self.predict_var:
            self.log_likelihood_fixed_var = s[2 + cx:2 + 2 * cx]
            self.log_likelihood_free_var = s[2 + 2 * cx:2 + 3 * cx]
            self.x_var = torch.exp(plan.x_log_var)
        self.x_mu = plan.x_mu
        self.z_mu = plan.stats4[0]
        self.z_log_var = plan.stats4[1]
        return elbo

    def sample_P(self, y, return_var=False, aux_label=None, z=None):
        """cvae.py:149-162 (always under no_grad)."""
        with torch.no_grad():
            y = torch.as_tensor(y, device=self.device, dtype=torch.float32)
            self._check_inputs(None, y)
            n = y.shape[0]
            aux = self._aux(aux_label, n)
            if self.L != 1:
                raise NotImplementedError("sample_P with L != 1")
            plan = self._plan(n, False, False)
            plan.load_inputs(y, aux)
            if z is None:
                plan.run_prior(self.training)
                plan.run_latent(self._draw_eps(n), use_q=False)
            else:
                zt = torch.as_tensor(z, device=self.device, dtype=torch.float32)
                if tuple(zt.shape) != (n, *self.dim_z):
                    raise ValueError(f"z has shape {tuple(zt.shape)}, expected {(n, *self.dim_z)}")
                lib, st = self._lib, _stream()
                zt = zt.contiguous()
                L.check(lib.bp_nchw_to_view(L.ptr(zt), self.dim_z[0], None, 0, C.byref(plan.z.view), st), "z layout")
            plan.run_generator(self.training)
            lib, st = self._lib, _stream()
            cx, H, W = self.dim_x
            mu = torch.empty((n, cx, H, W), device=self.device)
            self._head_to_nchw(plan.mu_head, plan.mu_softplus, mu)
            if self.predict_var and return_var:
                lv = torch.empty((n, cx, H, W), device=self.device)
                self._head_to_nchw(plan.var_head, False, lv)
                return mu, torch.exp(lv)
            return mu

    # ---- hipGraph-captured eval forward (BASELINE.json configs[4]: stream tiles through P)
    def sample_P_graphed(self, y, aux_label=None, z=None):
        """``sample_P`` (prior-sampled z, mean head) replayed from a captured hipGraph: one graph
        launch instead of ~110 kernel launches per batch.  Eval mode only; the graph is captured per
        batch size on first use and re-used while the parameters' storage is unchanged.  ``z`` (N, *dim_z), if
        given, replaces the prior sample as in ``sample_P(..., z=z)`` (cvae.py:149-154): a second captured graph
        without the prior network and the sampler."""
        if self.training:
            raise RuntimeError("sample_P_graphed is an eval-mode (paint) path: call model.train(False) first")
        y = torch.as_tensor(y, device=self.device, dtype=torch.float32)
        self._check_inputs(None, y)
        n = y.shape[0]
        aux = self._aux(aux_label, n)
        key = n if z is None else (n, "z")
        g = self._graphs.get(key)
        if g is None:
            g = self._capture_paint_graph(n, given_z=z is not None)
            self._graphs[key] = g
        for u in g["units"]:
            u.maybe_pack()                       # eager, a no-op unless the weights changed
            u.maybe_bn_eval()                    # ... or the running statistics
        g["y"].copy_(y)
        if aux is not None:
            g["aux"].copy_(aux)
        if z is not None:
            zt = torch.as_tensor(z, device=self.device, dtype=torch.float32)
            if tuple(zt.shape) != (n, *self.dim_z):
                raise ValueError(f"z has shape {tuple(zt.shape)}, expected {(n, *self.dim_z)}")
            g["z"].copy_(zt)
        elif self._eps_override is not None:
            raise RuntimeError("eps override is not supported on the graphed path (noise is drawn in-graph)")
        g["graph"].replay()
        return g["out"].clone()

    def paint_graph(self, n):
        """The captured paint pipeline for batches of ``n`` RAW tiles (configs[4]).  Returns a dict with ``slots``: TWO
        input / parameter / output buffer sets, each with its own captured graph over the SAME launch plans, so that a
        caller uploads batch b+1 straight into one slot and downloads batch b-1 straight out of it while the other
        slot's graph paints batch b -- no device-to-device staging copy.  A slot holds
          ``raw`` (n, cy, H, W) untransformed input tiles, ``out`` (n, cx, H, W) painted physical tiles,
          ``block`` one uint8 device buffer with the typed views ``xf_in`` / ``xf_out`` (n, 2) float64 {sigma, k} /
          {k, sigma} of the shift-log transform and its inverse, ``tile_ids`` (n,) int64 global tile numbers, ``seed``
          (1,) int64 Philox key (read by the kernel at run time: one graph serves every seed), ``aux`` (n, 1) float32
          redshifts (``block_layout``: name -> (byte offset, dtype, shape) for a pinned host mirror),
          ``graph``; ``graph.replay()`` leaves the painted tiles of ``raw`` in ``out``.
        The two graphs share activations: replay them on ONE stream."""
        if self.training:
            raise RuntimeError("paint_graph is an eval-mode (paint) path: call model.train(False) first")
        key = (n, "pipeline")
        g = self._graphs.get(key)
        if g is None:
            g = self._capture_paint_graph(n, pipeline=True)
            self._graphs[key] = g
        for u in g["units"]:
            u.maybe_pack()
            u.maybe_bn_eval()
        return g

    def _capture_paint_graph(self, n, given_z=False, pipeline=False):
        """Eval-mode layers do not couple the tiles of a batch (batch-norm runs on its running statistics), so the
        batch is painted as BP_PAINT_STREAMS (default 4) sub-batches on as many streams inside one graph: kernels of different layers share
        the CUs and fill each other's stalls (the effect the training step gets from its weight-gradient
        stream)."""
        cy, H, W = self.dim_y
        cx = self.dim_x[0]
        st = {"y": None if pipeline else torch.zeros((n, cy, H, W), device=self.device),
              "aux": torch.zeros((n, self.n_aux), device=self.device) if self.use_aux_label and not pipeline else None,
              "out": None if pipeline else torch.zeros((n, cx, H, W), device=self.device),
              "z": torch.zeros((n, *self.dim_z), device=self.device) if given_z else None}
        if pipeline:
            if self.L != 1 or self.prior_network is None:
                raise NotImplementedError("the paint pipeline needs L = 1 and a prior network")
            per_tile = self.dim_z[0] * self.dim_z[1] * self.dim_z[2]
            st["eps"] = torch.zeros((1, n, per_tile), device=self.device)
            # parameter block of a slot: one contiguous device buffer = one host-to-device copy per batch
            layout, off = {}, 0
            for name, dt, shape in (("xf_in", torch.float64, (n, 2)), ("xf_out", torch.float64, (n, 2)),
                                    ("tile_ids", torch.int64, (n,)), ("seed", torch.int64, (1,)),
                                    ("aux", torch.float32, (n, max(self.n_aux, 1)))):
                layout[name] = (off, dt, shape)
                nb = int(torch.tensor([], dtype=dt).element_size()) * int(torch.Size(shape).numel())
                off += (nb + 7) // 8 * 8
            st["block_layout"], st["block_bytes"] = layout, off

            def new_slot():
                sl = {"raw": torch.zeros((n, cy, H, W), device=self.device),
                      "out": torch.zeros((n, cx, H, W), device=self.device),
                      "block": torch.zeros(off, device=self.device, dtype=torch.uint8)}
                for name, (o, dt, shape) in layout.items():
                    nb = int(torch.tensor([], dtype=dt).element_size()) * int(torch.Size(shape).numel())
                    sl[name] = sl["block"][o:o + nb].view(dt).view(shape)
                sl["xf_in"].fill_(1.0)
                sl["xf_out"].fill_(1.0)
                return sl
            st["slots"] = [new_slot(), new_slot()]
        parts = int(os.environ.get("BP_PAINT_STREAMS", "4"))
        while parts > 1 and (n % parts != 0 or n // parts < 8):
            parts -= 1
        h = n // parts
        plans = [self._plan(h, False, False)] + [_Plan(self, h, False, False) for _ in range(parts - 1)]
        st["plans"] = plans
        units = []
        for plan in plans:
            for us in [plan.p_units] + plan.g_units + [plan.mu_units]:
                for u in us:
                    units += u.body if hasattr(u, "body") else [u]
        st["units"] = units
        others = [torch.cuda.Stream(device=self.device) for _ in range(parts - 1)]

        def paint_pipeline(plan, lo, sl):
            lib, sm = self._lib, _stream()
            plan.pack_all()
            auxp = L.ptr(sl["aux"][lo:lo + h]) if self.use_aux_label else None
            L.check(lib.bp_paint_load2(L.ptr(sl["raw"][lo:lo + h]), cy, L.ptr(sl["xf_in"][lo:lo + h]), auxp, plan.caux,
                                       C.byref(plan.y2.view), C.byref(plan.hy_slot.view), sm), "paint load")
            plan.run_prior(False)
            eps = st["eps"][:, lo:lo + h]
            L.check(lib.bp_philox_normal_dev(L.ptr(sl["seed"]), L.ptr(sl["tile_ids"][lo:lo + h]), h, 1, eps.shape[-1],
                                             L.ptr(eps), sm), "philox")
            plan.run_latent(eps.reshape(1, h, *self.dim_z), use_q=False)
            plan.run_generator(False)
            L.check(lib.bp_paint_store(C.byref(plan.mu_head.view), None, 1 if plan.mu_softplus else 0,
                                       L.ptr(sl["xf_out"][lo:lo + h]), L.ptr(sl["out"][lo:lo + h]), sm), "paint store")

        def paint(plan, lo, sl):
            if pipeline:
                return paint_pipeline(plan, lo, sl)
            plan.load_inputs(st["y"][lo:lo + h], None if st["aux"] is None else st["aux"][lo:lo + h])
            if given_z:
                L.check(self._lib.bp_nchw_to_view(L.ptr(st["z"][lo:lo + h]), self.dim_z[0], None, 0,
                                                  C.byref(plan.z.view), _stream()), "z layout")
            else:
                plan.run_prior(False)
                plan.run_latent(torch.randn(size=(self.L, h, *self.dim_z), device=self.device), use_q=False)
            plan.run_generator(False)
            self._head_to_nchw(plan.mu_head, plan.mu_softplus, st["out"][lo:lo + h])

        def run(sl=None):
            main = torch.cuda.current_stream(self.device)
            for k, s2 in enumerate(others):
                s2.wait_stream(main)
                with torch.cuda.stream(s2):
                    paint(plans[k + 1], (k + 1) * h, sl)
            paint(plans[0], 0, sl)
            for s2 in others:
                main.wait_stream(s2)

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        slots = st.get("slots", [None])
        with torch.cuda.stream(side), torch.no_grad():
            run(slots[0])                         # warm-up outside capture (packs weights, sizes workspaces)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graphs = []
        for sl in slots:
            graph = torch.cuda.CUDAGraph()
            # (thread-local capture: under data parallelism the process group's watchdog thread may query events
            #  while this thread captures; that is harmless and must not invalidate the capture)
            with torch.no_grad(), torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                run(sl)
            graphs.append(graph)
            if sl is not None:
                sl["graph"] = graph
        st["graph"] = graphs[0]
        return st

    # ---- the whole training step as one hipGraph (small minibatches are launch-bound: ~450 launches per step)
    def make_graphed_train_step(self, optimizer, n):
        """Capture forward + backward + Adam for minibatches of ``n`` tiles and return ``step(x, y, aux) -> ELBO``.

        ``step`` does what ``optimizer.zero_grad(); elbo = model(x, y, aux); (-elbo).backward(); optimizer.step()``
        does (the loop body of the reference's ``CVAEPainter.train``, painter.py:93-130) with identical arithmetic -
        the captured launches ARE the eager ones - but replays them from a hipGraph, so the host cost of a step is
        one graph launch plus the input copies.  ``optimizer`` must be a ``FlatAdam`` on this model; its learning
        rate may change between steps (schedulers), the latent noise is drawn in-graph."""
        from ..optim import FlatAdam
        if not isinstance(optimizer, FlatAdam) or optimizer.model is not self:
            raise TypeError("make_graphed_train_step needs the FlatAdam optimiser of this model")
        if self.sync is not None:
            raise NotImplementedError("graphed training steps under data parallelism (collectives inside the graph)")
        if not self.training:
            raise RuntimeError("graphed training step: model.train() first")
        plan = self._plan(n, True)
        dev = self.device
        cy, H, W = self.dim_y
        cx = self.dim_x[0]
        xs = torch.zeros((n, cx, H, W), device=dev)
        ys = torch.zeros((n, cy, H, W), device=dev)
        auxs = torch.zeros((n, self.n_aux), device=dev) if self.use_aux_label else None
        seed = torch.full((1,), -1.0, device=dev)            # d(loss)/d(ELBO), loss = -ELBO
        units = []
        for us in plan.q_units + [plan.p_units] + plan.g_units + [plan.mu_units, plan.var_units]:
            for u in us:
                units += u.body if hasattr(u, "body") else [u]

        holder = {}

        def run():
            for u in units:
                u._packed_version = None                     # the weights change every replay: always re-pack
            eps = torch.randn(size=(self.L, n, *self.dim_z), device=dev)
            holder["eps"] = eps
            plan.forward_train(xs, ys, auxs, eps, training=True)
            plan.backward(seed, self._grad_views_by_id)
            optimizer.device_step()

        # warm-up outside capture on a side stream (sizes workspaces, claims gradient buffers); it must not change
        # the training state, so parameters, Adam moments and batch-norm buffers are restored afterwards
        keep = [t.clone() for t in (self._flat_params, optimizer.exp_avg, optimizer.exp_avg_sq)]
        bufs = [(b, b.clone()) for b in self.buffers()]
        optimizer.upload_hyper(max(optimizer.n_steps, 1))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            run()
        with torch.no_grad():
            for t, k in zip((self._flat_params, optimizer.exp_avg, optimizer.exp_avg_sq), keep):
                t.copy_(k)
            for b, k in bufs:
                b.copy_(k)
        self._bump_param_versions()

        def step(x, y, aux_label=None):
            x = torch.as_tensor(x, device=dev, dtype=torch.float32)
            y = torch.as_tensor(y, device=dev, dtype=torch.float32)
            self._check_inputs(x, y)
            if y.shape[0] != n:
                raise ValueError(f"this graph was captured for minibatches of {n} tiles, got {y.shape[0]}")
            if self._eps_override is not None:
                raise RuntimeError("eps override is not supported on the graphed path (noise is drawn in-graph)")
            xs.copy_(x, non_blocking=True)
            ys.copy_(y, non_blocking=True)
            if auxs is not None:
                auxs.copy_(self._aux(aux_label, n), non_blocking=True)
            optimizer.n_steps += 1
            optimizer.upload_hyper(optimizer.n_steps)
            graph.replay()
            self._bump_param_versions()                      # eager paths (validation, paint) re-pack the new weights
            for p, gv in zip(self._params, self._grad_views):
                p.grad = gv
            self._last = plan
            s = plan.stats
            self.ELBO = s[0]
            self.KL_term = s[1]
            self.log_likelihood = s[2:2 + cx]
            self.x_mu = plan.x_mu
            self.z_mu = plan.stats4[0]
            self.z_log_var = plan.stats4[1]
            return s[0].clone()

        step.graph = graph
        step.last_eps = lambda: holder["eps"]                # the noise of the latest replay (tests)
        # every device tensor the captured launches read must outlive the capture: the graph holds raw pointers
        step._keepalive = (seed, xs, ys, auxs, holder, plan)
        return step

    def _head_to_nchw(self, slot, softplus, dst):
        L.check(self._lib.bp_view_to_nchw(C.byref(slot.view), None, 1 if softplus else 0, L.ptr(dst), _stream()),
                "head layout")

    def sample_prior(self, y, aux_label=None):
        raise NotImplementedError("use sample_P; the latent sample lives on the device plan")

    def get_stats(self):
        """cvae.py:164-171: same tuple order; forces a device->host sync like the reference's .item()."""
        s = self._last.stats.detach().cpu().numpy()
        cx = self.dim_x[0]
        if self.predict_var:
            return (float(s[0]), -float(s[1]), *s[2:2 + cx], *s[2 + cx:2 + 2 * cx], *s[2 + 2 * cx:2 + 3 * cx])
        return (float(s[0]), -float(s[1]), *s[2:2 + cx])

    def get_stats_labels(self):
        if self.predict_var:
            return (["ELBO", "KL_term"]
                    + ["log_likelihood_{}".format(i) for i in range(self.n_x_features)]
                    + ["log_likelihood_fixed_var_{}".format(i) for i in range(self.n_x_features)]
                    + ["log_likelihood_free_var_{}".format(i) for i in range(self.n_x_features)])
        return ["ELBO", "KL_term"] + ["log_likelihood_{}".format(i) for i in range(self.n_x_features)]

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def print_model_statistics(self, percentile=0.9):
        params = sorted([(p.numel(), name) for name, p in self.named_parameters() if p.requires_grad], reverse=True)
        total = sum(n for n, _ in params)
        print("Total number of parameters: {}".format(total))
        print("Top {}% of all parameters are in the following layers".format(percentile * 100))
        run = 0
        for n, name in params:
            run += n
            if run < total * percentile:
                print("{:<40s}   {:>8}".format(name, n))

    def check_gpu(self):
        for name, p in self.named_parameters():
            if "cuda" not in str(p.data.device):
                print("{} is not on the GPU!".format(name))
