"""Architecture description -> HIP launch plan.

The reference turns an architecture list into ``torch.nn`` modules with
``build_sequential`` (/root/reference/baryon_painter/models/utils.py:114-157) and lets
autograd derive the backward pass.  Here the same lists are compiled into *units*
(convolution [+ batch-norm] [+ activation], residual block) that own pre-allocated NHWC
device buffers and call the C ABI in ``include/bp_hip.h`` for forward and a hand-scheduled
backward.  PyTorch supplies device memory, streams and the parameter containers only.

Conventions
  * a ``Slot`` is an activation tensor: a RAW buffer view plus the pending per-channel
    affine+leaky-ReLU (``PW``) that its consumers apply while loading it;
  * every slot has at most two gradient contributions (``grad`` and ``grad2``), which the
    producer's backward adds while it applies the activation derivative.
"""
import ctypes as C
import os

import torch

from .. import _lib as L


# BP_EPILOGUE_STATS=0: batch-norm sums by separate streaming passes (bp_channel_sums / bp_act_backward) everywhere;
# "fwd" / "bwd": only the forward statistics / only the backward sums from the convolution epilogues
_ES = os.environ.get("BP_EPILOGUE_STATS", "auto")
EPILOGUE_STATS = _ES != "0"
EPILOGUE_FWD, EPILOGUE_BWD = _ES in ("1", "fwd", "auto"), _ES in ("1", "bwd", "auto")
FUSE_BN_BWD_FINALIZE = os.environ.get("BP_FUSE_BN_BWD_FINALIZE", "1") != "0"       # bp_act_backward_bn
# "auto": the backward sums only from the flattened-K data-gradient kernels (conv_flat.hip: weights in registers, sums
# per lane in LDS).  In the tiled igemm kernels the double-precision sums cost what the saved pass costs ("1": all).
_BWD_KERNEL_IDS = (710000,) if _ES == "auto" else None
FUSED_FINALIZE = os.environ.get("BP_FUSED_FINALIZE", "1") != "0"


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------ parameter containers
class _NoForward:
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("parameter container only: the arithmetic runs in libbp_hip.so")


class ParamConv2d(_NoForward, torch.nn.Conv2d):
    """Holds ``weight``/``bias`` with torch's own initialisation (same seed -> same initial
    weights as the reference's nn.Conv2d, utils.py:129).  Never called."""


class ParamConvTranspose2d(_NoForward, torch.nn.ConvTranspose2d):
    pass


class ParamBatchNorm2d(_NoForward, torch.nn.BatchNorm2d):
    pass


class ParamPReLU(_NoForward, torch.nn.PReLU):
    pass


class ParamLinear(_NoForward, torch.nn.Linear):
    pass


class SNConv2d(_NoForward, torch.nn.Conv2d):
    """Conv2d under spectral normalisation (``torch.nn.utils.spectral_norm`` semantics: parameter
    ``weight_orig``, buffers ``weight_u`` / ``weight_v``, one power iteration per training forward,
    ``weight = weight_orig / sigma``).  The power iteration and the chain rule through sigma are a few
    mat-vecs on the weight matrix -- parameter-side glue, done with torch ops; the convolution itself
    reads the normalised ``weight`` through the HIP kernels."""

    def __init__(self, **config):
        super().__init__(**config)
        w = self.weight.data
        del self.weight
        self.weight_orig = torch.nn.Parameter(w)
        self.register_buffer("weight", w.clone())
        co = w.shape[0]
        self.register_buffer("weight_u", torch.nn.functional.normalize(torch.randn(co), dim=0, eps=1e-12))
        self.register_buffer("weight_v", torch.nn.functional.normalize(torch.randn(w[0].numel()), dim=0, eps=1e-12))
        self.register_buffer("weight_grad", torch.zeros_like(w), persistent=False)
        self.sigma = None

    @torch.no_grad()
    def refresh(self, training):
        """One power iteration (training) and ``weight = weight_orig / sigma``.  The three mat-vecs are written as
        broadcast-multiply + reduce (element-wise ATen kernels): no BLAS library is on any path of this package."""
        wm = self.weight_orig.reshape(self.weight_orig.shape[0], -1)
        if training:
            self.weight_v.copy_(torch.nn.functional.normalize((wm * self.weight_u.unsqueeze(1)).sum(dim=0), dim=0, eps=1e-12))
            self.weight_u.copy_(torch.nn.functional.normalize((wm * self.weight_v.unsqueeze(0)).sum(dim=1), dim=0, eps=1e-12))
        self.sigma = (self.weight_u * (wm * self.weight_v.unsqueeze(0)).sum(dim=1)).sum()
        self.weight.copy_(self.weight_orig / self.sigma)

    @torch.no_grad()
    def finish_backward(self):
        """d/d weight_orig from d/d weight (u, v constant):  (G - <G, W> u v^T) / sigma."""
        g = self.weight_grad
        inner = (g * self.weight).sum()
        uv = torch.outer(self.weight_u, self.weight_v).view_as(g)
        r = (g - inner * uv) / self.sigma
        gv = getattr(self, "_grad_view", None)       # the owner's persistent view of a flat gradient buffer (CGAN)
        if gv is not None:
            gv.copy_(r)                              # ALWAYS into the flat buffer: data-parallel averaging reduces that
            if self.weight_orig.grad is None or self.weight_orig.grad.data_ptr() != gv.data_ptr():
                self.weight_orig.grad = gv           # (zero_grad(set_to_none=True) detached it)
        elif self.weight_orig.grad is None:
            self.weight_orig.grad = r
        else:
            self.weight_orig.grad.copy_(r)          # in place: the gradient may be a view of a flat buffer


class Marker(_NoForward, torch.nn.Module):
    """Parameter-free layer; keeps the reference's module indices in state_dict keys."""

    def __init__(self, kind, arg=None):
        super().__init__()
        self.kind, self.arg = kind, arg

    def extra_repr(self):
        return self.kind if self.arg is None else f"{self.kind}, {self.arg}"


class ResidualHolder(_NoForward, torch.nn.Module):
    def __init__(self, architecture):
        super().__init__()
        self.res_block = build_holders(architecture[0])
        name = architecture[1][0]
        if name is None:
            self.tail = (None, None)
        elif name.lower() == "relu":
            self.tail = ("relu", None)
        elif name.lower() == "leaky relu":
            self.tail = ("leaky relu", architecture[1][1])
        else:
            raise NotImplementedError("Layer {} not supported yet!".format(name))


_PLAIN = ("leaky relu", "relu", "tanh", "sigmoid", "softplus", "flatten", "unflatten")


def build_holders(architecture):
    """Same vocabulary and error behaviour as the reference's ``build_sequential``
    (utils.py:114-157); returns None for ``architecture is None`` (identity)."""
    if architecture is None:
        return None
    mods = []
    for layer in architecture:
        if len(layer) == 2:
            name, config = layer
        elif len(layer) == 1:
            name, config = layer[0], None
        else:
            raise RuntimeError("Layer definition ill-formed: {}.".format(layer))
        name = name.lower()
        if name == "conv":
            mods.append(ParamConv2d(**config))
        elif name == "sn conv":                      # extension: spectrally normalised conv (CGAN discriminator)
            mods.append(SNConv2d(**config))
        elif name == "transp conv":
            mods.append(ParamConvTranspose2d(**config))
        elif name == "linear":
            mods.append(ParamLinear(**config))
        elif name == "prelu":
            mods.append(ParamPReLU())
        elif name == "batchnorm":
            mods.append(ParamBatchNorm2d(**config))
        elif name == "residual block":
            mods.append(ResidualHolder(config))
        elif name in _PLAIN:
            mods.append(Marker(name, config))
        else:
            raise NotImplementedError("Layer {} not supported yet!".format(name))
    return torch.nn.Sequential(*mods)


# ------------------------------------------------------------------ plan data structures
class PW:
    """Per-channel (scale, shift, slope) arrays of a pending activation; may be slices of a
    wider array when the slot is part of a channel concatenation."""

    def __init__(self, scale, shift, slope):
        self.scale, self.shift, self.slope = scale, shift, slope
        self.struct = L.Pointwise(scale.data_ptr(), shift.data_ptr(), slope.data_ptr())

    @staticmethod
    def identity(c, device):
        return PW(torch.ones(c, device=device), torch.zeros(c, device=device), torch.ones(c, device=device))

    def slice(self, c0, c1):
        return PW(self.scale[c0:c1], self.shift[c0:c1], self.slope[c0:c1])


class Slot:
    def __init__(self, buf, n, h, w, c, coff=0, pw=None, parent=None):
        self.buf = buf                       # torch tensor (n,h,w,cstride), float32 or bfloat16
        self.n, self.h, self.w, self.c = n, h, w, c
        self.cstride = buf.shape[-1]
        self.coff = coff
        self.pw = pw                          # None = identity
        self.parent = parent
        self.dt = L.BF16 if buf.dtype == torch.bfloat16 else L.F32     # element type of the slot AND its gradients
        self.view = L.View(buf.data_ptr(), n, h, w, c, self.cstride, coff, self.dt)
        self.grad_buf = None
        self.grad = None                      # L.View
        self.grad2 = None                     # L.View (second contribution) or None
        self.n_consumers = 0

    @staticmethod
    def new(n, h, w, c, device, cstride=None, pw=None, bf16=False):
        cs = c if cstride is None else cstride
        return Slot(torch.zeros((n, h, w, cs), device=device, dtype=torch.bfloat16 if bf16 else torch.float32),
                    n, h, w, c, 0, pw)

    def sub(self, c0, c1, pw=None, dense_grad=False):
        """Channel slice sharing storage (and gradient storage, unless ``dense_grad``: a one-channel slice of a
        four-channel buffer costs every streaming pass over its gradient four times the bytes)."""
        s = Slot(self.buf, self.n, self.h, self.w, c1 - c0, self.coff + c0, pw, parent=self)
        s.dense_grad = dense_grad
        return s

    def pw_struct(self):
        return None if self.pw is None else C.byref(self.pw.struct)

    def shape(self):
        return (self.n, self.h, self.w, self.c)

    # ---- gradients
    dense_grad = False       # a channel slice that keeps d(loss)/d(slot) in a dense buffer of its own (see sub())

    def ensure_grad(self):
        if self.grad is None:
            if self.parent is not None and self.dense_grad:
                self.grad_buf = torch.zeros((self.n, self.h, self.w, self.c), device=self.buf.device, dtype=self.buf.dtype)
                self.grad = L.View(self.grad_buf.data_ptr(), self.n, self.h, self.w, self.c, self.c, 0, self.dt)
                return self.grad
            if self.parent is not None:
                self.parent.ensure_grad()
                self.grad_buf = self.parent.grad_buf
            else:
                self.grad_buf = torch.zeros_like(self.buf)
            self.grad = L.View(self.grad_buf.data_ptr(), self.n, self.h, self.w, self.c, self.cstride, self.coff,
                               self.dt)
        return self.grad

    def claim_grad(self):
        """A consumer asks where to write d(loss)/d(activated slot)."""
        self.n_consumers += 1
        if self.n_consumers == 1:
            return self.ensure_grad()
        if self.n_consumers == 2 and self.grad2 is None:
            self._grad2_buf = torch.zeros((self.n, self.h, self.w, self.c), device=self.buf.device, dtype=self.buf.dtype)
            self.grad2 = L.View(self._grad2_buf.data_ptr(), self.n, self.h, self.w, self.c, self.c, 0, self.dt)
            return self.grad2
        raise NotImplementedError("more than two consumers of one activation")

    def set_grad2_alias(self, view):
        self.n_consumers += 1
        if self.grad2 is not None or self.n_consumers > 2:
            raise NotImplementedError("more than two consumers of one activation")
        self.grad2 = view


class ConvUnit:
    """conv / transp conv [+ batchnorm] [+ relu | leaky relu | prelu]  (utils.py:128-147)."""

    def __init__(self, plan, name, holder, bn, act, act_arg, act_holder, inp, out_slot=None, out_pw=None,
                 need_dgrad=True):
        self.plan, self.name = plan, name
        dev = plan.device
        transposed = isinstance(holder, torch.nn.ConvTranspose2d)
        k, s, p = holder.kernel_size, holder.stride, holder.padding
        op = holder.output_padding if transposed else (0, 0)
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or op[0] != op[1] or holder.groups != 1 \
                or tuple(holder.dilation) != (1, 1):
            raise NotImplementedError(f"{name}: only square, undilated, ungrouped convolutions")
        self.cv = cv = L.Conv(1 if transposed else 0, holder.in_channels, holder.out_channels, k[0], s[0], p[0],
                              op[0])
        if inp.c != cv.cin:
            raise ValueError(f"{name}: input has {inp.c} channels, layer expects {cv.cin}")
        self.holder, self.bn, self.act_holder = holder, bn, act_holder
        self.act, self.act_arg = act, act_arg
        self.inp, self.need_dgrad = inp, need_dgrad
        if transposed:
            ho = (inp.h - 1) * cv.stride - 2 * cv.pad + cv.k + cv.out_pad
            wo = (inp.w - 1) * cv.stride - 2 * cv.pad + cv.k + cv.out_pad
        else:
            ho = (inp.h + 2 * cv.pad - cv.k) // cv.stride + 1
            wo = (inp.w + 2 * cv.pad - cv.k) // cv.stride + 1
        if ho <= 0 or wo <= 0:
            raise ValueError(f"{name}: empty output {ho}x{wo}")
        c = cv.cout
        self.has_pw = bn is not None or act is not None
        if out_pw is None and self.has_pw:
            out_pw = PW.identity(c, dev)
        if out_pw is not None:
            out_pw.scale.fill_(1.0)
            out_pw.shift.fill_(0.0)
            if act == "relu":
                out_pw.slope.fill_(0.0)
            elif act == "leaky relu":
                out_pw.slope.fill_(float(act_arg))
            else:
                out_pw.slope.fill_(1.0)      # none; prelu is refreshed every forward
        self.out_pw = out_pw
        # bf16 matrix-core kernels for this layer (plan policy: the generator trunk under dtype="bf16"), and bf16
        # storage of its output when the plan says every consumer reads bf16
        self.bf16 = bool(getattr(plan, "bf16_unit", lambda n_: False)(name)) and holder.bias is None \
            and plan.lib.bp_conv_bf16_supported(C.byref(cv), L.PACK_FWD, None, None) == 1 \
            and plan.lib.bp_conv_bf16_supported(C.byref(cv), L.PACK_BWD, None, None) == 1
        out_bf16 = self.bf16 and out_slot is None and bool(plan.bf16_out(name)) and c >= 8 and (c & (c - 1)) == 0
        if out_slot is None:
            self.out = Slot.new(inp.n, ho, wo, c, dev, pw=out_pw, bf16=out_bf16)
        else:
            if out_slot.shape() != (inp.n, ho, wo, c):
                raise ValueError(f"{name}: concat slot shape {out_slot.shape()} != {(inp.n, ho, wo, c)}")
            self.out = out_slot
            self.out.pw = out_pw
        self.sums = torch.zeros(3 * c, device=dev, dtype=torch.float64)
        if bn is not None:
            self.save_mean = torch.zeros(c, device=dev, dtype=torch.float64)
            self.save_invstd = torch.zeros(c, device=dev, dtype=torch.float64)
            self.abc = torch.zeros(4 * c, device=dev, dtype=torch.float64)
            self.count = 1.0
        lib = plan.lib
        if self.bf16:
            n_fwd = lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_FWD)
            n_bwd = lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_BWD)
            pdt = torch.bfloat16
        else:
            if inp.dt != L.F32:
                raise NotImplementedError(f"{name}: an fp32 layer cannot read a bf16 activation")
            n_fwd = lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD)
            n_bwd = lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD)
            pdt = torch.float32
        if n_fwd <= 0 or n_bwd <= 0:
            raise NotImplementedError(f"{name}: shape not supported by the gfx950 kernels")
        self.packed_fwd = torch.zeros(n_fwd, device=dev, dtype=pdt)
        self.packed_bwd = torch.zeros(n_bwd, device=dev, dtype=pdt) if need_dgrad else None
        self._packed_version = None
        self.dx = None
        self.dgrad_slice = None        # (c0, c1): only these input channels need a gradient
        self._sub = None
        plan.need_ws(lib.bp_channel_sums_workspace(C.byref(self.out.view)))
        # Batch-norm statistics out of the convolution's own epilogue (bp_conv_forward_stats) instead of a streaming
        # pass over the tensor it just wrote; the same for the two sums of a producer's batch-norm backward, taken
        # in this layer's data-gradient epilogue (bp_conv_backward_data_stats, decided at the first backward).
        self.out.producer = self
        self.fwd_stats = False
        self._fused_producer = None          # None: undecided; False: separate pass; a ConvUnit: fused
        self._sums_ready = False             # a consumer's data gradient has already filled self.sums
        self._g_ready = False                # ... and stored g itself (layers without batch-norm)
        self._act_producer = None            # None: undecided; False: separate pass; a ConvUnit: fused
        self._stats_impl = L.IMPL_BF16 if self.bf16 else L.IMPL_MFMA
        if EPILOGUE_FWD and bn is not None and holder.bias is None \
                and (self.bf16 or (self.out.dt == L.F32 and self._impl("fwd") in (L.IMPL_AUTO, L.IMPL_MFMA))):
            nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(inp.view), C.byref(self.out.view),
                                             self._stats_impl)
            if nb > 0:
                self.fwd_stats = True
                plan.need_ws(nb)

    def restrict_dgrad(self, c0, c1, target=None):
        """Only input channels [c0, c1) carry a gradient (the rest of a concatenated input is data): the data
        gradient then runs as a (c1-c0)-channel layer on a copy of that weight slice.  ``target``: the sub-slot
        of those channels when it keeps a DENSE gradient buffer of its own (``Slot.dense_grad``): the gradient is
        written there instead of into a channel slice of the wide buffer."""
        cv = self.cv
        if not (0 <= c0 < c1 <= cv.cin) or self.dx is not None:
            raise ValueError(f"{self.name}: bad gradient channel range or backward already prepared")
        if target is not None and target.grad is not None:
            raise ValueError(f"{self.name}: the target slot's gradient buffer already exists")
        if (c0, c1) == (0, cv.cin) or not self.need_dgrad:
            if target is not None:
                target.dense_grad = False       # (the gradient stays in the wide buffer)
            return
        lib, dev = self.plan.lib, self.plan.device
        sub = L.Conv(cv.transposed, c1 - c0, cv.cout, cv.k, cv.stride, cv.pad, cv.out_pad)
        if self.bf16:
            n_bwd = lib.bp_conv_bf16_packed_elems(C.byref(sub), L.PACK_BWD)
        else:
            n_bwd = lib.bp_conv_packed_floats(C.byref(sub), L.PACK_BWD)
        if n_bwd <= 0:
            if target is not None:
                target.dense_grad = False
            return                      # no kernel for the narrower layer: keep the full data gradient
        w = self.holder.weight
        shape = (c1 - c0, cv.cout, cv.k, cv.k) if cv.transposed else (cv.cout, c1 - c0, cv.k, cv.k)
        self.dgrad_slice = (c0, c1)
        self._sub = {"cv": sub, "w": torch.zeros(shape, device=dev, dtype=w.dtype),
                     "packed": torch.zeros(n_bwd, device=dev, dtype=torch.bfloat16 if self.bf16 else torch.float32),
                     "target": target if target is not None and target.dense_grad else None}
        self.packed_bwd = None

    ws_own = None       # this layer's own split-K workspace when the plan defers the reductions (cvae._Plan)
    ws_name = "ws"      # which of the plan's workspaces this unit's reductions use (branches that run on
                        # their own stream get their own: cvae._Plan)

    def _ws(self):
        return getattr(self.plan, self.ws_name)

    def _impl(self, kind):
        return L.IMPL_BF16 if self.bf16 else self.plan.impl_of(kind, self.name)

    def macs(self, kind="forward"):
        """Multiply-accumulates of one forward (= of each of the two gradients; fewer for a data gradient
        restricted to a channel slice)."""
        cv = self.cv
        dense = self.inp if cv.transposed else self.out       # the grid every tap visits
        cin = cv.cin
        if kind == "backward_data" and self.dgrad_slice is not None:
            cin = self.dgrad_slice[1] - self.dgrad_slice[0]
        return dense.n * dense.h * dense.w * cv.k * cv.k * cin * cv.cout

    def algorithmic_bytes(self, kind, nstreams=1):
        """HBM bytes one launch of this layer's kernel has to move (SURVEY.md 8d convention: every tensor it reads
        or writes exactly once; weights are negligible)."""
        def nbytes(slot, c=None):
            return slot.n * slot.h * slot.w * (slot.c if c is None else c) * (2 if slot.dt == L.BF16 else 4)
        if kind == "forward":
            return nbytes(self.inp) + nbytes(self.out)
        if kind == "backward_weight":
            return nbytes(self.inp) + nbytes(self.out)
        if kind == "backward_data":
            c = None if self.dgrad_slice is None else self.dgrad_slice[1] - self.dgrad_slice[0]
            return nbytes(self.out) + nbytes(self.inp, c)
        return nstreams * nbytes(self.out)          # streaming passes over tensors of the output's shape

    # ---- weights
    def maybe_pack(self):
        w = self.holder.weight
        ver = (w._version, w.data_ptr(), getattr(self.plan.model, "_param_epoch", 0))
        if ver == self._packed_version:
            return
        lib, st = self.plan.lib, _stream()
        pack = lib.bp_conv_bf16_pack if self.bf16 else lib.bp_conv_pack
        L.check(pack(C.byref(self.cv), L.PACK_FWD, L.ptr(w), L.ptr(self.packed_fwd), st), "pack")
        if self.packed_bwd is not None:
            L.check(pack(C.byref(self.cv), L.PACK_BWD, L.ptr(w), L.ptr(self.packed_bwd), st), "pack")
        if self._sub is not None:
            c0, c1 = self.dgrad_slice
            sub = self._sub
            with torch.no_grad():
                sub["w"].copy_(w[c0:c1] if self.cv.transposed else w[:, c0:c1])
            L.check(pack(C.byref(sub["cv"]), L.PACK_BWD, L.ptr(sub["w"]), L.ptr(sub["packed"]), st), "pack")
        self._packed_version = ver

    # ---- forward
    def _drive(self, steps):
        """Run a unit's step generator on its own: every statistic it yields is all-reduced right away."""
        try:
            while True:
                sums = next(steps)               # (StopIteration ends the unit; single-device units never yield)
                self.plan.sync.all_reduce_sum(sums)
        except StopIteration as done:
            return done.value

    def forward(self, training):
        return self._drive(self.forward_steps(training))

    def forward_steps(self, training):
        """Generator: the forward pass, yielding once - the batch-norm sums that must become global - under data
        parallelism.  A caller that drives several independent units in lock step (cvae._Plan, the recognition and
        prior branches) answers all their yields with ONE collective over the buffer their ``sums`` are views of."""
        plan, lib, st = self.plan, self.plan.lib, _stream()
        self.maybe_pack()
        hold = self.holder
        fused = training and self.fwd_stats
        if training and self.bn is not None:
            # (running statistics and this unit's pending scale / shift are about to change: eval-mode users recompute)
            plan.model._bn_epoch = getattr(plan.model, "_bn_epoch", 0) + 1
        # single device: the batch-norm finalize rides on the launch that sums the epilogue's partial rows
        # ... and under data parallelism too where the finalize kernel exchanges the channel sums itself (dist.Sync.fused:
        # peer memory; `count` is then the global pixel count)
        sync_bn = plan.sync is not None and plan.sync.sync_bn
        peer_fused = sync_bn and fused and self.bn is not None and FUSED_FINALIZE and plan.sync.fused(plan.device)
        fused_bn = fused and self.bn is not None and FUSED_FINALIZE and (not sync_bn or peer_fused)
        t0 = plan.prof_begin()
        if fused_bn:
            bn = self.bn
            self.count = float(self.out.n * self.out.h * self.out.w) * (plan.sync.world_size if peer_fused else 1)
            dp = lambda t: None if t is None else t.data_ptr()
            bt = L.BnTrain(self.count, dp(bn.weight), dp(bn.bias), float(bn.eps), float(bn.momentum),
                           dp(bn.running_mean), dp(bn.running_var), dp(bn.num_batches_tracked),
                           dp(self.out_pw.scale), dp(self.out_pw.shift), dp(self.save_mean), dp(self.save_invstd))
            if peer_fused:
                plan.sync.peer.bind(True)
                plan.sync.n_fused += 1
            try:
                L.check(lib.bp_conv_forward_bn(C.byref(self.cv), C.byref(self.inp.view), self.inp.pw_struct(),
                                               L.ptr(self.packed_fwd), C.byref(self.out.view), L.ptr(self.sums),
                                               C.byref(bt), L.ptr(self._ws()), plan.ws_bytes, self._stats_impl, st),
                        f"{self.name} forward + bn stats + finalize")
            finally:
                if peer_fused:
                    plan.sync.peer.bind(False)
        elif fused:
            L.check(lib.bp_conv_forward_stats(C.byref(self.cv), C.byref(self.inp.view), self.inp.pw_struct(),
                                              L.ptr(self.packed_fwd), C.byref(self.out.view), L.ptr(self.sums),
                                              L.ptr(self._ws()), plan.ws_bytes, self._stats_impl, st),
                    f"{self.name} forward + bn stats")
        else:
            L.check(lib.bp_conv_forward(C.byref(self.cv), C.byref(self.inp.view), self.inp.pw_struct(),
                                        L.ptr(self.packed_fwd), L.ptr(hold.weight), L.ptr(hold.bias),
                                        C.byref(self.out.view), self._impl("fwd"), st),
                    f"{self.name} forward")
        plan.prof_end(t0, self, "forward")
        c = self.cv.cout
        if self.act == "prelu":
            self.out_pw.slope.copy_(self.act_holder.weight.detach().expand(c))
        bn = self.bn
        if bn is None or fused_bn:
            return
        if training:
            if not fused:
                t0 = plan.prof_begin()
                L.check(lib.bp_channel_sums(C.byref(self.out.view), L.ptr(self.sums), L.ptr(self._ws()),
                                            plan.ws_bytes, st), f"{self.name} bn stats")
                plan.prof_end(t0, self, "bn_stats")
            count = float(self.out.n * self.out.h * self.out.w)
            if plan.sync is not None and plan.sync.sync_bn:
                yield self.sums[:2 * c]
                count *= plan.sync.world_size
            self.count = count
            L.check(lib.bp_bn_finalize(L.ptr(self.sums), count, c, L.ptr(bn.weight), L.ptr(bn.bias),
                                       float(bn.eps), float(bn.momentum), L.ptr(bn.running_mean),
                                       L.ptr(bn.running_var), L.ptr(bn.num_batches_tracked),
                                       L.ptr(self.out_pw.scale), L.ptr(self.out_pw.shift),
                                       L.ptr(self.save_mean), L.ptr(self.save_invstd), st), f"{self.name} bn")
        else:
            self.maybe_bn_eval()

    def maybe_bn_eval(self):
        """Eval mode: the per-channel (scale, shift) of this layer's batch-norm from its running statistics.  They only
        change when the parameters or the running statistics do, so the launch is skipped while neither has (one tiny
        launch per batch-norm layer and batch otherwise: 168 per replay of the paint graph).  A captured graph does
        not contain it: its owner calls this eagerly before a replay, next to ``maybe_pack``."""
        bn = self.bn
        if bn is None:
            return
        m = self.plan.model
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        ver = (getattr(m, "_param_epoch", 0), getattr(m, "_bn_epoch", 0)) + \
            tuple((t._version, t.data_ptr()) if t is not None else None for t in ts)
        if ver == getattr(self, "_bn_eval_version", None):
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"{self.name}: batch-norm statistics changed inside a graph capture")
        L.check(self.plan.lib.bp_bn_eval_pointwise(self.cv.cout, L.ptr(bn.weight), L.ptr(bn.bias),
                                                   L.ptr(bn.running_mean), L.ptr(bn.running_var), float(bn.eps),
                                                   L.ptr(self.out_pw.scale), L.ptr(self.out_pw.shift), _stream()),
                f"{self.name} bn eval")
        self._bn_eval_version = ver

    # ---- backward
    def prepare_backward(self):
        lib = self.plan.lib
        self.out.ensure_grad()
        self.plan.need_ws(lib.bp_act_backward_workspace(C.byref(self.out.view)))
        self._wgrad_ws_bytes = int(lib.bp_conv_backward_weight_workspace(C.byref(self.cv), C.byref(self.inp.view),
                                                                         C.byref(self.out.view)))
        self.plan.need_ws(self._wgrad_ws_bytes)
        if self.need_dgrad and self._sub is not None and self._sub["target"] is not None:
            self.inp.n_consumers += 1          # (the wide slot itself gets no gradient buffer)
            self.dx = self._sub["dx"] = self._sub["target"].ensure_grad()
        else:
            self.dx = self.inp.claim_grad() if self.need_dgrad else None
        if self.dx is not None and self._sub is None and EPILOGUE_BWD and not self.bf16 and self.inp.dt == L.F32 \
                and (_BWD_KERNEL_IDS is None
                     or lib.bp_conv_kernel_id(C.byref(self.cv), L.PACK_BWD) in _BWD_KERNEL_IDS):
            self.plan.need_ws(lib.bp_conv_stats_workspace(C.byref(self.cv), L.PACK_BWD, C.byref(self.inp.view),
                                                          C.byref(self.out.view), L.IMPL_MFMA))
        if self.dx is not None and self._sub is None and EPILOGUE_BWD and self.bf16 and self.inp.dt == L.BF16 \
                and os.environ.get("BP_BF16_BWD_STATS", "1") != "0":
            self.plan.need_ws(lib.bp_conv_stats_workspace(C.byref(self.cv), L.PACK_BWD, C.byref(self.inp.view),
                                                          C.byref(self.out.view), L.IMPL_BF16))
        if self.dx is not None and self._sub is None and EPILOGUE_BWD \
                and ((not self.bf16 and self.inp.dt == L.F32) or (self.bf16 and self.inp.dt == L.BF16 and self.out.dt == L.F32)):
            self.plan.need_ws(lib.bp_conv_backward_data_act_workspace(C.byref(self.cv), C.byref(self.out.view),
                                                                      C.byref(self.inp.view)))
        if self.dx is not None and self._sub is not None and self._sub["target"] is None:
            c0, c1 = self.dgrad_slice
            full = self.dx
            self._sub["dx"] = L.View(full.ptr, full.n, full.h, full.w, c1 - c0, full.cstride, full.coff + c0,
                                     full.dtype)

    def backward(self, grads):
        """``out.grad`` (+``out.grad2``) hold d/d(activated out).  Writes parameter gradients into
        ``grads[id(param)]`` and d/d(activated in) into the view claimed from the input slot."""
        return self._drive(self.backward_steps(grads))

    def backward_steps(self, grads):
        """Generator form of ``backward`` (see ``forward_steps``)."""
        out = self.out
        if self.has_pw or out.grad2 is not None:
            pw = None if self.out_pw is None else C.byref(self.out_pw.struct)
            yield from self.activation_backward_steps(out.grad, out.grad2, pw, None,
                                                      None if self.bn is not None else out.grad, out.grad, grads)
        self.conv_backward(grads)

    def activation_backward(self, *args):
        return self._drive(self.activation_backward_steps(*args))

    def activation_backward_steps(self, dout, dout2, pw, act_out, g_out, d_raw_out, grads):
        """g = (dout+dout2)*act'(t) -> g_out;  batch-norm backward of g -> d_raw_out.
        ``g_out`` None (batch-norm layers whose g nobody else reads): g is not stored, the apply pass
        recomputes it from dout / the activation mask.  Generator: yields the sums to make global (once)."""
        plan, lib, st = self.plan, self.plan.lib, _stream()
        c = self.cv.cout
        bn = self.bn
        if g_out is None and bn is None:
            raise ValueError("without batch-norm g IS the result: it must be stored")
        aout = None if act_out is None else C.byref(act_out)
        d2 = None if dout2 is None else C.byref(dout2)
        nstreams = 2 + (dout2 is not None) + (act_out is not None)       # tensors this pass and the apply pass read
        finalized = False
        if self._sums_ready and g_out is None and dout2 is None and act_out is None:
            self._sums_ready = False         # the consumer's data gradient left {sum g, sum g*raw} in self.sums
        elif self._g_ready and bn is None and dout2 is None and act_out is None and g_out is dout:
            self._g_ready = False            # ... or did the whole pass (bp_conv_backward_data_act): g is in place
        else:
            if self._g_ready:
                raise RuntimeError(f"{self.name}: the fused activation backward does not match this call")
            if self._sums_ready:
                raise RuntimeError(f"{self.name}: fused statistics do not match this activation backward")
            t0 = plan.prof_begin()
            sync_bn = plan.sync is not None and plan.sync.sync_bn
            peer_fused = sync_bn and bn is not None and FUSE_BN_BWD_FINALIZE and plan.sync.fused(plan.device)
            if bn is not None and (not sync_bn or peer_fused) and FUSE_BN_BWD_FINALIZE:
                # the batch-norm backward finalize rides on the launch that adds the partial sums; data parallel over peer
                # memory: that launch also exchanges the channel sums (count is global, losses are per-rank means)
                dp = lambda t: None if t is None else t.data_ptr()
                pscale = 1.0 / plan.sync.world_size if peer_fused else 1.0
                fin = L.BnBackwardFin(self.count, dp(bn.weight), dp(self.save_mean), dp(self.save_invstd), pscale,
                                      dp(grads[id(bn.weight)]), dp(grads[id(bn.bias)]), dp(self.abc))
                if peer_fused:
                    plan.sync.peer.bind(True)
                    plan.sync.n_fused += 1
                try:
                    L.check(lib.bp_act_backward_bn(C.byref(dout), d2, C.byref(self.out.view), pw, aout,
                                                   None if g_out is None else C.byref(g_out), L.ptr(self.sums),
                                                   C.byref(fin), L.ptr(self._ws()), plan.ws_bytes, st),
                            f"{self.name} act backward + bn backward finalize")
                finally:
                    if peer_fused:
                        plan.sync.peer.bind(False)
                finalized = True
            else:
                L.check(lib.bp_act_backward(C.byref(dout), d2, C.byref(self.out.view), pw, aout,
                                            None if g_out is None else C.byref(g_out), L.ptr(self.sums),
                                            L.ptr(self._ws()), plan.ws_bytes, st), f"{self.name} act backward")
            plan.prof_end(t0, self, "act_backward", nstreams + (g_out is not None))
        if self.act == "prelu":
            L.check(lib.bp_prelu_slope_grad(L.ptr(self.sums), c, L.ptr(grads[id(self.act_holder.weight)]), st),
                    f"{self.name} prelu grad")
        if bn is not None:
            pscale = 1.0
            if plan.sync is not None and plan.sync.sync_bn and not finalized:
                yield self.sums[:2 * c]
                pscale = 1.0 / plan.sync.world_size       # sums are global, losses are per-rank means
            if not finalized:
                L.check(lib.bp_bn_backward_finalize(L.ptr(self.sums), self.count, c, L.ptr(bn.weight),
                                                    L.ptr(self.save_mean), L.ptr(self.save_invstd), pscale,
                                                    L.ptr(grads[id(bn.weight)]), L.ptr(grads[id(bn.bias)]),
                                                    L.ptr(self.abc), st), f"{self.name} bn backward")
            t0 = plan.prof_begin()
            if g_out is None:
                L.check(lib.bp_act_bn_backward_apply(C.byref(dout), d2, C.byref(self.out.view), pw, aout,
                                                     L.ptr(self.abc), C.byref(d_raw_out), st),
                        f"{self.name} act+bn backward apply")
                plan.prof_end(t0, self, "bn_apply", nstreams + 1)
            else:
                L.check(lib.bp_bn_backward_apply(C.byref(g_out), C.byref(self.out.view), L.ptr(self.abc),
                                                 C.byref(d_raw_out), st), f"{self.name} bn backward apply")
                plan.prof_end(t0, self, "bn_apply", 3)
            return True
        return False

    def _producer_to_fuse(self):
        """The unit whose batch-norm backward sums this layer's data gradient can take in its epilogue: the only
        producer of ``inp`` when this layer is its only consumer (then d/d(activated inp) is exactly what this
        kernel writes) and its activation is a fixed-slope one.  Decided once, after every consumer has claimed."""
        if self._fused_producer is None:
            self._fused_producer = False
            p = getattr(self.inp, "producer", None)
            common = EPILOGUE_BWD and isinstance(p, ConvUnit) and p.out is self.inp and p.bn is not None \
                and p.act in (None, "relu", "leaky relu") and p.plan is self.plan and p.ws_name == self.ws_name \
                and self.inp.n_consumers == 1 and self.inp.grad2 is None and self._sub is None \
                and self.packed_bwd is not None
            if common and not self.bf16 and self.inp.dt == L.F32 and self._impl("dgrad") in (L.IMPL_AUTO, L.IMPL_MFMA):
                nb = self.plan.lib.bp_conv_stats_workspace(C.byref(self.cv), L.PACK_BWD, C.byref(self.inp.view),
                                                           C.byref(self.out.view), L.IMPL_MFMA)
                if _BWD_KERNEL_IDS is not None \
                        and self.plan.lib.bp_conv_kernel_id(C.byref(self.cv), L.PACK_BWD) not in _BWD_KERNEL_IDS:
                    nb = 0
                if 0 < nb <= self.plan.ws_bytes:
                    self._fused_producer = p
            elif common and self.bf16 and self.inp.dt == L.BF16 and os.environ.get("BP_BF16_BWD_STATS", "1") != "0":
                # the two full-resolution flattened-K bf16 data gradients (conv_bf16_flat.hip, STATS == 2)
                nb = self.plan.lib.bp_conv_stats_workspace(C.byref(self.cv), L.PACK_BWD, C.byref(self.inp.view),
                                                           C.byref(self.out.view), L.IMPL_BF16)
                if 0 < nb <= self.plan.ws_bytes:
                    self._fused_producer = p
        return self._fused_producer

    def _producer_act_to_fuse(self):
        """The producer WITHOUT batch-norm whose whole activation backward this layer's data gradient can do in its
        epilogue (bp_conv_backward_data_act): only producer / only consumer of ``inp`` and a kernel that has the
        epilogue (the data gradient of the heads' 8 -> 1 k5 layer: on the vector ALUs between fp32 slots, on the
        matrix cores -- conv_bf16_head.hip -- where the 8-channel slot is stored as bf16)."""
        if self._act_producer is None:
            self._act_producer = False
            p = getattr(self.inp, "producer", None)
            if EPILOGUE_BWD and isinstance(p, ConvUnit) and p.out is self.inp and p.bn is None and p.has_pw \
                    and p.act in ("relu", "leaky relu", "prelu") and p.plan is self.plan and p.ws_name == self.ws_name \
                    and self.inp.n_consumers == 1 and self.inp.grad2 is None \
                    and self.out.dt == L.F32 and self._sub is None and self.packed_bwd is not None \
                    and ((not self.bf16 and self.inp.dt == L.F32 and self._impl("dgrad") in (L.IMPL_AUTO, L.IMPL_MFMA))
                         or (self.bf16 and self.inp.dt == L.BF16)) and self.dx is not None:
                nb = self.plan.lib.bp_conv_backward_data_act_workspace(C.byref(self.cv), C.byref(self.out.view),
                                                                       C.byref(self.inp.view))
                if 0 < nb <= self.plan.ws_bytes:
                    self._act_producer = p
        return self._act_producer

    def conv_backward(self, grads):
        """Weight gradient and data gradient of this layer from d_raw (``out.grad``).

        With a side stream (``plan.side``) the weight gradient is forked here and only joined at the end of the
        backward pass: it depends on nothing the chain below produces.  Two different matrix-core kernels (this
        weight gradient, the data gradients that follow) then share the CUs and fill each other's barrier / staging
        stalls, and the HBM-bound activation / batch-norm passes run beside MFMA work: 54.0 -> 50.4 ms per step."""
        plan, lib, st = self.plan, self.plan.lib, _stream()
        g, hold = self.out.grad, self.holder
        dbias = None if hold.bias is None else grads[id(hold.bias)]
        side, side_ws = plan.side_of(self) if hasattr(plan, "side_of") else (getattr(plan, "side", None), None)
        if side is not None and side_ws is None:
            side_ws = plan.ws2

        def wgrad(ws):
            t0 = plan.prof_begin()
            nbytes, flags = plan.ws_bytes, (L.IMPL_SHARED if side is not None else 0)
            if self.ws_own is not None and getattr(plan, "deferring", False):
                ws, nbytes = self.ws_own, self.ws_own.numel() * 8      # (its reduction waits for the plan's flush)
                flags |= L.IMPL_DEFER
            L.check(lib.bp_conv_backward_weight(C.byref(self.cv), C.byref(self.inp.view), self.inp.pw_struct(),
                                                C.byref(g), L.ptr(grads[id(hold.weight)]), L.ptr(dbias),
                                                L.ptr(ws), nbytes,
                                                self._impl("wgrad") | flags, _stream()),
                    f"{self.name} backward_weight")
            plan.prof_end(t0, self, "backward_weight")

        if getattr(plan, "skip_wgrad", False):
            pass        # only the data gradient is wanted (generator step through the discriminator)
        elif side is None:
            wgrad(self._ws())
        else:
            side.wait_stream(torch.cuda.current_stream())     # fork: d_raw of this layer is complete
            with torch.cuda.stream(side):
                wgrad(side_ws)
        if self.dx is not None:
            prod = self._producer_to_fuse()
            aprod = None if prod else self._producer_act_to_fuse()
            t0 = plan.prof_begin()
            if aprod:
                L.check(lib.bp_conv_backward_data_act(C.byref(self.cv), C.byref(g), L.ptr(self.packed_bwd),
                                                      C.byref(self.dx), C.byref(self.inp.view), self.inp.pw_struct(),
                                                      L.ptr(aprod.sums), L.ptr(self._ws()), plan.ws_bytes, st),
                        f"{self.name} backward_data + {aprod.name} activation backward")
                aprod._g_ready = True
            elif prod:
                L.check(lib.bp_conv_backward_data_stats(C.byref(self.cv), C.byref(g), L.ptr(self.packed_bwd),
                                                        C.byref(self.dx), C.byref(self.inp.view),
                                                        self.inp.pw_struct(), L.ptr(prod.sums), L.ptr(self._ws()),
                                                        plan.ws_bytes, st),
                        f"{self.name} backward_data + {prod.name} activation sums")
                prod._sums_ready = True
            elif self._sub is not None:
                sub = self._sub
                L.check(lib.bp_conv_backward_data(C.byref(sub["cv"]), C.byref(g), L.ptr(sub["packed"]),
                                                  L.ptr(sub["w"]), C.byref(sub["dx"]), self._impl("dgrad"), st),
                        f"{self.name} backward_data (channels {self.dgrad_slice})")
            else:
                L.check(lib.bp_conv_backward_data(C.byref(self.cv), C.byref(g), L.ptr(self.packed_bwd),
                                                  L.ptr(hold.weight), C.byref(self.dx), self._impl("dgrad"), st),
                        f"{self.name} backward_data")
            plan.prof_end(t0, self, "backward_data")


class PackBatch:
    """All weight re-layouts of a plan in one launch (``bp_conv_pack_jobs``) instead of two per layer.

    The job table holds raw pointers: it is rebuilt when the parameter storage moves.  Layers whose packing is
    not batchable (vector-ALU kernels) keep their own ``bp_conv_pack`` call."""

    def __init__(self, plan, units):
        self.plan, self.units = plan, units
        self._storage = None

    def _build(self):
        lib, dev = self.plan.lib, self.plan.device
        nb = lib.bp_conv_pack_job_bytes()
        recs, counts, self.rest, self.rest_late = [], [], [], []     # (rest_late: units marked ``pack_late`` by the plan)
        self.own = [u for u in self.units if u.bf16]            # (bf16 images: the layer's own pack launches)
        for u in self.units:
            if u.bf16:
                continue
            jobs = [(u.cv, L.PACK_FWD, u.holder.weight, u.packed_fwd)]
            if u.packed_bwd is not None:
                jobs.append((u.cv, L.PACK_BWD, u.holder.weight, u.packed_bwd))
            if u._sub is not None:
                jobs.append((u._sub["cv"], L.PACK_BWD, u._sub["w"], u._sub["packed"]))
            for cv, d, w, packed in jobs:
                buf = (C.c_char * nb)()
                n = C.c_int64(0)
                rc = lib.bp_conv_pack_job(C.byref(cv), d, L.ptr(w), L.ptr(packed), buf, C.byref(n))
                if rc == L.BP_OK:
                    recs.append(bytes(buf))
                    counts.append(n.value)
                elif getattr(u, "pack_late", False):
                    self.rest_late.append((cv, d, w, packed))
                else:
                    self.rest.append((cv, d, w, packed))
        first = [0]
        for c in counts:
            first.append(first[-1] + c)
        self.njobs, self.total_blocks = len(recs), first[-1]
        self.jobs = torch.frombuffer(bytearray(b"".join(recs) or b"\0"), dtype=torch.uint8).to(dev)
        self.first = torch.tensor(first, dtype=torch.int64, device=dev)

    def run(self):
        plan, lib, st = self.plan, self.plan.lib, _stream()
        storage = tuple(u.holder.weight.data_ptr() for u in self.units)
        if storage != self._storage:
            self._build()
            self._storage = storage
        # bf16 images keep their own pack launches (two per layer).  In a training plan they go to the weight-gradient
        # stream, idle during the forward pass: the first bf16 layer (run_generator) waits for them, the recognition /
        # prior networks in front of it do not.
        side = getattr(plan, "side", None)
        plan._own_packed = None
        for u in self.units:                 # (weight slices of restricted data gradients: before anything packs them)
            if u._sub is not None and not u.bf16:
                c0, c1 = u.dgrad_slice
                w = u.holder.weight
                with torch.no_grad():
                    u._sub["w"].copy_(w[c0:c1] if u.cv.transposed else w[:, c0:c1])

        def late(stream):
            for u in self.own:
                u._packed_version = None
                u.maybe_pack()
            # ... and the single-launch packs of fp32 layers the generator reads first (``pack_late`` units): the
            # recognition / prior networks at the front of the step do not wait for them
            for cv, d, w, packed in self.rest_late:
                L.check(lib.bp_conv_pack(C.byref(cv), d, L.ptr(w), L.ptr(packed), stream), "pack")
        if side is not None and (self.own or self.rest_late) and not torch.cuda.is_current_stream_capturing():
            side.wait_stream(torch.cuda.current_stream(plan.device))          # the optimizer's update
            with torch.cuda.stream(side):
                late(_stream())
                plan._own_packed = torch.cuda.Event()
                plan._own_packed.record(side)
        else:
            late(st)
        L.check(lib.bp_conv_pack_jobs(L.ptr(self.jobs), L.ptr(self.first), self.njobs, self.total_blocks, st),
                "batched pack")
        for cv, d, w, packed in self.rest:
            L.check(lib.bp_conv_pack(C.byref(cv), d, L.ptr(w), L.ptr(packed), st), "pack")
        for u in self.units:
            w = u.holder.weight
            u._packed_version = (w._version, w.data_ptr(), getattr(plan.model, "_param_epoch", 0))


class ResidualUnit:
    """activation(res_block(x) + x)  (utils.py:22-38)."""

    def __init__(self, plan, name, body_units, inp, tail):
        self.plan, self.name = plan, name
        self.body, self.inp = body_units, inp
        kind, arg = tail
        self.slope = {"relu": 0.0, None: 1.0}.get(kind, arg)
        last = body_units[-1].out
        if last.shape() != inp.shape():
            raise ValueError(f"{name}: residual branch changes the shape")
        if last.dt != inp.dt:
            raise NotImplementedError(f"{name}: residual branch and skip differ in element type")
        self.out = Slot.new(inp.n, inp.h, inp.w, inp.c, plan.device, pw=None, bf16=inp.dt == L.BF16)
        lp = last.pw if last.pw is not None else PW.identity(inp.c, plan.device)
        # mask pointwise of the tail: the branch's affine, the tail's slope
        self.tail_pw = PW(lp.scale, lp.shift, torch.full((inp.c,), float(self.slope), device=plan.device))

    def forward(self, training):
        for u in self.body:
            u.forward(training)
        last = self.body[-1].out
        t0 = self.plan.prof_begin()
        L.check(self.plan.lib.bp_residual_forward(C.byref(last.view), last.pw_struct(), C.byref(self.inp.view),
                                                  self.inp.pw_struct(), float(self.slope),
                                                  C.byref(self.out.view), _stream()), f"{self.name} tail")
        self.plan.prof_end(t0, self.body[-1], "residual", 3)

    def prepare_backward(self):
        self.out.ensure_grad()
        for u in reversed(self.body):
            u.prepare_backward()
        # skip path: d/d(act(inp)) also receives the tail's masked gradient, kept in out.grad
        self.inp.set_grad2_alias(self.out.grad)

    def backward(self, grads):
        out, last = self.out, self.body[-1]
        # g_t in place in out.grad (also the skip gradient); branch d_raw -> last.out.grad
        had_bn = last.activation_backward(out.grad, out.grad2, C.byref(self.tail_pw.struct), out.view, out.grad,
                                          last.out.grad, grads)
        if not had_bn:
            last.out.grad_buf.copy_(out.grad_buf)
        last.conv_backward(grads)
        for u in reversed(self.body[:-1]):
            u.backward(grads)


# ------------------------------------------------------------------ sequential compiler
def compile_sequential(plan, prefix, architecture, holders, inp, out_slot=None, out_pw=None,
                       need_input_grad=True):
    """Group the layer list into units.  Returns (units, output slot, trailing) where
    ``trailing`` lists parameter-free layers after the last convolution that are not
    expressible as a pending pointwise (softplus / tanh / sigmoid / unflatten / flatten)."""
    units, trailing = [], []
    if architecture is None:
        return units, inp, trailing
    layers = []
    for layer in architecture:
        name = layer[0].lower()
        layers.append((name, layer[1] if len(layer) == 2 else None))
    # index of the last conv-like layer (its output may go to a concat slot)
    conv_like = [i for i, (n, _) in enumerate(layers) if n in ("conv", "sn conv", "transp conv", "residual block")]
    last_conv = conv_like[-1] if conv_like else -1
    cur = inp
    i = 0
    first = True
    while i < len(layers):
        name, cfg = layers[i]
        if name in ("conv", "sn conv", "transp conv"):
            holder = holders[i]
            j = i + 1
            bn = act = act_arg = act_holder = None
            if j < len(layers) and layers[j][0] == "batchnorm":
                bn = holders[j]
                j += 1
            if j < len(layers) and layers[j][0] in ("relu", "leaky relu", "prelu"):
                act, act_arg = layers[j][0], layers[j][1]
                act_holder = holders[j] if act == "prelu" else None
                j += 1
            is_last = (i == last_conv)
            u = ConvUnit(plan, f"{prefix}{i}", holder, bn, act, act_arg, act_holder, cur,
                         out_slot=out_slot if is_last else None, out_pw=out_pw if is_last else None,
                         need_dgrad=(need_input_grad or not first))
            units.append(u)
            cur = u.out
            i = j
        elif name == "residual block":
            rh = holders[i]
            body, body_out, tr = compile_sequential(plan, f"{prefix}{i}.res_block.", cfg[0], rh.res_block, cur)
            if tr:
                raise NotImplementedError("residual branch must end in a convolution/batch-norm")
            if i == last_conv and out_slot is not None:
                raise NotImplementedError("residual block as the last layer of a concatenated branch")
            u = ResidualUnit(plan, f"{prefix}{i}", body, cur, rh.tail)
            units.append(u)
            cur = u.out
            i += 1
        elif name in ("softplus", "tanh", "sigmoid", "unflatten", "flatten"):
            if i < last_conv:
                raise NotImplementedError(f"{prefix}{i}: '{name}' between convolutions is not supported "
                                          "by the HIP path")
            trailing.append((name, cfg))
            i += 1
        else:
            raise NotImplementedError(f"{prefix}{i}: layer '{name}' is not supported by the HIP path here")
        first = False
    return units, cur, trailing


def probe_output(architecture, c, h, w):
    """(c,h,w) produced by a layer list, without building anything."""
    for layer in architecture or []:
        name = layer[0].lower()
        if name in ("conv", "sn conv", "transp conv"):
            cfg = layer[1]
            k, st, pd = cfg["kernel_size"], cfg.get("stride", 1), cfg.get("padding", 0)
            if name in ("conv", "sn conv"):
                h, w = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
            else:
                op = cfg.get("output_padding", 0)
                h, w = (h - 1) * st - 2 * pd + k + op, (w - 1) * st - 2 * pd + k + op
            c = cfg["out_channels"]
    return c, h, w
