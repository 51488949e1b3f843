"""The reference's module graph re-assembled from stock ``torch.nn.functional`` calls on the
CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

This is what the reference executes on a host: ATen/oneDNN convolutions, batch-norm and
autograd (cvae.py:63-146, utils.py:114-182) -- here driven by the architecture lists directly,
with parameters in a flat dict under the reference's state_dict keys.  It is used
  * as the timed ``cpu_baseline`` ("port") of bench.py on the GPU box's host cores, and
  * for full-size parity checks where the NumPy oracle would take minutes.
It is pinned to the golden fixtures by tests/test_oracle_golden.py.
"""
import math

import torch
import torch.nn.functional as F


class _StoreBf16(torch.autograd.Function):
    """A tensor that the bf16 mode keeps in HBM as bf16: rounded (to nearest even) on the way forward, and its gradient
    rounded on the way back -- whatever the arithmetic type of the graph around it (float64 in the tests)."""

    @staticmethod
    def forward(ctx, x, both):
        ctx.both = both
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.both else g), None


class _StageGradBf16(torch.autograd.Function):
    """The gradient of a matrix-core layer's fp32 output is rounded to bf16 when the data- and weight-gradient kernels
    stage it (the value itself stays fp32)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _bf16_layer(bf16, p):
    """The bf16 policy of baryon_painter_amd.models.cvae._Plan.bf16_unit / bf16_out for the layer with state_dict
    prefix ``p``: (runs on bf16 matrix cores, output stored as bf16)."""
    if not bf16:
        return False, False
    name = p[:-1]
    head0 = name in ("p_mu_out.0", "p_var_out.0")
    return (name.startswith("p_y_z_in.") or head0 or name in ("p_mu_out.2", "p_var_out.2"),
            name.startswith("p_y_z_in.") or head0)


def _seq(architecture, x, P, prefix, training, tap=None, bf16=False):
    """``tap`` (debug aid, tools/chain_bisect.py): dict that receives every convolution's raw output under the
    layer's state_dict prefix, with ``retain_grad()`` so that d(loss)/d(raw) can be read after backward.
    ``bf16``: a ROUNDING TWIN of the throughput mode (BASELINE.json configs[3]) -- the graph stays in its own dtype, but
    every tensor the HIP path stores or stages as bf16 is rounded where that path rounds it: the operands of the trunk's
    matrix-core layers (staged activation, packed weight), their stored raw outputs and residual sums, and the gradients
    stored in those slots.  Not bit-equal to the kernels (accumulation order, fp32 vs this graph's dtype), but the same
    rounding noise at the same places: what a correct bf16 execution's distance from the float64 truth looks like, per
    tensor, conditioning included."""
    for i, layer in enumerate(architecture or []):
        name = layer[0].lower()
        cfg = layer[1] if len(layer) == 2 else None
        p = f"{prefix}{i}."
        if name in ("conv", "transp conv"):
            w = P[p + "weight"]
            mm, stored = _bf16_layer(bf16, p)
            if mm:
                # the stem's input is an fp32 slot whose gradient stays fp32 (dense one-channel buffer)
                x = _StoreBf16.apply(x, not p.startswith("p_y_z_in.0."))
                w = _StoreBf16.apply(w, False)
            if name == "conv":
                x = F.conv2d(x, w, P.get(p + "bias"), stride=cfg.get("stride", 1), padding=cfg.get("padding", 0))
            else:
                x = F.conv_transpose2d(x, w, P.get(p + "bias"), stride=cfg.get("stride", 1),
                                       padding=cfg.get("padding", 0), output_padding=cfg.get("output_padding", 0))
            if stored:
                x = _StoreBf16.apply(x, True)
            elif mm:
                x = _StageGradBf16.apply(x)
            if tap is not None:
                tap[p] = x
                if x.requires_grad:
                    x.retain_grad()
        elif name == "batchnorm":
            x = F.batch_norm(x, P[p + "running_mean"], P[p + "running_var"], P[p + "weight"], P[p + "bias"],
                             training=training, momentum=0.1, eps=1e-5)
            if training:
                P[p + "num_batches_tracked"] += 1
        elif name == "relu":
            x = F.relu(x)
        elif name == "leaky relu":
            x = F.leaky_relu(x, cfg)
        elif name == "prelu":
            x = F.prelu(x, P[p + "weight"])
        elif name == "softplus":
            x = F.softplus(x)
        elif name == "tanh":
            x = torch.tanh(x)
        elif name == "sigmoid":
            x = torch.sigmoid(x)
        elif name == "unflatten":
            x = x.view(x.size(0), *cfg)
        elif name == "flatten":
            x = x.view(x.size(0), -1)
        elif name == "residual block":
            h = _seq(cfg[0], x, P, p + "res_block.", training, tap, bf16) + x
            tail = cfg[1][0]
            if tail is None:
                x = h
            elif tail.lower() == "relu":
                x = F.relu(h)
            elif tail.lower() == "leaky relu":
                x = F.leaky_relu(h, cfg[1][1])
            else:
                raise NotImplementedError(tail)
            if _bf16_layer(bf16, p)[1]:
                x = _StoreBf16.apply(x, True)
        else:
            raise NotImplementedError(name)
    return x


class TorchRefCVAE:
    def __init__(self, architecture, params, buffers=None, dtype=torch.float32, tap=None, bf16=False):
        """params: name -> array/tensor (learnables).  Buffers default to torch's initial values.
        ``dtype`` float64 gives the "true value" of every tensor; ``tap``, ``bf16`` (rounding twin of the throughput
        mode): see ``_seq``."""
        self.a = architecture
        self.dtype, self.tap, self.bf16 = dtype, tap, bf16
        self.P = {}
        for k, v in params.items():
            t = torch.as_tensor(v, dtype=dtype).clone()
            t.requires_grad_(True)
            self.P[k] = t
        self._init_buffers(buffers or {})
        self.training = True
        self.alpha_var, self.beta_KL = 1.0, 1.0
        self.L = architecture.get("L", 1)
        self.min_z_var = architecture.get("min_z_var", 1e-7)
        self.likelihood_scaling = architecture.get("likelihood_scaling", 1.0)
        self.predict_var = len(architecture["p_y_z_out"]) > 1
        self.dim_z = tuple(architecture["dim_z"])

    def _init_buffers(self, given):
        def walk(arch, prefix):
            for i, layer in enumerate(arch or []):
                name = layer[0].lower()
                p = f"{prefix}{i}."
                if name == "batchnorm":
                    c = layer[1]["num_features"]
                    self.P[p + "running_mean"] = torch.as_tensor(given.get(p + "running_mean", torch.zeros(c)),
                                                                 dtype=self.dtype).clone()
                    self.P[p + "running_var"] = torch.as_tensor(given.get(p + "running_var", torch.ones(c)),
                                                                dtype=self.dtype).clone()
                    self.P[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
                elif name == "residual block":
                    walk(layer[1][0], p + "res_block.")
        a = self.a
        for net, key in (("q_x_in", "q_x_in"), ("q_y_in", "q_y_in"), ("q_out", "q_x_y_out"), ("p_y_in", "p_y_in"),
                         ("p_z_in", "p_z_in"), ("p_y_z_in", "p_y_z_in"), ("prior_network", "prior_z_y")):
            if key in a:
                walk(a[key], net + ".")
        walk(a["p_y_z_out"][0], "p_mu_out.")
        if len(a["p_y_z_out"]) > 1:
            walk(a["p_y_z_out"][1], "p_var_out.")

    def parameters(self):
        return [t for t in self.P.values() if t.requires_grad]

    @staticmethod
    def _merge(y, aux):
        aux = aux.reshape(-1, 1) if aux.dim() <= 1 else aux
        planes = aux.reshape(*aux.shape, 1, 1).expand(*aux.shape, *y.shape[-2:])
        return torch.cat((y, planes), dim=1)

    def forward(self, x, y, aux, eps):
        a, P, tr = self.a, self.P, self.training
        dt, tap = self.dtype, self.tap
        x, y = torch.as_tensor(x, dtype=dt), torch.as_tensor(y, dtype=dt)
        aux, eps = torch.as_tensor(aux, dtype=dt), torch.as_tensor(eps, dtype=dt)
        y2 = self._merge(y, aux) if a["aux_label"] else y
        h = torch.cat([_seq(a["q_x_in"], x, P, "q_x_in.", tr, tap), _seq(a["q_y_in"], y2, P, "q_y_in.", tr, tap)], 1)
        h = _seq(a["q_x_y_out"], h, P, "q_out.", tr, tap)
        self.z_mu, self.z_log_var = h[:, 0], h[:, 1]
        z = (self.z_mu + eps * (torch.exp(self.z_log_var / 2) + self.min_z_var)).view(-1, *self.dim_z)
        M = x.size(0)
        if "prior_z_y" in a:
            hp = _seq(a["prior_z_y"], y2, P, "prior_network.", tr, tap)
            p_mu, p_lv = hp[:, 0], hp[:, 1]
        else:
            p_mu = torch.zeros_like(self.z_mu)
            p_lv = torch.zeros_like(self.z_mu)
        p_var = torch.exp(p_lv)
        self.KL_term = 0.5 / M * torch.sum((p_mu - self.z_mu) ** 2 / p_var + torch.exp(self.z_log_var) / p_var
                                           + p_lv - self.z_log_var - 1)
        self.x_mu, x_lv = self._P(z, y2)
        xr = x.repeat(self.L, 1, 1, 1)
        c0 = -0.5 * math.log(2 * math.pi)
        self.log_likelihood_fixed_var = c0 + (-0.5 * (xr - self.x_mu) ** 2).sum(dim=[3, 2, 0]) / (M * self.L)
        if self.predict_var:
            xv = torch.exp(x_lv)
            self.log_likelihood_free_var = c0 + (-0.5 * x_lv - 0.5 * (xr - self.x_mu) ** 2 / xv
                                                 ).sum(dim=[3, 2, 0]) / (M * self.L)
            self.log_likelihood = ((1 - self.alpha_var) * self.log_likelihood_fixed_var
                                   + self.alpha_var * self.log_likelihood_free_var)
        else:
            self.log_likelihood = self.log_likelihood_fixed_var
        self.ELBO = -self.KL_term * self.beta_KL + self.likelihood_scaling * self.log_likelihood.sum()
        return self.ELBO

    def _P(self, z, y2):
        a, P, tr, tap = self.a, self.P, self.training, self.tap
        h_y = _seq(a["p_y_in"], y2, P, "p_y_in.", tr, tap)
        h_z = _seq(a["p_z_in"], z, P, "p_z_in.", tr, tap)
        b = self.bf16
        h = _seq(a["p_y_z_in"], torch.cat([h_z, h_y.repeat(self.L, 1, 1, 1)], 1), P, "p_y_z_in.", tr, tap, b)
        x_mu = _seq(a["p_y_z_out"][0], h, P, "p_mu_out.", tr, tap, b)
        x_lv = _seq(a["p_y_z_out"][1], h, P, "p_var_out.", tr, tap, b) if self.predict_var else None
        return x_mu, x_lv

    def sample_P(self, y, aux, eps=None, z=None):
        a, P = self.a, self.P
        with torch.no_grad():
            y = torch.as_tensor(y, dtype=self.dtype)
            aux = torch.as_tensor(aux, dtype=self.dtype)
            y2 = self._merge(y, aux) if a["aux_label"] else y
            if z is None:
                hp = _seq(a["prior_z_y"], y2, P, "prior_network.", self.training)
                eps = torch.as_tensor(eps, dtype=self.dtype)
                z = (hp[:, 0] + eps * (torch.exp(hp[:, 1] / 2) + self.min_z_var)).view(-1, *self.dim_z)
            else:
                z = torch.as_tensor(z, dtype=self.dtype)
            return self._P(z, y2)[0]

    def get_stats(self):
        out = (float(self.ELBO), -float(self.KL_term), *self.log_likelihood.detach().numpy())
        if self.predict_var:
            out += (*self.log_likelihood_fixed_var.detach().numpy(), *self.log_likelihood_free_var.detach().numpy())
        return out
