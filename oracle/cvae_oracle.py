"""NumPy restatement of the reference CVAE (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/baryon_painter/models/cvae.py:9-171 and
models/utils.py:22-182.  State-dict key names are the reference's
(``q_x_in.0.weight`` ... ``running_var``).  Backward is hand-derived; the
reference uses autograd on ``-ELBO`` (painter.py:226-228) -- here
``backward(seed)`` returns d(seed*ELBO)/d(param).
"""
import math

import numpy as np

from . import ops


# ---------------------------------------------------------------- layer objects
class _Layer:
    params = ()          # names of learnable arrays, relative to this layer
    buffers = ()

    def backward(self, dy, grads, prefix):
        raise NotImplementedError


class Conv(_Layer):
    params = ("weight", "bias")

    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self.s = cfg.get("stride", 1)
        self.p = cfg.get("padding", 0)
        k = cfg["kernel_size"]
        self.k = (k, k) if isinstance(k, int) else tuple(k)
        self.has_bias = cfg.get("bias", True)
        self.shapes = {"weight": (cfg["out_channels"], cfg["in_channels"], *self.k)}
        if self.has_bias:
            self.shapes["bias"] = (cfg["out_channels"],)

    def forward(self, x, P, prefix, train):
        self.x = x
        self.w = P[prefix + "weight"]
        y = ops.conv2d_fwd(x, self.w, self.s, self.p)
        if self.has_bias:
            y = y + P[prefix + "bias"][None, :, None, None]
        return y

    def backward(self, dy, grads, prefix):
        grads[prefix + "weight"] = ops.conv2d_bwd_weight(self.x, dy, self.s, self.p, *self.k)
        if self.has_bias:
            grads[prefix + "bias"] = dy.sum(axis=(0, 2, 3))
        return ops.conv2d_bwd_data(dy, self.w, self.s, self.p, *self.x.shape[2:])


class ConvT(Conv):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.op = cfg.get("output_padding", 0)
        self.shapes["weight"] = (cfg["in_channels"], cfg["out_channels"], *self.k)

    def forward(self, x, P, prefix, train):
        self.x = x
        self.w = P[prefix + "weight"]
        y = ops.convT2d_fwd(x, self.w, self.s, self.p, self.op)
        if self.has_bias:
            y = y + P[prefix + "bias"][None, :, None, None]
        return y

    def backward(self, dy, grads, prefix):
        grads[prefix + "weight"] = ops.convT2d_bwd_weight(self.x, dy, self.s, self.p, *self.k)
        if self.has_bias:
            grads[prefix + "bias"] = dy.sum(axis=(0, 2, 3))
        return ops.convT2d_bwd_data(dy, self.w, self.s, self.p)


class BatchNorm(_Layer):
    params = ("weight", "bias")
    buffers = ("running_mean", "running_var", "num_batches_tracked")

    def __init__(self, cfg):
        self.C = cfg["num_features"]
        self.eps = cfg.get("eps", 1e-5)
        self.momentum = cfg.get("momentum", 0.1)
        self.shapes = {"weight": (self.C,), "bias": (self.C,)}

    def forward(self, x, P, prefix, train):
        g, b = P[prefix + "weight"], P[prefix + "bias"]
        self.g = g
        if train:
            y, (self.xhat, self.invstd, mean, var) = ops.batchnorm_train_fwd(x, g, b, self.eps)
            n = x.shape[0] * x.shape[2] * x.shape[3]
            rm, rv = ops.batchnorm_running_update(P[prefix + "running_mean"], P[prefix + "running_var"],
                                                  mean, var, n, self.momentum)
            P[prefix + "running_mean"] = rm.astype(P[prefix + "running_mean"].dtype)
            P[prefix + "running_var"] = rv.astype(P[prefix + "running_var"].dtype)
            P[prefix + "num_batches_tracked"] = P[prefix + "num_batches_tracked"] + 1
            return y
        return ops.batchnorm_eval_fwd(x, g, b, P[prefix + "running_mean"], P[prefix + "running_var"], self.eps)

    def backward(self, dy, grads, prefix):
        dx, dg, db = ops.batchnorm_bwd(dy, self.xhat, self.invstd, self.g)
        grads[prefix + "weight"] = dg
        grads[prefix + "bias"] = db
        return dx


class Act(_Layer):
    def __init__(self, kind, arg=None):
        self.kind, self.arg = kind, arg
        self.shapes = {"weight": (1,)} if kind == "prelu" else {}
        if kind == "prelu":
            self.params = ("weight",)

    def forward(self, x, P, prefix, train):
        self.x = x
        k = self.kind
        if k == "relu":
            return ops.relu(x)
        if k == "leaky relu":
            return ops.leaky_relu(x, self.arg)
        if k == "prelu":
            self.a = P[prefix + "weight"][0]
            return ops.leaky_relu(x, self.a)
        if k == "tanh":
            self.y = np.tanh(x)
            return self.y
        if k == "sigmoid":
            self.y = ops.sigmoid(x)
            return self.y
        if k == "softplus":
            return ops.softplus(x)
        raise NotImplementedError(k)

    def backward(self, dy, grads, prefix):
        k = self.kind
        if k == "relu":
            return dy * (self.x > 0)
        if k == "leaky relu":
            return np.where(self.x > 0, dy, dy * self.arg)
        if k == "prelu":
            dx, da = ops.prelu_bwd(dy, self.x, self.a)
            grads[prefix + "weight"] = np.array([da])
            return dx
        if k == "tanh":
            return dy * (1 - self.y ** 2)
        if k == "sigmoid":
            return dy * self.y * (1 - self.y)
        if k == "softplus":
            return dy * ops.softplus_grad(self.x)
        raise NotImplementedError(k)


class UnFlatten(_Layer):
    shapes = {}

    def __init__(self, dim):
        self.dim = tuple(dim)

    def forward(self, x, P, prefix, train):
        self.in_shape = x.shape
        return x.reshape(x.shape[0], *self.dim)

    def backward(self, dy, grads, prefix):
        return dy.reshape(self.in_shape)


class Flatten(UnFlatten):
    def __init__(self):
        pass

    def forward(self, x, P, prefix, train):
        self.in_shape = x.shape
        return x.reshape(x.shape[0], -1)


class Linear(_Layer):
    params = ("weight", "bias")

    def __init__(self, cfg):
        self.has_bias = cfg.get("bias", True)
        self.shapes = {"weight": (cfg["out_features"], cfg["in_features"])}
        if self.has_bias:
            self.shapes["bias"] = (cfg["out_features"],)

    def forward(self, x, P, prefix, train):
        self.x, self.w = x, P[prefix + "weight"]
        y = x @ self.w.T
        return y + P[prefix + "bias"] if self.has_bias else y

    def backward(self, dy, grads, prefix):
        grads[prefix + "weight"] = dy.T @ self.x
        if self.has_bias:
            grads[prefix + "bias"] = dy.sum(axis=0)
        return dy @ self.w


class Residual(_Layer):
    """models/utils.py:22-38: activation(res_block(x) + x)."""
    shapes = {}

    def __init__(self, architecture):
        self.body = Sequential(architecture[0])
        name = architecture[1][0]
        if name is None:
            self.act = None
        elif name.lower() == "relu":
            self.act = Act("relu")
        elif name.lower() == "leaky relu":
            self.act = Act("leaky relu", architecture[1][1])
        else:
            raise NotImplementedError("Layer {} not supported yet!".format(name))

    def named_layers(self, prefix):
        return self.body.named_layers(prefix + "res_block.")

    def forward(self, x, P, prefix, train):
        h = self.body.forward(x, P, prefix + "res_block.", train) + x
        return self.act.forward(h, P, prefix, train) if self.act else h

    def backward(self, dy, grads, prefix):
        if self.act:
            dy = self.act.backward(dy, grads, prefix)
        return self.body.backward(dy, grads, prefix + "res_block.") + dy


class Sequential:
    """models/utils.py:114-157 ``build_sequential``: same layer vocabulary."""

    def __init__(self, architecture):
        self.layers = []
        for layer in architecture or []:
            if len(layer) == 2:
                name, config = layer
            elif len(layer) == 1:
                name, config = layer[0], None
            else:
                raise RuntimeError("Layer definition ill-formed: {}.".format(layer))
            name = name.lower()
            if name == "conv":
                self.layers.append(Conv(config))
            elif name == "transp conv":
                self.layers.append(ConvT(config))
            elif name == "linear":
                self.layers.append(Linear(config))
            elif name in ("leaky relu", "relu", "prelu", "tanh", "sigmoid", "softplus"):
                self.layers.append(Act(name, config))
            elif name == "batchnorm":
                self.layers.append(BatchNorm(config))
            elif name == "residual block":
                self.layers.append(Residual(config))
            elif name == "flatten":
                self.layers.append(Flatten())
            elif name == "unflatten":
                self.layers.append(UnFlatten(config))
            else:
                raise NotImplementedError("Layer {} not supported yet!".format(name))

    def named_layers(self, prefix):
        for i, l in enumerate(self.layers):
            if isinstance(l, Residual):
                yield from l.named_layers(f"{prefix}{i}.")
            else:
                yield f"{prefix}{i}.", l

    def forward(self, x, P, prefix, train):
        for i, l in enumerate(self.layers):
            x = l.forward(x, P, f"{prefix}{i}.", train)
        return x

    def backward(self, dy, grads, prefix):
        for i in reversed(range(len(self.layers))):
            dy = self.layers[i].backward(dy, grads, f"{prefix}{i}.")
        return dy


# ---------------------------------------------------------------------- the CVAE
class CVAEOracle:
    """cvae.py:8-61.  ``params`` is a dict name -> ndarray with the reference's
    state_dict keys (learnables and BN buffers)."""

    def __init__(self, architecture, params=None, dtype=np.float64):
        a = architecture
        if a["type"] != "Type-1":
            raise NotImplementedError("Architecture {} not supported yet!".format(a["type"]))
        self.architecture = a
        self.dtype = dtype
        self.dim_x, self.dim_y, self.dim_z = tuple(a["dim_x"]), tuple(a["dim_y"]), tuple(a["dim_z"])
        self.L = a.get("L", 1)
        self.n_x_features = a["n_x_features"]
        self.nets = {
            "q_x_in": Sequential(a["q_x_in"]),
            "q_y_in": Sequential(a["q_y_in"]),
            "q_out": Sequential(a["q_x_y_out"]),
            "p_y_in": Sequential(a["p_y_in"]),
            "p_z_in": Sequential(a["p_z_in"]),
            "p_y_z_in": Sequential(a["p_y_z_in"]),
            "p_mu_out": Sequential(a["p_y_z_out"][0]),
        }
        self.predict_var = len(a["p_y_z_out"]) > 1
        if self.predict_var:
            self.nets["p_var_out"] = Sequential(a["p_y_z_out"][1])
            self.min_x_var = a.get("min_x_var", 1e-7)
        self.use_aux_label = a["aux_label"]
        self.has_prior = "prior_z_y" in a
        if self.has_prior:
            self.nets["prior_network"] = Sequential(a["prior_z_y"])
        self.min_z_var = a.get("min_z_var", 1e-7)
        self.likelihood_scaling = a.get("likelihood_scaling", 1.0)
        self.alpha_var = 1.0
        self.beta_KL = 1.0
        self.training = True
        self.P = {}
        if params is not None:
            self.load_params(params)

    # -- parameters
    def param_shapes(self):
        """name -> shape for learnables, in the reference's registration order."""
        out = {}
        for net in ("q_x_in", "q_y_in", "q_out", "p_y_in", "p_z_in", "p_y_z_in", "p_mu_out",
                    "p_var_out", "prior_network"):
            if net not in self.nets:
                continue
            for prefix, layer in self.nets[net].named_layers(net + "."):
                for pname in layer.params:
                    if pname in layer.shapes:
                        out[prefix + pname] = layer.shapes[pname]
        return out

    def buffer_shapes(self):
        out = {}
        for net, seq in self.nets.items():
            for prefix, layer in seq.named_layers(net + "."):
                if isinstance(layer, BatchNorm):
                    out[prefix + "running_mean"] = (layer.C,)
                    out[prefix + "running_var"] = (layer.C,)
                    out[prefix + "num_batches_tracked"] = ()
        return out

    def load_params(self, params):
        self.P = {}
        for k, v in params.items():
            v = np.asarray(v)
            self.P[k] = v.astype(np.int64) if k.endswith("num_batches_tracked") else v.astype(self.dtype)
        for k, shp in self.buffer_shapes().items():
            if k not in self.P:
                if k.endswith("running_var"):
                    self.P[k] = np.ones(shp, self.dtype)
                elif k.endswith("num_batches_tracked"):
                    self.P[k] = np.zeros(shp, np.int64)
                else:
                    self.P[k] = np.zeros(shp, self.dtype)

    # -- pieces (cvae.py:63-120)
    def _merge(self, y, aux):
        if aux is not None and self.use_aux_label:
            return ops.merge_aux_label(y, aux)
        return y

    def sample_z(self, z_mu, z_log_var, eps):
        """cvae.py:63-66; ``eps`` (L,N,*dim_z) replaces torch.randn.  NB: min_z_var
        is added to the standard deviation, as the reference does."""
        self._eps = eps
        self._std = np.exp(z_log_var / 2)
        z = z_mu[None] + eps * (self._std[None] + self.min_z_var)
        return z.reshape(-1, *self.dim_z)

    def Q(self, x, y, aux, eps):
        y2 = self._merge(y, aux)
        h_x = self.nets["q_x_in"].forward(x, self.P, "q_x_in.", self.training)
        h_y = self.nets["q_y_in"].forward(y2, self.P, "q_y_in.", self.training)
        self._cx = h_x.shape[1]
        h = self.nets["q_out"].forward(np.concatenate([h_x, h_y], 1), self.P, "q_out.", self.training)
        self.z_mu, self.z_log_var = h[:, 0], h[:, 1]
        assert self.z_mu.shape[1:] == self.dim_z
        return self.sample_z(self.z_mu, self.z_log_var, eps)

    def prior(self, y, aux):
        if not self.has_prior:
            z = np.zeros((y.shape[0], *self.dim_z), self.dtype)
            return z, z.copy()
        h = self.nets["prior_network"].forward(self._merge(y, aux), self.P, "prior_network.", self.training)
        return h[:, 0], h[:, 1]

    def Pnet(self, z, y, L, aux):
        y2 = self._merge(y, aux)
        h_y = self.nets["p_y_in"].forward(y2, self.P, "p_y_in.", self.training)
        h_z = self.nets["p_z_in"].forward(z, self.P, "p_z_in.", self.training)
        self._cz = h_z.shape[1]
        h = np.concatenate([h_z, np.tile(h_y, (L, 1, 1, 1))], 1)
        h = self.nets["p_y_z_in"].forward(h, self.P, "p_y_z_in.", self.training)
        x_mu = self.nets["p_mu_out"].forward(h, self.P, "p_mu_out.", self.training)
        if self.predict_var:
            return x_mu, self.nets["p_var_out"].forward(h, self.P, "p_var_out.", self.training)
        return (x_mu,)

    # -- forward (cvae.py:122-147)
    def forward(self, x, y, aux, eps):
        x, y = x.astype(self.dtype), y.astype(self.dtype)
        aux = None if aux is None else np.asarray(aux, self.dtype)
        eps = np.asarray(eps, self.dtype)
        z = self.Q(x, y, aux, eps)
        M = x.shape[0]
        self.M = M
        self.p_mu, self.p_lv = self.prior(y, aux)
        p_var = np.exp(self.p_lv)
        self.KL_term = 0.5 / M * np.sum((self.p_mu - self.z_mu) ** 2 / p_var + np.exp(self.z_log_var) / p_var
                                        + self.p_lv - self.z_log_var - 1)
        params = self.Pnet(z, y, self.L, aux)
        self.x_mu = params[0]
        self._x_rep = np.tile(x, (self.L, 1, 1, 1))
        diff = self._x_rep - self.x_mu
        c = -0.5 * math.log(2 * math.pi)
        self.log_likelihood_fixed_var = c + (-0.5 * diff ** 2).sum(axis=(3, 2, 0)) / (M * self.L)
        if self.predict_var:
            self.log_x_var = params[1]
            self.x_var = np.exp(self.log_x_var)
            self.log_likelihood_free_var = c + (-0.5 * self.log_x_var - 0.5 * diff ** 2 / self.x_var
                                                ).sum(axis=(3, 2, 0)) / (M * self.L)
            self.log_likelihood = ((1 - self.alpha_var) * self.log_likelihood_fixed_var
                                   + self.alpha_var * self.log_likelihood_free_var)
        else:
            self.log_likelihood = self.log_likelihood_fixed_var
        self.ELBO = -self.KL_term * self.beta_KL + self.likelihood_scaling * self.log_likelihood.sum()
        return self.ELBO

    # -- backward of seed*ELBO
    def backward(self, seed=1.0):
        g = {}
        M, L = self.M, self.L
        diff = self._x_rep - self.x_mu
        s = seed * self.likelihood_scaling / (M * L)
        if self.predict_var:
            a = self.alpha_var
            d_xmu = s * ((1 - a) * diff + a * diff / self.x_var)
            d_lv = s * a * (-0.5 + 0.5 * diff ** 2 / self.x_var)
            dh = self.nets["p_var_out"].backward(d_lv, g, "p_var_out.")
            dh = dh + self.nets["p_mu_out"].backward(d_xmu, g, "p_mu_out.")
        else:
            dh = self.nets["p_mu_out"].backward(s * diff, g, "p_mu_out.")
        dcat = self.nets["p_y_z_in"].backward(dh, g, "p_y_z_in.")
        self._dcat = dcat          # kept for debugging tools
        dz = self.nets["p_z_in"].backward(dcat[:, :self._cz], g, "p_z_in.")
        if self.nets["p_y_in"].layers:
            d_hy = dcat[:, self._cz:]
            d_hy = d_hy.reshape(L, M, *d_hy.shape[1:]).sum(0)
            self.nets["p_y_in"].backward(d_hy, g, "p_y_in.")
        # reparametrisation (cvae.py:63-66)
        dz = dz.reshape(L, M, *self.dim_z)
        d_zmu = dz.sum(0)
        d_zlv = (dz * self._eps).sum(0) * 0.5 * self._std
        # KL (cvae.py:129-130), ELBO = -beta*KL + ...
        k = -seed * self.beta_KL * 0.5 / M
        p_var = np.exp(self.p_lv)
        dm = self.p_mu - self.z_mu
        d_zmu = d_zmu + k * (-2 * dm / p_var)
        d_zlv = d_zlv + k * (np.exp(self.z_log_var) / p_var - 1)
        d_pmu = k * (2 * dm / p_var)
        d_plv = k * (-(dm ** 2) / p_var - np.exp(self.z_log_var) / p_var + 1)
        if self.has_prior:
            self.nets["prior_network"].backward(np.stack([d_pmu, d_plv], 1), g, "prior_network.")
        dh = self.nets["q_out"].backward(np.stack([d_zmu, d_zlv], 1), g, "q_out.")
        self.nets["q_x_in"].backward(dh[:, :self._cx], g, "q_x_in.")
        self.nets["q_y_in"].backward(dh[:, self._cx:], g, "q_y_in.")
        return g

    # -- inference (cvae.py:97-100,149-162)
    def sample_P(self, y, aux=None, z=None, eps=None, return_var=False):
        y = y.astype(self.dtype)
        aux = None if aux is None else np.asarray(aux, self.dtype)
        if z is None:
            p_mu, p_lv = self.prior(y, aux)
            z = self.sample_z(p_mu, p_lv, np.asarray(eps, self.dtype))
        else:
            z = np.asarray(z, self.dtype)
        p = self.Pnet(z, y, 1, aux)
        if len(p) == 2 and return_var:
            return p[0], np.exp(p[1])
        return p[0]

    def get_stats(self):
        """cvae.py:164-171 (same tuple order)."""
        if self.predict_var:
            return (float(self.ELBO), -float(self.KL_term), *self.log_likelihood,
                    *self.log_likelihood_fixed_var, *self.log_likelihood_free_var)
        return (float(self.ELBO), -float(self.KL_term), *self.log_likelihood)
