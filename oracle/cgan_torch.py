"""torch CPU restatement of the CGAN training iteration (TEST INFRASTRUCTURE).

The reference has no CGAN code (SURVEY.md key fact 2), so this is NOT pinned to the reference:
it restates, with stock ``torch.nn.functional`` calls and autograd, the same model and losses as
``baryon_painter_amd/models/cgan.py`` (layer tables of trained_models/README.md:106-128).  Parity of
the HIP path against it is therefore "vs. own restatement" only.
"""
import torch
import torch.nn.functional as F

from .torch_ref import _seq


def _sn_weight(P, prefix, training):
    w = P[prefix + "weight_orig"]
    wm = w.reshape(w.shape[0], -1)
    u, v = P[prefix + "weight_u"], P[prefix + "weight_v"]
    with torch.no_grad():
        if training:
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=1e-12))
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def discriminator(arch, x, P, prefix, training):
    for i, layer in enumerate(arch):
        name = layer[0].lower()
        p = f"{prefix}{i}."
        if name == "sn conv":
            cfg = layer[1]
            x = F.conv2d(x, _sn_weight(P, p, training), P.get(p + "bias"), stride=cfg["stride"], padding=cfg["padding"])
        elif name == "leaky relu":
            x = F.leaky_relu(x, layer[1])
        elif name == "sigmoid":
            pass                        # losses are evaluated on the logits
        else:
            raise NotImplementedError(name)
    return x


class TorchCGAN:
    def __init__(self, g_arch, d_arch, state, lambda_perceptual=2.5, dtype=torch.float32):
        """``dtype=torch.float64``: the same graph in double -- the "true value" the fp32 executions (this class in
        float32 under different thread counts, and the HIP path) are measured against."""
        self.g_arch, self.d_arch, self.lam, self.dtype = g_arch, d_arch, lambda_perceptual, dtype
        self.P = {}
        for k, v in state.items():
            t = torch.as_tensor(v).detach().cpu().clone()
            if t.dtype.is_floating_point:
                t = t.to(dtype)
            if t.dtype.is_floating_point and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v")) \
                    and not (k.startswith("discriminator") and k.endswith(".weight")):
                t.requires_grad_(True)
            self.P[k] = t

    def g_params(self):
        return {k: v for k, v in self.P.items() if k.startswith("generator.") and v.requires_grad}

    def d_params(self):
        return {k: v for k, v in self.P.items() if k.startswith("discriminator.") and v.requires_grad}

    def iteration(self, x, y, z, lr_g=5e-5, lr_d=5e-5):
        """One alternating iteration with Adam(betas=(0.5,0.999)); returns (losses, grads_d, grads_g)."""
        x, y = torch.as_tensor(x).to(self.dtype), torch.as_tensor(y).to(self.dtype)
        zc = torch.as_tensor(z, dtype=torch.float32).to(self.dtype).reshape(-1, 1, 1, 1) - 1.0
        cond = torch.cat([y, zc.expand(-1, 1, *y.shape[-2:])], 1)
        fake = torch.tanh(_seq(self.g_arch[:-1], cond, self.P, "generator.", True))
        opt_d = torch.optim.Adam(list(self.d_params().values()), lr=lr_d, betas=(0.5, 0.999))
        opt_g = torch.optim.Adam(list(self.g_params().values()), lr=lr_g, betas=(0.5, 0.999))
        d_in = torch.cat([torch.cat([cond, x], 1), torch.cat([cond, fake.detach()], 1)], 0)
        out = discriminator(self.d_arch, d_in, self.P, "discriminator.", True)
        n = x.shape[0]
        loss_d = 0.5 * (F.softplus(-out[:n]).mean() + F.softplus(out[n:]).mean())
        opt_d.zero_grad()
        loss_d.backward()
        grads_d = {k: v.grad.clone() for k, v in self.d_params().items()}
        opt_d.step()
        out = discriminator(self.d_arch, torch.cat([cond, fake], 1), self.P, "discriminator.", True)
        adv = 0.5 * F.softplus(-out).mean()
        perc = (fake - x).abs().mean()
        opt_g.zero_grad()
        (adv + self.lam * perc).backward()
        grads_g = {k: v.grad.clone() for k, v in self.g_params().items()}
        opt_g.step()
        return ({"D": float(loss_d), "G_adv": float(adv), "G_perceptual": float(perc)}, grads_d, grads_g,
                fake.detach())
