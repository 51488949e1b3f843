"""NumPy restatement of the operators on the CVAE hot path (TEST INFRASTRUCTURE).

Each function states the arithmetic that the reference obtains from a stock
``torch.nn`` module appended by ``build_sequential``
(/root/reference/baryon_painter/models/utils.py:114-157).  Tensors are NCHW like
the reference.  ``dtype`` follows the inputs: pass float64 arrays to get a
"true value" checker, float32 to mimic the reference's precision.

Backward functions are hand-derived (the reference relies on autograd,
painter.py:227); they are pinned by the gradient goldens in tests/golden/.
"""
import numpy as np


# --------------------------------------------------------------------------- conv
def conv_out_size(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def conv2d_fwd(x, w, stride=1, padding=0):
    """torch.nn.Conv2d(bias=False) forward (models/utils.py:128-129).
    x (N,Ci,H,W), w (Co,Ci,kh,kw)."""
    N, Ci, H, W = x.shape
    Co, Ci2, kh, kw = w.shape
    assert Ci == Ci2
    s, p = stride, padding
    Ho, Wo = conv_out_size(H, kh, s, p), conv_out_size(W, kw, s, p)
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    out = np.zeros((N, Co, Ho * Wo), dtype=np.result_type(x, w))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky:ky + s * (Ho - 1) + 1:s, kx:kx + s * (Wo - 1) + 1:s]
            out += np.matmul(w[:, :, ky, kx], patch.reshape(N, Ci, Ho * Wo))
    return out.reshape(N, Co, Ho, Wo)


def conv2d_bwd_data(dy, w, stride, padding, H, W):
    """d/dx of conv2d_fwd.  dy (N,Co,Ho,Wo), w (Co,Ci,kh,kw) -> (N,Ci,H,W)."""
    N, Co, Ho, Wo = dy.shape
    Co2, Ci, kh, kw = w.shape
    assert Co == Co2
    s, p = stride, padding
    dxp = np.zeros((N, Ci, H + 2 * p, W + 2 * p), dtype=np.result_type(dy, w))
    dyf = dy.reshape(N, Co, Ho * Wo)
    for ky in range(kh):
        for kx in range(kw):
            contrib = np.matmul(w[:, :, ky, kx].T, dyf).reshape(N, Ci, Ho, Wo)
            dxp[:, :, ky:ky + s * (Ho - 1) + 1:s, kx:kx + s * (Wo - 1) + 1:s] += contrib
    return dxp[:, :, p:p + H, p:p + W]


def conv2d_bwd_weight(x, dy, stride, padding, kh, kw):
    """d/dw of conv2d_fwd -> (Co,Ci,kh,kw)."""
    N, Ci, H, W = x.shape
    _, Co, Ho, Wo = dy.shape
    s, p = stride, padding
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    dw = np.zeros((Co, Ci, kh, kw), dtype=np.result_type(x, dy))
    dyf = dy.reshape(N, Co, Ho * Wo)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky:ky + s * (Ho - 1) + 1:s, kx:kx + s * (Wo - 1) + 1:s]
            pf = patch.reshape(N, Ci, Ho * Wo)
            dw[:, :, ky, kx] = np.einsum("nop,ncp->oc", dyf, pf, optimize=True)
    return dw


def convT_out_size(n, k, s, p, op=0):
    return (n - 1) * s - 2 * p + k + op


def convT2d_fwd(x, w, stride, padding, output_padding=0):
    """torch.nn.ConvTranspose2d(bias=False) forward (models/utils.py:130-131).
    x (N,Ci,H,W), w (Ci,Co,kh,kw).  It is the data-gradient of a conv whose
    weight is ``w`` read as (Co_conv=Ci, Ci_conv=Co, kh, kw)."""
    N, Ci, H, W = x.shape
    kh, kw = w.shape[2:]
    Ho = convT_out_size(H, kh, stride, padding, output_padding)
    Wo = convT_out_size(W, kw, stride, padding, output_padding)
    return conv2d_bwd_data(x, w, stride, padding, Ho, Wo)


def convT2d_bwd_data(dout, w, stride, padding):
    return conv2d_fwd(dout, w, stride, padding)


def convT2d_bwd_weight(x, dout, stride, padding, kh, kw):
    """-> (Ci,Co,kh,kw), the torch ConvTranspose2d weight layout."""
    return conv2d_bwd_weight(dout, x, stride, padding, kh, kw)


# --------------------------------------------------------------------- batch norm
def batchnorm_train_fwd(x, gamma, beta, eps=1e-5):
    """torch.nn.BatchNorm2d in train mode (models/utils.py:146-147): batch mean
    and *biased* variance over (N,H,W).  Returns y and the cache for backward."""
    mean = x.mean(axis=(0, 2, 3))
    var = x.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
    y = xhat * gamma[None, :, None, None] + beta[None, :, None, None]
    return y, (xhat, invstd, mean, var)


def batchnorm_running_update(running_mean, running_var, mean, var, n, momentum=0.1):
    """Running statistics as torch does: momentum 0.1, *unbiased* variance."""
    unbiased = var * (n / max(n - 1, 1))
    return ((1 - momentum) * running_mean + momentum * mean,
            (1 - momentum) * running_var + momentum * unbiased)


def batchnorm_eval_fwd(x, gamma, beta, running_mean, running_var, eps=1e-5):
    scale = gamma / np.sqrt(running_var + eps)
    shift = beta - running_mean * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def batchnorm_bwd(dy, xhat, invstd, gamma):
    n = dy.shape[0] * dy.shape[2] * dy.shape[3]
    dbeta = dy.sum(axis=(0, 2, 3))
    dgamma = (dy * xhat).sum(axis=(0, 2, 3))
    c = (gamma * invstd)[None, :, None, None]
    dx = c * (dy - dbeta[None, :, None, None] / n - xhat * dgamma[None, :, None, None] / n)
    return dx, dgamma, dbeta


# -------------------------------------------------------------------- activations
def relu(x):
    return np.maximum(x, 0)


def leaky_relu(x, slope):
    return np.where(x > 0, x, x * slope)


def prelu_bwd(dy, x, a):
    """PReLU with ONE shared slope (torch.nn.PReLU() default, utils.py:137)."""
    dx = np.where(x > 0, dy, dy * a)
    da = np.sum(np.where(x > 0, 0.0, dy * x))
    return dx, da


def softplus(x, beta=1.0, threshold=20.0):
    """torch.nn.Softplus(beta=1, threshold=20) (utils.py:143)."""
    bx = x * beta
    return np.where(bx > threshold, x, np.log1p(np.exp(np.minimum(bx, threshold))) / beta)


def softplus_grad(x, beta=1.0, threshold=20.0):
    bx = x * beta
    return np.where(bx > threshold, 1.0, 1.0 / (1.0 + np.exp(-np.minimum(bx, threshold))))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ------------------------------------------------------------------ aux-label merge
def merge_aux_label(y, aux_label):
    """models/utils.py:159-182: broadcast the scalar label(s) to constant
    feature maps and append them after y's channels."""
    aux = np.asarray(aux_label, dtype=y.dtype)
    if aux.ndim == 0 or aux.ndim == 1:
        aux = aux.reshape(-1, 1)
    if aux.shape[0] != y.shape[0]:
        raise ValueError("aux_label batch size needs to match that of y")
    planes = np.broadcast_to(aux[:, :, None, None], (*aux.shape, *y.shape[-2:]))
    return np.concatenate([y, planes], axis=1)
