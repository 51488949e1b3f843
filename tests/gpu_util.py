"""Helpers for the -m gpu parity tests: call the C ABI on NHWC device buffers."""
import ctypes as C

import numpy as np
import torch

from baryon_painter_amd import _lib as L


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_nhwc(x_nchw, cstride=None, coff=0):
    """NCHW numpy -> (torch NHWC buffer with `cstride` channels, bp_view)."""
    n, c, h, w = x_nchw.shape
    cs = c if cstride is None else cstride
    buf = torch.full((n, h, w, cs), 7.5, dtype=torch.float32, device="cuda")      # poison the unused channels
    buf[..., coff:coff + c] = torch.from_numpy(np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1))).cuda()
    return buf, L.View(buf.data_ptr(), n, h, w, c, cs, coff)


def empty_nhwc(n, h, w, c, cstride=None, coff=0):
    cs = c if cstride is None else cstride
    buf = torch.full((n, h, w, cs), float("nan"), dtype=torch.float32, device="cuda")
    return buf, L.View(buf.data_ptr(), n, h, w, c, cs, coff)


def from_nhwc(buf, c, coff=0):
    return buf[..., coff:coff + c].permute(0, 3, 1, 2).contiguous().cpu().numpy()


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def pointwise(scale, shift, slope):
    ts = [dev(np.asarray(v, np.float32)) for v in (scale, shift, slope)]
    return ts, L.Pointwise(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr())


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
