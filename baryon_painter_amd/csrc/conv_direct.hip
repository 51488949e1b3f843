// Direct (VALU, no LDS) convolution kernels.  These are the plain statement of the arithmetic on
// the GPU: one thread per output pixel and group of produced channels, torch-layout weights read
// as wave-uniform scalars.  They serve as an on-device second opinion for the MFMA kernels and
// for shapes the MFMA kernels do not cover; the hot path uses conv_igemm.hip / conv_wgrad.hip.
#include "common.hpp"

namespace {

constexpr int DB = 8;  // produced channels per thread

struct DirectArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_h, out_w, out_cs, out_co;
  int n;
  const float* w; int64_t sa, sb;
  const float* bias;
  int k, stride, pad, cin_g, cout_g, transposed;
  PW pw;
};

__global__ __launch_bounds__(256) void direct_gather_kernel(DirectArgs a) {
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t npix = (int64_t)a.n * a.out_h * a.out_w;
  if (pix >= npix) return;
  const int b0 = blockIdx.y * DB;
  const int X = pix % a.out_w;
  const int Y = (pix / a.out_w) % a.out_h;
  const int n = pix / ((int64_t)a.out_w * a.out_h);
  float acc[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < a.k; ++ky) {
    int iy;
    if (!a.transposed) {
      iy = Y * a.stride + ky - a.pad;
    } else {
      const int t = Y + a.pad - ky;
      if (t < 0 || t % a.stride) continue;
      iy = t / a.stride;
    }
    if (iy < 0 || iy >= a.in_h) continue;
    for (int kx = 0; kx < a.k; ++kx) {
      int ix;
      if (!a.transposed) {
        ix = X * a.stride + kx - a.pad;
      } else {
        const int t = X + a.pad - kx;
        if (t < 0 || t % a.stride) continue;
        ix = t / a.stride;
      }
      if (ix < 0 || ix >= a.in_w) continue;
      const float* ip = a.in + (((int64_t)n * a.in_h + iy) * a.in_w + ix) * a.in_cs + a.in_co;
      const float* wp = a.w + ky * a.k + kx;
      for (int ci = 0; ci < a.cin_g; ++ci) {
        const float v = pw_apply(a.pw, ci, ip[ci]);
#pragma unroll
        for (int j = 0; j < DB; ++j) {
          const int b = b0 + j;
          if (b < a.cout_g) acc[j] = fmaf(v, wp[ci * a.sa + b * a.sb], acc[j]);
        }
      }
    }
  }
  float* op = a.out + (((int64_t)n * a.out_h + Y) * a.out_w + X) * a.out_cs + a.out_co;
#pragma unroll
  for (int j = 0; j < DB; ++j) {
    const int b = b0 + j;
    if (b < a.cout_g) op[b] = acc[j] + (a.bias ? a.bias[b] : 0.f);
  }
}

// dst[cy][cx][ky][kx] = sum_{n,q} X[n, q*s + k - p, cx] * Y[n, q, cy]
struct DirectWgradArgs {
  const float* X; int xh, xw, xcs, xco, cx;
  const float* Y; int yh, yw, ycs, yco, cy;
  int n, k, stride, pad;
  PW pwx, pwy;
  float* dst;
};

__global__ __launch_bounds__(256) void direct_wgrad_kernel(DirectWgradArgs a) {
  const int kk = a.k * a.k;
  int id = blockIdx.x;
  const int kx = id % a.k; id /= a.k;
  const int ky = id % a.k; id /= a.k;
  const int cx = id % a.cx; id /= a.cx;
  const int cy = id;
  const int64_t npix = (int64_t)a.n * a.yh * a.yw;
  double acc = 0.0;
  for (int64_t p = threadIdx.x; p < npix; p += blockDim.x) {
    const int qx = p % a.yw;
    const int qy = (p / a.yw) % a.yh;
    const int n = p / ((int64_t)a.yw * a.yh);
    const int iy = qy * a.stride + ky - a.pad, ix = qx * a.stride + kx - a.pad;
    if (iy < 0 || iy >= a.xh || ix < 0 || ix >= a.xw) continue;
    const float xv = pw_apply(a.pwx, cx, a.X[(((int64_t)n * a.xh + iy) * a.xw + ix) * a.xcs + a.xco + cx]);
    const float yv = pw_apply(a.pwy, cy, a.Y[(((int64_t)n * a.yh + qy) * a.yw + qx) * a.ycs + a.yco + cy]);
    acc += (double)xv * (double)yv;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.dst[((int64_t)cy * a.cx + cx) * kk + ky * a.k + kx] = (float)red[0];
}

}  // namespace

int bp_direct_gather(const ConvGeom& g, const WeightMap& wm, const bp_view* in, const PW& pw,
                     const float* w_torch, const float* bias, const bp_view* out, hipStream_t st) {
  DirectArgs a{};
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.n = in->n; a.w = w_torch; a.sa = wm.sa; a.sb = wm.sb; a.bias = bias;
  a.k = g.k; a.stride = g.stride; a.pad = g.pad; a.cin_g = g.cin_g; a.cout_g = g.cout_g;
  a.transposed = g.gather_transposed; a.pw = pw;
  const int64_t npix = (int64_t)out->n * out->h * out->w;
  dim3 grid((unsigned)((npix + 255) / 256), (unsigned)bp_ceil_div(g.cout_g, DB));
  hipLaunchKernelGGL(direct_gather_kernel, grid, dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_direct_wgrad(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y,
                    const PW& pwy, float* dst, hipStream_t st) {
  DirectWgradArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff; a.cy = Y->c;
  a.n = X->n; a.k = cv->k; a.stride = cv->stride; a.pad = cv->pad; a.pwx = pwx; a.pwy = pwy;
  a.dst = dst;
  const int64_t blocks = (int64_t)a.cx * a.cy * cv->k * cv->k;
  hipLaunchKernelGGL(direct_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
