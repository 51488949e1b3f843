// HBM-bound kernels of the path: per-channel reductions (batch-norm statistics, activation /
// batch-norm backward sums), the residual tail, layout glue, the latent sampler + KL, the Gaussian
// log-likelihood head and Adam.  All reductions are two-stage (per-block partials in a caller
// workspace, then a fixed-order sum in double), so results are bitwise reproducible.
#include "common.hpp"
#include "peer_dev.hpp"
#include <math.h>

namespace {

constexpr int RB = 256;          // threads per block
constexpr int MAX_RBLOCKS = 1024;  // stage-1 blocks of a reduction

static inline int h_next_pow2(int c) {
  int p = 1;
  while (p < c) p <<= 1;
  return p;
}

struct ViewD {
  float* p; int h, w, c, cs, co; int64_t npix;
};
static inline ViewD vd(const bp_view* v) {
  ViewD d{};
  if (v) { d.p = v->ptr; d.h = v->h; d.w = v->w; d.c = v->c; d.cs = v->cstride; d.co = v->coff; d.npix = bp_view_pixels(v); }
  return d;
}

// ---------------------------------------------------------------- channel sums {sum x, sum x^2}
__global__ __launch_bounds__(RB) void channel_sums_kernel(ViewD x, int cbase, int CP, int64_t pix_per_block,
                                                          double* partial /* [nblk][2][c] */) {
  __shared__ double sh[RB];
  const int tid = threadIdx.x;
  const int ch = cbase + tid % CP, slot = tid / CP, slots = RB / CP;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  int64_t p1 = p0 + pix_per_block;
  if (p1 > x.npix) p1 = x.npix;
  double v[2] = {0.0, 0.0};
  if (ch < x.c) {
    for (int64_t p = p0 + slot; p < p1; p += slots) {
      const double t = x.p[p * x.cs + x.co + ch];
      v[0] += t;
      v[1] += t * t;
    }
  }
  // channels of this pass are [cbase, cbase+CP); write at their absolute index
  __syncthreads();
  for (int s = 0; s < 2; ++s) {
    __syncthreads();
    sh[tid] = v[s];
    __syncthreads();
    if (slot == 0 && ch < x.c) {
      double t = 0.0;
      for (int k = 0; k < slots; ++k) t += sh[k * CP + tid % CP];
      partial[((int64_t)blockIdx.x * 2 + s) * x.c + ch] = t;
    }
  }
}

// ================================================================= vectorised fast paths
// Dense views (cstride == c, coff == 0) with a power-of-two channel count are streamed as flat
// float4s: thread t of a block always sees the same 4 channels ((4*t + j) % c) because every block
// starts at a multiple of 1024 elements, so the per-channel sums live in registers.  The block
// tree-reduces over threads that share a channel group (stride = c/4 threads) in LDS.
__host__ __device__ __forceinline__ bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static inline bool dense_ok(const bp_view* v) {
  return v && v->cstride == v->c && v->coff == 0 && is_pow2(v->c) && v->c <= 1024 &&
         (reinterpret_cast<uintptr_t>(v->ptr) % 16 == 0) && ((bp_view_pixels(v) * v->c) % 4 == 0);
}

template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NS][4], int c, double* out /* [NS][c] */, double* sh) {
  const int tid = threadIdx.x;
  const int G = c >= 4 ? c / 4 : 1;     // threads per distinct channel group
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[tid * 4 + j] = acc[s][j];
    __syncthreads();
    for (int st = RB / 2; st >= G; st >>= 1) {
      if (tid < st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[tid * 4 + j] += sh[(tid + st) * 4 + j];
      }
      __syncthreads();
    }
    if (c >= 4) {
      if (tid < G) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[s * c + tid * 4 + j] = sh[tid * 4 + j];
      }
    } else if (tid == 0) {
      if (c == 1) out[s] = sh[0] + sh[1] + sh[2] + sh[3];
      else { out[s * 2] = sh[0] + sh[2]; out[s * 2 + 1] = sh[1] + sh[3]; }
    }
  }
}

__global__ __launch_bounds__(RB) void channel_sums_fast_kernel(const float4* __restrict__ x, int c, int64_t total4,
                                                               int64_t chunk4, double* partial) {
  __shared__ double sh[RB * 4];
  const int64_t b0 = (int64_t)blockIdx.x * chunk4;
  int64_t b1 = b0 + chunk4;
  if (b1 > total4) b1 = total4;
  double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int64_t i = b0 + threadIdx.x; i < b1; i += RB) {
    const float4 v = x[i];
    acc[0][0] += v.x; acc[0][1] += v.y; acc[0][2] += v.z; acc[0][3] += v.w;
    acc[1][0] += (double)v.x * v.x; acc[1][1] += (double)v.y * v.y;
    acc[1][2] += (double)v.z * v.z; acc[1][3] += (double)v.w * v.w;
  }
  block_reduce_store<2>(acc, c, partial + (int64_t)blockIdx.x * 2 * c, sh);
}

struct ActBwdFast {
  const float4* dout; const float4* dout2; const float4* raw; const float4* aout; float4* g;
  PW pw; int c; int64_t total4, chunk4; double* partial;
};

__global__ __launch_bounds__(RB) void act_backward_fast_kernel(ActBwdFast a) {
  __shared__ double sh[RB * 4];
  const int tid = threadIdx.x;
  const int c = a.c;
  float sc[4], sf[4], sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = (4 * tid + j) % c;
    sc[j] = a.pw.scale ? a.pw.scale[ch] : 1.f;
    sf[j] = a.pw.scale ? a.pw.shift[ch] : 0.f;
    sl[j] = a.pw.scale ? a.pw.slope[ch] : 1.f;
  }
  const int64_t b0 = (int64_t)blockIdx.x * a.chunk4;
  int64_t b1 = b0 + a.chunk4;
  if (b1 > a.total4) b1 = a.total4;
  double acc[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int64_t i = b0 + tid; i < b1; i += RB) {
    float4 d4 = a.dout[i];
    if (a.dout2) { const float4 e = a.dout2[i]; d4.x += e.x; d4.y += e.y; d4.z += e.z; d4.w += e.w; }
    const float4 r4 = a.raw[i];
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.aout) s4 = a.aout[i];
    const float d[4] = {d4.x, d4.y, d4.z, d4.w};
    const float r[4] = {r4.x, r4.y, r4.z, r4.w};
    const float so[4] = {s4.x, s4.y, s4.z, s4.w};
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = fmaf(r[j], sc[j], sf[j]);
      const bool pos = (a.aout ? so[j] : t) > 0.f;
      g[j] = pos ? d[j] : d[j] * sl[j];
      acc[0][j] += g[j];
      acc[1][j] += (double)g[j] * r[j];
      if (!pos) acc[2][j] += (double)d[j] * t;
    }
    if (a.g) a.g[i] = make_float4(g[0], g[1], g[2], g[3]);
  }
  block_reduce_store<3>(acc, c, a.partial + (int64_t)blockIdx.x * 3 * c, sh);
}

__global__ __launch_bounds__(RB) void bn_backward_apply_fast_kernel(const float4* g, const float4* raw, const double* abc,
                                                                    float4* out, int c, int64_t total4) {
  const int tid = threadIdx.x;
  double A[4], G[4], B[4], M[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = (4 * tid + j) % c;
    A[j] = abc[ch]; G[j] = abc[c + ch]; B[j] = abc[2 * c + ch]; M[j] = abc[3 * c + ch];
  }
  for (int64_t i = (int64_t)blockIdx.x * RB + tid; i < total4; i += (int64_t)gridDim.x * RB) {
    const float4 gv = g[i], r = raw[i];
    out[i] = make_float4((float)(A[0] * ((double)gv.x - G[0]) + B[0] * ((double)r.x - M[0])),
                         (float)(A[1] * ((double)gv.y - G[1]) + B[1] * ((double)r.y - M[1])),
                         (float)(A[2] * ((double)gv.z - G[2]) + B[2] * ((double)r.z - M[2])),
                         (float)(A[3] * ((double)gv.w - G[3]) + B[3] * ((double)r.w - M[3])));
  }
}

// The same map with g recomputed from the incoming gradient(s) and the activation mask, so that
// bp_act_backward need not write g at all (one tensor write and no re-read of it saved per layer).
struct ActApplyFast {
  const float4* dout; const float4* dout2; const float4* raw; const float4* aout; float4* out;
  PW pw; const double* abc; int c; int64_t total4;
};

__global__ __launch_bounds__(RB) void act_bn_backward_apply_fast_kernel(ActApplyFast a) {
  const int tid = threadIdx.x;
  const int c = a.c;
  double A[4], G[4], B[4], M[4];
  float sc[4], sf[4], sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = (4 * tid + j) % c;
    A[j] = a.abc[ch]; G[j] = a.abc[c + ch]; B[j] = a.abc[2 * c + ch]; M[j] = a.abc[3 * c + ch];
    sc[j] = a.pw.scale ? a.pw.scale[ch] : 1.f;
    sf[j] = a.pw.scale ? a.pw.shift[ch] : 0.f;
    sl[j] = a.pw.scale ? a.pw.slope[ch] : 1.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * RB + tid; i < a.total4; i += (int64_t)gridDim.x * RB) {
    float4 d4 = a.dout[i];
    if (a.dout2) { const float4 e = a.dout2[i]; d4.x += e.x; d4.y += e.y; d4.z += e.z; d4.w += e.w; }
    const float4 r4 = a.raw[i];
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.aout) s4 = a.aout[i];
    const float d[4] = {d4.x, d4.y, d4.z, d4.w};
    const float r[4] = {r4.x, r4.y, r4.z, r4.w};
    const float so[4] = {s4.x, s4.y, s4.z, s4.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = fmaf(r[j], sc[j], sf[j]);
      const bool pos = (a.aout ? so[j] : t) > 0.f;
      const float g = pos ? d[j] : d[j] * sl[j];
      o[j] = (float)(A[j] * ((double)g - G[j]) + B[j] * ((double)r[j] - M[j]));
    }
    a.out[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__global__ __launch_bounds__(RB) void residual_forward_fast_kernel(const float4* raw, PW pw, const float4* skip, PW spw,
                                                                   float slope, float4* out, int c, int64_t total4) {
  const int tid = threadIdx.x;
  float sc[4], sf[4], ksc[4], ksf[4], ksl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = (4 * tid + j) % c;
    sc[j] = pw.scale ? pw.scale[ch] : 1.f; sf[j] = pw.scale ? pw.shift[ch] : 0.f;
    ksc[j] = spw.scale ? spw.scale[ch] : 1.f; ksf[j] = spw.scale ? spw.shift[ch] : 0.f;
    ksl[j] = spw.scale ? spw.slope[ch] : 1.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * RB + tid; i < total4; i += (int64_t)gridDim.x * RB) {
    const float4 r4 = raw[i], k4 = skip[i];
    const float r[4] = {r4.x, r4.y, r4.z, r4.w};
    const float k[4] = {k4.x, k4.y, k4.z, k4.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = fmaf(r[j], sc[j], sf[j]);
      float u = fmaf(k[j], ksc[j], ksf[j]);
      u = u > 0.f ? u : u * ksl[j];
      t += u;
      o[j] = t > 0.f ? t : t * slope;
    }
    out[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// out[i] = sum_b partial[b][i]: one wave per output, lanes stride over the blocks (fixed order).
__global__ __launch_bounds__(64) void sum_partials_wave_kernel(const double* partial, int nblk, int n, double* out) {
  const int i = blockIdx.x;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) t += partial[(int64_t)b * n + i];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) t += __shfl_down(t, s, 64);
  if (threadIdx.x == 0) out[i] = t;
}

// ... rows of `stride` doubles of which the first n are wanted
__global__ __launch_bounds__(64) void sum_partials_strided_kernel(const double* partial, int nblk, int stride, double* out) {
  const int i = blockIdx.x;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) t += partial[(int64_t)b * stride + i];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) t += __shfl_down(t, s, 64);
  if (threadIdx.x == 0) out[i] = t;
}

struct FastPlan { int nblk; int64_t total4, chunk4; };
static inline FastPlan fast_plan(int64_t total_elems) {
  FastPlan f{};
  f.total4 = total_elems / 4;
  int64_t nb = (f.total4 + 2047) / 2048;          // >= 8 float4 per thread
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  int64_t chunk = (f.total4 + nb - 1) / nb;
  chunk = (chunk + RB - 1) / RB * RB;             // block starts stay multiples of 1024 elements
  f.chunk4 = chunk;
  f.nblk = (int)((f.total4 + chunk - 1) / chunk);
  return f;
}

// ---------------------------------------------------------------- batch-norm finalize
__global__ void bn_finalize_kernel(const double* sums, double count, int c, const float* gamma,
                                   const float* beta, float eps, float momentum, float* rm, float* rv,
                                   int64_t* nbt, float* scale, float* shift, double* smean, double* sinv) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch == 0 && nbt) *nbt += 1;
  if (ch >= c) return;
  const double mean = sums[ch] / count;
  double var = sums[c + ch] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[ch] : 1.0, b = beta ? (double)beta[ch] : 0.0;
  scale[ch] = (float)(g * invstd);
  shift[ch] = (float)(b - mean * g * invstd);
  if (smean) smean[ch] = mean;
  if (sinv) sinv[ch] = invstd;
  if (rm) rm[ch] = (float)((1.0 - momentum) * rm[ch] + momentum * mean);
  if (rv) {
    const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rv[ch] = (float)((1.0 - momentum) * rv[ch] + momentum * unb);
  }
}

__global__ void bn_eval_kernel(int c, const float* gamma, const float* beta, const float* rm,
                               const float* rv, float eps, float* scale, float* shift) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
  const float s = g / sqrtf(rv[ch] + eps);
  scale[ch] = s;
  shift[ch] = b - rm[ch] * s;
}

// ---------------------------------------------------------------- activation backward + sums
struct ActBwdArgs {
  ViewD dout, dout2, raw, aout, g;
  PW pw;
  int cbase, CP;
  int64_t pix_per_block;
  double* partial;  // [nblk][3][c]
};

__global__ __launch_bounds__(RB) void act_backward_kernel(ActBwdArgs a) {
  __shared__ double sh[RB];
  const int tid = threadIdx.x;
  const int CP = a.CP;
  const int ch = a.cbase + tid % CP, slot = tid / CP, slots = RB / CP;
  const int c = a.raw.c;
  const int64_t p0 = (int64_t)blockIdx.x * a.pix_per_block;
  int64_t p1 = p0 + a.pix_per_block;
  if (p1 > a.raw.npix) p1 = a.raw.npix;
  double v[3] = {0.0, 0.0, 0.0};
  if (ch < c) {
    float sc = 1.f, sf = 0.f, sl = 1.f;
    if (a.pw.scale) { sc = a.pw.scale[ch]; sf = a.pw.shift[ch]; sl = a.pw.slope[ch]; }
    for (int64_t p = p0 + slot; p < p1; p += slots) {
      float d = a.dout.p[p * a.dout.cs + a.dout.co + ch];
      if (a.dout2.p) d += a.dout2.p[p * a.dout2.cs + a.dout2.co + ch];
      const float r = a.raw.p[p * a.raw.cs + a.raw.co + ch];
      const float t = fmaf(r, sc, sf);
      const float sgn = a.aout.p ? a.aout.p[p * a.aout.cs + a.aout.co + ch] : t;
      const bool pos = sgn > 0.f;
      const float g = pos ? d : d * sl;
      if (a.g.p) a.g.p[p * a.g.cs + a.g.co + ch] = g;
      v[0] += g;
      v[1] += (double)g * r;
      if (!pos) v[2] += (double)d * t;
    }
  }
  for (int s = 0; s < 3; ++s) {
    __syncthreads();
    sh[tid] = v[s];
    __syncthreads();
    if (slot == 0 && ch < c) {
      double t = 0.0;
      for (int k = 0; k < slots; ++k) t += sh[k * CP + tid % CP];
      a.partial[((int64_t)blockIdx.x * 3 + s) * c + ch] = t;
    }
  }
}

__global__ void bn_backward_finalize_kernel(const double* sums, double count, int c, const float* gamma,
                                            const double* smean, const double* sinv, float pscale,
                                            float* dgamma, float* dbeta, double* coef) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const double S0 = sums[ch], S1 = sums[c + ch];
  const double mean = smean[ch], inv = sinv[ch], g = gamma ? (double)gamma[ch] : 1.0;
  const double dg = inv * (S1 - mean * S0);
  if (dgamma) dgamma[ch] = (float)(dg * pscale);
  if (dbeta) dbeta[ch] = (float)(S0 * pscale);
  // d_raw = g*inv*((gr - S0/n) - xhat*dg/n),  xhat = (raw-mean)*inv
  //       = A*(gr - mg) + B*(raw - mean)     evaluated in double by the apply kernel (the reference's
  //         CPU batch_norm_backward also runs this elementwise step in its double accumulate type)
  coef[ch] = g * inv;
  coef[c + ch] = S0 / count;
  coef[2 * c + ch] = -g * inv * inv * dg / count;
  coef[3 * c + ch] = mean;
}

__global__ __launch_bounds__(RB) void bn_backward_apply_kernel(ViewD g, ViewD raw, const double* abc, ViewD out,
                                                               int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = g.c;
  const int ch = i % c;
  const int64_t p = i / c;
  const float gv = g.p[p * g.cs + g.co + ch];
  const float r = raw.p[p * raw.cs + raw.co + ch];
  out.p[p * out.cs + out.co + ch] =
      (float)(abc[ch] * ((double)gv - abc[c + ch]) + abc[2 * c + ch] * ((double)r - abc[3 * c + ch]));
}

__global__ __launch_bounds__(RB) void act_bn_backward_apply_kernel(ViewD dout, ViewD dout2, ViewD raw, ViewD aout, PW pw,
                                                                   const double* abc, ViewD out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = raw.c;
  const int ch = i % c;
  const int64_t p = i / c;
  float d = dout.p[p * dout.cs + dout.co + ch];
  if (dout2.p) d += dout2.p[p * dout2.cs + dout2.co + ch];
  const float r = raw.p[p * raw.cs + raw.co + ch];
  float sc = 1.f, sf = 0.f, sl = 1.f;
  if (pw.scale) { sc = pw.scale[ch]; sf = pw.shift[ch]; sl = pw.slope[ch]; }
  const float t = fmaf(r, sc, sf);
  const float sgn = aout.p ? aout.p[p * aout.cs + aout.co + ch] : t;
  const float g = sgn > 0.f ? d : d * sl;
  out.p[p * out.cs + out.co + ch] =
      (float)(abc[ch] * ((double)g - abc[c + ch]) + abc[2 * c + ch] * ((double)r - abc[3 * c + ch]));
}

__global__ void prelu_slope_grad_kernel(const double* sums, int c, float* dslope) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int ch = 0; ch < c; ++ch) t += sums[2 * c + ch];
    *dslope = (float)t;
  }
}

__global__ void sums_to_float_kernel(const double* sums, int c, float* dst) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < c) dst[ch] = (float)sums[ch];
}

// ---------------------------------------------------------------- residual tail
__global__ __launch_bounds__(RB) void residual_forward_kernel(ViewD raw, PW pw, ViewD skip, PW spw, float slope,
                                                              ViewD out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = raw.c;
  const int ch = i % c;
  const int64_t p = i / c;
  float t = raw.p[p * raw.cs + raw.co + ch];
  if (pw.scale) t = fmaf(t, pw.scale[ch], pw.shift[ch]);
  t += pw_apply(spw, ch, skip.p[p * skip.cs + skip.co + ch]);
  out.p[p * out.cs + out.co + ch] = t > 0.f ? t : t * slope;
}

// ---------------------------------------------------------------- layout glue
__global__ __launch_bounds__(RB) void nchw_to_view_kernel(const float* src, int c, const float* aux, int caux,
                                                          ViewD out, int n, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int ct = c + caux;
  const int ch = i % ct;
  const int64_t p = i / ct;
  const int64_t hw = (int64_t)out.h * out.w;
  const int64_t nn = p / hw, yx = p % hw;
  float v;
  if (ch < c) v = src[(nn * c + ch) * hw + yx];
  else v = aux[nn * caux + (ch - c)];
  out.p[p * out.cs + out.co + ch] = v;
}

// ... four consecutive pixels per thread (h*w a multiple of 4, source 16-byte aligned): one 16-byte load per channel,
// 32-bit index arithmetic; CT = c + caux <= 4 known at compile time.  (The element-per-thread kernel above spends its
// time in 64-bit divisions: 94 us for the 64 x 512^2 y + aux image.)
template <int CT>
__global__ __launch_bounds__(RB) void nchw_to_view4_kernel(const float* src, int c, const float* aux, ViewD out, int hw4,
                                                           int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total4) return;
  const int nn = (int)(i / hw4), q = (int)(i - (int64_t)nn * hw4);
  float v[CT][4];
#pragma unroll
  for (int ch = 0; ch < CT; ++ch) {
    if (ch < c) {
      const float4 t = *reinterpret_cast<const float4*>(src + ((int64_t)nn * c + ch) * hw4 * 4 + 4 * q);
      v[ch][0] = t.x; v[ch][1] = t.y; v[ch][2] = t.z; v[ch][3] = t.w;
    } else {
      const float t = aux[nn * (CT - c) + (ch - c)];
      v[ch][0] = t; v[ch][1] = t; v[ch][2] = t; v[ch][3] = t;
    }
  }
  float* o = out.p + ((int64_t)nn * hw4 * 4 + 4 * q) * out.cs + out.co;
  if (CT == 2 && out.cs == 2 && out.co == 0) {          // dense two-channel image: 32 contiguous bytes
    reinterpret_cast<float4*>(o)[0] = make_float4(v[0][0], v[1][0], v[0][1], v[1][1]);
    reinterpret_cast<float4*>(o)[1] = make_float4(v[0][2], v[1][2], v[0][3], v[1][3]);
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int ch = 0; ch < CT; ++ch) o[p * out.cs + ch] = v[ch][p];
  }
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float softplus_grad_f(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(RB) void view_to_nchw_kernel(ViewD src, PW pw, int softplus, float* dst, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = src.c;
  const int64_t hw = (int64_t)src.h * src.w;
  // i indexes the NCHW destination (coalesced writes)
  const int64_t yx = i % hw;
  const int ch = (i / hw) % c;
  const int64_t nn = i / (hw * c);
  const float v = pw_apply(pw, ch, src.p[(nn * hw + yx) * src.cs + src.co + ch]);
  dst[i] = softplus ? softplus_f(v) : v;
}

__global__ __launch_bounds__(RB) void fill_kernel(float* dst, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i < n) dst[i] = v;
}

// ---------------------------------------------------------------- latent sampler + KL
struct LatentArgs {
  bp_latent lt;
  ViewD q, p, z;
  PW qpw, ppw;
  const float* eps;
  float* stats4;
  double* partial;
  int64_t nelem;  // N*zc*zh*zw
};

__global__ __launch_bounds__(RB) void latent_forward_kernel(LatentArgs a) {
  __shared__ double sh[RB];
  const int zc = a.lt.zc, zh = a.lt.zh, zw = a.lt.zw;
  const int64_t hw = (int64_t)zh * zw;
  double kl = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < a.nelem; i += (int64_t)gridDim.x * RB) {
    // i indexes (n, c, h, w)
    const int64_t yx = i % hw;
    const int c = (i / hw) % zc;
    const int64_t n = i / (hw * zc);
    const int64_t pix = n * hw + yx;
    const float mu = pw_apply(a.qpw, c, a.q.p[pix * a.q.cs + a.q.co + c]);
    const float lv = pw_apply(a.qpw, zc + c, a.q.p[pix * a.q.cs + a.q.co + zc + c]);
    float pm = 0.f, plv = 0.f;
    if (a.p.p) {
      pm = pw_apply(a.ppw, c, a.p.p[pix * a.p.cs + a.p.co + c]);
      plv = pw_apply(a.ppw, zc + c, a.p.p[pix * a.p.cs + a.p.co + zc + c]);
    }
    a.stats4[i] = mu;
    a.stats4[a.nelem + i] = lv;
    a.stats4[2 * a.nelem + i] = pm;
    a.stats4[3 * a.nelem + i] = plv;
    const float pvar = expf(plv);
    const float dm = pm - mu;
    kl += (double)(dm * dm / pvar + expf(lv) / pvar + plv - lv - 1.f);
    const float sd = expf(lv * 0.5f) + a.lt.min_z_var;
    for (int l = 0; l < a.lt.L; ++l) {
      const float e = a.eps[(int64_t)l * a.nelem + i];
      const int64_t zp = ((int64_t)l * a.lt.n + n) * hw + yx;
      a.z.p[zp * a.z.cs + a.z.co + c] = fmaf(e, sd, mu);
    }
  }
  sh[threadIdx.x] = kl;
  __syncthreads();
  for (int s = RB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.partial[blockIdx.x] = sh[0];
}

struct LatentBwdArgs {
  bp_latent lt;
  ViewD dz, dq, dp;
  const float* stats4; const float* eps; const float* seed;
  float beta_kl;
  int64_t nelem;
};

__global__ __launch_bounds__(RB) void latent_backward_kernel(LatentBwdArgs a) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= a.nelem) return;
  const int zc = a.lt.zc;
  const int64_t hw = (int64_t)a.lt.zh * a.lt.zw;
  const int64_t yx = i % hw;
  const int c = (i / hw) % zc;
  const int64_t n = i / (hw * zc);
  const int64_t pix = n * hw + yx;
  const float mu = a.stats4[i], lv = a.stats4[a.nelem + i];
  const float pm = a.stats4[2 * a.nelem + i], plv = a.stats4[3 * a.nelem + i];
  const float sd = expf(lv * 0.5f);
  float dmu = 0.f, dlv = 0.f;
  for (int l = 0; l < a.lt.L; ++l) {
    const int64_t zp = ((int64_t)l * a.lt.n + n) * hw + yx;
    const float d = a.dz.p[zp * a.dz.cs + a.dz.co + c];
    dmu += d;
    dlv += d * a.eps[(int64_t)l * a.nelem + i];
  }
  dlv *= 0.5f * sd;
  const float k = -(*a.seed) * a.beta_kl * 0.5f / (float)a.lt.n;
  const float pvar = expf(plv), dm = pm - mu, ev = expf(lv);
  dmu += k * (-2.f * dm / pvar);
  dlv += k * (ev / pvar - 1.f);
  a.dq.p[pix * a.dq.cs + a.dq.co + c] = dmu;
  a.dq.p[pix * a.dq.cs + a.dq.co + zc + c] = dlv;
  if (a.dp.p) {
    a.dp.p[pix * a.dp.cs + a.dp.co + c] = k * (2.f * dm / pvar);
    a.dp.p[pix * a.dp.cs + a.dp.co + zc + c] = k * (-dm * dm / pvar - ev / pvar + 1.f);
  }
}

// ---------------------------------------------------------------- Gaussian log-likelihood head

// pixel index of the (L*M, H, W) grid -> (lm, yx, m): 64-bit divisions by run-time values cost ~100 instructions each and
// were what bounded these two kernels (0.12 / 0.10 ms for 0.2 GB); H*W is a power of two for every tile size in use and
// L = 1 in the fiducial model, otherwise 32-bit divisions (the grid has < 2^31 pixels: checked by the entry points)
struct PixSplit {
  unsigned hw, n; int shift; bool one_l;
  __device__ __forceinline__ void operator()(unsigned p, unsigned& lm, unsigned& yx, unsigned& m) const {
    if (shift >= 0) { lm = p >> shift; yx = p & (hw - 1u); }
    else { lm = p / hw; yx = p - lm * hw; }
    m = one_l ? lm : lm % n;
  }
};
__device__ __forceinline__ PixSplit pix_split(const bp_loglik& ll) {
  PixSplit s;
  s.hw = (unsigned)ll.h * (unsigned)ll.w; s.n = (unsigned)ll.n;
  s.shift = (s.hw & (s.hw - 1u)) == 0u ? __ffs((int)s.hw) - 1 : -1;
  s.one_l = ll.L == 1;
  return s;
}

struct LoglikArgs {
  bp_loglik ll;
  const float* x;
  ViewD mu, var;
  float* x_mu; float* x_lv;
  double* partial;  // [nblk][2][c]
  int CP;
  int64_t pix_per_block, npix;  // pixels of the (L*M,H,W) grid
};

__global__ __launch_bounds__(RB) void loglik_forward_kernel(LoglikArgs a) {
  __shared__ double sh[RB];
  const int tid = threadIdx.x;
  const int CP = a.CP, c = a.ll.c;
  const int ch = tid % CP, slot = tid / CP, slots = RB / CP;
  const int64_t hw = (int64_t)a.ll.h * a.ll.w;
  const PixSplit split = pix_split(a.ll);
  const int64_t p0 = (int64_t)blockIdx.x * a.pix_per_block;
  int64_t p1 = p0 + a.pix_per_block;
  if (p1 > a.npix) p1 = a.npix;
  double v[2] = {0.0, 0.0};
  if (ch < c) {
    for (int64_t p = p0 + slot; p < p1; p += slots) {
      unsigned lm32, yx32, m32;
      split((unsigned)p, lm32, yx32, m32);
      const int64_t lm = lm32, yx = yx32, m = m32;
      const float raw = a.mu.p[p * a.mu.cs + a.mu.co + ch];
      const float xm = a.ll.mu_softplus ? softplus_f(raw) : raw;
      a.x_mu[(lm * c + ch) * hw + yx] = xm;
      const float d = a.x[(m * c + ch) * hw + yx] - xm;
      v[0] += (double)(-0.5f * d * d);
      if (a.ll.predict_var) {
        const float lv = a.var.p[p * a.var.cs + a.var.co + ch];
        if (a.x_lv) a.x_lv[(lm * c + ch) * hw + yx] = lv;
        v[1] += (double)(-0.5f * lv - 0.5f * d * d / expf(lv));
      }
    }
  }
  for (int s = 0; s < 2; ++s) {
    __syncthreads();
    sh[tid] = v[s];
    __syncthreads();
    if (slot == 0 && ch < c) {
      double t = 0.0;
      for (int k = 0; k < slots; ++k) t += sh[k * CP + ch];
      a.partial[((int64_t)blockIdx.x * 2 + s) * c + ch] = t;
    }
  }
}

// One wave: the partial rows are summed by 64 lanes (lane l takes rows l, l + 64, ...) and a fixed-order shuffle
// tree -- one thread walking 2 x 1024 rows was 87 us of dependent loads between the forward and the backward pass.
__global__ __launch_bounds__(64) void loglik_finalize_kernel(bp_loglik ll, const double* partial, int nblk,
                                                             const double* kl_sum, float* stats) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  const int c = ll.c;
  const double norm = (double)ll.n * ll.L;
  const double c0 = -0.5 * log(2.0 * M_PI);
  double total = 0.0;
  for (int ch = 0; ch < c; ++ch) {
    double sf = 0.0, sv = 0.0;
    for (int b = lane; b < nblk; b += 64) {
      sf += partial[((int64_t)b * 2 + 0) * c + ch];
      sv += partial[((int64_t)b * 2 + 1) * c + ch];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { sf += __shfl_down(sf, s, 64); sv += __shfl_down(sv, s, 64); }
    if (lane == 0) {
      const double fixed = c0 + sf / norm;
      const double freev = c0 + sv / norm;
      const double llk = ll.predict_var ? (1.0 - ll.alpha_var) * fixed + ll.alpha_var * freev : fixed;
      stats[2 + ch] = (float)llk;
      stats[2 + c + ch] = (float)fixed;
      stats[2 + 2 * c + ch] = (float)(ll.predict_var ? freev : 0.0);
      total += llk;
    }
  }
  if (lane == 0) {
    const double kl = kl_sum ? 0.5 / (double)ll.n * (*kl_sum) : 0.0;
    stats[1] = (float)kl;
    stats[0] = (float)(-kl * ll.beta_kl + ll.likelihood_scaling * total);
  }
}

struct LoglikBwdArgs {
  bp_loglik ll;
  const float* x; const float* seed;
  ViewD mu, var, dmu, dvar;
  int64_t total;
};

__global__ __launch_bounds__(RB) void loglik_backward_kernel(LoglikBwdArgs a) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= a.total) return;
  const int c = a.ll.c;
  const int ch = c == 1 ? 0 : (int)(i % c);
  const int64_t p = c == 1 ? i : i / c;
  const int64_t hw = (int64_t)a.ll.h * a.ll.w;
  unsigned lm32, yx32, m32;
  pix_split(a.ll)((unsigned)p, lm32, yx32, m32);
  const int64_t yx = yx32, m = m32;
  const float raw = a.mu.p[p * a.mu.cs + a.mu.co + ch];
  const float xm = a.ll.mu_softplus ? softplus_f(raw) : raw;
  const float dact = a.ll.mu_softplus ? softplus_grad_f(raw) : 1.f;
  const float d = a.x[(m * c + ch) * hw + yx] - xm;
  const float s = (*a.seed) * a.ll.likelihood_scaling / ((float)a.ll.n * (float)a.ll.L);
  if (a.ll.predict_var) {
    const float lv = a.var.p[p * a.var.cs + a.var.co + ch];
    const float xv = expf(lv);
    const float al = a.ll.alpha_var;
    a.dmu.p[p * a.dmu.cs + a.dmu.co + ch] = s * ((1.f - al) * d + al * d / xv) * dact;
    a.dvar.p[p * a.dvar.cs + a.dvar.co + ch] = s * al * (-0.5f + 0.5f * d * d / xv);
  } else {
    a.dmu.p[p * a.dmu.cs + a.dmu.co + ch] = s * d * dact;
  }
}

// ---------------------------------------------------------------- GAN heads (CGAN rows g1-g5 of SURVEY.md 8a)
// out = tanh(act(in)) or sigmoid(act(in)), view -> view (e.g. the generator's Tanh output written
// straight into the pressure channel of the discriminator's input concatenation).
__global__ __launch_bounds__(RB) void unary_forward_kernel(ViewD in, PW pw, int kind, ViewD out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = in.c;
  const int ch = i % c;
  const int64_t p = i / c;
  const float v = pw_apply(pw, ch, in.p[p * in.cs + in.co + ch]);
  out.p[p * out.cs + out.co + ch] = kind == 1 ? tanhf(v) : (kind == 2 ? 1.f / (1.f + expf(-v)) : v);
}

// Binary cross entropy on logits: target 1 -> softplus(-x), target 0 -> softplus(x); partial sums.
__global__ __launch_bounds__(RB) void bce_logits_kernel(ViewD raw, int n0, int n1, float target, double* partial) {
  __shared__ double sh[RB];
  const int64_t per = (int64_t)raw.h * raw.w * raw.c;
  const int64_t total = (int64_t)(n1 - n0) * per;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < total; i += (int64_t)gridDim.x * RB) {
    const int ch = i % raw.c;
    const int64_t p = (int64_t)n0 * raw.h * raw.w + i / raw.c;
    const float x = raw.p[p * raw.cs + raw.co + ch];
    const float z = target > 0.5f ? -x : x;
    acc += (double)(z > 20.f ? z : log1pf(expf(z)));
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = RB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// d/dx of scale * BCE: scale * (sigmoid(x) - target), samples [n0, n1) of the view.
__global__ __launch_bounds__(RB) void bce_logits_grad_kernel(ViewD raw, int n0, int n1, float target, float scale,
                                                             ViewD d) {
  const int64_t per = (int64_t)raw.h * raw.w * raw.c;
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= (int64_t)(n1 - n0) * per) return;
  const int ch = i % raw.c;
  const int64_t p = (int64_t)n0 * raw.h * raw.w + i / raw.c;
  const float x = raw.p[p * raw.cs + raw.co + ch];
  d.p[p * d.cs + d.co + ch] = scale * (1.f / (1.f + expf(-x)) - target);
}

// L1 distance between fake (a view) and the real field (NCHW): partial sums of |fake - x|.
__global__ __launch_bounds__(RB) void l1_kernel(ViewD fake, const float* x, double* partial) {
  __shared__ double sh[RB];
  const int c = fake.c;
  const int64_t hw = (int64_t)fake.h * fake.w, total = fake.npix * c;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < total; i += (int64_t)gridDim.x * RB) {
    const int ch = i % c;
    const int64_t p = i / c, n = p / hw, yx = p % hw;
    acc += fabs((double)fake.p[p * fake.cs + fake.co + ch] - (double)x[(n * c + ch) * hw + yx]);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = RB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// d_graw = (d_fake + l1_scale * sign(fake - x)) * (1 - fake^2)     (through the Tanh output layer)
__global__ __launch_bounds__(RB) void tanh_l1_backward_kernel(ViewD fake, const float* x, ViewD dfake, float l1_scale,
                                                              ViewD d, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = fake.c;
  const int ch = i % c;
  const int64_t hw = (int64_t)fake.h * fake.w;
  const int64_t p = i / c, n = p / hw, yx = p % hw;
  const float f = fake.p[p * fake.cs + fake.co + ch];
  const float diff = f - x[(n * c + ch) * hw + yx];
  float g = dfake.p ? dfake.p[p * dfake.cs + dfake.co + ch] : 0.f;
  g += l1_scale * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
  d.p[p * d.cs + d.co + ch] = g * (1.f - f * f);
}

// ---------------------------------------------------------------- device-side batch assembly
// One descriptor per (sample, slab): where the tile starts in the HBM-resident stack and how the
// dihedral tile permutation maps output (r,c) to source (row,col):  row = r0 + rr*r + rc*c, ...
struct TileDesc {
  const float* base;      // &stack[slice][tile_y*t][tile_x*t]
  int32_t pitch;          // n_grid
  int32_t r0, rr, rc, c0, cr, cc;
  int32_t pad_;
};
struct SampleXform {      // x -> log(scale*x * inv_sigma + 1) * inv_k   (mode 1), or scale*x (mode 0)
  double scale, inv_sigma, inv_k;
  int32_t mode, pad_;
};

__global__ __launch_bounds__(RB) void gather_tiles_kernel(const TileDesc* d100, const TileDesc* d150,
                                                          const SampleXform* xf, float* out, int t, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  const int c = i % t;
  const int r = (i / t) % t;
  const int n = i / ((int64_t)t * t);
  const TileDesc a = d100[n], b = d150[n];
  const float va = a.base[(int64_t)(a.r0 + a.rr * r + a.rc * c) * a.pitch + (a.c0 + a.cr * r + a.cc * c)];
  const float vb = b.base[(int64_t)(b.r0 + b.rr * r + b.rc * c) * b.pitch + (b.c0 + b.cr * r + b.cc * c)];
  const float s = va + vb;                                   // float32 add, like the host path
  const SampleXform x = xf[n];
  // (python float * float32 array = a float32 multiply with the scalar rounded to float32, datasets.py:399)
  double v = x.scale == 1.0 ? (double)s : (double)((float)x.scale * s);
  if (x.mode == 1) v = log(v * x.inv_sigma + 1.0) * x.inv_k;
  out[i] = (float)v;
}

// ---------------------------------------------------------------- Adam
__global__ __launch_bounds__(RB) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                  float lr, float b1, float b2, float eps, float bc1,
                                                  float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;      // exp_avg.lerp_(grad, 1-beta1)
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;  // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

struct RedPlan { int CP, npass, nblk; int64_t ppb; };
static inline RedPlan red_plan(int c, int64_t npix) {
  RedPlan r{};
  r.CP = h_next_pow2(c < RB ? c : RB);
  r.npass = bp_ceil_div(c, r.CP);
  // (256 pixels per block at least: with 2048 the 16 x 16 latent-level tensors of the recognition / prior networks ran
  //  on 8 workgroups, 256 dependent trips per thread: 0.14 ms for a 2 MB tensor)
  int64_t nb = (npix + 255) / 256;
  if (nb > MAX_RBLOCKS) nb = MAX_RBLOCKS;
  if (nb < 1) nb = 1;
  r.ppb = (npix + nb - 1) / nb;
  r.nblk = (int)((npix + r.ppb - 1) / r.ppb);
  return r;
}
static inline unsigned nblocks(int64_t total) { return (unsigned)((total + RB - 1) / RB); }

__global__ __launch_bounds__(RB) void adam_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                      const float* hyper) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= n) return;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2_sqrt = hyper[5];
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace

// sum_partials_wave_kernel for the two columns {c, C + c} of a channel (same order of additions: bitwise the same sums)
// followed by bn_finalize_kernel's arithmetic for that channel.  Data parallel (pd.world > 0, peer_dev.hpp): the channel's
// two sums are exchanged with the other ranks in between, so that the statistics are those of the global batch.
__global__ __launch_bounds__(128) void sum_partials_bn_kernel(const double* partial, int nblk, int c, double* out, BnFin f,
                                                              PeerDev pd) {
  __shared__ double sh[2];
  const int ch = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = w * c + ch;
  double t = 0.0;
  for (int b = lane; b < nblk; b += 64) t += partial[(int64_t)b * 2 * c + col];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) t += __shfl_down(t, s, 64);
  if (lane == 0) { out[col] = t; sh[w] = t; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (pd.world > 0) {
    double v[2] = {sh[0], sh[1]};
    peer_exchange<2>(pd, ch, c, v);
    sh[0] = v[0]; sh[1] = v[1];
    out[ch] = v[0]; out[c + ch] = v[1];
  }
  if (ch == 0 && f.nbt) *f.nbt += 1;
  const double mean = sh[0] / f.count;
  double var = sh[1] / f.count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)f.eps);
  const double g = f.gamma ? (double)f.gamma[ch] : 1.0, b = f.beta ? (double)f.beta[ch] : 0.0;
  f.scale[ch] = (float)(g * invstd);
  f.shift[ch] = (float)(b - mean * g * invstd);
  if (f.smean) f.smean[ch] = mean;
  if (f.sinv) f.sinv[ch] = invstd;
  if (f.rm) f.rm[ch] = (float)((1.0 - f.momentum) * f.rm[ch] + f.momentum * mean);
  if (f.rv) {
    const double unb = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
    f.rv[ch] = (float)((1.0 - f.momentum) * f.rv[ch] + f.momentum * unb);
  }
}

int bp_sum_partials_req(const double* partial, int nblk, int n, const IgemmStatsReq* sr, hipStream_t st) {
  if (!sr->fin) return bp_sum_partials(partial, nblk, n, sr->sums, st);
  PeerDev pd;
  if (bp_peer_next(&pd) && n > PC_MAXN) return BP_EUNSUPPORTED;        // (a bound communicator: the sums become global)
  hipLaunchKernelGGL(sum_partials_bn_kernel, dim3(n / 2), dim3(128), 0, st, partial, nblk, n / 2, sr->sums, *sr->fin, pd);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_sum_partials_strided(const double* partial, int nblk, int stride, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_partials_strided_kernel, dim3(n), dim3(64), 0, st, partial, nblk, stride, out);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// bp_act_backward_bn: the last stage of the activation backward's three sums with the batch-norm backward finalize of
// the same layer folded in.  One workgroup per channel, wave q sums column q*c + ch exactly as sum_partials_wave_kernel
// does (bitwise the same sums), then one thread evaluates bn_backward_finalize_kernel's formulas for the channel.
struct BnBwdFin {
  double count;
  const float* gamma; const double* smean; const double* sinv;
  float pscale;
  float* dgamma; float* dbeta; double* coef;
};
__global__ __launch_bounds__(192) void sum_partials_bnbwd_kernel(const double* partial, int nblk, int c, double* sums, BnBwdFin f,
                                                                 PeerDev pd) {
  __shared__ double sh[3];
  const int ch = blockIdx.x, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = q * c + ch;
  double t = 0.0;
  for (int b = lane; b < nblk; b += 64) t += partial[(int64_t)b * 3 * c + i];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) t += __shfl_down(t, s, 64);
  if (lane == 0) { sums[i] = t; sh[q] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (pd.world > 0) {        // data parallel: {sum g, sum g * raw} of the global batch (the third sum stays local)
      double v[2] = {sh[0], sh[1]};
      peer_exchange<2>(pd, ch, c, v);
      sh[0] = v[0]; sh[1] = v[1];
      sums[ch] = v[0]; sums[c + ch] = v[1];
    }
    const double S0 = sh[0], S1 = sh[1];
    const double mean = f.smean[ch], inv = f.sinv[ch], g = f.gamma ? (double)f.gamma[ch] : 1.0;
    const double dg = inv * (S1 - mean * S0);
    if (f.dgamma) f.dgamma[ch] = (float)(dg * f.pscale);
    if (f.dbeta) f.dbeta[ch] = (float)(S0 * f.pscale);
    f.coef[ch] = g * inv;
    f.coef[c + ch] = S0 / f.count;
    f.coef[2 * c + ch] = -g * inv * inv * dg / f.count;
    f.coef[3 * c + ch] = mean;
  }
}
static thread_local const BnBwdFin* t_bnbwd = nullptr;       // set by bp_act_backward_bn around bp_act_backward

// the three sums of an activation backward, partial[nblk][3c] -> sums[3c] (+ the pending finalize)
int bp_sum_partials3(const double* partial, int nblk, int c, double* sums, hipStream_t st) {
  if (t_bnbwd) {
    PeerDev pd;
    if (bp_peer_next(&pd) && 2 * c > PC_MAXN) return BP_EUNSUPPORTED;
    hipLaunchKernelGGL(sum_partials_bnbwd_kernel, dim3(c), dim3(192), 0, st, partial, nblk, c, sums, *t_bnbwd, pd);
  } else hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(3 * c), dim3(64), 0, st, partial, nblk, 3 * c, sums);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// out[i] = sum over blocks of partial[b][i], fixed order (shared with pointwise_bf16.hip)
int bp_sum_partials(const double* partial, int nblk, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(n), dim3(64), 0, st, partial, nblk, n, out);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// pointwise_bf16.hip: the same passes over dense bf16 views
bool bp_bf16_dense_ok(const bp_view* v);
size_t bp_bf16_reduce_workspace(const bp_view* x, int nsums);
int bp_bf16_channel_sums(const bp_view* x, double* sums, void* workspace, hipStream_t st);
int bp_bf16_act_backward(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const PW& pw,
                         const bp_view* act_out, const bp_view* g, double* sums, void* workspace, hipStream_t st);
int bp_bf16_bn_backward_apply(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const PW& pw,
                              const bp_view* act_out, const double* abc, const bp_view* out, bool recompute_g,
                              hipStream_t st);
int bp_bf16_residual_forward(const bp_view* raw, const PW& pw, const bp_view* skip, const PW& spw, float slope,
                             const bp_view* out, hipStream_t st);

// A call whose views are bf16: every view must be a dense bf16 view of the same grid (the layouts the launch
// plan produces for the generator trunk); anything else has no bf16 form.
static bool is_bf16(const bp_view* v) { return v && v->dtype == BP_BF16; }
static bool bf16_set_ok(const bp_view* ref, const bp_view* const* vs, int n) {
  if (!bp_view_ok_any(ref) || !bp_bf16_dense_ok(ref)) return false;
  for (int i = 0; i < n; ++i) {
    const bp_view* v = vs[i];
    if (!v) continue;
    if (!bp_view_ok_any(v) || !bp_bf16_dense_ok(v) || v->n != ref->n || v->h != ref->h || v->w != ref->w || v->c != ref->c)
      return false;
  }
  return true;
}

extern "C" {

size_t bp_channel_sums_workspace(const bp_view* x) {
  if (is_bf16(x)) return bp_view_ok_any(x) && bp_bf16_dense_ok(x) ? bp_bf16_reduce_workspace(x, 2) : 0;
  if (!bp_view_ok(x)) return 0;
  const RedPlan r = red_plan(x->c, bp_view_pixels(x));
  const FastPlan f = fast_plan(bp_view_pixels(x) * x->c);
  const int nb = r.nblk > f.nblk ? r.nblk : f.nblk;
  return (size_t)nb * 2 * x->c * sizeof(double);
}

int bp_channel_sums(const bp_view* x, double* sums, void* workspace, size_t workspace_bytes, void* stream) {
  if (is_bf16(x)) {
    if (!sums) return BP_EINVAL;
    if (!bf16_set_ok(x, nullptr, 0)) return BP_EUNSUPPORTED;
    if (!workspace || workspace_bytes < bp_bf16_reduce_workspace(x, 2)) return BP_EWORKSPACE;
    return bp_bf16_channel_sums(x, sums, workspace, bp_stream(stream));
  }
  if (!bp_view_ok(x) || !sums) return BP_EINVAL;
  if (!workspace || workspace_bytes < bp_channel_sums_workspace(x)) return BP_EWORKSPACE;
  const RedPlan r = red_plan(x->c, bp_view_pixels(x));
  hipStream_t st = bp_stream(stream);
  double* partial = reinterpret_cast<double*>(workspace);
  if (dense_ok(x)) {
    const FastPlan f = fast_plan(bp_view_pixels(x) * x->c);
    hipLaunchKernelGGL(channel_sums_fast_kernel, dim3(f.nblk), dim3(RB), 0, st,
                       reinterpret_cast<const float4*>(x->ptr), x->c, f.total4, f.chunk4, partial);
    BP_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(2 * x->c), dim3(64), 0, st, partial, f.nblk, 2 * x->c, sums);
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  for (int pass = 0; pass < r.npass; ++pass) {
    hipLaunchKernelGGL(channel_sums_kernel, dim3(r.nblk), dim3(RB), 0, st, vd(x), pass * r.CP, r.CP, r.ppb, partial);
    BP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(2 * x->c), dim3(64), 0, st, partial, r.nblk, 2 * x->c, sums);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bn_finalize(const double* sums, double count, int32_t c, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                   float* scale, float* shift, double* save_mean, double* save_invstd, void* stream) {
  if (!sums || c <= 0 || count <= 0 || !scale || !shift) return BP_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(bp_ceil_div(c, 64)), dim3(64), 0, bp_stream(stream), sums, count, c,
                     gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, scale, shift,
                     save_mean, save_invstd);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bn_eval_pointwise(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift, void* stream) {
  if (c <= 0 || !running_mean || !running_var || !scale || !shift) return BP_EINVAL;
  hipLaunchKernelGGL(bn_eval_kernel, dim3(bp_ceil_div(c, 64)), dim3(64), 0, bp_stream(stream), c, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

size_t bp_act_backward_workspace(const bp_view* raw) {
  if (is_bf16(raw)) return bp_view_ok_any(raw) && bp_bf16_dense_ok(raw) ? bp_bf16_reduce_workspace(raw, 3) : 0;
  if (!bp_view_ok(raw)) return 0;
  const RedPlan r = red_plan(raw->c, bp_view_pixels(raw));
  const FastPlan f = fast_plan(bp_view_pixels(raw) * raw->c);
  const int nb = r.nblk > f.nblk ? r.nblk : f.nblk;
  return (size_t)nb * 3 * raw->c * sizeof(double);
}

static bool same_grid(const bp_view* a, const bp_view* b) {
  return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c;
}

int bp_act_backward(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const bp_pointwise* pw,
                    const bp_view* act_out, const bp_view* g, double* sums, void* workspace,
                    size_t workspace_bytes, void* stream) {
  if (is_bf16(raw) || is_bf16(dout) || is_bf16(dout2) || is_bf16(act_out) || is_bf16(g)) {
    if (!sums || !dout) return BP_EINVAL;
    const bp_view* vs[4] = {dout, dout2, act_out, g};
    if (!bf16_set_ok(raw, vs, 4)) return BP_EUNSUPPORTED;
    if (!workspace || workspace_bytes < bp_bf16_reduce_workspace(raw, 3)) return BP_EWORKSPACE;
    return bp_bf16_act_backward(dout, dout2, raw, bp_pw(pw), act_out, g, sums, workspace, bp_stream(stream));
  }
  if (!bp_view_ok(dout) || !bp_view_ok(raw) || (g && !bp_view_ok(g)) || !sums) return BP_EINVAL;
  if (!same_grid(dout, raw) || (g && !same_grid(g, raw))) return BP_EINVAL;
  if (dout2 && (!bp_view_ok(dout2) || !same_grid(dout2, raw))) return BP_EINVAL;
  if (act_out && (!bp_view_ok(act_out) || !same_grid(act_out, raw))) return BP_EINVAL;
  if (!workspace || workspace_bytes < bp_act_backward_workspace(raw)) return BP_EWORKSPACE;
  const RedPlan r = red_plan(raw->c, bp_view_pixels(raw));
  hipStream_t st = bp_stream(stream);
  if (dense_ok(raw) && dense_ok(dout) && (!g || dense_ok(g)) && (!dout2 || dense_ok(dout2)) && (!act_out || dense_ok(act_out))) {
    const FastPlan f = fast_plan(bp_view_pixels(raw) * raw->c);
    ActBwdFast fa{};
    fa.dout = reinterpret_cast<const float4*>(dout->ptr);
    fa.dout2 = dout2 ? reinterpret_cast<const float4*>(dout2->ptr) : nullptr;
    fa.raw = reinterpret_cast<const float4*>(raw->ptr);
    fa.aout = act_out ? reinterpret_cast<const float4*>(act_out->ptr) : nullptr;
    fa.g = g ? reinterpret_cast<float4*>(g->ptr) : nullptr;
    fa.pw = bp_pw(pw); fa.c = raw->c; fa.total4 = f.total4; fa.chunk4 = f.chunk4;
    fa.partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(act_backward_fast_kernel, dim3(f.nblk), dim3(RB), 0, st, fa);
    BP_CHECK_LAUNCH();
    return bp_sum_partials3(fa.partial, f.nblk, raw->c, sums, st);
  }
  ActBwdArgs a{};
  a.dout = vd(dout); a.dout2 = vd(dout2); a.raw = vd(raw); a.aout = vd(act_out); a.g = vd(g);
  a.pw = bp_pw(pw); a.CP = r.CP; a.pix_per_block = r.ppb; a.partial = reinterpret_cast<double*>(workspace);
  for (int pass = 0; pass < r.npass; ++pass) {
    a.cbase = pass * r.CP;
    hipLaunchKernelGGL(act_backward_kernel, dim3(r.nblk), dim3(RB), 0, st, a);
    BP_CHECK_LAUNCH();
  }
  return bp_sum_partials3(a.partial, r.nblk, raw->c, sums, st);
}

int bp_act_backward_bn(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const bp_pointwise* pw,
                       const bp_view* act_out, const bp_view* g, double* sums, const bp_bn_backward_fin* fin,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!fin || !raw || fin->count <= 0 || !fin->save_mean || !fin->save_invstd || !fin->coef_abc) return BP_EINVAL;
  const BnBwdFin f{fin->count, fin->gamma, fin->save_mean, fin->save_invstd, fin->param_grad_scale, fin->dgamma,
                   fin->dbeta, fin->coef_abc};
  t_bnbwd = &f;
  const int rc = bp_act_backward(dout, dout2, raw, pw, act_out, g, sums, workspace, workspace_bytes, stream);
  t_bnbwd = nullptr;
  return rc;
}

int bp_bn_backward_finalize(const double* sums, double count, int32_t c, const float* gamma,
                            const double* save_mean, const double* save_invstd, float param_grad_scale,
                            float* dgamma, float* dbeta, double* coef_abc, void* stream) {
  if (!sums || c <= 0 || count <= 0 || !save_mean || !save_invstd || !coef_abc) return BP_EINVAL;
  hipLaunchKernelGGL(bn_backward_finalize_kernel, dim3(bp_ceil_div(c, 64)), dim3(64), 0, bp_stream(stream), sums,
                     count, c, gamma, save_mean, save_invstd, param_grad_scale, dgamma, dbeta, coef_abc);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bn_backward_apply(const bp_view* g, const bp_view* raw, const double* coef_abc, const bp_view* out,
                         void* stream) {
  if (is_bf16(g) || is_bf16(raw) || is_bf16(out)) {
    if (!coef_abc || !g || !out) return BP_EINVAL;
    const bp_view* vs[2] = {g, out};
    if (!bf16_set_ok(raw, vs, 2)) return BP_EUNSUPPORTED;
    return bp_bf16_bn_backward_apply(g, nullptr, raw, PW{nullptr, nullptr, nullptr}, nullptr, coef_abc, out, false,
                                     bp_stream(stream));
  }
  if (!bp_view_ok(g) || !bp_view_ok(raw) || !bp_view_ok(out) || !same_grid(g, raw) || !same_grid(g, out) ||
      !coef_abc)
    return BP_EINVAL;
  const int64_t total = bp_view_pixels(g) * g->c;
  if (dense_ok(g) && dense_ok(raw) && dense_ok(out)) {
    const int64_t total4 = total / 4;
    int64_t nb = (total4 + RB * 4 - 1) / (RB * 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(bn_backward_apply_fast_kernel, dim3((unsigned)nb), dim3(RB), 0, bp_stream(stream),
                       reinterpret_cast<const float4*>(g->ptr), reinterpret_cast<const float4*>(raw->ptr), coef_abc,
                       reinterpret_cast<float4*>(out->ptr), g->c, total4);
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(g), vd(raw),
                     coef_abc, vd(out), total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_act_bn_backward_apply(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const bp_pointwise* pw,
                             const bp_view* act_out, const double* coef_abc, const bp_view* out, void* stream) {
  if (is_bf16(raw) || is_bf16(dout) || is_bf16(dout2) || is_bf16(act_out) || is_bf16(out)) {
    if (!coef_abc || !dout || !out) return BP_EINVAL;
    const bp_view* vs[4] = {dout, dout2, act_out, out};
    if (!bf16_set_ok(raw, vs, 4)) return BP_EUNSUPPORTED;
    return bp_bf16_bn_backward_apply(dout, dout2, raw, bp_pw(pw), act_out, coef_abc, out, true, bp_stream(stream));
  }
  if (!bp_view_ok(dout) || !bp_view_ok(raw) || !bp_view_ok(out) || !coef_abc) return BP_EINVAL;
  if (!same_grid(dout, raw) || !same_grid(out, raw)) return BP_EINVAL;
  if (dout2 && (!bp_view_ok(dout2) || !same_grid(dout2, raw))) return BP_EINVAL;
  if (act_out && (!bp_view_ok(act_out) || !same_grid(act_out, raw))) return BP_EINVAL;
  const int64_t total = bp_view_pixels(raw) * raw->c;
  if (dense_ok(raw) && dense_ok(dout) && dense_ok(out) && (!dout2 || dense_ok(dout2)) && (!act_out || dense_ok(act_out))) {
    ActApplyFast fa{};
    fa.dout = reinterpret_cast<const float4*>(dout->ptr);
    fa.dout2 = dout2 ? reinterpret_cast<const float4*>(dout2->ptr) : nullptr;
    fa.raw = reinterpret_cast<const float4*>(raw->ptr);
    fa.aout = act_out ? reinterpret_cast<const float4*>(act_out->ptr) : nullptr;
    fa.out = reinterpret_cast<float4*>(out->ptr);
    fa.pw = bp_pw(pw); fa.abc = coef_abc; fa.c = raw->c; fa.total4 = total / 4;
    int64_t nb = (fa.total4 + RB * 4 - 1) / (RB * 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(act_bn_backward_apply_fast_kernel, dim3((unsigned)nb), dim3(RB), 0, bp_stream(stream), fa);
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  hipLaunchKernelGGL(act_bn_backward_apply_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(dout),
                     vd(dout2), vd(raw), vd(act_out), bp_pw(pw), coef_abc, vd(out), total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_prelu_slope_grad(const double* sums, int32_t c, float* dslope, void* stream) {
  if (!sums || c <= 0 || !dslope) return BP_EINVAL;
  hipLaunchKernelGGL(prelu_slope_grad_kernel, dim3(1), dim3(64), 0, bp_stream(stream), sums, c, dslope);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_sums_to_float(const double* sums, int32_t c, float* dst, void* stream) {
  if (!sums || c <= 0 || !dst) return BP_EINVAL;
  hipLaunchKernelGGL(sums_to_float_kernel, dim3(bp_ceil_div(c, 64)), dim3(64), 0, bp_stream(stream), sums, c, dst);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_residual_forward(const bp_view* raw, const bp_pointwise* pw, const bp_view* skip,
                        const bp_pointwise* skip_pw, float slope, const bp_view* out, void* stream) {
  if (is_bf16(raw) || is_bf16(skip) || is_bf16(out)) {
    if (!skip || !out) return BP_EINVAL;
    const bp_view* vs[2] = {skip, out};
    if (!bf16_set_ok(raw, vs, 2)) return BP_EUNSUPPORTED;
    return bp_bf16_residual_forward(raw, bp_pw(pw), skip, bp_pw(skip_pw), slope, out, bp_stream(stream));
  }
  if (!bp_view_ok(raw) || !bp_view_ok(skip) || !bp_view_ok(out)) return BP_EINVAL;
  if (!same_grid(raw, skip) || !same_grid(raw, out)) return BP_EINVAL;
  const int64_t total = bp_view_pixels(raw) * raw->c;
  if (dense_ok(raw) && dense_ok(skip) && dense_ok(out)) {
    const int64_t total4 = total / 4;
    int64_t nb = (total4 + RB * 4 - 1) / (RB * 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(residual_forward_fast_kernel, dim3((unsigned)nb), dim3(RB), 0, bp_stream(stream),
                       reinterpret_cast<const float4*>(raw->ptr), bp_pw(pw), reinterpret_cast<const float4*>(skip->ptr),
                       bp_pw(skip_pw), slope, reinterpret_cast<float4*>(out->ptr), raw->c, total4);
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  hipLaunchKernelGGL(residual_forward_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(raw),
                     bp_pw(pw), vd(skip), bp_pw(skip_pw), slope, vd(out), total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_nchw_to_view(const float* src_nchw, int32_t c, const float* aux, int32_t caux, const bp_view* out,
                    void* stream) {
  if (!src_nchw || c <= 0 || caux < 0 || (caux > 0 && !aux) || !out || !out->ptr) return BP_EINVAL;
  if (out->coff < 0 || out->coff + c + caux > out->cstride) return BP_EINVAL;
  const int64_t total = bp_view_pixels(out) * (c + caux);
  const int64_t hw = (int64_t)out->h * out->w;
  const int ct = c + caux;
  if (hw % 4 == 0 && hw / 4 < 0x7fffffff && ct >= 1 && ct <= 4 && reinterpret_cast<uintptr_t>(src_nchw) % 16 == 0 &&
      (!(ct == 2 && out->cstride == 2 && out->coff == 0) || reinterpret_cast<uintptr_t>(out->ptr) % 16 == 0)) {
    const int hw4 = (int)(hw / 4);
    const int64_t total4 = (int64_t)out->n * hw4;
    const dim3 grid(nblocks(total4)), block(RB);
    hipStream_t st = bp_stream(stream);
    switch (ct) {
      case 1: hipLaunchKernelGGL(nchw_to_view4_kernel<1>, grid, block, 0, st, src_nchw, c, aux, vd(out), hw4, total4); break;
      case 2: hipLaunchKernelGGL(nchw_to_view4_kernel<2>, grid, block, 0, st, src_nchw, c, aux, vd(out), hw4, total4); break;
      case 3: hipLaunchKernelGGL(nchw_to_view4_kernel<3>, grid, block, 0, st, src_nchw, c, aux, vd(out), hw4, total4); break;
      default: hipLaunchKernelGGL(nchw_to_view4_kernel<4>, grid, block, 0, st, src_nchw, c, aux, vd(out), hw4, total4); break;
    }
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  hipLaunchKernelGGL(nchw_to_view_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), src_nchw, c, aux,
                     caux, vd(out), out->n, total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_view_to_nchw(const bp_view* src, const bp_pointwise* pw, int32_t softplus, float* dst_nchw, void* stream) {
  if (!bp_view_ok(src) || !dst_nchw) return BP_EINVAL;
  const int64_t total = bp_view_pixels(src) * src->c;
  hipLaunchKernelGGL(view_to_nchw_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(src), bp_pw(pw),
                     softplus, dst_nchw, total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_fill(float* dst, int64_t n, float value, void* stream) {
  if (!dst || n < 0) return BP_EINVAL;
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(nblocks(n)), dim3(RB), 0, bp_stream(stream), dst, n, value);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

static bool latent_view_ok(const bp_latent* lt, const bp_view* v, int n, int c) {
  return bp_view_ok(v) && v->n == n && v->h == lt->zh && v->w == lt->zw && v->c == c;
}

int bp_latent_forward(const bp_latent* lt, const bp_view* q_raw, const bp_pointwise* q_pw, const bp_view* p_raw,
                      const bp_pointwise* p_pw, const float* eps, float* stats4, const bp_view* z,
                      double* kl_sum, void* workspace, size_t workspace_bytes, void* stream) {
  if (!lt || lt->n <= 0 || lt->L <= 0 || lt->zc <= 0 || !eps || !stats4 || !kl_sum) return BP_EINVAL;
  if (!latent_view_ok(lt, q_raw, lt->n, 2 * lt->zc)) return BP_EINVAL;
  if (p_raw && !latent_view_ok(lt, p_raw, lt->n, 2 * lt->zc)) return BP_EINVAL;
  if (!latent_view_ok(lt, z, lt->n * lt->L, lt->zc)) return BP_EINVAL;
  LatentArgs a{};
  a.lt = *lt; a.q = vd(q_raw); a.p = vd(p_raw); a.z = vd(z); a.qpw = bp_pw(q_pw); a.ppw = bp_pw(p_pw);
  a.eps = eps; a.stats4 = stats4; a.nelem = (int64_t)lt->n * lt->zc * lt->zh * lt->zw;
  int nblk = (int)((a.nelem + RB - 1) / RB);
  if (nblk > 256) nblk = 256;
  if (!workspace || workspace_bytes < (size_t)nblk * sizeof(double)) return BP_EWORKSPACE;
  a.partial = reinterpret_cast<double*>(workspace);
  hipStream_t st = bp_stream(stream);
  hipLaunchKernelGGL(latent_forward_kernel, dim3(nblk), dim3(RB), 0, st, a);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(1), dim3(64), 0, st, a.partial, nblk, 1, kl_sum);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_latent_backward(const bp_latent* lt, const bp_view* dz, const float* stats4, const float* eps,
                       const float* seed, float beta_kl, const bp_view* dq_act, const bp_view* dp_act,
                       void* stream) {
  if (!lt || !stats4 || !eps || !seed) return BP_EINVAL;
  if (!latent_view_ok(lt, dz, lt->n * lt->L, lt->zc) || !latent_view_ok(lt, dq_act, lt->n, 2 * lt->zc))
    return BP_EINVAL;
  if (dp_act && !latent_view_ok(lt, dp_act, lt->n, 2 * lt->zc)) return BP_EINVAL;
  LatentBwdArgs a{};
  a.lt = *lt; a.dz = vd(dz); a.dq = vd(dq_act); a.dp = vd(dp_act); a.stats4 = stats4; a.eps = eps; a.seed = seed;
  a.beta_kl = beta_kl; a.nelem = (int64_t)lt->n * lt->zc * lt->zh * lt->zw;
  hipLaunchKernelGGL(latent_backward_kernel, dim3(nblocks(a.nelem)), dim3(RB), 0, bp_stream(stream), a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

size_t bp_loglik_workspace(const bp_loglik* ll) {
  if (!ll) return 0;
  const RedPlan r = red_plan(ll->c, (int64_t)ll->n * ll->L * ll->h * ll->w);
  return (size_t)r.nblk * 2 * ll->c * sizeof(double);
}

static bool ll_view_ok(const bp_loglik* ll, const bp_view* v) {
  return bp_view_ok(v) && v->n == ll->n * ll->L && v->h == ll->h && v->w == ll->w && v->c == ll->c;
}

int bp_loglik_forward(const bp_loglik* ll, const float* x_nchw, const bp_view* mu_raw, const bp_view* var_raw,
                      const double* kl_sum, float* x_mu_nchw, float* x_log_var_nchw, float* stats,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!ll || !x_nchw || !x_mu_nchw || !stats || !ll_view_ok(ll, mu_raw)) return BP_EINVAL;
  if (ll->c > RB || (int64_t)ll->n * ll->L * ll->h * ll->w >= (int64_t)1 << 31) return BP_EUNSUPPORTED;
  if (ll->predict_var && !ll_view_ok(ll, var_raw)) return BP_EINVAL;
  if (!workspace || workspace_bytes < bp_loglik_workspace(ll)) return BP_EWORKSPACE;
  LoglikArgs a{};
  a.ll = *ll; a.x = x_nchw; a.mu = vd(mu_raw); a.var = vd(ll->predict_var ? var_raw : nullptr);
  a.x_mu = x_mu_nchw; a.x_lv = x_log_var_nchw; a.partial = reinterpret_cast<double*>(workspace);
  a.npix = (int64_t)ll->n * ll->L * ll->h * ll->w;
  const RedPlan r = red_plan(ll->c, a.npix);
  a.CP = r.CP; a.pix_per_block = r.ppb;
  hipStream_t st = bp_stream(stream);
  hipLaunchKernelGGL(loglik_forward_kernel, dim3(r.nblk), dim3(RB), 0, st, a);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(loglik_finalize_kernel, dim3(1), dim3(64), 0, st, *ll, a.partial, r.nblk, kl_sum, stats);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_loglik_backward(const bp_loglik* ll, const float* x_nchw, const bp_view* mu_raw, const bp_view* var_raw,
                       const float* seed, const bp_view* d_mu_raw, const bp_view* d_var_raw, void* stream) {
  if (!ll || !x_nchw || !seed || !ll_view_ok(ll, mu_raw) || !ll_view_ok(ll, d_mu_raw)) return BP_EINVAL;
  if (ll->predict_var && (!ll_view_ok(ll, var_raw) || !ll_view_ok(ll, d_var_raw))) return BP_EINVAL;
  if ((int64_t)ll->n * ll->L * ll->h * ll->w >= (int64_t)1 << 31) return BP_EUNSUPPORTED;
  LoglikBwdArgs a{};
  a.ll = *ll; a.x = x_nchw; a.seed = seed; a.mu = vd(mu_raw); a.var = vd(ll->predict_var ? var_raw : nullptr);
  a.dmu = vd(d_mu_raw); a.dvar = vd(ll->predict_var ? d_var_raw : nullptr);
  a.total = (int64_t)ll->n * ll->L * ll->h * ll->w * ll->c;
  hipLaunchKernelGGL(loglik_backward_kernel, dim3(nblocks(a.total)), dim3(RB), 0, bp_stream(stream), a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_unary_forward(const bp_view* in, const bp_pointwise* pw, int32_t kind, const bp_view* out, void* stream) {
  if (!bp_view_ok(in) || !bp_view_ok(out) || !same_grid(in, out) || kind < 0 || kind > 2) return BP_EINVAL;
  const int64_t total = bp_view_pixels(in) * in->c;
  hipLaunchKernelGGL(unary_forward_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(in), bp_pw(pw),
                     kind, vd(out), total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bce_logits(const bp_view* raw, int32_t n0, int32_t n1, float target, double* sum, void* workspace,
                  size_t workspace_bytes, void* stream) {
  if (!bp_view_ok(raw) || n0 < 0 || n1 > raw->n || n0 >= n1 || !sum) return BP_EINVAL;
  const int nblk = 256;
  if (!workspace || workspace_bytes < nblk * sizeof(double)) return BP_EWORKSPACE;
  hipStream_t st = bp_stream(stream);
  hipLaunchKernelGGL(bce_logits_kernel, dim3(nblk), dim3(RB), 0, st, vd(raw), n0, n1, target,
                     reinterpret_cast<double*>(workspace));
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<double*>(workspace), nblk, 1, sum);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bce_logits_grad(const bp_view* raw, int32_t n0, int32_t n1, float target, float scale, const bp_view* d_raw,
                       void* stream) {
  if (!bp_view_ok(raw) || !bp_view_ok(d_raw) || !same_grid(raw, d_raw) || n0 < 0 || n1 > raw->n || n0 >= n1)
    return BP_EINVAL;
  const int64_t total = (int64_t)(n1 - n0) * raw->h * raw->w * raw->c;
  hipLaunchKernelGGL(bce_logits_grad_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(raw), n0, n1,
                     target, scale, vd(d_raw));
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_l1_sum(const bp_view* fake, const float* x_nchw, double* sum, void* workspace, size_t workspace_bytes,
              void* stream) {
  if (!bp_view_ok(fake) || !x_nchw || !sum) return BP_EINVAL;
  const int nblk = 256;
  if (!workspace || workspace_bytes < nblk * sizeof(double)) return BP_EWORKSPACE;
  hipStream_t st = bp_stream(stream);
  hipLaunchKernelGGL(l1_kernel, dim3(nblk), dim3(RB), 0, st, vd(fake), x_nchw, reinterpret_cast<double*>(workspace));
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_wave_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<double*>(workspace), nblk, 1, sum);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_tanh_l1_backward(const bp_view* fake, const float* x_nchw, const bp_view* d_fake, float l1_scale,
                        const bp_view* d_raw, void* stream) {
  if (!bp_view_ok(fake) || !x_nchw || !bp_view_ok(d_raw) || !same_grid(fake, d_raw)) return BP_EINVAL;
  if (d_fake && (!bp_view_ok(d_fake) || !same_grid(fake, d_fake))) return BP_EINVAL;
  const int64_t total = bp_view_pixels(fake) * fake->c;
  hipLaunchKernelGGL(tanh_l1_backward_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), vd(fake), x_nchw,
                     vd(d_fake), l1_scale, vd(d_raw), total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_gather_tiles(const void* desc100, const void* desc150, const void* xform, int32_t n, int32_t tile,
                    float* out_nchw, void* stream) {
  if (!desc100 || !desc150 || !xform || n <= 0 || tile <= 0 || !out_nchw) return BP_EINVAL;
  const int64_t total = (int64_t)n * tile * tile;
  hipLaunchKernelGGL(gather_tiles_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream),
                     reinterpret_cast<const TileDesc*>(desc100), reinterpret_cast<const TileDesc*>(desc150),
                     reinterpret_cast<const SampleXform*>(xform), out_nchw, tile, total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                     const float* hyper, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || !hyper) return BP_EINVAL;
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(nblocks(n)), dim3(RB), 0, bp_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, n, hyper);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                 float beta1, float beta2, float eps, int32_t step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return BP_EINVAL;
  if (n == 0) return BP_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(RB), 0, bp_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
  BP_CHECK_LAUNCH();
  return BP_OK;
}

}  // extern "C"
