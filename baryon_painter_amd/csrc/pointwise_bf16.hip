// HBM-bound passes over bf16 tensors (BASELINE.json configs[3]): batch-norm statistics, activation / batch-norm
// backward and the residual tail for dense NHWC bf16 views with a power-of-two channel count >= 8.
// Same contracts and the same two-stage fixed-order reductions (double partials) as the fp32 kernels in
// pointwise.hip; a thread moves 16 bytes = 8 channels per access and keeps its 8 channels' sums in registers
// (every block starts at a multiple of 2048 elements, so thread t always sees channels (8t + j) % c).
// Arithmetic is fp32 per element (inputs are 8-bit-mantissa values; the reference's double evaluation of the
// batch-norm backward map would be rounded away by the bf16 store), sums are double.
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int RB = 256;
typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

__device__ __forceinline__ void unpack8(const uint4 t, float (&v)[8]) {
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// sum over the threads that share a channel group (stride c/8 threads), then one row of c values per block
template <int NS>
__device__ __forceinline__ void block_reduce_store8(double (&acc)[NS][8], int c, double* out /* [NS][c] */, double* sh) {
  const int tid = threadIdx.x;
  const int G = c / 8;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[tid * 8 + j] = acc[s][j];
    __syncthreads();
    for (int st = RB / 2; st >= G; st >>= 1) {
      if (tid < st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[tid * 8 + j] += sh[(tid + st) * 8 + j];
      }
      __syncthreads();
    }
    if (tid < G) {
#pragma unroll
      for (int j = 0; j < 8; ++j) out[s * c + tid * 8 + j] = sh[tid * 8 + j];
    }
  }
}

struct Pw8 {
  float sc[8], sf[8], sl[8];
};
__device__ __forceinline__ Pw8 pw8_load(const PW& pw, int c) {
  Pw8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = (8 * threadIdx.x + j) % c;
    r.sc[j] = pw.scale ? pw.scale[ch] : 1.f;
    r.sf[j] = pw.scale ? pw.shift[ch] : 0.f;
    r.sl[j] = pw.scale ? pw.slope[ch] : 1.f;
  }
  return r;
}

__global__ __launch_bounds__(RB) void channel_sums_bf16_kernel(const uint4* __restrict__ x, int c, int64_t total8,
                                                               int64_t chunk8, double* partial) {
  __shared__ double sh[RB * 8];
  const int64_t b0 = (int64_t)blockIdx.x * chunk8;
  int64_t b1 = b0 + chunk8;
  if (b1 > total8) b1 = total8;
  double acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.0; acc[1][j] = 0.0; }
  // four elements per trip: their loads are in flight together, their sums are taken in fp32 (the elements carry 8
  // mantissa bits) and added to the double accumulators once per trip -- a quarter of the double-rate instructions
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 4 * RB) {
    uint4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = x[i + u * RB < b1 ? i + u * RB : i];
    float p0[8], p1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { p0[j] = 0.f; p1[j] = 0.f; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * RB >= b1) break;
      float v[8];
      unpack8(q[u], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { p0[j] += v[j]; p1[j] = fmaf(v[j], v[j], p1[j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[0][j] += (double)p0[j]; acc[1][j] += (double)p1[j]; }
  }
  block_reduce_store8<2>(acc, c, partial + (int64_t)blockIdx.x * 2 * c, sh);
}

struct ActBwd8 {
  const uint4* dout; const uint4* dout2; const uint4* raw; const uint4* aout; uint4* g;
  PW pw; int c; int64_t total8, chunk8; double* partial;
};

// g = (dout [+ dout2]) * act'(t), sums {sum g, sum g*raw, sum d*t*[t<=0]}  (bp_act_backward's contract)
template <bool D2, bool AO, bool WG>
__global__ __launch_bounds__(RB) void act_backward_bf16_kernel(ActBwd8 a) {
  __shared__ double sh[RB * 8];
  const Pw8 p = pw8_load(a.pw, a.c);
  const int64_t b0 = (int64_t)blockIdx.x * a.chunk8;
  int64_t b1 = b0 + a.chunk8;
  if (b1 > a.total8) b1 = a.total8;
  double acc[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.0; acc[1][j] = 0.0; acc[2][j] = 0.0; }
  // two elements per trip (loads in flight together), fp32 partial sums added to the double accumulators per trip
  for (int64_t i0 = b0 + threadIdx.x; i0 < b1; i0 += 2 * RB) {
    uint4 qd[2], qr[2], q2[2], qa[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t i = i0 + u * RB < b1 ? i0 + u * RB : i0;
      qd[u] = a.dout[i]; qr[u] = a.raw[i];
      if constexpr (D2) q2[u] = a.dout2[i];
      if constexpr (AO) qa[u] = a.aout[i];
    }
    float p0[8], p1[8], p2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { p0[j] = 0.f; p1[j] = 0.f; p2[j] = 0.f; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t i = i0 + u * RB;
      if (i >= b1) break;
      float d[8], r[8], so[8], g[8];
      unpack8(qd[u], d);
      if constexpr (D2) {
        float e[8];
        unpack8(q2[u], e);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] += e[j];
      }
      unpack8(qr[u], r);
      if constexpr (AO) unpack8(qa[u], so);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = fmaf(r[j], p.sc[j], p.sf[j]);
        const bool pos = (AO ? so[j] : t) > 0.f;
        g[j] = pos ? d[j] : d[j] * p.sl[j];
        p0[j] += g[j];
        p1[j] = fmaf(g[j], r[j], p1[j]);
        if (!pos) p2[j] = fmaf(d[j], t, p2[j]);
      }
      if constexpr (WG) a.g[i] = pack8(g);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[0][j] += (double)p0[j]; acc[1][j] += (double)p1[j]; acc[2][j] += (double)p2[j]; }
  }
  block_reduce_store8<3>(acc, a.c, a.partial + (int64_t)blockIdx.x * 3 * a.c, sh);
}

struct Apply8 {
  const uint4* dout; const uint4* dout2; const uint4* raw; const uint4* aout; uint4* out;
  PW pw; const double* abc; int c; int64_t total8; int recompute_g;
};

// out = A*(g - mg) + B*(raw - mean), g either given (dout IS g) or recomputed from (dout [+ dout2], mask).
// Two 16-byte units per thread and trip with every load of the trip issued before the first use (the variants are
// template parameters: a load under a run-time condition is waited for on the spot), raw words kept until then.
template <bool D2, bool AO, bool RG>
__global__ __launch_bounds__(RB) void bn_backward_apply_bf16_kernel(Apply8 a) {
  const int c = a.c;
  const Pw8 p = pw8_load(a.pw, c);
  float A[8], G[8], B[8], M[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = (8 * threadIdx.x + j) % c;
    A[j] = (float)a.abc[ch]; G[j] = (float)a.abc[c + ch]; B[j] = (float)a.abc[2 * c + ch]; M[j] = (float)a.abc[3 * c + ch];
  }
  const int64_t stride = (int64_t)gridDim.x * RB;
  auto one = [&](const uint4 wd, const uint4 w2, const uint4 wr, const uint4 wa) {
    float d[8], r[8], so[8], o[8];
    unpack8(wd, d);
    if constexpr (D2) {
      float e[8];
      unpack8(w2, e);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] += e[j];
    }
    unpack8(wr, r);
    if constexpr (AO) unpack8(wa, so);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g = d[j];
      if constexpr (RG) {
        const float t = fmaf(r[j], p.sc[j], p.sf[j]);
        const bool pos = (AO ? so[j] : t) > 0.f;
        g = pos ? d[j] : d[j] * p.sl[j];
      }
      o[j] = fmaf(A[j], g - G[j], B[j] * (r[j] - M[j]));
    }
    return pack8(o);
  };
  int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  for (; i + stride < a.total8; i += 2 * stride) {
    const int64_t k = i + stride;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 d0 = a.dout[i], d1 = a.dout[k];
    const uint4 e0 = D2 ? a.dout2[i] : z, e1 = D2 ? a.dout2[k] : z;
    const uint4 r0 = a.raw[i], r1 = a.raw[k];
    const uint4 s0 = AO ? a.aout[i] : z, s1 = AO ? a.aout[k] : z;
    a.out[i] = one(d0, e0, r0, s0);
    a.out[k] = one(d1, e1, r1, s1);
  }
  if (i < a.total8) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    a.out[i] = one(a.dout[i], D2 ? a.dout2[i] : z, a.raw[i], AO ? a.aout[i] : z);
  }
}

__global__ __launch_bounds__(RB) void residual_forward_bf16_kernel(const uint4* raw, PW pw, const uint4* skip, PW spw,
                                                                   float slope, uint4* out, int c, int64_t total8) {
  const Pw8 p = pw8_load(pw, c), k = pw8_load(spw, c);
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < total8; i += (int64_t)gridDim.x * RB) {
    float r[8], s[8], o[8];
    unpack8(raw[i], r);
    unpack8(skip[i], s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = fmaf(r[j], p.sc[j], p.sf[j]);          // (slope of `pw` is not applied: the tail activates the sum)
      float u = fmaf(s[j], k.sc[j], k.sf[j]);
      u = u > 0.f ? u : u * k.sl[j];
      t += u;
      o[j] = t > 0.f ? t : t * slope;
    }
    out[i] = pack8(o);
  }
}

struct FastPlan8 { int nblk; int64_t total8, chunk8; };
static inline FastPlan8 fast_plan8(int64_t total_elems) {
  FastPlan8 f{};
  f.total8 = total_elems / 8;
  static const int per = getenv("BP_PW8_UNITS") ? atoi(getenv("BP_PW8_UNITS")) : 16;
  int64_t nb = (f.total8 + 256 * per - 1) / (256 * per);   // >= 16 accesses per thread: a thread's fixed costs (24
  //                                                 activation parameters, the block's three tree reductions, a row of
  //                                                 partial sums) were most of the launch on the 67 MB trunk tensors
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  int64_t chunk = (f.total8 + nb - 1) / nb;
  chunk = (chunk + RB - 1) / RB * RB;             // block starts stay multiples of 2048 elements
  f.chunk8 = chunk;
  f.nblk = (int)((f.total8 + chunk - 1) / chunk);
  return f;
}

static inline const uint4* u4(const bp_view* v) { return v ? reinterpret_cast<const uint4*>(v->ptr) : nullptr; }
static inline unsigned stream_blocks(int64_t total8) {
  static const int per = getenv("BP_PW8_UNITS") ? atoi(getenv("BP_PW8_UNITS")) : 16;
  int64_t nb = (total8 + RB * per - 1) / (RB * per);       // (>= 16 units per thread: 56 parameter loads per thread)
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

}  // namespace

// pointwise.hip
int bp_sum_partials(const double* partial, int nblk, int n, double* out, hipStream_t st);
int bp_sum_partials3(const double* partial, int nblk, int c, double* sums, hipStream_t st);

bool bp_bf16_dense_ok(const bp_view* v) {
  return v && v->dtype == BP_BF16 && v->cstride == v->c && v->coff == 0 && v->c >= 8 && v->c <= 1024 &&
         (v->c & (v->c - 1)) == 0 && reinterpret_cast<uintptr_t>(v->ptr) % 16 == 0;
}

size_t bp_bf16_reduce_workspace(const bp_view* x, int nsums) {
  const FastPlan8 f = fast_plan8(bp_view_pixels(x) * x->c);
  return (size_t)f.nblk * nsums * x->c * sizeof(double);
}

int bp_bf16_channel_sums(const bp_view* x, double* sums, void* workspace, hipStream_t st) {
  const FastPlan8 f = fast_plan8(bp_view_pixels(x) * x->c);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(channel_sums_bf16_kernel, dim3(f.nblk), dim3(RB), 0, st, u4(x), x->c, f.total8, f.chunk8, partial);
  BP_CHECK_LAUNCH();
  return bp_sum_partials(partial, f.nblk, 2 * x->c, sums, st);
}

int bp_bf16_act_backward(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const PW& pw,
                         const bp_view* act_out, const bp_view* g, double* sums, void* workspace, hipStream_t st) {
  const FastPlan8 f = fast_plan8(bp_view_pixels(raw) * raw->c);
  ActBwd8 a{};
  a.dout = u4(dout); a.dout2 = u4(dout2); a.raw = u4(raw); a.aout = u4(act_out);
  a.g = g ? reinterpret_cast<uint4*>(g->ptr) : nullptr;
  a.pw = pw; a.c = raw->c; a.total8 = f.total8; a.chunk8 = f.chunk8; a.partial = reinterpret_cast<double*>(workspace);
  const dim3 grid(f.nblk), block(RB);
  switch ((a.dout2 ? 4 : 0) | (a.aout ? 2 : 0) | (a.g ? 1 : 0)) {
    case 0: hipLaunchKernelGGL((act_backward_bf16_kernel<false, false, false>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((act_backward_bf16_kernel<false, false, true>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((act_backward_bf16_kernel<false, true, false>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((act_backward_bf16_kernel<false, true, true>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((act_backward_bf16_kernel<true, false, false>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((act_backward_bf16_kernel<true, false, true>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((act_backward_bf16_kernel<true, true, false>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((act_backward_bf16_kernel<true, true, true>), grid, block, 0, st, a); break;
  }
  BP_CHECK_LAUNCH();
  return bp_sum_partials3(a.partial, f.nblk, raw->c, sums, st);
}

int bp_bf16_bn_backward_apply(const bp_view* dout, const bp_view* dout2, const bp_view* raw, const PW& pw,
                              const bp_view* act_out, const double* abc, const bp_view* out, bool recompute_g,
                              hipStream_t st) {
  Apply8 a{};
  a.dout = u4(dout); a.dout2 = u4(dout2); a.raw = u4(raw); a.aout = u4(act_out);
  a.out = reinterpret_cast<uint4*>(out->ptr); a.pw = pw; a.abc = abc; a.c = raw->c;
  a.total8 = bp_view_pixels(raw) * raw->c / 8; a.recompute_g = recompute_g ? 1 : 0;
  const dim3 grid(stream_blocks(a.total8)), block(RB);
  const int variant = (a.dout2 ? 4 : 0) | (a.aout ? 2 : 0) | (a.recompute_g ? 1 : 0);
  switch (variant) {
    case 0: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<false, false, false>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<false, false, true>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<false, true, false>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<false, true, true>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<true, false, false>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<true, false, true>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<true, true, false>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((bn_backward_apply_bf16_kernel<true, true, true>), grid, block, 0, st, a); break;
  }
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bf16_residual_forward(const bp_view* raw, const PW& pw, const bp_view* skip, const PW& spw, float slope,
                             const bp_view* out, hipStream_t st) {
  const int64_t total8 = bp_view_pixels(raw) * raw->c / 8;
  hipLaunchKernelGGL(residual_forward_bf16_kernel, dim3(stream_blocks(total8)), dim3(RB), 0, st, u4(raw), pw, u4(skip),
                     spw, slope, reinterpret_cast<uint4*>(out->ptr), raw->c, total8);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
