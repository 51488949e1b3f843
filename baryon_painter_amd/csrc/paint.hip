// paint() throughput pipeline (BASELINE.json configs[4], SURVEY.md 8d metric (B), 8f-2): the pieces that sit on
// either side of the captured eval-mode forward so that a RAW dark-matter tile goes in and a physical pressure tile
// comes out without a host-side NumPy pass per tile.
//   bp_paint_load     raw NCHW tile -> shift-log transform (data_transforms.py:76) -> NHWC view + redshift planes
//                     (merge_aux_label, utils.py:159-182): the fused form of the host transform + bp_nchw_to_view
//   bp_paint_store    head view -> [softplus] -> inverse shift-log (data_transforms.py:97) -> NCHW tile: the fused
//                     form of bp_view_to_nchw + the host inverse transform
//   bp_philox_normal  the prior noise of cvae.py:64-65 from a counter-based generator keyed on (seed, GLOBAL tile id):
//                     a tile's sample does not depend on which batch, stream or rank paints it
// Float semantics are those of the host path (the reference's NumPy expressions on float32 tiles, evaluated in
// double where NumPy promotes to double), so device and host transforms agree to the last float32 bit except for
// libm differences of exp (<= 1 ulp of the exponential).
#include "common.hpp"
#include <math.h>

namespace {

constexpr int RB = 256;

// (out2: an optional second destination of the same pixels -- the generator reads the transformed tile in its
//  concatenated input as well as the prior network: the double-precision logarithm is taken once, not once per view)
__global__ __launch_bounds__(RB) void paint_load_kernel(const float* src, int c, const double* sigma_k, const float* aux,
                                                        int caux, float* out, int out_cs, int out_co, float* out2,
                                                        int out2_cs, int out2_co, int64_t hw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= total) return;
  // (32-bit index arithmetic, shifts where the sizes are powers of two: four 64-bit divisions by run-time values per
  //  element were what this kernel spent its time on -- the entry points check total < 2^31)
  const unsigned ct = (unsigned)(c + caux), iu = (unsigned)i, hwu = (unsigned)hw;
  const unsigned pu = ct == 2u ? iu >> 1 : iu / ct;
  const int ch = (int)(iu - pu * ct);
  const unsigned nu = (hwu & (hwu - 1u)) == 0u ? pu >> (__ffs((int)hwu) - 1) : pu / hwu;
  const int64_t p = pu, n = nu, yx = pu - nu * hwu;
  float v;
  if (ch < c) {
    // np.log(x / std + 1) / k  with float32 x and float64 std: evaluated in double, stored as float32
    const double x = (double)src[(n * c + ch) * hw + yx];
    v = (float)(log(x / sigma_k[2 * n] + 1.0) / sigma_k[2 * n + 1]);
  } else {
    v = aux[n * caux + (ch - c)];
  }
  out[p * out_cs + out_co + ch] = v;
  if (out2) out2[p * out2_cs + out2_co + ch] = v;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(RB) void paint_store_kernel(const float* src, int src_cs, int src_co, int c, PW pw,
                                                         int softplus, const double* k_sigma, float* dst, int64_t hw,
                                                         int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;     // NCHW destination index (coalesced writes)
  if (i >= total) return;
  const unsigned iu = (unsigned)i, hwu = (unsigned)hw;          // (32-bit arithmetic: see paint_load_kernel)
  const unsigned pl = (hwu & (hwu - 1u)) == 0u ? iu >> (__ffs((int)hwu) - 1) : iu / hwu;        // plane = n * c + ch
  const int64_t yx = iu - pl * hwu;
  const unsigned nu = c == 1 ? pl : pl / (unsigned)c;
  const int ch = (int)(pl - nu * (unsigned)c);
  const int64_t n = nu;
  float v = pw_apply(pw, ch, src[(n * hw + yx) * src_cs + src_co + ch]);
  if (softplus) v = softplus_f(v);
  // (np.exp(x * k) - 1) * std: float32 product, float32 exp, float32 subtraction, double product
  const float t = v * (float)k_sigma[2 * n];
  const float e = (float)exp((double)t);            // correctly rounded float32 exponential
  const float r = e - 1.0f;
  dst[i] = (float)((double)r * k_sigma[2 * n + 1]);
}

// Philox4x32-10 (Salmon et al. 2011): counter (c0..c3), key (k0, k1)
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
  c[0] = h1 ^ c[1] ^ k0; c[1] = l1; c[2] = h0 ^ c[3] ^ k1; c[3] = l0;
}

// eps[(l * n + s) * per_tile + i], i = 4 * g + r: r-th normal of Philox block (g, l, tile id) under key = seed.
// Box-Muller in double on the uniforms u1 = (x + 1) / 2^32 in (0, 1], u2 = x / 2^32 in [0, 1).
// (seed_dev != nullptr: the key is read from device memory, so that a captured graph serves every seed)
__global__ __launch_bounds__(RB) void philox_normal_kernel(unsigned long long seed, const unsigned long long* seed_dev,
                                                           const long long* tile_ids, int n, int L, int per_tile,
                                                           float* eps) {
  if (seed_dev) seed = *seed_dev;
  const int groups = (per_tile + 3) / 4;
  const int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x;
  if (i >= (int64_t)L * n * groups) return;
  const int g = i % groups;
  const int s = (i / groups) % n;
  const int l = i / ((int64_t)groups * n);
  const unsigned long long tid = (unsigned long long)tile_ids[s];
  unsigned c[4] = {(unsigned)g, (unsigned)l, (unsigned)tid, (unsigned)(tid >> 32)};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const double two32 = 4294967296.0, twopi = 6.283185307179586476925286766559;
  double z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double u1 = ((double)c[2 * h] + 1.0) / two32, u2 = (double)c[2 * h + 1] / two32;
    const double rad = sqrt(-2.0 * log(u1));
    z[2 * h] = rad * cos(twopi * u2);
    z[2 * h + 1] = rad * sin(twopi * u2);
  }
  float* o = eps + ((int64_t)l * n + s) * per_tile + 4 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * g + r < per_tile) o[r] = (float)z[r];
}

static inline unsigned nblocks(int64_t total) { return (unsigned)((total + RB - 1) / RB); }

}  // namespace

extern "C" {

int bp_paint_load(const float* raw_nchw, int32_t c, const double* sigma_k, const float* aux, int32_t caux,
                  const bp_view* out, void* stream) {
  if (!raw_nchw || !sigma_k || !bp_view_ok(out) || c <= 0 || caux < 0 || out->c != c + caux || (caux > 0 && !aux))
    return BP_EINVAL;
  const int64_t hw = (int64_t)out->h * out->w, total = (int64_t)out->n * hw * (c + caux);
  if (total >= (int64_t)1 << 31) return BP_EUNSUPPORTED;
  hipLaunchKernelGGL(paint_load_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), raw_nchw, c, sigma_k, aux,
                     caux, out->ptr, out->cstride, out->coff, (float*)nullptr, 0, 0, hw, total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_paint_load2(const float* raw_nchw, int32_t c, const double* sigma_k, const float* aux, int32_t caux,
                   const bp_view* out, const bp_view* out2, void* stream) {
  if (!raw_nchw || !sigma_k || !bp_view_ok(out) || !bp_view_ok(out2) || c <= 0 || caux < 0 || out->c != c + caux ||
      out2->c != out->c || out2->n != out->n || out2->h != out->h || out2->w != out->w || (caux > 0 && !aux))
    return BP_EINVAL;
  const int64_t hw = (int64_t)out->h * out->w, total = (int64_t)out->n * hw * (c + caux);
  if (total >= (int64_t)1 << 31) return BP_EUNSUPPORTED;
  hipLaunchKernelGGL(paint_load_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), raw_nchw, c, sigma_k, aux,
                     caux, out->ptr, out->cstride, out->coff, out2->ptr, out2->cstride, out2->coff, hw, total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_paint_store(const bp_view* src, const bp_pointwise* pw, int32_t softplus, const double* k_sigma, float* dst_nchw,
                   void* stream) {
  if (!bp_view_ok(src) || !k_sigma || !dst_nchw) return BP_EINVAL;
  const int64_t hw = (int64_t)src->h * src->w, total = (int64_t)src->n * hw * src->c;
  if (total >= (int64_t)1 << 31) return BP_EUNSUPPORTED;
  hipLaunchKernelGGL(paint_store_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), src->ptr, src->cstride,
                     src->coff, src->c, bp_pw(pw), softplus, k_sigma, dst_nchw, hw, total);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_philox_normal(uint64_t seed, const int64_t* tile_ids, int32_t n, int32_t L, int32_t per_tile, float* eps,
                     void* stream) {
  if (!tile_ids || n <= 0 || L <= 0 || per_tile <= 0 || !eps) return BP_EINVAL;
  const int64_t total = (int64_t)L * n * ((per_tile + 3) / 4);
  hipLaunchKernelGGL(philox_normal_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream),
                     (unsigned long long)seed, (const unsigned long long*)nullptr,
                     reinterpret_cast<const long long*>(tile_ids), n, L, per_tile, eps);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_philox_normal_dev(const uint64_t* seed_dev, const int64_t* tile_ids, int32_t n, int32_t L, int32_t per_tile,
                         float* eps, void* stream) {
  if (!seed_dev || !tile_ids || n <= 0 || L <= 0 || per_tile <= 0 || !eps) return BP_EINVAL;
  const int64_t total = (int64_t)L * n * ((per_tile + 3) / 4);
  hipLaunchKernelGGL(philox_normal_kernel, dim3(nblocks(total)), dim3(RB), 0, bp_stream(stream), 0ull,
                     reinterpret_cast<const unsigned long long*>(seed_dev),
                     reinterpret_cast<const long long*>(tile_ids), n, L, per_tile, eps);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

}  // extern "C"
