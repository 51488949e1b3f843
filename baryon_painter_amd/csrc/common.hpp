// Shared host/device helpers for the gfx950 kernels (see include/bp_hip.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bp_hip.h"

#define BP_CHECK_LAUNCH()                                   \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return BP_ELAUNCH; \
  } while (0)

static inline hipStream_t bp_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// geometry of a view, any element type
static inline bool bp_view_ok_any(const bp_view* v) {
  return v && v->ptr && v->n > 0 && v->h > 0 && v->w > 0 && v->c > 0 && v->cstride >= v->c &&
         v->coff >= 0 && v->coff + v->c <= v->cstride && (v->dtype == BP_F32 || v->dtype == BP_BF16);
}
// ... of an fp32 view: what every entry point without a bf16 form checks
static inline bool bp_view_ok(const bp_view* v) { return bp_view_ok_any(v) && v->dtype == BP_F32; }

// every pixel's channel slice starts on a 16-byte boundary
static inline bool bp_view_vec4(const bp_view* v) {
  return v->cstride % 4 == 0 && v->coff % 4 == 0 && reinterpret_cast<uintptr_t>(v->ptr) % 16 == 0;
}

static inline int64_t bp_view_pixels(const bp_view* v) { return (int64_t)v->n * v->h * v->w; }

static inline int bp_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int bp_round_up(int a, int b) { return bp_ceil_div(a, b) * b; }

// Device copy of a pointwise; scale == nullptr -> identity.
struct PW {
  const float* scale;
  const float* shift;
  const float* slope;
};
static inline PW bp_pw(const bp_pointwise* p) {
  PW r{nullptr, nullptr, nullptr};
  if (p && p->scale) r = PW{p->scale, p->shift, p->slope};
  return r;
}

// Request for per-channel sums out of an igemm epilogue (conv_igemm.hip, IgemmArgs::stat).  mode 1: {sum y,
// sum y^2} of the produced tensor; mode 2: the produced tensor is d(loss)/d(activated slot), `raw` / `spw` are that
// slot's raw values and pending activation: {sum g, sum g*raw} with g = d * act'(spw(raw)).  sums[2c].
// mode 3: as mode 2, but g ITSELF is stored instead of d (the slot has no batch-norm: g is what its layer's gradients
// read) and sums[3c] = {sum g, sum g*raw, sum_{t<=0} d*t} as bp_act_backward leaves them (conv_small.hip only).
// Training-mode batch-norm finalize (bp_bn_finalize's arguments) folded into the launch that sums the partial rows
// of a mode-1 request: one small launch less per batch-norm layer and step.
struct BnFin {
  double count;
  const float* gamma; const float* beta;
  float eps, momentum;
  float* rm; float* rv; int64_t* nbt;
  float* scale; float* shift;
  double* smean; double* sinv;
};
struct IgemmStatsReq {
  int mode;
  const bp_view* raw;
  PW spw;
  double* sums;
  void* ws;
  size_t ws_bytes;
  const BnFin* fin;         // mode 1 only; nullptr: sums only
};
int bp_sum_partials(const double* partial, int nblk, int n, double* out, hipStream_t st);
int bp_sum_partials_strided(const double* partial, int nblk, int stride, int n, double* out, hipStream_t st);
// the last stage of a statistics request: partial[nblk][n] -> sr->sums (and, with sr->fin, the finalize)
int bp_sum_partials_req(const double* partial, int nblk, int n, const IgemmStatsReq* sr, hipStream_t st);

__device__ __forceinline__ float pw_apply(const PW& pw, int ch, float x) {
  if (pw.scale == nullptr) return x;
  float t = fmaf(x, pw.scale[ch], pw.shift[ch]);
  return t > 0.f ? t : t * pw.slope[ch];
}

// ReLU that keeps a NaN a NaN, as torch.relu does (fmaxf(NaN, 0) is 0: a diverged run would paint finite tiles)
__device__ __forceinline__ float bp_relu_nan(float t) { return t < 0.f ? 0.f : t; }

// The pending activation of 4 consecutive channels held in registers (staging loops: a thread
// always handles the same channel quad, so the parameters are loaded once, not per element).
struct PW4 {
  float sc[4], sf[4], sl[4];
  bool on;
};
__device__ __forceinline__ PW4 pw4_load(const PW& pw, int ch, int cmax) {
  PW4 r;
  r.on = pw.scale != nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = r.on && (ch + j) < cmax;
    r.sc[j] = ok ? pw.scale[ch + j] : 1.f;
    r.sf[j] = ok ? pw.shift[ch + j] : 0.f;
    r.sl[j] = ok ? pw.slope[ch + j] : 1.f;
  }
  return r;
}
__device__ __forceinline__ float pw4_apply(const PW4& p, int j, float x) {
  const float t = fmaf(x, p.sc[j], p.sf[j]);
  return t > 0.f ? t : t * p.sl[j];
}
__device__ __forceinline__ float4 pw4_apply4(const PW4& p, float4 v) {
  if (!p.on) return v;
  return make_float4(pw4_apply(p, 0, v.x), pw4_apply(p, 1, v.y), pw4_apply(p, 2, v.z), pw4_apply(p, 3, v.w));
}

// ---- LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of
// consecutive LDS, no staging registers.  Issued from inline assembly: hipcc would otherwise treat every ds_read
// that follows a __builtin_amdgcn_global_load_lds as possibly aliasing it and wait vmcnt(0) right there, which
// serialises the pipeline.  All waits on these loads are therefore written by hand; the compiler's own counted
// waits for ordinary loads can only over-wait because of them (retirement is in order).  Keep ordinary global
// loads out of loops that hold a DMA in flight: their destination registers make the compiler drain vmcnt.
//   sbase          wave-uniform base (SGPR pair);  voff_bytes  per-lane byte offset (one VGPR)
//   lds_float_off  wave-uniform offset in floats into the dynamic LDS array, which starts right after the
//                  kernel's static LDS (lane l lands at that offset + 4*l floats)
__device__ __forceinline__ void bp_glds16(const float* sbase, unsigned voff_bytes, int lds_float_off) {
  const unsigned l = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_groupstaticsize() + 4u * (unsigned)lds_float_off);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_bytes), "s"(l), "s"(sbase) : "memory");
}
// Retire this wave's DMAs (its own lanes' bytes are then readable by this wave).
__device__ __forceinline__ void bp_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... then the workgroup barrier that publishes the landed tiles (and this wave's LDS writes) to the other waves.
__device__ __forceinline__ void bp_wait_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Geometry of one (transposed) convolution expressed as a stride-IS correlation over an output
// sub-grid ("phase").  Conv2d: one phase, IS = stride, taps = k.  ConvTranspose2d / data-gradient
// of a strided Conv2d: stride^2 phases with ceil(k/stride) taps each and IS = 1.
//   out[n, oy0 + OS*qy, ox0 + OS*qx, :] = sum_{ty,tx,ci} in[n, IS*qy + iy0 + ty, IS*qx + ix0 + tx, ci]
//                                                        * Wp[phase][ty][tx][ci][:]
struct ConvGeom {
  int gather_transposed;  // 0: conv form (F), 1: transposed form (T)
  int k, stride, pad;
  int cin_g, cout_g;      // channels of the gathered tensor / of the produced tensor
  int nphase;             // phases per dimension (1 or stride)
  int taps;               // taps per dimension per phase
  int IS, OS;
};

// Which torch-layout weight element feeds gemm element (a = gathered channel, b = produced
// channel, ky, kx):  w[a*sa + b*sb + ky*k + kx].
struct WeightMap {
  int64_t sa, sb;
};

// Forward of Conv2d or ConvTranspose2d.
static inline ConvGeom bp_geom_forward(const bp_conv* cv) {
  ConvGeom g{};
  g.k = cv->k; g.stride = cv->stride; g.pad = cv->pad;
  g.cin_g = cv->cin; g.cout_g = cv->cout;
  if (!cv->transposed) {
    g.gather_transposed = 0; g.nphase = 1; g.taps = cv->k; g.IS = cv->stride; g.OS = 1;
  } else {
    g.gather_transposed = 1; g.nphase = cv->stride; g.taps = (cv->k + cv->stride - 1) / cv->stride;
    g.IS = 1; g.OS = cv->stride;
  }
  return g;
}
// Data gradient: the roles of the channel counts swap and the form flips.
static inline ConvGeom bp_geom_backward_data(const bp_conv* cv) {
  ConvGeom g{};
  g.k = cv->k; g.stride = cv->stride; g.pad = cv->pad;
  g.cin_g = cv->cout; g.cout_g = cv->cin;
  if (!cv->transposed) {
    g.gather_transposed = 1; g.nphase = cv->stride; g.taps = (cv->k + cv->stride - 1) / cv->stride;
    g.IS = 1; g.OS = cv->stride;
  } else {
    g.gather_transposed = 0; g.nphase = 1; g.taps = cv->k; g.IS = cv->stride; g.OS = 1;
  }
  return g;
}
static inline WeightMap bp_wmap(const bp_conv* cv, int dir) {
  const int64_t K = (int64_t)cv->k * cv->k;
  WeightMap m{};
  if (!cv->transposed) {           // [cout][cin][k][k]
    if (dir == BP_PACK_FWD) { m.sa = K; m.sb = (int64_t)cv->cin * K; }   // a = ci, b = co
    else { m.sa = (int64_t)cv->cin * K; m.sb = K; }                      // a = co, b = ci
  } else {                         // [cin][cout][k][k]
    if (dir == BP_PACK_FWD) { m.sa = (int64_t)cv->cout * K; m.sb = K; }  // a = ci, b = co
    else { m.sa = K; m.sb = (int64_t)cv->cout * K; }                     // a = co, b = ci
  }
  return m;
}

// Output extent of the module's forward.
static inline int bp_conv_out_extent(const bp_conv* cv, int in) {
  if (!cv->transposed) return (in + 2 * cv->pad - cv->k) / cv->stride + 1;
  return (in - 1) * cv->stride - 2 * cv->pad + cv->k + cv->out_pad;
}

// For the transposed form, phase p (0..stride-1) of the produced grid:
//   residue r = (p + pad) % stride selects taps ky = r + stride*j (j = 0..taps-1, ky < k),
//   gathered row = q + c - j with c = (p + pad) / stride.
// Written as a correlation with ascending tap index t = taps-1-j:
//   gathered row = q + (c - (taps-1)) + t,  ky(t) = r + stride*(taps-1-t).
__host__ __device__ __forceinline__ int bp_t_i0(int p, int pad, int stride, int taps) {
  return (p + pad) / stride - (taps - 1);
}
__host__ __device__ __forceinline__ int bp_t_ky(int p, int pad, int stride, int taps, int t) {
  return (p + pad) % stride + stride * (taps - 1 - t);
}
