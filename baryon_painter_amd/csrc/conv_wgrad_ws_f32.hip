// Output-stationary fp32 weight gradient of the generator's residual trunk: Conv2d k3 s1 p1, 128 <-> 128 channels
// (/root/reference/baryon_painter/models/utils.py:22-38; eight layers, 5.4 ms of the fp32 step in
// wgrad_tiles_dma_kernel<3,3,1,2,2,2,2,1,3> at 0.73 of the matrix peak with 1.40 x its algorithmic HBM bytes).
//
//   dW[t][co][ci] = sum over pixels p of  act(X)[p + t][ci] * dY[p][co]
//
// The row-ring scheme of conv_ws_f32.hip turned around: what stays in registers is the OUTPUT.  A workgroup owns one
// (64 ci x 64 co) block of dW for all nine taps -- 36 accumulator tiles of v_mfma_f32_16x16x4_f32 per wave = 144 AGPRs --
// and walks an image top to bottom: a ring of four X rows (its 64 gathered channels) and two dY rows (its 64 produced
// channels) in LDS, one new row of each per produced row, every input byte fetched once per workgroup.  With fp32 MFMAs a
// lane holds ONE k element per operand, so neither operand needs a transposed read: k = 4 consecutive pixels, a lane (lm, kq)
// reads pixel kq of the quad and four consecutive channels 4 lm .. 4 lm + 3 as one float4 -- the four components are the A
// operands of four MFMA row tiles (tile j = channels {4 i + j}), and a dY float is the B operand of the wave's column tile
// (wave w: channels {4 i + w}).  Per k-step: ten 16-byte LDS reads for 36 MFMAs.  Partial sums per (image, band) go to the
// split-K workspace in the layout wgrad_reduce expects; a lane's 16 values of a tap are 16 consecutive ci: float4 stores.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WW_C = 128, WW_B = 64, WW_R = 4;

struct WwArgs {
  const float* X; int x_cs, x_co;
  const float* Y; int y_cs, y_co;
  int n, h;
  int iw, strips;                    // image width, column strips of W = 16 G pixels per row (iw = W * strips)
  PW pwx;                            // pending activation of X (this layer's input) or nullptr
  float* ws;                         // [split][tap 9][co 128][ci 128]
  int BR, bands;
  int nsplit;                        // images x bands x strips
};

template <int G> struct WwGeom {
  static constexpr int W = 16 * G, RPX = W + 2;
  static constexpr size_t x_floats = (size_t)WW_R * RPX * WW_B;          // [ring row][pixel + 2][64]
  static constexpr size_t y_floats = (size_t)2 * W * WW_B;               // [buffer 2][pixel][64]
  static constexpr size_t lds_bytes = (x_floats + y_floats + 3 * WW_B) * sizeof(float);
};

template <int G, bool ACT>
__global__ __launch_bounds__(256) void wgrad_ws_f32_kernel(WwArgs a) {
  using GM = WwGeom<G>;
  constexpr int W = GM::W, RPX = GM::RPX;
  constexpr int NU = G;                                 // 16-byte staging units per thread, row and tensor: W * 16 / 256
  constexpr int KS = W / 4;                             // k-steps (pixel quads) per row
  extern __shared__ __attribute__((aligned(16))) float smem_w[];
  float4* xs = reinterpret_cast<float4*>(smem_w);                         // 16 quads per pixel
  float4* ys = reinterpret_cast<float4*>(smem_w + GM::x_floats);
  float* ysf = smem_w + GM::x_floats;
  float* lpw = smem_w + GM::x_floats + GM::y_floats;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  // block id = split_lo + 8 * (pair + 4 * split_hi): workgroups go to the XCDs round-robin by id, so the four channel-block
  // pairs of a split -- which read the same X and dY rows -- share an XCD's L2 (577 -> ~300 MB of HBM per launch)
  const int split = (int)(blockIdx.x & 7) + 8 * (int)(blockIdx.x >> 5), pair = (blockIdx.x >> 3) & 3;
  if (split >= a.nsplit) return;
  const int cib = pair & 1, cob = pair >> 1;
  const int strip = split % a.strips, nbd = split / a.strips;
  const int n = nbd / a.bands, band = nbd % a.bands;
  const int y0 = band * a.BR;
  const int y1 = min(y0 + a.BR, a.h);
  const int x0 = strip * W;          // (wider images: the X ring's pad columns hold the neighbouring strips' pixels)

  // staging: unit i of a row = pixel (tid >> 4) + 16 i, channel quad tid & 15 (16 lanes = the 256 contiguous bytes of a pixel's half)
  const int s_q = tid & 15, s_p = tid >> 4;
  const char* x_img = reinterpret_cast<const char*>(a.X + ((int64_t)n * a.h * a.iw + x0) * a.x_cs + a.x_co + WW_B * cib);
  const char* y_img = reinterpret_cast<const char*>(a.Y + ((int64_t)n * a.h * a.iw + x0) * a.y_cs + a.y_co + WW_B * cob);
  const unsigned x_row = (unsigned)(a.iw * a.x_cs) * 4u, y_row = (unsigned)(a.iw * a.y_cs) * 4u;
  // halo columns x0 - 1 and x0 + W of an X row: threads 0..31 = (side, quad); outside the image they stay zero
  const bool h_on = a.strips > 1 && tid < 32;
  const int h_side = (tid >> 4) & 1;
  const bool h_in = h_side ? x0 + W < a.iw : x0 > 0;
  const unsigned h_keep = h_in ? 0xffffffffu : 0u;
  const int h_off = ((h_in ? (h_side ? W : -1) : 0) * a.x_cs + 4 * s_q) * 4;
  const unsigned x_off = (unsigned)(s_p * a.x_cs + 4 * s_q) * 4u, x_st = (unsigned)(16 * a.x_cs) * 4u;
  const unsigned y_off = (unsigned)(s_p * a.y_cs + 4 * s_q) * 4u, y_st = (unsigned)(16 * a.y_cs) * 4u;
  auto load_x = [&](int r, float4 (&raw)[NU], float4& hraw) {
    const char* rowp = x_img + (size_t)((unsigned)r * x_row);
#pragma unroll
    for (int i = 0; i < NU; ++i) raw[i] = *reinterpret_cast<const float4*>(rowp + (x_off + (unsigned)i * x_st));
    if (h_on) hraw = *reinterpret_cast<const float4*>(rowp + h_off);
  };
  auto load_y = [&](int r, float4 (&raw)[NU]) {
    const char* rowp = y_img + (size_t)((unsigned)r * y_row);
#pragma unroll
    for (int i = 0; i < NU; ++i) raw[i] = *reinterpret_cast<const float4*>(rowp + (y_off + (unsigned)i * y_st));
  };
  auto act4 = [&](float4 v) {
    if constexpr (ACT) {
      const float4 sc = *reinterpret_cast<const float4*>(lpw + 4 * s_q);
      const float4 sf = *reinterpret_cast<const float4*>(lpw + WW_B + 4 * s_q);
      const float4 sl = *reinterpret_cast<const float4*>(lpw + 2 * WW_B + 4 * s_q);
      float t;
      t = fmaf(v.x, sc.x, sf.x); v.x = t > 0.f ? t : t * sl.x;          // (a NaN stays a NaN, as torch.relu)
      t = fmaf(v.y, sc.y, sf.y); v.y = t > 0.f ? t : t * sl.y;
      t = fmaf(v.z, sc.z, sf.z); v.z = t > 0.f ? t : t * sl.z;
      t = fmaf(v.w, sc.w, sf.w); v.w = t > 0.f ? t : t * sl.w;
    }
    return v;
  };
  auto put_x = [&](int r, unsigned keep, const float4 (&raw)[NU], const float4& hraw) {      // X row r -> ring slot (r + 1) & 3, columns 1 .. W
    const int rr = (r + 1) & (WW_R - 1);
    auto m = [&](float f) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) & keep); };
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const float4 v = act4(raw[i]);
      xs[(rr * RPX + 1 + s_p + 16 * i) * 16 + s_q] = make_float4(m(v.x), m(v.y), m(v.z), m(v.w));
    }
    if (h_on) {
      const float4 v = act4(hraw);
      const unsigned k = keep & h_keep;
      auto mh = [&](float f) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) & k); };
      xs[(rr * RPX + (h_side ? RPX - 1 : 0)) * 16 + s_q] = make_float4(mh(v.x), mh(v.y), mh(v.z), mh(v.w));
    }
  };
  auto put_y = [&](int r, const float4 (&raw)[NU]) {                     // dY row r -> buffer r & 1
#pragma unroll
    for (int i = 0; i < NU; ++i) ys[((r & 1) * W + s_p + 16 * i) * 16 + s_q] = raw[i];
  };

  // ---- prologue
  float4 rx[NU], ry[NU];
  float4 hx = make_float4(0.f, 0.f, 0.f, 0.f), hy = hx;
  // (with strips those columns are the halo, written with every row by threads 0..31 -- zeros where the image ends: no
  //  fill then, it would race with those writes)
  if (a.strips == 1 && tid < 128) {                     // zero columns 0 and W + 1 of the four ring rows
    const int rr = tid >> 5, side = (tid >> 4) & 1, q = tid & 15;
    xs[(rr * RPX + (side ? RPX - 1 : 0)) * 16 + q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (ACT) {
    for (int i = tid; i < WW_B; i += 256) {
      lpw[i] = a.pwx.scale[WW_B * cib + i]; lpw[WW_B + i] = a.pwx.shift[WW_B * cib + i]; lpw[2 * WW_B + i] = a.pwx.slope[WW_B * cib + i];
    }
    __syncthreads();
  }
  {
    const bool in0 = y0 - 1 >= 0, in2 = y0 + 1 < a.h;
    load_x(in0 ? y0 - 1 : y0, rx, hx);
    load_y(y0, ry);
    put_x(y0 - 1, in0 ? 0xffffffffu : 0u, rx, hx);
    put_y(y0, ry);
    load_x(y0, rx, hx);
    load_x(in2 ? y0 + 1 : y0, ry, hy);                  // (ry as a second X buffer)
    put_x(y0, 0xffffffffu, rx, hx);
    put_x(y0 + 1, in2 ? 0xffffffffu : 0u, ry, hy);
  }
  __syncthreads();

  v4f acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = v4f{0.f, 0.f, 0.f, 0.f};

  const int xl = kq * 16 + lm;                          // float4 index of (pixel kq, quad lm) inside a row
  for (int y = y0; y < y1; ++y) {
    const bool more = y + 1 < y1;
    const bool in_next = y + 2 < a.h;
    load_x(in_next ? y + 2 : a.h - 1, rx, hx);
    load_y(more ? y + 1 : y, ry);
    const unsigned keep = in_next ? 0xffffffffu : 0u;
    int xb[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) xb[ty] = ((y + ty) & (WW_R - 1)) * RPX * 16 + xl;     // X row y - 1 + ty
    const int yb = (y & 1) * W * WW_B + kq * WW_B + 4 * lm + wave;                        // float index: (pixel kq, channel 4 lm + wave)
    float4 af[2][9];
    float bf[2];
#pragma unroll
    for (int t = 0; t < 9; ++t) af[0][t] = xs[xb[t / 3] + (t % 3) * 16];
    bf[0] = ysf[yb];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int t = 0; t < 9; ++t) af[(ks + 1) & 1][t] = xs[xb[t / 3] + (4 * (ks + 1) + t % 3) * 16];
        bf[(ks + 1) & 1] = ysf[yb + 4 * (ks + 1) * WW_B];
      }
      // the next rows go to LDS in the middle of the row (their loads were issued at its top); slots nobody reads now
      if (ks == KS / 2) {
        put_x(y + 2, keep, rx, hx);
        put_y(y + 1, ry);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 x = af[ks & 1][t];
        const float b = bf[ks & 1];
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, b, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, b, acc[t][3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- partial sums: D[row 4 kq + r][col lm] of tile (t, j) = dW[t][co 64 cob + 4 lm + wave][ci 64 cib + 4 (4 kq + r) + j]
  float* dst = a.ws + ((int64_t)split * 9 * WW_C + (WW_B * cob + 4 * lm + wave)) * WW_C + WW_B * cib + 16 * kq;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(dst + (int64_t)t * WW_C * WW_C + 4 * r) = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
}

bool ww_enabled() {
  static const bool off = getenv("BP_F32_WGRAD_WS") && atoi(getenv("BP_F32_WGRAD_WS")) == 0;
  return !off;
}
int g_ww_override = -1;

// strip width / 16: the image itself up to 64 pixels, 64-pixel strips of wider images (multiples of 64)
int ww_G(int w) { return w == 64 ? 4 : w == 32 ? 2 : w == 16 ? 1 : (w > 64 && w % 64 == 0) ? 4 : 0; }
int ww_strips(int w) { return w > 64 ? w / 64 : 1; }

void ww_bands(int n, int h, int strips, int* BR, int* bands) {      // ~256 workgroups = 64 splits x 4 channel-block pairs
  int br = h;
  while (br > 8 && (int64_t)n * strips * bp_ceil_div(h, br) < 64) br = bp_ceil_div(br, 2);
  *BR = br;
  *bands = bp_ceil_div(h, br);
}

template <int G, bool ACT>
int ww_launch(const WwArgs& a, unsigned grid, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ws_f32_kernel<G, ACT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)WwGeom<G>::lds_bytes);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((wgrad_ws_f32_kernel<G, ACT>), dim3(grid), dim3(256), WwGeom<G>::lds_bytes, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

}  // namespace

void bp_f32_wgrad_ws_set(int v) { g_ww_override = v; }

// Same contract as bp_wgrad_tiles (conv_wgrad_tiles.hip): BP_EUNSUPPORTED -> the next kernel in the chain.
int bp_wgrad_ws_f32(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                    size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  if (!(g_ww_override < 0 ? ww_enabled() : g_ww_override != 0)) return BP_EUNSUPPORTED;
  if (cv->transposed || cv->k != 3 || cv->stride != 1 || cv->pad != 1 || cv->cin != WW_C || cv->cout != WW_C) return BP_EUNSUPPORTED;
  if (X->dtype != BP_F32 || Y->dtype != BP_F32 || X->c != WW_C || Y->c != WW_C || pwy.scale) return BP_EUNSUPPORTED;
  if (X->n != Y->n || X->h != Y->h || X->w != Y->w || !ww_G(X->w) || !bp_view_vec4(X) || !bp_view_vec4(Y)) return BP_EUNSUPPORTED;
  if ((int64_t)X->h * X->w * X->cstride * 4 >= (int64_t)1 << 31 || (int64_t)Y->h * Y->w * Y->cstride * 4 >= (int64_t)1 << 31)
    return BP_EUNSUPPORTED;
  WwArgs a{};
  a.iw = X->w; a.strips = ww_strips(X->w);
  ww_bands(X->n, X->h, a.strips, &a.BR, &a.bands);
  const int64_t splits = (int64_t)X->n * a.bands * a.strips;
  if (splits * 4 > 0x7fffffff) return BP_EUNSUPPORTED;
  *nsplit = (int)splits; *cxp = WW_C; *cyp = WW_C;
  *need = (size_t)splits * 9 * WW_C * WW_C * sizeof(float);
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.X = X->ptr; a.x_cs = X->cstride; a.x_co = X->coff;
  a.Y = Y->ptr; a.y_cs = Y->cstride; a.y_co = Y->coff;
  a.n = X->n; a.h = X->h; a.pwx = pwx; a.ws = ws;
  a.nsplit = (int)splits;
  const bool act = pwx.scale != nullptr;
  const unsigned grid = (unsigned)(bp_ceil_div((int)splits, 8) * 32);          // (split_lo 8) x (pair 4) x split_hi
  switch (ww_G(X->w)) {
    case 4: return act ? ww_launch<4, true>(a, grid, st) : ww_launch<4, false>(a, grid, st);
    case 2: return act ? ww_launch<2, true>(a, grid, st) : ww_launch<2, false>(a, grid, st);
    default: return act ? ww_launch<1, true>(a, grid, st) : ww_launch<1, false>(a, grid, st);
  }
}
