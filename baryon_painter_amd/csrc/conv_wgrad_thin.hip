// Weight gradients of the unit-stride layers that END in one channel (the tails of the generator heads:
// 8 -> 1 k5 and 1 -> 1 k3 at full resolution), fp32.  Both are bound by reading X and dY once; the tap-packed
// kernel of conv_wgrad_small.hip spent 0.46 ms (three 16x16x4 MFMAs per 4-pixel k-step, staging not overlapped) and
// 0.23 ms on them against HBM floors of 0.12 and 0.03 ms (batch 64 of 512 x 512).
//
//   dW[ky][kx][cx] = sum_{n,r,x} act(X)[n, r+ky-p, x+kx-p, cx] * Y[n, r, x]
//
// (1) wgrad_cy1_kernel<K, CXS, BH>: one MFMA per k-step.  Rows of the 16x16 tile are (tpx, cx) -- TPM = 16/CXS
//     neighbouring x-taps of all channels, 16 CONSECUTIVE floats of the NHWC tile -- and columns are (ky, gx): every
//     y-tap and every group of TPM x-taps, moved from X to Y by substituting r' = r + ky, x' = x + TPM*gx:
//         A[(tpx,cx)][(r',x')] = X[r'-p][x'+tpx-p][cx]        B[(r',x')][(ky,gx)] = Y[r'-ky][x'-TPM*gx]
//     with Y zero outside the tile.  K*ceil(K/TPM) <= 16 columns (k5, 8 channels: 15).  The next tile's global loads
//     are issued before the k-steps of the current one and land in registers behind them.
// (2) wgrad_c1_kernel<K>: one channel on both sides -- 9 outputs: no matrix unit.  A thread walks a 4-pixel-wide
//     column band with the K input rows it needs in registers and K*K fp32 accumulators; blocks are reduced through
//     LDS.  Each input row is loaded once per band (plus K-1 halo rows per 16).
// Partials go to the workspace layout of conv_wgrad.hip ([split][ky][kx][cy][cx]) and are reduced there in fixed
// order.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct ThinArgs {
  const float* X; int xh, xw, xcs, xco, cx;
  const float* Y; int yh, yw, ycs, yco;
  int n, pad;
  PW pwx, pwy;
  float* ws;
  int nsplit, tiles_x, tiles_y;
  int xvec;
  // single-channel kernel
  int qpr, nbands;
  int64_t total;
};

// ------------------------------------------------------------------------------------------ (1) cy = 1, MFMA
template <int K, int CXS, int BH>
struct Cy1Cfg {
  static constexpr int BW = 32;
  static constexpr int TPM = 16 / CXS;
  static constexpr int GXN = (K + TPM - 1) / TPM;
  static_assert(K * GXN <= 16, "columns (ky, gx) must fit one MFMA tile");
  static constexpr int RS = BH + K - 1;                     // k-step rows r'
  static constexpr int XE = (GXN - 1) * TPM;                // k-step columns beyond the tile
  static constexpr int NG = (BW + XE + 3) / 4;              // 4-pixel groups per row
  static constexpr int XW = 4 * NG + TPM - 1;               // X columns staged
  static constexpr int XP = XW + 1;                         // X row pitch in pixels
  static constexpr int YL = XE;                             // zero columns left of the Y tile
  static constexpr int YP = 40;                             // Y row pitch: (ky rows) x (8-wide windows) on distinct banks
  static_assert(YL + 4 * NG <= YP && BW == 32, "Y pitch");
  static constexpr int YR = BH + 2 * (K - 1);
  static constexpr int C4 = CXS / 4;
  static constexpr int NUX = RS * XW * C4;                  // float4 units of an X tile
  static constexpr int UX = (NUX + 255) / 256;
  static constexpr int NUY = BH * BW;
  static constexpr int UY = (NUY + 255) / 256;
  static constexpr int XF = RS * XP * CXS, YF = YR * YP;
  static constexpr size_t LDS = (size_t)(XF + YF) * 4;
};

template <int K, int CXS, int BH>
__global__ __launch_bounds__(256) void wgrad_cy1_kernel(ThinArgs a) {
  using C = Cy1Cfg<K, CXS, BH>;
  constexpr int TPM = C::TPM, GXN = C::GXN, RS = C::RS, NG = C::NG, XW = C::XW, XP = C::XP, YP = C::YP, C4 = C::C4;
  constexpr int UX = C::UX, UY = C::UY;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + C::XF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int split = blockIdx.x;

  for (int e = tid; e < C::YF; e += 256) ys[e] = 0.f;       // the zero frame stays; the interior is rewritten per tile
  // the pad column of each X row is never read

  // this thread's staging units: tile-independent positions
  int xr[UX], xc[UX];
  const int ch = (tid % C4) * 4;
#pragma unroll
  for (int j = 0; j < UX; ++j) {
    const int e = min(tid + j * 256, C::NUX - 1);
    const int pix = e / C4;
    xr[j] = pix / XW; xc[j] = pix % XW;
  }
  const PW4 px4 = pw4_load(a.pwx, ch, a.cx);
  const bool yon = a.pwy.scale != nullptr;
  const float ysc = yon ? a.pwy.scale[0] : 1.f, ysf = yon ? a.pwy.shift[0] : 0.f, ysl = yon ? a.pwy.slope[0] : 1.f;

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.n * tiles_per_img;
  float4 xv[UX];
  float yv[UY];
  unsigned okx = 0, oky = 0;

  // offsets of the units inside an interior tile (no clamping, every unit valid): one add per load
  int xoff[UX], yoff[UY];
#pragma unroll
  for (int j = 0; j < UX; ++j) xoff[j] = (xr[j] * a.xw + xc[j]) * a.xcs + ch;
#pragma unroll
  for (int j = 0; j < UY; ++j) {
    const int e = min(tid + j * 256, C::NUY - 1);
    yoff[j] = ((e >> 5) * a.yw + (e & 31)) * a.ycs;
  }
  const bool fastx = a.xvec && a.cx == CXS;

  auto issue = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int qy0 = ty_ * BH, qx0 = tx_ * C::BW;
    if (fastx && qy0 >= a.pad && qx0 >= a.pad && qy0 - a.pad + RS <= a.xh && qx0 - a.pad + XW <= a.xw &&
        qy0 + BH <= a.yh && qx0 + C::BW <= a.yw) {
      const float* Xt = a.X + (((int64_t)n * a.xh + (qy0 - a.pad)) * a.xw + (qx0 - a.pad)) * a.xcs + a.xco;
      const float* Yt = a.Y + (((int64_t)n * a.yh + qy0) * a.yw + qx0) * a.ycs + a.yco;
#pragma unroll
      for (int j = 0; j < UX; ++j) xv[j] = *reinterpret_cast<const float4*>(Xt + xoff[j]);
#pragma unroll
      for (int j = 0; j < UY; ++j) yv[j] = Yt[yoff[j]];
      okx = oky = ~0u;
      return;
    }
    const float* Xn = a.X + (int64_t)n * a.xh * a.xw * a.xcs + a.xco + (ch < a.cx ? ch : 0);
    okx = 0;
#pragma unroll
    for (int j = 0; j < UX; ++j) {
      const int iy = qy0 - a.pad + xr[j], ix = qx0 - a.pad + xc[j];
      if (ch < a.cx && iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw) okx |= 1u << j;
      const int cy = min(max(iy, 0), a.xh - 1), cx_ = min(max(ix, 0), a.xw - 1);
      const float* p = Xn + ((int64_t)cy * a.xw + cx_) * a.xcs;
      if (a.xvec) xv[j] = *reinterpret_cast<const float4*>(p);
      else {
        // (channels past the view are masked below; clamp their addresses into it)
        xv[j].x = p[0];
        xv[j].y = p[ch + 1 < a.cx ? 1 : 0];
        xv[j].z = p[ch + 2 < a.cx ? 2 : 0];
        xv[j].w = p[ch + 3 < a.cx ? 3 : 0];
      }
    }
    const float* Yn = a.Y + (int64_t)n * a.yh * a.yw * a.ycs + a.yco;
    oky = 0;
#pragma unroll
    for (int j = 0; j < UY; ++j) {
      const int e = tid + j * 256;
      const int r = e >> 5, c = e & 31;
      const int qy = qy0 + r, qx = qx0 + c;
      if (e < C::NUY && qy < a.yh && qx < a.yw) oky |= 1u << j;
      yv[j] = Yn[((int64_t)min(qy, a.yh - 1) * a.yw + min(qx, a.yw - 1)) * a.ycs];
    }
  };

  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < UX; ++j) {
      if (tid + j * 256 < C::NUX) {
        float4 w = pw4_apply4(px4, xv[j]);
        if (ch + 1 >= a.cx) w.y = 0.f;
        if (ch + 2 >= a.cx) w.z = 0.f;
        if (ch + 3 >= a.cx) w.w = 0.f;
        if (!((okx >> j) & 1)) w = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xs + (xr[j] * XP + xc[j]) * CXS + ch) = w;
      }
    }
#pragma unroll
    for (int j = 0; j < UY; ++j) {
      const int e = tid + j * 256;
      if (e < C::NUY) {
        float v = yv[j];
        if (yon) { v = fmaf(v, ysc, ysf); v = v > 0.f ? v : v * ysl; }
        ys[((K - 1) + (e >> 5)) * YP + C::YL + (e & 31)] = ((oky >> j) & 1) ? v : 0.f;
      }
    }
  };

  v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // A: 16 consecutive floats from pixel (r', 4g + kq);  B: column (ky, gx) = li -> Y[r' - ky][4g + kq - TPM*gx]
  const int a_lane = kq * CXS + li;
  const int jky = li / GXN, jgx = li % GXN;
  const bool jok = li < K * GXN;
  const int b_lane = jok ? ((K - 1) - jky) * YP + C::YL + kq - TPM * jgx : 0;

  int tile = split;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += a.nsplit) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + a.nsplit < ntiles) issue(tile + a.nsplit);
    for (int r = wk; r < RS; r += 4) {
      const float* xp = xs + r * XP * CXS + a_lane;
      const float* yp = ys + r * YP + b_lane;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float af = xp[4 * g * CXS];
        float bf = yp[4 * g];
        bf = jok ? bf : 0.f;
        if (g & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc0, 0, 0, 0);
      }
    }
  }

  // ---- reduce the 4 waves: D[row = 4*(lane>>4) + r : (tpx, cx)][col = lane & 15 : (ky, gx)]
  __syncthreads();
  float* red = smem;   // [4][64][4]
  const v4f acc = acc0 + acc1;
  *reinterpret_cast<float4*>(red + (wk * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  {
    const int l = tid >> 2, rr = tid & 3;
    const float sum = ((red[(0 * 64 + l) * 4 + rr] + red[(1 * 64 + l) * 4 + rr]) + red[(2 * 64 + l) * 4 + rr]) +
                      red[(3 * 64 + l) * 4 + rr];
    const int i = 4 * (l >> 4) + rr, j = l & 15;
    const int ky = j / GXN, kx = (j % GXN) * TPM + i / CXS, cxi = i % CXS;
    if (ky < K && kx < K) a.ws[(((int64_t)split * K + ky) * K + kx) * CXS + cxi] = sum;
  }
}

// ------------------------------------------------------------------------------------------ (2) cx = cy = 1
constexpr int C1_BR = 16;    // rows per band

template <int K>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(ThinArgs a) {
  constexpr int W = K + 3;
  __shared__ float red[4][K * K];
  const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
  const int64_t gidx = (int64_t)blockIdx.x * 256 + tid;
  const bool live = gidx < a.total;
  const int64_t gi = live ? gidx : 0;
  const int q = (int)(gi % a.qpr);
  const int band = (int)((gi / a.qpr) % a.nbands);
  const int img = (int)(gi / ((int64_t)a.qpr * a.nbands));
  const int x0 = 4 * q, r0 = band * C1_BR;
  const float* Xn = a.X + (int64_t)img * a.xh * a.xw * a.xcs + a.xco;
  const float* Yn = a.Y + (int64_t)img * a.yh * a.yw * a.ycs + a.yco;
  const bool xon = a.pwx.scale != nullptr, yon = a.pwy.scale != nullptr;
  const float xsc = xon ? a.pwx.scale[0] : 1.f, xsf = xon ? a.pwx.shift[0] : 0.f, xsl = xon ? a.pwx.slope[0] : 1.f;
  const float ysc = yon ? a.pwy.scale[0] : 1.f, ysf = yon ? a.pwy.shift[0] : 0.f, ysl = yon ? a.pwy.slope[0] : 1.f;

  float acc[K][K];
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[i][j] = 0.f;

  float xw[K][W];
  auto load_row = [&](int iy, float* dst) {
    const bool rok = live && iy >= 0 && iy < a.xh;
    const float* row = Xn + (int64_t)min(max(iy, 0), a.xh - 1) * a.xw * a.xcs;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int ix = x0 - a.pad + j;
      float v = row[(int64_t)min(max(ix, 0), a.xw - 1) * a.xcs];
      if (xon) { v = fmaf(v, xsc, xsf); v = v > 0.f ? v : v * xsl; }
      dst[j] = (rok && ix >= 0 && ix < a.xw) ? v : 0.f;
    }
  };
#pragma unroll
  for (int j = 0; j < K - 1; ++j) load_row(r0 - a.pad + j, xw[j]);
  const int rend = min(r0 + C1_BR, a.yh);
  for (int r = r0; r < rend; ++r) {
    load_row(r - a.pad + K - 1, xw[K - 1]);
    float y[4];
    const float* yrow = Yn + (int64_t)r * a.yw * a.ycs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x0 + j;
      float v = yrow[(int64_t)min(x, a.yw - 1) * a.ycs];
      if (yon) { v = fmaf(v, ysc, ysf); v = v > 0.f ? v : v * ysl; }
      y[j] = (live && x < a.yw) ? v : 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ky][kx] = fmaf(xw[ky][j + kx], y[j], acc[ky][kx]);
#pragma unroll
    for (int i = 0; i < K - 1; ++i)
#pragma unroll
      for (int j = 0; j < W; ++j) xw[i][j] = xw[i + 1][j];
  }
  // ---- block sum in a fixed order: butterfly inside the wave, then the four waves
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      float t = acc[ky][kx];
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) t += __shfl_down(t, s, 64);
      if (lane == 0) red[wk][ky * K + kx] = t;
    }
  __syncthreads();
  if (tid < K * K) a.ws[(int64_t)blockIdx.x * K * K + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

template <int K, int CXS, int BH>
int launch_cy1(const ThinArgs& a0, float* ws, size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp,
               hipStream_t st, bool dry) {
  using C = Cy1Cfg<K, CXS, BH>;
  static_assert(C::LDS <= 64 * 1024, "LDS budget");
  ThinArgs a = a0;
  a.tiles_x = bp_ceil_div(a.yw, C::BW);
  a.tiles_y = bp_ceil_div(a.yh, BH);
  const int64_t ntiles = (int64_t)a.n * a.tiles_x * a.tiles_y;
  static const int cap = getenv("BP_THIN_NSPLIT") ? atoi(getenv("BP_THIN_NSPLIT")) : 1280;     // five workgroups per CU
  a.nsplit = (int)(ntiles < cap ? ntiles : cap);
  *need = (size_t)a.nsplit * K * K * CXS * sizeof(float);
  *nsplit = a.nsplit; *cxp = CXS; *cyp = 1;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.ws = ws;
  hipLaunchKernelGGL((wgrad_cy1_kernel<K, CXS, BH>), dim3((unsigned)a.nsplit), dim3(256), C::LDS, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int K>
int launch_c1(const ThinArgs& a0, float* ws, size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp,
              hipStream_t st, bool dry) {
  ThinArgs a = a0;
  a.qpr = bp_ceil_div(a.yw, 4);
  a.nbands = bp_ceil_div(a.yh, C1_BR);
  a.total = (int64_t)a.n * a.nbands * a.qpr;
  const int64_t nblk = (a.total + 255) / 256;
  if (nblk > 65535) return BP_EUNSUPPORTED;       // (the reduction's split count)
  a.nsplit = (int)nblk;
  *need = (size_t)a.nsplit * K * K * sizeof(float);
  *nsplit = a.nsplit; *cxp = 1; *cyp = 1;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.ws = ws;
  hipLaunchKernelGGL(wgrad_c1_kernel<K>, dim3((unsigned)nblk), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

}  // namespace

// BP_EUNSUPPORTED unless: stride 1, one produced channel (Y), and an instantiation below.
int bp_wgrad_thin(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                  size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  static const bool off = getenv("BP_NOTHIN") != nullptr;
  if (off || cv->stride != 1 || Y->c != 1) return BP_EUNSUPPORTED;
  ThinArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff;
  a.n = X->n; a.pad = cv->pad; a.pwx = pwx; a.pwy = pwy;
  a.xvec = bp_view_vec4(X) ? 1 : 0;
  if (X->c == 1) {
    if (cv->k == 3) return launch_c1<3>(a, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
    if (cv->k == 5) return launch_c1<5>(a, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
    return BP_EUNSUPPORTED;
  }
  if (cv->k == 5 && X->c > 4 && X->c <= 8) return launch_cy1<5, 8, 16>(a, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  return BP_EUNSUPPORTED;
}
