// Weight gradient of unit-stride convolutions with FEW channels on one or both sides
// (the full-resolution ends of the generator: 3->16 k5, 16->8 k7, 8->1 k5, 1->1 k3), fp32 MFMA 16x16x4.
//
//   dW[ky][kx][cx][cy] = sum_{n,r,x} act(X)[n, r+ky-p, x+kx-p, cx] * Y[n, r, x, cy]
//
// A 16x16 MFMA tile would be mostly padding if M = cx and N = cy.  Instead the tile's rows and
// columns are filled with kernel taps:
//   M row    i = (tpx, cx):  TPM = 16/CXS neighbouring x-taps  -> A[i][pixel] = X[r'+ky0-p][x+kx0+tpx-p][cx]
//   N column j = (tpy, cy):  TPN = 16/CYS neighbouring y-taps  -> B[pixel][j] = Y[r'-tpy][x][cy]
// (substituting r' = r + tpy moves the y-tap from X to Y, so one B fragment serves every tap group).
// With the tiles stored compactly in LDS ([row][x][CXS] / [row][x][CYS]) both fragments are 16
// CONSECUTIVE floats per pixel: conflict-free b32 reads, no per-lane gather tables.  Y gets TPN-1
// zero rows above and TPNe-1 below so that shifted reads outside the tile's own rows contribute 0.
// MFMAs per 4-pixel k-step: ceil(k/TPM)*ceil(k/TPN) instead of k*k (49 -> 28, 25 -> 10, 25 -> 3, 9 -> 1).
//
// Workgroup = 4 waves splitting the k-steps of a BH x 32 pixel tile; pixel tiles are split over
// workgroups; partials go to the workspace layout of conv_wgrad.hip and are reduced there.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct WsArgs {
  const float* X; int xh, xw, xcs, xco, cx;
  const float* Y; int yh, yw, ycs, yco, cy;
  int n, pad;
  PW pwx, pwy;
  float* ws;
  int nsplit, tiles_x, tiles_y;
  int xvec, yvec;
};

template <int K, int S, int CXS, int CYS, int BH>
struct WsCfg {
  static constexpr int TPM = 16 / CXS, TPN = 16 / CYS;
  static constexpr int GX = (K + TPM - 1) / TPM;
  // y-taps: ky = gy + S*(h*TPN + tpy): residue gy (0..S-1), TPN shifted rows per MFMA column set, H sets
  static constexpr int KS = (K + S - 1) / S;               // taps per residue
  static constexpr int H = (KS + TPN - 1) / TPN;
  static constexpr int GY = S * H;
  static constexpr int TPNe = TPN < KS ? TPN : KS;
  static constexpr int RS = BH + TPNe - 1;                 // k-step rows per tile
  static constexpr int XR = S * (RS - 1) + (S - 1) + S * (H - 1) * TPN + 1;   // X rows staged
  static constexpr int XW = S * 31 + GX * TPM;             // X columns staged
  static constexpr int XWS = XW + 3 * S;                   // slack for the discarded lanes of the last group
  static constexpr int PADR = TPN - 1;                     // zero rows above the Y tile
  static constexpr int YR = PADR + BH + TPNe - 1;
  static constexpr int YWS = 34;                           // row pitch: keeps the shifted B reads conflict-free
  static constexpr int XF = XR * XWS * CXS, YF = YR * YWS * CYS;
  static constexpr size_t LDS_MAIN = (size_t)(XF + YF) * 4;
  static constexpr size_t LDS_RED = (size_t)4 * 64 * 4 * 4;
  static constexpr size_t LDS = LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED;
};

template <int K, int S, int CXS, int CYS, int BH>
__global__ __launch_bounds__(256) void wgrad_small_kernel(WsArgs a) {
  using C = WsCfg<K, S, CXS, CYS, BH>;
  constexpr int TPM = C::TPM, TPN = C::TPN, GX = C::GX, GY = C::GY, RS = C::RS, XR = C::XR, XW = C::XW, H = C::H;
  constexpr int XWS = C::XWS, PADR = C::PADR, YWS = C::YWS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + C::XF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int split = blockIdx.x;

  v4f acc[GY][GX];
#pragma unroll
  for (int i = 0; i < GY; ++i)
#pragma unroll
    for (int j = 0; j < GX; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  // zero the whole Y image once: the pad rows stay zero, the interior is rewritten per tile
  for (int e = tid; e < C::YF; e += 256) ys[e] = 0.f;

  const int a_lane = kq * S * CXS + li;                             // (4 pixels, S apart) x (16 consecutive floats)
  const int b_lane = (kq - (li / CYS) * YWS) * CYS + (li % CYS);    // column (tpy, cy): row shifted up by tpy

  const PW4 px4 = pw4_load(a.pwx, (tid % (CXS >= 4 ? CXS / 4 : 1)) * 4, a.cx);
  const PW4 py4 = pw4_load(a.pwy, (tid % (CYS >= 4 ? CYS / 4 : 1)) * 4, a.cy);
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.n * tiles_per_img;

  // The 3*S slack columns of each X row are only read by k-steps whose outputs are discarded, but must hold finite
  // numbers: zeroed once, never written again.
  for (int e = tid; e < XR * 3 * S * CXS; e += 256) {
    const int ch = e % CXS, c = XW + (e / CXS) % (3 * S), r = e / (3 * S * CXS);
    xs[(r * XWS + c) * CXS + ch] = 0.f;
  }

  // Staging in two halves: issue() loads a tile's X and Y into registers (clamped addresses, validity in bit masks),
  // commit() applies the pending activations and writes LDS.  The loop below issues tile t+1 before the k-steps of
  // tile t, so the loads fly behind the MFMAs instead of in front of them (this kernel used to spend as long waiting
  // for its three or four dependent load trips per tile as computing).
  constexpr bool XV = CXS >= 4, YV = CYS >= 4;
  constexpr int C4X = XV ? CXS / 4 : 1, C4Y = YV ? CYS / 4 : 1;
  constexpr int NUX = XV ? XR * XW * C4X : XR * XW * CXS;       // float4 units (vector path) / floats (scalar path)
  constexpr int NUY = YV ? BH * 32 * C4Y : BH * 32 * CYS;
  constexpr int UX = (NUX + 255) / 256, UY = (NUY + 255) / 256;
  static_assert(UX <= 32 && UY <= 32, "validity masks");
  float4 xv4[XV ? UX : 1];
  float xv1[XV ? 1 : UX];
  float4 yv4[YV ? UY : 1];
  float yv1[YV ? 1 : UY];
  unsigned okx = 0, oky = 0;
  const bool xvec = XV && a.xvec, yvec = YV && a.yvec;

  auto issue = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int qy0 = ty_ * BH, qx0 = tx_ * 32;
    const float* Xn = a.X + (int64_t)n * a.xh * a.xw * a.xcs + a.xco;
    const float* Yn = a.Y + (int64_t)n * a.yh * a.yw * a.ycs + a.yco;
    okx = oky = 0;
    if (XV) {
      const int ch = (tid % C4X) * 4;
      const int chs = ch < a.cx ? ch : 0;                  // (a quad past the view is zeroed in commit: any address inside it)
#pragma unroll
      for (int j = 0; j < UX; ++j) {
        const int e = min(tid + j * 256, NUX - 1);
        const int pix = e / C4X;
        const int c = pix % XW, r = pix / XW;
        const int iy = S * qy0 - a.pad + r, ix = S * qx0 - a.pad + c;
        if (ch < a.cx && iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw) okx |= 1u << j;
        const float* p = Xn + ((int64_t)min(max(iy, 0), a.xh - 1) * a.xw + min(max(ix, 0), a.xw - 1)) * a.xcs + chs;
        if (xvec) xv4[XV ? j : 0] = *reinterpret_cast<const float4*>(p);
        else      // (unaligned view: channels past it are masked in commit; clamp their addresses into it)
          xv4[XV ? j : 0] = make_float4(p[0], p[chs + 1 < a.cx ? 1 : 0], p[chs + 2 < a.cx ? 2 : 0], p[chs + 3 < a.cx ? 3 : 0]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < UX; ++j) {
        const int e = min(tid + j * 256, NUX - 1);
        const int ch = e % CXS;
        const int pix = e / CXS;
        const int c = pix % XW, r = pix / XW;
        const int iy = S * qy0 - a.pad + r, ix = S * qx0 - a.pad + c;
        if (ch < a.cx && iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw) okx |= 1u << j;
        xv1[XV ? 0 : j] = Xn[((int64_t)min(max(iy, 0), a.xh - 1) * a.xw + min(max(ix, 0), a.xw - 1)) * a.xcs + (ch < a.cx ? ch : 0)];
      }
    }
    if (YV) {
      const int ch = (tid % C4Y) * 4;
      const int chs = ch < a.cy ? ch : 0;
#pragma unroll
      for (int j = 0; j < UY; ++j) {
        const int e = min(tid + j * 256, NUY - 1);
        const int pix = e / C4Y;
        const int c = pix & 31, r = pix >> 5;
        const int qy = qy0 + r, qx = qx0 + c;
        if (ch < a.cy && qy < a.yh && qx < a.yw) oky |= 1u << j;
        const float* p = Yn + ((int64_t)min(qy, a.yh - 1) * a.yw + min(qx, a.yw - 1)) * a.ycs + chs;
        if (yvec) yv4[YV ? j : 0] = *reinterpret_cast<const float4*>(p);
        else
          yv4[YV ? j : 0] = make_float4(p[0], p[chs + 1 < a.cy ? 1 : 0], p[chs + 2 < a.cy ? 2 : 0], p[chs + 3 < a.cy ? 3 : 0]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < UY; ++j) {
        const int e = min(tid + j * 256, NUY - 1);
        const int ch = e % CYS;
        const int pix = e / CYS;
        const int c = pix & 31, r = pix >> 5;
        const int qy = qy0 + r, qx = qx0 + c;
        if (ch < a.cy && qy < a.yh && qx < a.yw) oky |= 1u << j;
        yv1[YV ? 0 : j] = Yn[((int64_t)min(qy, a.yh - 1) * a.yw + min(qx, a.yw - 1)) * a.ycs + (ch < a.cy ? ch : 0)];
      }
    }
  };

  auto commit = [&]() {
    if (XV) {
      const int ch = (tid % C4X) * 4;
#pragma unroll
      for (int j = 0; j < UX; ++j) {
        const int e = tid + j * 256;
        if (e < NUX) {
          const int pix = e / C4X;
          const int c = pix % XW, r = pix / XW;
          float4 w = pw4_apply4(px4, xv4[XV ? j : 0]);
          if (ch + 1 >= a.cx) w.y = 0.f;
          if (ch + 2 >= a.cx) w.z = 0.f;
          if (ch + 3 >= a.cx) w.w = 0.f;
          *reinterpret_cast<float4*>(xs + (r * XWS + c) * CXS + ch) = ((okx >> j) & 1) ? w : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < UX; ++j) {
        const int e = tid + j * 256;
        if (e < NUX) {
          const int ch = e % CXS;
          const int pix = e / CXS;
          const int c = pix % XW, r = pix / XW;
          xs[(r * XWS + c) * CXS + ch] = ((okx >> j) & 1) ? pw_apply(a.pwx, ch, xv1[XV ? 0 : j]) : 0.f;
        }
      }
    }
    if (YV) {
      const int ch = (tid % C4Y) * 4;
#pragma unroll
      for (int j = 0; j < UY; ++j) {
        const int e = tid + j * 256;
        if (e < NUY) {
          const int pix = e / C4Y;
          const int c = pix & 31, r = pix >> 5;
          float4 w = pw4_apply4(py4, yv4[YV ? j : 0]);
          if (ch + 1 >= a.cy) w.y = 0.f;
          if (ch + 2 >= a.cy) w.z = 0.f;
          if (ch + 3 >= a.cy) w.w = 0.f;
          *reinterpret_cast<float4*>(ys + ((PADR + r) * YWS + c) * CYS + ch) = ((oky >> j) & 1) ? w : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < UY; ++j) {
        const int e = tid + j * 256;
        if (e < NUY) {
          const int ch = e % CYS;
          const int pix = e / CYS;
          const int c = pix & 31, r = pix >> 5;
          ys[((PADR + r) * YWS + c) * CYS + ch] = ((oky >> j) & 1) ? pw_apply(a.pwy, ch, yv1[YV ? 0 : j]) : 0.f;
        }
      }
    }
  };

  // (with 32 accumulator tiles -- k8 stride 4, 8 x 16 channels -- the prefetch registers cost the second wave per SIMD
  //  and more than the overlap wins: 0.10 -> 0.15 ms; such shapes load and commit back to back)
  constexpr bool PIPE = GY * GX <= 16;
  int tile = split;
  if (PIPE && tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += a.nsplit) {
    if (!PIPE) issue(tile);
    __syncthreads();
    commit();
    __syncthreads();
    if (PIPE && tile + a.nsplit < ntiles) issue(tile + a.nsplit);
    // ---- k-steps: row r' (0..RS-1), pixel group g (0..7); this wave takes every 4th
    for (int s = wk; s < RS * 8; s += 4) {
      const int r = s >> 3, g = s & 7;
      const float bf = ys[((PADR + r) * YWS + 4 * g) * CYS + b_lane];
      const float* xp = xs + (S * r * XWS + S * 4 * g) * CXS + a_lane;
#pragma unroll
      for (int gy = 0; gy < GY; ++gy)
#pragma unroll
        for (int gx = 0; gx < GX; ++gx) {
          // group gy = (residue gy % S, set gy / S): X row offset = residue + S*set*TPN
          const float af = xp[(((gy % S) + S * (gy / S) * TPN) * XWS + gx * TPM) * CXS];
          acc[gy][gx] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[gy][gx], 0, 0, 0);
        }
    }
  }

  // ---- reduce the 4 waves and write: D[row = 4*(lane>>4)+r : (tpx,cx)][col = lane&15 : (tpy,cy)]
  float* red = smem;   // [4][64][4]
#pragma unroll
  for (int gy = 0; gy < GY; ++gy)
#pragma unroll
    for (int gx = 0; gx < GX; ++gx) {
      __syncthreads();
      *reinterpret_cast<float4*>(red + (wk * 64 + lane) * 4) =
          make_float4(acc[gy][gx][0], acc[gy][gx][1], acc[gy][gx][2], acc[gy][gx][3]);
      __syncthreads();
      {
        const int l = tid >> 2, rr = tid & 3;          // 64 lanes x 4 registers
        const float sum = ((red[(0 * 64 + l) * 4 + rr] + red[(1 * 64 + l) * 4 + rr]) +
                           red[(2 * 64 + l) * 4 + rr]) + red[(3 * 64 + l) * 4 + rr];
        const int i = 4 * (l >> 4) + rr, j = l & 15;
        const int kx = gx * TPM + i / CXS, cxi = i % CXS;
        const int ky = (gy % S) + S * ((gy / S) * TPN + j / CYS), cyi = j % CYS;
        if (kx < K && ky < K)
          a.ws[((((int64_t)split * K + ky) * K + kx) * CYS + cyi) * CXS + cxi] = sum;
      }
    }
}

template <int K, int S, int CXS, int CYS, int BH>
int launch(const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, int pad, float* ws, size_t ws_bytes,
           size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  using C = WsCfg<K, S, CXS, CYS, BH>;
  static_assert(C::LDS <= 64 * 1024, "LDS budget");
  WsArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff; a.cy = Y->c;
  a.n = X->n; a.pad = pad; a.pwx = pwx; a.pwy = pwy; a.ws = ws;
  a.tiles_x = bp_ceil_div(Y->w, 32);
  a.tiles_y = bp_ceil_div(Y->h, BH);
  int64_t ntiles = (int64_t)Y->n * a.tiles_x * a.tiles_y;
  static const int cap = getenv("BP_WS_NSPLIT") ? atoi(getenv("BP_WS_NSPLIT")) : 1024;
  int64_t ns = ntiles < cap ? ntiles : cap;          // ~4 workgroups per CU
  a.nsplit = (int)ns;
  *need = (size_t)a.nsplit * K * K * CYS * CXS * sizeof(float);
  *nsplit = a.nsplit; *cxp = CXS; *cyp = CYS;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.xvec = (X->cstride % 4 == 0 && X->coff % 4 == 0 && reinterpret_cast<uintptr_t>(X->ptr) % 16 == 0) ? 1 : 0;
  a.yvec = (Y->cstride % 4 == 0 && Y->coff % 4 == 0 && reinterpret_cast<uintptr_t>(Y->ptr) % 16 == 0) ? 1 : 0;
  hipLaunchKernelGGL((wgrad_small_kernel<K, S, CXS, CYS, BH>), dim3((unsigned)a.nsplit), dim3(256), C::LDS, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int pow2_at_least(int c) { return c <= 1 ? 1 : (c <= 2 ? 2 : (c <= 4 ? 4 : (c <= 8 ? 8 : 16))); }

}  // namespace

// BP_EUNSUPPORTED unless both channel counts are <= 16, at least one is < 16 and (k, stride, padded
// channel counts) has an instantiation below.
int bp_wgrad_small(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                   size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  if (X->c > 16 || Y->c > 16 || (X->c == 16 && Y->c == 16)) return BP_EUNSUPPORTED;
  const int cxs = pow2_at_least(X->c), cys = pow2_at_least(Y->c);
#define BP_WS(K_, S_, CX_, CY_, BH_) \
  if (cv->k == K_ && cv->stride == S_ && cxs == CX_ && cys == CY_) \
    return launch<K_, S_, CX_, CY_, BH_>(X, pwx, Y, pwy, cv->pad, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry)
  BP_WS(7, 1, 16, 8, 8);
  BP_WS(5, 1, 4, 16, 8);
  BP_WS(5, 1, 8, 1, 16);
  BP_WS(3, 1, 1, 1, 16);
  BP_WS(5, 1, 16, 8, 8);
  BP_WS(3, 1, 16, 8, 8);
  BP_WS(3, 1, 8, 16, 8);
  BP_WS(3, 1, 4, 16, 8);
  BP_WS(3, 1, 8, 8, 8);
  BP_WS(5, 1, 8, 8, 8);
  BP_WS(5, 1, 1, 1, 16);
  BP_WS(7, 1, 1, 1, 16);
  // k9 stem / head of the CGAN generator, per 16-channel chunk of the wide side (conv_wgrad.hip: wide_side_chunks)
  BP_WS(9, 1, 2, 16, 8);
  BP_WS(9, 1, 1, 16, 8);
  BP_WS(9, 1, 16, 1, 8);
  BP_WS(9, 1, 16, 2, 4);
  // strided ends of the encoders / the latent up-sampler
  BP_WS(4, 2, 1, 8, 8);
  BP_WS(4, 2, 2, 8, 8);
  BP_WS(4, 2, 1, 1, 16);
  BP_WS(8, 4, 8, 16, 2);
  BP_WS(8, 4, 1, 1, 8);
#undef BP_WS
  return BP_EUNSUPPORTED;
}
