// Weight gradient, register-blocked over kernel taps (fp32 MFMA 16x16x4), NHWC.
//
//   dst[cy][cx][ky][kx] = sum_{n,q} act(X)[n, q*S + k - p, cx] * act(Y)[n, q, cy]
//
// Compared with the generic kernel in conv_wgrad.hip this one keeps the accumulators of ALL taps
// of KHB kernel rows x NTX x NTY channel tiles in registers (up to 144 accumulator VGPRs per
// lane), so one staged pixel tile feeds KHB*KW*NTX*NTY MFMAs per k-step per wave instead of KW*NTY:
// the kernel is MFMA-bound, not LDS/L2-bound.  Everything that indexes LDS is a template constant
// (taps, stride, tile rows), so all tap offsets fold into ds_read immediates.
//
// Workgroup = 4 waves arranged WX x WY x WK: WX*NTX X-channel tiles, WY*NTY Y-channel tiles, and
// WK-way split of the k-steps (pixel groups) of a tile, summed through LDS at the end.
// Per tile of BH x 16 coarse-grid pixels the block stages
//   X rows  [CXC/16][(BH-1)*S + KHB][S][IWq][16]   (x de-interleaved by the stride -> the 4 pixels of
//   Y       [CYC/16][BH][16][16]                     an MFMA k-step are contiguous: conflict-free)
// through registers, applying the pending activation of whichever operand is a layer input.
// Partials per pixel split go to a workspace; wgrad_reduce_kernel (conv_wgrad.hip) sums them in a
// fixed order (bitwise reproducible).
#include "common.hpp"
#include <cstdlib>

// Workgroup-count target of the next launches on this thread (0: the default).  bp_wgrad_mfma sets it for a launch that
// runs on a side stream beside the data-gradient chain (BP_IMPL_SHARED): measured on the fiducial step, 512 workgroups
// (two per CU, the best alone) is the WORST choice there -- 320 ... 448 and 640 ... 768 all give a shorter step, 448 the
// shortest (43.4 -> 42.9 ms): a grid that does not fill every CU twice leaves room for the main chain's workgroups.
static thread_local int t_wt_target = 0;
void bp_wgrad_tiles_target(int target) { t_wt_target = target; }

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct WtArgs {
  const float* X; int xh, xw, xcs, xco, cx;
  const float* Y; int yh, yw, ycs, yco, cy;
  int n, k, pad;
  PW pwx, pwy;
  float* ws;
  int ncxb, nsplit, tiles_x, tiles_y, CXP, CYP;
  int xvec, yvec;
};

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int WK, int BH>
struct WtCfg {
  static constexpr int CXC = 16 * NTX * WX;
  static constexpr int CYC = 16 * NTY * WY;
  static constexpr int XR = (BH - 1) * S + KHB;
  static constexpr int IW = 15 * S + KW;
  static constexpr int IWq = (IW + S - 1) / S;
  static constexpr int XT = XR * S * IWq * 16;    // floats per X channel tile
  static constexpr int YT = BH * 16 * 16;         // floats per Y channel tile
  static constexpr int TAPS = KHB * KW;
  static constexpr int STEPS = BH * 4 / WK;
  static constexpr size_t LDS_MAIN = (size_t)(CXC / 16 * XT + CYC / 16 * YT) * 4;
  static constexpr size_t LDS_RED = (WK > 1) ? (size_t)WK * NTX * NTY * 64 * 4 * 4 : 0;
  static constexpr size_t LDS = LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED;
};

// XCD-aware workgroup -> (channel-block pair, tap-row group, pixel split) mapping.  Workgroups go to the 8 XCDs
// round-robin by linear id; the channel-block pairs and tap-row groups of ONE split read the same pixel tiles, so
// they are made consecutive members of one XCD's share (each XCD has its own L2) instead of neighbours on
// different XCDs: the tiles then come from HBM once instead of once per pair.
__device__ __forceinline__ void wt_block_coords(int* bx, int* by, int* bz) {
  const int gx = gridDim.x, gy = gridDim.y;
  const int n = gx * gy * gridDim.z;
  const int L = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
  const int q = n >> 3, r = n & 7;
  const int xcd = L & 7, idx = L >> 3;
  const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  *bx = Lp % gx;
  *by = (Lp / gx) % gy;
  *bz = Lp / (gx * gy);
}

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int WK, int BH>
__global__ __launch_bounds__(256) void wgrad_tiles_kernel(WtArgs a) {
  using Cfg = WtCfg<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>;
  static_assert(WX * WY * WK == 4, "4 waves");
  static_assert(BH * 4 % WK == 0, "k-steps must divide");
  constexpr int CXC = Cfg::CXC, CYC = Cfg::CYC, XR = Cfg::XR, IW = Cfg::IW, IWq = Cfg::IWq;
  constexpr int XT = Cfg::XT, YT = Cfg::YT, TAPS = Cfg::TAPS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + CXC / 16 * XT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave / (WY * WK), wy = (wave / WK) % WY, wk = wave % WK;
  const int li = lane & 15, kq = lane >> 4;

  int bx, by, split;
  wt_block_coords(&bx, &by, &split);
  const int cxb = bx % a.ncxb, cyb = bx / a.ncxb;
  const int ky0 = by * KHB;
  const int cx0 = cxb * CXC, cy0 = cyb * CYC;

  v4f acc[TAPS][NTX][NTY];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < NTX; ++i)
#pragma unroll
      for (int j = 0; j < NTY; ++j) acc[t][i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  const int xbase = (wx * NTX) * XT + kq * 16 + li;
  const int ybase = (wy * NTY) * YT + kq * 16 + li;

  const PW4 px4 = pw4_load(a.pwx, cx0 + (tid % (CXC / 4)) * 4, a.cx);
  const PW4 py4 = pw4_load(a.pwy, cy0 + (tid % (CYC / 4)) * 4, a.cy);
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.n * tiles_per_img;
  for (int tile = split; tile < ntiles; tile += a.nsplit) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int qy0 = ty_ * BH, qx0 = tx_ * 16;
    const int gy0 = qy0 * S + ky0 - a.pad, gx0 = qx0 * S - a.pad;
    __syncthreads();
    // ---- stage X: a float4 of channels per thread-iteration (fixed channel quad per thread)
    {
      const float* Xn = a.X + (int64_t)n * a.xh * a.xw * a.xcs + a.xco;
      constexpr int C4 = CXC / 4;
      const int c4 = tid % C4;
      const int chl = c4 * 4, ch = cx0 + chl;
      for (int e = tid; e < XR * IW * C4; e += 256) {
        const int pix = e / C4;
        const int c = pix % IW, r = pix / IW;
        const int iy = gy0 + r, ix = gx0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw && ch < a.cx) {
          const float* p = Xn + ((int64_t)iy * a.xw + ix) * a.xcs + ch;
          if (a.xvec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (ch + 1 < a.cx) v.y = p[1];
            if (ch + 2 < a.cx) v.z = p[2];
            if (ch + 3 < a.cx) v.w = p[3];
          }
          v = pw4_apply4(px4, v);
          if (ch + 1 >= a.cx) v.y = 0.f;
          if (ch + 2 >= a.cx) v.z = 0.f;
          if (ch + 3 >= a.cx) v.w = 0.f;
        }
        const int li_ = (chl >> 4) * XT + ((r * S + c % S) * IWq + c / S) * 16 + (chl & 15);
        *reinterpret_cast<float4*>(xs + li_) = v;
      }
    }
    // ---- stage Y
    {
      const float* Yn = a.Y + (int64_t)n * a.yh * a.yw * a.ycs + a.yco;
      constexpr int C4 = CYC / 4;
      const int c4 = tid % C4;
      const int chl = c4 * 4, ch = cy0 + chl;
      for (int e = tid; e < BH * 16 * C4; e += 256) {
        const int pix = e / C4;
        const int c = pix & 15, r = pix >> 4;
        const int qy = qy0 + r, qx = qx0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qy < a.yh && qx < a.yw && ch < a.cy) {
          const float* p = Yn + ((int64_t)qy * a.yw + qx) * a.ycs + ch;
          if (a.yvec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (ch + 1 < a.cy) v.y = p[1];
            if (ch + 2 < a.cy) v.z = p[2];
            if (ch + 3 < a.cy) v.w = p[3];
          }
          v = pw4_apply4(py4, v);
          if (ch + 1 >= a.cy) v.y = 0.f;
          if (ch + 2 >= a.cy) v.z = 0.f;
          if (ch + 3 >= a.cy) v.w = 0.f;
        }
        *reinterpret_cast<float4*>(ys + (chl >> 4) * YT + (r * 16 + c) * 16 + (chl & 15)) = v;
      }
    }
    __syncthreads();
    // ---- MFMA: this wave's k-steps (row r, pixel group g of 4)
#pragma unroll 2
    for (int st = 0; st < Cfg::STEPS; ++st) {
      const int step = st * WK + wk;
      const int r = step >> 2, g = step & 3;
      const float* xp = xs + xbase + (r * S * S * IWq + 4 * g) * 16;
      const float* yp = ys + ybase + (r * 16 + 4 * g) * 16;
      float bf[NTY];
#pragma unroll
      for (int j = 0; j < NTY; ++j) bf[j] = yp[j * YT];
#pragma unroll
      for (int kyl = 0; kyl < KHB; ++kyl)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          constexpr int dummy = 0;
          (void)dummy;
          const int toff = ((kyl * S + kx % S) * IWq + kx / S) * 16;
#pragma unroll
          for (int i = 0; i < NTX; ++i) {
            const float af = xp[i * XT + toff];
#pragma unroll
            for (int j = 0; j < NTY; ++j)
              acc[kyl * KW + kx][i][j] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[j], acc[kyl * KW + kx][i][j], 0, 0, 0);
          }
        }
    }
  }

  // ---- write this split's partial tiles: D[row = 4*(lane>>4)+r : X channel][col = lane&15 : Y channel]
  if constexpr (WK == 1) {
#pragma unroll
    for (int kyl = 0; kyl < KHB; ++kyl) {
      const int ky = ky0 + kyl;
      if (ky < a.k) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
          for (int i = 0; i < NTX; ++i)
#pragma unroll
            for (int j = 0; j < NTY; ++j) {
              const int cx = cx0 + (wx * NTX + i) * 16 + 4 * kq;
              const int cy = cy0 + (wy * NTY + j) * 16 + li;
              const v4f v = acc[kyl * KW + kx][i][j];
              *reinterpret_cast<float4*>(a.ws + ((((int64_t)split * a.k + ky) * a.k + kx) * a.CYP + cy) * a.CXP + cx) =
                  make_float4(v[0], v[1], v[2], v[3]);
            }
      }
    }
  } else {
    float* red = smem;  // [WK][NTX][NTY][64][4]
#pragma unroll
    for (int kyl = 0; kyl < KHB; ++kyl)
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NTX; ++i)
#pragma unroll
          for (int j = 0; j < NTY; ++j) {
            const v4f v = acc[kyl * KW + kx][i][j];
            *reinterpret_cast<float4*>(red + (((wk * NTX + i) * NTY + j) * 64 + lane) * 4) =
                make_float4(v[0], v[1], v[2], v[3]);
          }
        __syncthreads();
        const int ky = ky0 + kyl;
        if (ky < a.k) {
          // WX*WY == 1 whenever WK == 4; for WK == 2 the (wx,wy) pair owns its own slice below
          for (int e = tid; e < NTX * NTY * 64; e += 256 / 1) {
            const int l = e & 63, ij = e >> 6;
            const int i = ij / NTY, j = ij % NTY;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < WK; ++w) {
              const float4 t = *reinterpret_cast<const float4*>(red + (((w * NTX + i) * NTY + j) * 64 + l) * 4);
              s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            const int cx = cx0 + i * 16 + 4 * (l >> 4);
            const int cy = cy0 + j * 16 + (l & 15);
            *reinterpret_cast<float4*>(a.ws + ((((int64_t)split * a.k + ky) * a.k + kx) * a.CYP + cy) * a.CXP + cx) = s;
          }
        }
      }
  }
}


// ------------------------------------------------------------------------------------------------
// The same kernel with both tiles brought in by LDS-DMA one tile ahead (two LDS buffers):
//     barrier | issue DMA(tile t+1 -> other buffer) | first half of tile t's MFMAs |
//     wait own DMAs, rewrite own elements of tile t+1 | second half of tile t's MFMAs
// A DMA can neither apply the pending activation nor zero what lies outside the image / channel range, so each
// thread rewrites in place the elements its own lanes fetched (sources are clamped into the tensor so every lane
// fetches something).  The barrier at the top of the next tile publishes the rewritten tile.  The LDS images are
// the ones of wgrad_tiles_kernel, enumerated in LDS order (a wave-instruction lands on 1 KiB of consecutive LDS).
// Requires 16-byte addressable views and channel counts that are multiples of 4.
template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int WK, int BH>
struct WtDma {
  using Cfg = WtCfg<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>;
  static constexpr int XF4 = Cfg::CXC / 16 * Cfg::XT / 4;     // float4 groups of the X image
  static constexpr int YF4 = Cfg::CYC / 16 * Cfg::YT / 4;
  static constexpr int XP4 = (XF4 + 63) / 64 * 64;            // padded to whole 1-KiB pieces
  static constexpr int YP4 = (YF4 + 63) / 64 * 64;
  static constexpr int BUF = (XP4 + YP4) * 4;                 // floats per buffer
  static constexpr int XS = (XP4 + 255) / 256, YS = (YP4 + 255) / 256;   // elements per thread
  static constexpr size_t LDS_PIPE = (size_t)(2 * BUF + 3 * (Cfg::CXC + Cfg::CYC)) * 4;
  static constexpr size_t LDS = LDS_PIPE > Cfg::LDS_RED ? LDS_PIPE : Cfg::LDS_RED;
};

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int WK, int BH>
__global__ __launch_bounds__(256) void wgrad_tiles_dma_kernel(WtArgs a) {
  using Cfg = WtCfg<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>;
  using D = WtDma<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>;
  static_assert(WX * WY * WK == 4, "4 waves");
  static_assert(Cfg::STEPS % 2 == 0, "the tile's k-steps are split in two halves");
  static_assert(D::XS <= 16 && D::YS <= 16, "mask bits");
  constexpr int CXC = Cfg::CXC, CYC = Cfg::CYC, XR = Cfg::XR, IW = Cfg::IW, IWq = Cfg::IWq;
  constexpr int XT = Cfg::XT, YT = Cfg::YT, TAPS = Cfg::TAPS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lpx = smem + 2 * D::BUF;   // [scale | shift | slope][CXC]
  float* lpy = lpx + 3 * CXC;       // [scale | shift | slope][CYC]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave / (WY * WK), wy = (wave / WK) % WY, wk = wave % WK;
  const int li = lane & 15, kq = lane >> 4;

  int bx, by, split;
  wt_block_coords(&bx, &by, &split);
  const int cxb = bx % a.ncxb, cyb = bx / a.ncxb;
  const int ky0 = by * KHB;
  const int cx0 = cxb * CXC, cy0 = cyb * CYC;

  v4f acc[TAPS][NTX][NTY];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < NTX; ++i)
#pragma unroll
      for (int j = 0; j < NTY; ++j) acc[t][i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  const int xbase = (wx * NTX) * XT + kq * 16 + li;
  const int ybase = (wy * NTY) * YT + kq * 16 + li;

  // pending activations -> LDS (no ordinary global loads inside the tile loop)
  const bool px_on = a.pwx.scale != nullptr, py_on = a.pwy.scale != nullptr;
  for (int i = tid; i < CXC; i += 256) {
    const bool ok = px_on && cx0 + i < a.cx;
    lpx[i] = ok ? a.pwx.scale[cx0 + i] : 1.f;
    lpx[CXC + i] = ok ? a.pwx.shift[cx0 + i] : 0.f;
    lpx[2 * CXC + i] = ok ? a.pwx.slope[cx0 + i] : 1.f;
  }
  for (int i = tid; i < CYC; i += 256) {
    const bool ok = py_on && cy0 + i < a.cy;
    lpy[i] = ok ? a.pwy.scale[cy0 + i] : 1.f;
    lpy[CYC + i] = ok ? a.pwy.shift[cy0 + i] : 0.f;
    lpy[2 * CYC + i] = ok ? a.pwy.slope[cy0 + i] : 1.f;
  }

  // This thread's elements of the two images (float4 index e = i * 256 + tid, LDS order), tile-invariant part:
  // row | column << 8 | channel offset in the block << 16 | valid << 31.
  unsigned xd[D::XS], yd[D::YS];
#pragma unroll
  for (int i = 0; i < D::XS; ++i) {
    const int e = i * 256 + tid;
    const int c4 = e & 3;
    int p = e >> 2;
    const int xq = p % IWq; p /= IWq;
    const int xm = p % S; p /= S;
    const int r = p % XR;
    const int ct = p / XR;
    const int c = xq * S + xm;
    const bool valid = e < D::XF4 && c < IW;
    xd[i] = valid ? ((unsigned)r | ((unsigned)c << 8) | ((unsigned)(ct * 16 + c4 * 4) << 16) | 0x80000000u) : 0u;
  }
#pragma unroll
  for (int i = 0; i < D::YS; ++i) {
    const int e = i * 256 + tid;
    const int c4 = e & 3;
    int p = e >> 2;
    const int c = p & 15; p >>= 4;
    const int r = p % BH;
    const int ct = p / BH;
    const bool valid = e < D::YF4;
    yd[i] = valid ? ((unsigned)r | ((unsigned)c << 8) | ((unsigned)(ct * 16 + c4 * 4) << 16) | 0x80000000u) : 0u;
  }

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.n * tiles_per_img;

  // Issue the DMAs of one tile into buffer `buf`; returns the "outside" masks (bit i: element i must be zero).
  auto issue = [&](int tile, int buf, unsigned& xmask, unsigned& ymask) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int qy0 = ty_ * BH, qx0 = tx_ * 16;
    const int gy0 = qy0 * S + ky0 - a.pad, gx0 = qx0 * S - a.pad;
    const float* Xn = a.X + (int64_t)n * a.xh * a.xw * a.xcs + a.xco + cx0;
    const float* Yn = a.Y + (int64_t)n * a.yh * a.yw * a.ycs + a.yco + cy0;
    xmask = 0; ymask = 0;
#pragma unroll
    for (int i = 0; i < D::XS; ++i) {
      const int ebase = i * 256 + wave * 64;   // wave-uniform
      if (ebase < D::XP4) {
        const unsigned d = xd[i];
        const int r = d & 0xff, c = (d >> 8) & 0xff, chl = (d >> 16) & 0x7fff;
        const int iy = gy0 + r, ix = gx0 + c;
        const bool in = (d >> 31) && iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw && cx0 + chl < a.cx;
        if (!in) xmask |= 1u << i;
        const int yy = min(max(iy, 0), a.xh - 1), xx = min(max(ix, 0), a.xw - 1);
        const int cc = cx0 + chl < a.cx ? chl : 0;
        bp_glds16(Xn, (unsigned)(((yy * a.xw + xx) * a.xcs + cc) * 4), buf * D::BUF + ebase * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < D::YS; ++i) {
      const int ebase = i * 256 + wave * 64;
      if (ebase < D::YP4) {
        const unsigned d = yd[i];
        const int r = d & 0xff, c = (d >> 8) & 0xff, chl = (d >> 16) & 0x7fff;
        const int qy = qy0 + r, qx = qx0 + c;
        const bool in = (d >> 31) && qy < a.yh && qx < a.yw && cy0 + chl < a.cy;
        if (!in) ymask |= 1u << i;
        const int yy = min(qy, a.yh - 1), xx = min(qx, a.yw - 1);
        const int cc = cy0 + chl < a.cy ? chl : 0;
        bp_glds16(Yn, (unsigned)(((yy * a.yw + xx) * a.ycs + cc) * 4), buf * D::BUF + D::XP4 * 4 + ebase * 4);
      }
    }
  };
  // Activation / zeroing of this thread's own elements of buffer `buf` (after bp_wait_dma()).
  auto rewrite = [&](int buf, unsigned xmask, unsigned ymask) {
#pragma unroll
    for (int i = 0; i < D::XS; ++i) {
      const int e = i * 256 + tid;
      if (e < D::XF4) {
        float4* q = reinterpret_cast<float4*>(smem + buf * D::BUF + e * 4);
        if ((xmask >> i) & 1u) {
          *q = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (px_on) {
          const int chl = (xd[i] >> 16) & 0x7fff;
          const float4 sc = *reinterpret_cast<const float4*>(lpx + chl);
          const float4 sf = *reinterpret_cast<const float4*>(lpx + CXC + chl);
          const float4 sl = *reinterpret_cast<const float4*>(lpx + 2 * CXC + chl);
          float4 v = *q;
          v.x = fmaf(v.x, sc.x, sf.x); v.x = v.x > 0.f ? v.x : v.x * sl.x;
          v.y = fmaf(v.y, sc.y, sf.y); v.y = v.y > 0.f ? v.y : v.y * sl.y;
          v.z = fmaf(v.z, sc.z, sf.z); v.z = v.z > 0.f ? v.z : v.z * sl.z;
          v.w = fmaf(v.w, sc.w, sf.w); v.w = v.w > 0.f ? v.w : v.w * sl.w;
          *q = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < D::YS; ++i) {
      const int e = i * 256 + tid;
      if (e < D::YF4) {
        float4* q = reinterpret_cast<float4*>(smem + buf * D::BUF + D::XP4 * 4 + e * 4);
        if ((ymask >> i) & 1u) {
          *q = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (py_on) {
          const int chl = (yd[i] >> 16) & 0x7fff;
          const float4 sc = *reinterpret_cast<const float4*>(lpy + chl);
          const float4 sf = *reinterpret_cast<const float4*>(lpy + CYC + chl);
          const float4 sl = *reinterpret_cast<const float4*>(lpy + 2 * CYC + chl);
          float4 v = *q;
          v.x = fmaf(v.x, sc.x, sf.x); v.x = v.x > 0.f ? v.x : v.x * sl.x;
          v.y = fmaf(v.y, sc.y, sf.y); v.y = v.y > 0.f ? v.y : v.y * sl.y;
          v.z = fmaf(v.z, sc.z, sf.z); v.z = v.z > 0.f ? v.z : v.z * sl.z;
          v.w = fmaf(v.w, sc.w, sf.w); v.w = v.w > 0.f ? v.w : v.w * sl.w;
          *q = v;
        }
      }
    }
  };
  // k-steps [s0, s1) of the tile in buffer `buf`
  auto compute = [&](int buf, int s0, int s1) {
    const float* xs = smem + buf * D::BUF;
    const float* ys = xs + D::XP4 * 4;
#pragma unroll 2
    for (int st = s0; st < s1; ++st) {
      const int step = st * WK + wk;
      const int r = step >> 2, g = step & 3;
      const float* xp = xs + xbase + (r * S * S * IWq + 4 * g) * 16;
      const float* yp = ys + ybase + (r * 16 + 4 * g) * 16;
      float bf[NTY];
#pragma unroll
      for (int j = 0; j < NTY; ++j) bf[j] = yp[j * YT];
#pragma unroll
      for (int kyl = 0; kyl < KHB; ++kyl)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int toff = ((kyl * S + kx % S) * IWq + kx / S) * 16;
#pragma unroll
          for (int i = 0; i < NTX; ++i) {
            const float af = xp[i * XT + toff];
#pragma unroll
            for (int j = 0; j < NTY; ++j)
              acc[kyl * KW + kx][i][j] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[j], acc[kyl * KW + kx][i][j], 0, 0, 0);
          }
        }
    }
  };

  __syncthreads();   // activation parameters are in LDS
  {
    unsigned xm, ym;
    issue(split, 0, xm, ym);
    bp_wait_dma();
    rewrite(0, xm, ym);
  }
  int buf = 0;
  for (int tile = split; tile < ntiles; tile += a.nsplit) {
    __syncthreads();   // tile `tile` is rewritten by everybody; the other buffer is read out
    const int next = tile + a.nsplit;
    unsigned xm = 0, ym = 0;
    if (next < ntiles) issue(next, buf ^ 1, xm, ym);
    compute(buf, 0, Cfg::STEPS / 2);
    if (next < ntiles) {
      bp_wait_dma();
      rewrite(buf ^ 1, xm, ym);
    }
    compute(buf, Cfg::STEPS / 2, Cfg::STEPS);
    buf ^= 1;
  }

  // ---- write this split's partial tiles: D[row = 4*(lane>>4)+r : X channel][col = lane&15 : Y channel]
  if constexpr (WK == 1) {
#pragma unroll
    for (int kyl = 0; kyl < KHB; ++kyl) {
      const int ky = ky0 + kyl;
      if (ky < a.k) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
          for (int i = 0; i < NTX; ++i)
#pragma unroll
            for (int j = 0; j < NTY; ++j) {
              const int cx = cx0 + (wx * NTX + i) * 16 + 4 * kq;
              const int cy = cy0 + (wy * NTY + j) * 16 + li;
              const v4f v = acc[kyl * KW + kx][i][j];
              *reinterpret_cast<float4*>(a.ws + ((((int64_t)split * a.k + ky) * a.k + kx) * a.CYP + cy) * a.CXP + cx) =
                  make_float4(v[0], v[1], v[2], v[3]);
            }
      }
    }
  } else {
    float* red = smem;  // [WK][NTX][NTY][64][4]
#pragma unroll
    for (int kyl = 0; kyl < KHB; ++kyl)
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NTX; ++i)
#pragma unroll
          for (int j = 0; j < NTY; ++j) {
            const v4f v = acc[kyl * KW + kx][i][j];
            *reinterpret_cast<float4*>(red + (((wk * NTX + i) * NTY + j) * 64 + lane) * 4) =
                make_float4(v[0], v[1], v[2], v[3]);
          }
        __syncthreads();
        const int ky = ky0 + kyl;
        if (ky < a.k) {
          for (int e = tid; e < NTX * NTY * 64; e += 256) {
            const int l = e & 63, ij = e >> 6;
            const int i = ij / NTY, j = ij % NTY;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < WK; ++w) {
              const float4 t = *reinterpret_cast<const float4*>(red + (((w * NTX + i) * NTY + j) * 64 + l) * 4);
              s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            const int cx = cx0 + i * 16 + 4 * (l >> 4);
            const int cy = cy0 + j * 16 + (l & 15);
            *reinterpret_cast<float4*>(a.ws + ((((int64_t)split * a.k + ky) * a.k + kx) * a.CYP + cy) * a.CXP + cx) = s;
          }
        }
      }
  }
}

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int WK, int BH, bool DMA = false>
int launch(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
           size_t ws_bytes, size_t* need, int* nsplit_out, int* cxp, int* cyp, hipStream_t st, bool dry) {
  using Cfg = WtCfg<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>;
  using Dm = WtDma<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>;
  static_assert(WK == 1 || (WX == 1 && WY == 1), "k-split variants own the whole channel block");
  static_assert(DMA || Cfg::LDS <= 64 * 1024, "LDS budget");
  static_assert(!DMA || Dm::LDS <= 80 * 1024, "LDS budget: two workgroups per CU");
  WtArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff; a.cy = Y->c;
  a.n = X->n; a.k = cv->k; a.pad = cv->pad; a.pwx = pwx; a.pwy = pwy; a.ws = ws;
  a.ncxb = bp_ceil_div(X->c, Cfg::CXC);
  const int ncyb = bp_ceil_div(Y->c, Cfg::CYC);
  a.CXP = a.ncxb * Cfg::CXC;
  a.CYP = ncyb * Cfg::CYC;
  a.tiles_x = bp_ceil_div(Y->w, 16);
  a.tiles_y = bp_ceil_div(Y->h, BH);
  const int kyg = bp_ceil_div(cv->k, KHB);
  const int64_t ntiles = (int64_t)Y->n * a.tiles_x * a.tiles_y;
  const int64_t base = (int64_t)a.ncxb * ncyb * kyg;
  // ~2 workgroups per CU alone; fewer when the launch shares the GPU with the data-gradient chain (bp_wgrad_tiles_target)
  static const int wt_target = getenv("BP_WT_TARGET") ? atoi(getenv("BP_WT_TARGET")) : 512;
  // (only launches of about a millisecond: a long one -- the CGAN's 256 -> 512 layers run 5 - 18 ms -- loses more to the
  //  uneven grid than the main chain gains: CGAN iteration 345.8 ms at 512, 348.7 at 448)
  const double flop = 2.0 * (double)Y->n * Y->h * Y->w * cv->k * cv->k * X->c * Y->c;
  const int target = (!dry && t_wt_target > 0 && t_wt_target < wt_target && flop < 1.5e11) ? t_wt_target : wt_target;
  int64_t ns = (target + base - 1) / base;
  if (ns > ntiles) ns = ntiles;
  if (ns < 1) ns = 1;
  if (ns > 65535) ns = 65535;
  a.nsplit = (int)ns;
  *need = (size_t)a.nsplit * cv->k * cv->k * a.CYP * a.CXP * sizeof(float);
  *nsplit_out = a.nsplit; *cxp = a.CXP; *cyp = a.CYP;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.xvec = bp_view_vec4(X) ? 1 : 0;
  a.yvec = bp_view_vec4(Y) ? 1 : 0;
  dim3 grid((unsigned)(a.ncxb * ncyb), (unsigned)kyg, (unsigned)a.nsplit);
  if constexpr (DMA) {
    static const hipError_t optin = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&wgrad_tiles_dma_kernel<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (optin != hipSuccess) return BP_ELAUNCH;
    hipLaunchKernelGGL((wgrad_tiles_dma_kernel<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>), grid, dim3(256), Dm::LDS, st, a);
  } else {
    hipLaunchKernelGGL((wgrad_tiles_kernel<KHB, KW, S, NTX, NTY, WX, WY, WK, BH>), grid, dim3(256), Cfg::LDS, st, a);
  }
  BP_CHECK_LAUNCH();
  return BP_OK;
}

}  // namespace

// Returns BP_EUNSUPPORTED when no tap-blocked variant fits (caller falls back to conv_wgrad.hip).
int bp_wgrad_tiles(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                   size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  const int k = cv->k, s = cv->stride, cx = X->c, cy = Y->c;
#define BP_WT(...) return launch<__VA_ARGS__>(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry)
  // LDS-DMA pipelined variants need 16-byte addressable views and whole channel quads
  static const bool no_dma = getenv("BP_WT_NODMA") != nullptr;
  const bool dma = !no_dma && bp_view_vec4(X) && bp_view_vec4(Y) && cx % 4 == 0 && cy % 4 == 0;
  if (k == 3 && s == 1) {
    if (cx > 16 && cy > 16) {
      if (dma) BP_WT(3, 3, 1, 2, 2, 2, 2, 1, 3, true);
      BP_WT(3, 3, 1, 2, 2, 2, 2, 1, 4);
    }
    if (cy > 16) BP_WT(3, 3, 1, 1, 2, 1, 1, 4, 8);
    BP_WT(3, 3, 1, 1, 1, 1, 1, 4, 8);
  }
  if (k == 4 && s == 2) {
    if (cx > 16 && cy > 32) {
      if (dma) BP_WT(4, 4, 2, 1, 2, 2, 2, 1, 2, true);
      BP_WT(4, 4, 2, 1, 2, 2, 2, 1, 4);
    }
    if (cy > 16) {
      if (dma) BP_WT(4, 4, 2, 1, 2, 1, 1, 4, 4, true);
      BP_WT(4, 4, 2, 1, 2, 1, 1, 4, 4);
    }
    BP_WT(4, 4, 2, 1, 1, 1, 1, 4, 4);
  }
  // CGAN kernel sizes (trained_models/README.md:106-128): k3s2 encoders/decoders, k4s1 PatchGAN tail, k9 stem/head
  if (k == 3 && s == 2) {
    if (cx > 16 && cy > 32) {
      if (dma) BP_WT(3, 3, 2, 1, 2, 2, 2, 1, 2, true);
      BP_WT(3, 3, 2, 1, 2, 2, 2, 1, 4);
    }
    if (cy > 16) BP_WT(3, 3, 2, 1, 2, 1, 1, 4, 4);
    BP_WT(3, 3, 2, 1, 1, 1, 1, 4, 4);
  }
  if (k == 4 && s == 1) {
    if (cx > 16 && cy > 32) {
      if (dma) BP_WT(4, 4, 1, 1, 2, 2, 2, 1, 4, true);
      BP_WT(4, 4, 1, 1, 2, 2, 2, 1, 4);
    }
    if (cy > 16) BP_WT(4, 4, 1, 1, 2, 1, 1, 4, 8);
    BP_WT(4, 4, 1, 1, 1, 1, 1, 4, 8);
  }
  if (k == 9 && s == 1) BP_WT(3, 9, 1, 1, 1, 1, 1, 4, 8);
  if (k == 5 && s == 1) BP_WT(5, 5, 1, 1, 1, 1, 1, 4, 8);
  if (k == 7 && s == 1) BP_WT(4, 7, 1, 1, 1, 1, 1, 4, 8);
  if (k == 8 && s == 4) BP_WT(4, 8, 4, 1, 1, 1, 1, 4, 2);
#undef BP_WT
  return BP_EUNSUPPORTED;
}
