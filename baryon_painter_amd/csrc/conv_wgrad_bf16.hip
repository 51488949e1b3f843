// bf16 matrix-core weight gradient (see conv_bf16.hpp for the forward / data-gradient kernels and the shared helpers).
#include "conv_bf16.hpp"

using namespace bpbf16;
// ---------------------------------------------------------------------------------------------- weight gradient
//   ws[split][ky][kx][cy][cx] = sum over this split's pixels of act(X)[n, q*S + k - p, cx] * act(Y)[n, q, cy]
// (X: the layer's fine-grid tensor, Y: its coarse-grid tensor, as in conv_wgrad_tiles.hip; the partials are summed
// in fixed order by the reduce kernel of conv_wgrad.hip.)  A workgroup owns a (16*NTX) x (16*NTY) channel block
// and KHB rows of taps; its four waves take the rows of a BH x 32 pixel tile in turn (one MFMA k-step = the 32
// pixels of one row) and are summed through LDS at the end.
// LDS images:  X  [CXC/16][rows][x % S][x / S][16]      Y  [CYC/16][BH][32][16]        (bf16)
// Both MFMA fragments are "8 pixels of one channel": ds_read_b64_tr_b16 turns a 4-pixel x 16-channel block into
// "lane i holds channel i of the 4 pixels"; two reads per fragment.  The k index of a fragment is a dummy index,
// so lane group g reads pixels 4g..4g+3 and 16+4g..16+4g+3 -- each 32-lane half then touches 8 consecutive
// 32-byte rows = 256 contiguous bytes: conflict-free.
namespace {


struct WbArgs {
  const void* X; int xh, xw, xcs, xco, cx, x_bf16;
  const void* Y; int yh, yw, ycs, yco, cy, y_bf16;
  int n, k, pad;
  PW pwx, pwy;
  float* ws;
  int ncxb, nsplit, tiles_x, tiles_y, CXP, CYP;
};

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int BH>
struct WbCfg {
  static constexpr int WK = 4 / (WX * WY);          // waves that share a channel sub-block and split the tile's rows
  static constexpr int CXC = 16 * NTX * WX, CYC = 16 * NTY * WY;
  static constexpr int XR = (BH - 1) * S + KHB;
  static constexpr int IW = 31 * S + KW;
  static constexpr int IWq = (IW + S - 1) / S;
  static constexpr int XT = XR * S * IWq * 16;     // bf16 per X channel tile
  static constexpr int YT = BH * 32 * 16;          // bf16 per Y channel tile
  static constexpr int TAPS = KHB * KW;
  static constexpr size_t LDS_MAIN = (size_t)(CXC / 16 * XT + CYC / 16 * YT) * 2;
  static constexpr size_t LDS_RED = WK > 1 ? (size_t)WK * NTX * NTY * 64 * 4 * 4 : 0;
  static constexpr size_t LDS_TILES = LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED;
  static constexpr size_t LDS_PW = (size_t)3 * (CXC + CYC) * sizeof(float);     // pending activations of both sides
  static constexpr size_t LDS = LDS_TILES + LDS_PW;
};

__device__ __forceinline__ s4 lds_tr(const u16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf8 frag_of(s4 a, s4 b) {
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf8, v);
}

template <bool BF>
__device__ __forceinline__ void wb_load8(const void* base, int64_t off, int cmax, int ch, bool vec, float (&v)[8]) {
  // 8 channels [ch, ch+8) of one pixel; channels >= cmax read as 0 (never dereferenced)
  if (vec && ch + 7 < cmax) {
    load_unit<8, BF>(base, off, v);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
      if (ch + j < cmax) {
        if constexpr (BF) t = bf2f(reinterpret_cast<const u16*>(base)[off + j]);
        else t = reinterpret_cast<const float*>(base)[off + j];
      }
      v[j] = t;
    }
  }
}

// Workgroup = 4 waves as WX x WY x WK: WX*NTX X-channel tiles, WY*NTY Y-channel tiles; the WK waves of one channel
// sub-block take the rows of a tile in turn and are summed through LDS at the end.  The wide layers use 2 x 2 x 1:
// a 64 x 64 channel block per workgroup -- every pixel tile is staged (loaded, activated, rounded to bf16) once per
// FOUR channel-block pairs instead of once per sixteen.
// WT ("wave per tap row", KHB = 4 = the waves of the workgroup): every wave walks ALL rows of the tile for ONE tap row and
// the whole channel block -- no cross-wave sum, and a layer whose channels fit one block (32 <-> 64) stages every
// pixel tile once instead of once per (channel block, tap-row group) = four times.
template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int BH, bool XB, bool YB, bool WT = false>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WbArgs a) {
  using Cfg = WbCfg<KHB, KW, S, NTX, NTY, WX, WY, BH>;
  constexpr int CXC = Cfg::CXC, CYC = Cfg::CYC, XR = Cfg::XR, IWq = Cfg::IWq, XT = Cfg::XT, YT = Cfg::YT;
  constexpr int TAPS = WT ? KW : Cfg::TAPS, WK = WT ? 1 : Cfg::WK;
  static_assert(BH % WK == 0, "rows are dealt to the waves of a channel sub-block");
  static_assert(WT || WK == 1 || (WX == 1 && WY == 1), "row-split variants own the whole channel block");
  static_assert(!WT || (KHB == 4 && WX == 1 && WY == 1), "one tap row per wave");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* xs = smem;
  u16* ys = smem + CXC / 16 * XT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = WT ? 0 : wave / (WY * WK), wy = WT ? 0 : (wave / WK) % WY, wk = WT ? 0 : wave % WK;
  const int li = lane & 15, kq = lane >> 4;

  // XCD-aware block coordinates (see wt_block_coords in conv_wgrad_tiles.hip)
  int bx, by, split;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nb = gx * gy * gridDim.z;
    const int L = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const int q = nb >> 3, r = nb & 7;
    const int xcd = L & 7, idx = L >> 3;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = Lp % gx; by = (Lp / gx) % gy; split = Lp / (gx * gy);
  }
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

  // transposing-read address of this lane inside a [pixel][16] image: pixel 4*kq + (li >> 2), channels 4*(li & 3)
  const int trl = (4 * kq + (li >> 2)) * 16 + 4 * (li & 3);

  // staging: fixed 8-channel group per thread
  constexpr int XU = CXC / 8, YU = CYC / 8;          // units per pixel
  const int xcu = tid % XU, ycu = tid % YU;
  float xsc[8], xsf[8], xsl[8], ysc[8], ysf[8], ysl[8];
  const bool xon = a.pwx.scale != nullptr, yon = a.pwy.scale != nullptr;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int cxj = cx0 + xcu * 8 + j, cyj = cy0 + ycu * 8 + j;
    const bool okx = xon && cxj < a.cx, oky = yon && cyj < a.cy;
    xsc[j] = okx ? a.pwx.scale[cxj] : 1.f; xsf[j] = okx ? a.pwx.shift[cxj] : 0.f; xsl[j] = okx ? a.pwx.slope[cxj] : 1.f;
    ysc[j] = oky ? a.pwy.scale[cyj] : 1.f; ysf[j] = oky ? a.pwy.shift[cyj] : 0.f; ysl[j] = oky ? a.pwy.slope[cyj] : 1.f;
  }
  const bool xvec = (a.cx & 7) == 0, yvec = (a.cy & 7) == 0;

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.n * tiles_per_img;
  auto multiply = [&]() {
    // ---- this wave's rows of the tile
#pragma unroll 1
    for (int r = wk; r < BH; r += WK) {
      bf8 yf[NTY];
#pragma unroll
      for (int j = 0; j < NTY; ++j) {
        const u16* p = ys + (wy * NTY + j) * YT + r * 32 * 16 + trl;
        yf[j] = frag_of(lds_tr(p), lds_tr(p + 16 * 16));
      }
      if constexpr (WT) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int toff = (((r * S + wave) * S + kx % S) * IWq + kx / S) * 16;          // (tap row = this wave)
#pragma unroll
          for (int i = 0; i < NTX; ++i) {
            const u16* p = xs + i * XT + toff + trl;
            const bf8 xf = frag_of(lds_tr(p), lds_tr(p + 16 * 16));
#pragma unroll
            for (int j = 0; j < NTY; ++j)
              acc[kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf[j], acc[kx][i][j], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
      for (int kyl = 0; kyl < KHB; ++kyl)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int toff = (((r * S + kyl) * S + kx % S) * IWq + kx / S) * 16;
#pragma unroll
          for (int i = 0; i < NTX; ++i) {
            const u16* p = xs + (wx * NTX + i) * XT + toff + trl;
            const bf8 xf = frag_of(lds_tr(p), lds_tr(p + 16 * 16));
#pragma unroll
            for (int j = 0; j < NTY; ++j)
              acc[kyl * KW + kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf[j], acc[kyl * KW + kx][i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  // Staging, pipelined form (channel counts that are multiples of 8: 16-byte units).  A thread's units of the NEXT tile
  // are fetched -- all of them back to back, as raw 16-byte words, coordinates clamped into the image -- before this
  // tile is multiplied, and activated / rounded / written to LDS after it.  (Loading under a per-lane condition, or
  // converting right behind the load, makes the compiler wait for every load in turn: one HBM round trip per unit.)
  constexpr int NXS = (XR * S * IWq * XU + 255) / 256, NYS = (BH * 32 * YU + 255) / 256;
  if (xvec && yvec) {
    // the activation parameters of this thread's channel groups, parked in LDS (48 registers otherwise)
    float* lpw = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + Cfg::LDS_TILES);
    if (tid < XU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { lpw[tid * 8 + j] = xsc[j]; lpw[CXC + tid * 8 + j] = xsf[j]; lpw[2 * CXC + tid * 8 + j] = xsl[j]; }
    }
    if (tid < YU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lpw[3 * CXC + tid * 8 + j] = ysc[j]; lpw[3 * CXC + CYC + tid * 8 + j] = ysf[j]; lpw[3 * CXC + 2 * CYC + tid * 8 + j] = ysl[j];
      }
    }
    RawUnit<8, XB> xr[NXS];
    RawUnit<8, YB> yr[NYS];
    unsigned xin = 0, yin = 0;
    const int xch = cx0 + xcu * 8, ych = cy0 + ycu * 8;
    auto fetch = [&](int tile) {
      const int n = tile / tiles_per_img;
      const int trem = tile - n * tiles_per_img;
      const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
      const int qy0 = ty_ * BH, qx0 = tx_ * 32;
      const int gy0 = qy0 * S + ky0 - a.pad, gx0 = qx0 * S - a.pad;
      const int64_t ximg = (int64_t)n * a.xh * a.xw * a.xcs + a.xco + (xch < a.cx ? xch : 0);
      const int64_t yimg = (int64_t)n * a.yh * a.yw * a.ycs + a.yco + (ych < a.cy ? ych : 0);
      xin = 0; yin = 0;
#pragma unroll
      for (int i = 0; i < NXS; ++i) {
        const int e = tid + i * 256;
        const int pi = e / XU;
        const int xq = pi % IWq;
        const int t = pi / IWq;
        const int xm = t % S, r = t / S;
        const int iy = gy0 + r, ix = gx0 + xq * S + xm;
        if (e < XR * S * IWq * XU && iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw && xch < a.cx) xin |= 1u << i;
        const int cy = min(max(iy, 0), a.xh - 1), cx = min(max(ix, 0), a.xw - 1);
        load_unit_raw<8, XB>(a.X, ximg + ((int64_t)cy * a.xw + cx) * a.xcs, xr[i]);
      }
#pragma unroll
      for (int i = 0; i < NYS; ++i) {
        const int e = tid + i * 256;
        const int pi = e / YU;
        const int c = pi & 31, r = pi >> 5;
        const int qy = qy0 + r, qx = qx0 + c;
        if (e < BH * 32 * YU && qy < a.yh && qx < a.yw && ych < a.cy) yin |= 1u << i;
        const int cy = min(qy, a.yh - 1), cx = min(qx, a.yw - 1);
        load_unit_raw<8, YB>(a.Y, yimg + ((int64_t)cy * a.yw + cx) * a.ycs, yr[i]);
      }
    };
    auto commit = [&]() {
      // A unit's 24 activation parameters are re-read from LDS for every unit (the stores in between may alias them
      // as far as the compiler knows, and keeping them in registers across the loop spills: the accumulators fill the
      // file): as six 16-byte reads instead of 24 scalar ones, four channels at a time -- the commit was LDS-
      // instruction bound: weight gradients of the layers whose X carries a pending activation 0.32 -> 0.22 ms
      // (16 -> 32), 0.31 -> 0.21 (64 -> 128), 0.18 -> 0.15 (trunk); the Y side (transposed layers) measured no gain
      // from the same change and keeps its scalar reads.
#pragma unroll
      for (int i = 0; i < NXS; ++i) {
        const int e = tid + i * 256;
        if (e >= XR * S * IWq * XU) continue;
        u16* dst = xs + (xcu >> 1) * XT + (e / XU) * 16 + (xcu & 1) * 8;
        if constexpr (XB) {
          if (!xon) {        // bf16 in, no pending activation: the raw words ARE the LDS image (no unpack / round trip)
            *reinterpret_cast<uint4*>(dst) = ((xin >> i) & 1u) ? xr[i].q[0] : make_uint4(0u, 0u, 0u, 0u);
            continue;
          }
        }
        float v[8], raw[8];
        unpack_unit<8, XB>(xr[i], raw);
        // (the 32 <-> 64 wave-per-tap-row instance holds 128 accumulator registers: any other shape of this loop makes
        //  it spill, and its transposed layer's weight gradient goes 0.16 -> 0.25 ms: it keeps the scalar reads)
        if constexpr (WT && NTX * NTY >= 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float u = raw[j];
            if (xon) { u = fmaf(u, lpw[xcu * 8 + j], lpw[CXC + xcu * 8 + j]); u = u > 0.f ? u : u * lpw[2 * CXC + xcu * 8 + j]; }
            v[j] = ((xin >> i) & 1u) ? u : 0.f;
          }
        } else
#pragma unroll
        for (int h = 0; h < 2; ++h) {            // (four channels at a time: twelve parameter registers live, not 24)
          float c[4] = {1.f, 1.f, 1.f, 1.f}, f[4] = {0.f, 0.f, 0.f, 0.f}, l[4] = {1.f, 1.f, 1.f, 1.f};
          if (xon) {
            const float4 sc = *reinterpret_cast<const float4*>(lpw + xcu * 8 + 4 * h);
            const float4 sf = *reinterpret_cast<const float4*>(lpw + CXC + xcu * 8 + 4 * h);
            const float4 sl = *reinterpret_cast<const float4*>(lpw + 2 * CXC + xcu * 8 + 4 * h);
            c[0] = sc.x; c[1] = sc.y; c[2] = sc.z; c[3] = sc.w; f[0] = sf.x; f[1] = sf.y; f[2] = sf.z; f[3] = sf.w;
            l[0] = sl.x; l[1] = sl.y; l[2] = sl.z; l[3] = sl.w;
          }
          // (batch-norm + ReLU -- every trunk layer: slope 0 -- is fma + max instead of fma + compare + multiply + select)
          const bool relu = l[0] == 0.f && l[1] == 0.f && l[2] == 0.f && l[3] == 0.f;
          if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * h + j] = ((xin >> i) & 1u) ? bp_relu_nan(fmaf(raw[4 * h + j], c[j], f[j])) : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float u = raw[4 * h + j];
              if (xon) { u = fmaf(u, c[j], f[j]); u = u > 0.f ? u : u * l[j]; }
              v[4 * h + j] = ((xin >> i) & 1u) ? u : 0.f;
            }
          }
        }
        lds_store_unit<8>(dst, v);
      }
#pragma unroll
      for (int i = 0; i < NYS; ++i) {
        const int e = tid + i * 256;
        if (e >= BH * 32 * YU) continue;
        u16* dst = ys + (ycu >> 1) * YT + (e / YU) * 16 + (ycu & 1) * 8;
        if constexpr (YB) {
          if (!yon) {        // (a layer's d_raw: always this case)
            *reinterpret_cast<uint4*>(dst) = ((yin >> i) & 1u) ? yr[i].q[0] : make_uint4(0u, 0u, 0u, 0u);
            continue;
          }
        }
        float v[8], raw[8];
        unpack_unit<8, YB>(yr[i], raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float u = raw[j];
          if (yon) {       // (scalar reads here: the 16-byte form of the X side made this side slower, 0.15 -> 0.23 ms)
            u = fmaf(u, lpw[3 * CXC + ycu * 8 + j], lpw[3 * CXC + CYC + ycu * 8 + j]);
            u = u > 0.f ? u : u * lpw[3 * CXC + 2 * CYC + ycu * 8 + j];
          }
          v[j] = ((yin >> i) & 1u) ? u : 0.f;
        }
        lds_store_unit<8>(dst, v);
      }
    };
    int tile = split;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += a.nsplit) {
      __syncthreads();                // the previous tile's readers are done
      commit();
      __syncthreads();
      if (tile + a.nsplit < ntiles) fetch(tile + a.nsplit);
      multiply();
    }
  } else
  for (int tile = split; tile < ntiles; tile += a.nsplit) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int qy0 = ty_ * BH, qx0 = tx_ * 32;
    const int gy0 = qy0 * S + ky0 - a.pad, gx0 = qx0 * S - a.pad;
    __syncthreads();
    {   // ---- X tile
      const int64_t img = (int64_t)n * a.xh * a.xw * a.xcs + a.xco;
      const int ch = cx0 + xcu * 8;
      for (int e = tid; e < XR * S * IWq * XU; e += 256) {
        const int pi = e / XU;
        const int xq = pi % IWq;
        const int t = pi / IWq;
        const int xm = t % S, r = t / S;
        const int iy = gy0 + r, ix = gx0 + xq * S + xm;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw && ch < a.cx) {
          wb_load8<XB>(a.X, img + ((int64_t)iy * a.xw + ix) * a.xcs + ch, a.cx, ch, xvec, v);
          if (xon) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = fmaf(v[j], xsc[j], xsf[j]);
              v[j] = (ch + j < a.cx) ? (u > 0.f ? u : u * xsl[j]) : 0.f;
            }
          }
        }
        lds_store_unit<8>(xs + (xcu >> 1) * XT + pi * 16 + (xcu & 1) * 8, v);
      }
    }
    {   // ---- Y tile
      const int64_t img = (int64_t)n * a.yh * a.yw * a.ycs + a.yco;
      const int ch = cy0 + ycu * 8;
      for (int e = tid; e < BH * 32 * YU; e += 256) {
        const int pi = e / YU;
        const int c = pi & 31, r = pi >> 5;
        const int qy = qy0 + r, qx = qx0 + c;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (qy < a.yh && qx < a.yw && ch < a.cy) {
          wb_load8<YB>(a.Y, img + ((int64_t)qy * a.yw + qx) * a.ycs + ch, a.cy, ch, yvec, v);
          if (yon) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = fmaf(v[j], ysc[j], ysf[j]);
              v[j] = (ch + j < a.cy) ? (u > 0.f ? u : u * ysl[j]) : 0.f;
            }
          }
        }
        lds_store_unit<8>(ys + (ycu >> 1) * YT + pi * 16 + (ycu & 1) * 8, v);
      }
    }
    __syncthreads();
    multiply();
  }

  // ---- partial tiles of this split: D[row = 4*kq + r : X channel][col = li : Y channel]
  if constexpr (WT) {
    const int ky = ky0 + wave;
    if (ky < a.k) {
#pragma unroll
      for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int i = 0; i < NTX; ++i)
#pragma unroll
          for (int j = 0; j < NTY; ++j) {
            const int cx = cx0 + i * 16 + 4 * kq;
            const int cy = cy0 + j * 16 + li;
            const v4f v = acc[kx][i][j];
            *reinterpret_cast<float4*>(a.ws + ((((int64_t)split * a.k + ky) * a.k + kx) * a.CYP + cy) * a.CXP + cx) =
                make_float4(v[0], v[1], v[2], v[3]);
          }
    }
  } else if constexpr (WK == 1) {
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
    float* red = reinterpret_cast<float*>(smem);      // [WK][NTX][NTY][64][4]: the row-split waves summed through LDS
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
            *reinterpret_cast<float4*>(red + (((wk * NTX + i) * NTY + j) * 64 + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
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

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int BH, bool WT = false>
int wb_launch(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
              size_t ws_bytes, size_t* need, int* nsplit_out, int* cxp, int* cyp, hipStream_t st, bool dry) {
  using Cfg = WbCfg<KHB, KW, S, NTX, NTY, WX, WY, BH>;
  static_assert(Cfg::LDS <= 80 * 1024, "LDS budget: two workgroups per CU");
  WbArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c; a.x_bf16 = X->dtype == BP_BF16;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff; a.cy = Y->c; a.y_bf16 = Y->dtype == BP_BF16;
  a.n = X->n; a.k = cv->k; a.pad = cv->pad; a.pwx = pwx; a.pwy = pwy; a.ws = ws;
  a.ncxb = bp_ceil_div(X->c, Cfg::CXC);
  const int ncyb = bp_ceil_div(Y->c, Cfg::CYC);
  a.CXP = a.ncxb * Cfg::CXC;
  a.CYP = ncyb * Cfg::CYC;
  a.tiles_x = bp_ceil_div(Y->w, 32);
  a.tiles_y = bp_ceil_div(Y->h, BH);
  const int kyg = bp_ceil_div(cv->k, KHB);
  const int64_t ntiles = (int64_t)Y->n * a.tiles_x * a.tiles_y;
  const int64_t base = (int64_t)a.ncxb * ncyb * kyg;
  static const int wb_target = getenv("BP_WB_TARGET") ? atoi(getenv("BP_WB_TARGET")) : 512;
  int64_t ns = (wb_target + base - 1) / base;     // ~2 workgroups per CU (1024: 20.75 ms per bf16 step, 512: 20.59, 256: 21.1 -- partial sums are 590 KB per split of a trunk layer)
  if (ns > ntiles) ns = ntiles;
  if (ns < 1) ns = 1;
  if (ns > 65535) ns = 65535;
  a.nsplit = (int)ns;
  *need = (size_t)a.nsplit * cv->k * cv->k * a.CYP * a.CXP * sizeof(float);
  *nsplit_out = a.nsplit; *cxp = a.CXP; *cyp = a.CYP;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  dim3 grid((unsigned)(a.ncxb * ncyb), (unsigned)kyg, (unsigned)a.nsplit);
  const bool xb = a.x_bf16, yb = a.y_bf16;
#define BP_WB(XB_, YB_)                                                                                             \
  do {                                                                                                              \
    static const hipError_t optin = hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(&wgrad_bf16_kernel<KHB, KW, S, NTX, NTY, WX, WY, BH, XB_, YB_, WT>),           \
        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                                                     \
    if (optin != hipSuccess) return BP_ELAUNCH;                                                                     \
    hipLaunchKernelGGL((wgrad_bf16_kernel<KHB, KW, S, NTX, NTY, WX, WY, BH, XB_, YB_, WT>), grid, dim3(256), Cfg::LDS, st, a); \
  } while (0)
  if (xb && yb) BP_WB(true, true);
  else if (xb) BP_WB(true, false);
  else if (yb) BP_WB(false, true);
  else BP_WB(false, false);
#undef BP_WB
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- k7 head, flattened
// Weight gradient of the heads' first layer (unit-stride k7, 16 bf16 channels in, 8 fp32 gradient channels out) -- after
// conv_bf16_flat.hip took its forward and data gradient, the largest single launch of the bf16 step (1.2 ms in
// wgrad_bf16_kernel<4,7,...>: one 16 x 16 channel tile per MFMA, half of it padding, 28 of 49 taps per workgroup, the
// pixel tile staged twice).  Here:
//   * MFMA columns = (output row select rs, gradient channel co): the dY image interleaves row PAIRS as 16 "channels",
//     so ONE X fragment (input row i, tap column kx) feeds tap row ky = i - y + 3 of output row y AND ky - 1 of row
//     y + 1: accumulator "pair p" = [dW[p] | dW[p-1]], 8 x 7 accumulators instead of 7 x 7 half-empty ones, and 56
//     MFMAs per 64 pixel-rows instead of 98;
//   * the four waves own two pairs each (14 accumulators), every wave walks the whole 64 x 16 pixel tile: one dY
//     fragment per 14 MFMAs, the X fragments (ds_read_b64_tr_b16 of the [row][pixel][16] image the forward kernel
//     stages) are the only per-MFMA LDS traffic;
//   * persistent workgroups (two per CU) accumulate over their tiles and leave ONE partial each: the two halves of a
//     tap meet through LDS, partials go to the workspace layout of conv_wgrad.hip (fixed-order reduce).
struct WfArgs {
  const u16* X; int h, w, xcs, xco;
  const void* Y; int ycs, yco;          // fp32, or bf16 where the head's 8-channel slot is stored as bf16 (YB)
  int n;
  PW pwx;
  float* ws;
  int tiles_x, tiles_y;
};

constexpr int WF_K = 7, WF_TW = 64, WF_TH = 16, WF_LW = WF_TW + WF_K - 1, WF_LH = WF_TH + WF_K - 1;
constexpr int WF_XE = WF_LH * WF_LW * 16, WF_YE = (WF_TH / 2) * WF_TW * 16;          // bf16 elements
constexpr size_t WF_RED = (size_t)(WF_K + 1) * WF_K * 256 * sizeof(float);
constexpr size_t WF_TILES = (size_t)(WF_XE + WF_YE) * 2 > WF_RED ? (size_t)(WF_XE + WF_YE) * 2 : WF_RED;
constexpr size_t WF_LDS = WF_TILES + 3 * 16 * sizeof(float);

template <bool YB>
__global__ __launch_bounds__(256, 2) void wgrad_flatb_k7_kernel(WfArgs a) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* xs = smem;
  u16* ys = smem + WF_XE;
  float* lpw = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + WF_TILES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int trl = (4 * kq + (li >> 2)) * 16 + 4 * (li & 3);

  const bool on = a.pwx.scale != nullptr;
  if (tid < 16) {
    lpw[tid] = on ? a.pwx.scale[tid] : 1.f; lpw[16 + tid] = on ? a.pwx.shift[tid] : 0.f; lpw[32 + tid] = on ? a.pwx.slope[tid] : 1.f;
  }
  __syncthreads();
  const int cu = tid & 1;                       // X staging: this thread's channel octet
  float sc[8], sf[8], sl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = lpw[cu * 8 + j]; sf[j] = lpw[16 + cu * 8 + j]; sl[j] = lpw[32 + cu * 8 + j]; }

  v4f acc[2][WF_K];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int kx = 0; kx < WF_K; ++kx) acc[pp][kx] = v4f{0.f, 0.f, 0.f, 0.f};

  constexpr int NUX = WF_LH * WF_LW * 2, XS = (NUX + 255) / 256;      // 8-channel units of the X halo tile
  constexpr int NUY = WF_TH * WF_TW, YS = NUY / 256;                  // pixels of the dY tile (8 fp32 channels each)
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n = t / per_img, tr = t - n * per_img;
    const int ty0 = (tr / a.tiles_x) * WF_TH, tx0 = (tr % a.tiles_x) * WF_TW;
    // ---- loads first (all in flight, raw words, clamped coordinates), then convert into LDS
    uint4 xr[XS];
    float4 yr[YS][YB ? 1 : 2];
    unsigned xin = 0, yin = 0;
    const int64_t ximg = (int64_t)n * a.h * a.w * a.xcs + a.xco + cu * 8;
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = tid + i * 256, pi = e >> 1;
      const int row = pi / WF_LW, px = pi - row * WF_LW;
      const int gy = ty0 - 3 + row, gx = tx0 - 3 + px;
      if (e < NUX && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) xin |= 1u << i;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
      xr[i] = *reinterpret_cast<const uint4*>(a.X + ximg + ((int64_t)cy * a.w + cx) * a.xcs);
    }
    const int64_t yimg = (int64_t)n * a.h * a.w * a.ycs + a.yco;
#pragma unroll
    for (int i = 0; i < YS; ++i) {
      const int e = tid + i * 256;
      const int row = e / WF_TW, px = e - row * WF_TW;
      const int gy = ty0 + row, gx = tx0 + px;
      if (gy < a.h && gx < a.w) yin |= 1u << i;
      const int cy = min(gy, a.h - 1), cx = min(gx, a.w - 1);
      const int64_t yo = yimg + ((int64_t)cy * a.w + cx) * a.ycs;
      if constexpr (YB) {
        yr[i][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const u16*>(a.Y) + yo);
      } else {
        const float* q = reinterpret_cast<const float*>(a.Y) + yo;
        yr[i][0] = *reinterpret_cast<const float4*>(q);
        yr[i][1] = *reinterpret_cast<const float4*>(q + 4);
      }
    }
    __syncthreads();                            // the previous tile's readers are done with the images
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = tid + i * 256;
      if (e >= NUX) continue;
      const unsigned w[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = v[j];
        if (on) { x = fmaf(x, sc[j], sf[j]); x = x > 0.f ? x : x * sl[j]; }
        v[j] = ((xin >> i) & 1u) ? x : 0.f;
      }
      lds_store_unit<8>(xs + e * 8, v);
    }
#pragma unroll
    for (int i = 0; i < YS; ++i) {
      const int e = tid + i * 256;
      const int row = e / WF_TW, px = e - row * WF_TW;
      const bool ok = (yin >> i) & 1u;
      u16* yd = ys + (((row >> 1) * WF_TW + px) * 16) + (row & 1) * 8;                   // row pairs interleaved
      if constexpr (YB) {
        const uint4 w = __builtin_bit_cast(uint4, yr[i][0]);
        *reinterpret_cast<uint4*>(yd) = ok ? w : make_uint4(0u, 0u, 0u, 0u);
      } else {
        const float v[8] = {ok ? yr[i][0].x : 0.f, ok ? yr[i][0].y : 0.f, ok ? yr[i][0].z : 0.f, ok ? yr[i][0].w : 0.f,
                            ok ? yr[i][1].x : 0.f, ok ? yr[i][1].y : 0.f, ok ? yr[i][1].z : 0.f, ok ? yr[i][1].w : 0.f};
        lds_store_unit<8>(yd, v);
      }
    }
    __syncthreads();
    // ---- multiply: this wave's two pairs over the whole tile
#pragma unroll 1
    for (int rp = 0; rp < WF_TH / 2; ++rp) {
#pragma unroll
      for (int seg = 0; seg < WF_TW / 32; ++seg) {
        const u16* py = ys + (rp * WF_TW + seg * 32) * 16 + trl;
        const bf8 yf = frag_of(lds_tr(py), lds_tr(py + 16 * 16));
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const u16* px_ = xs + ((2 * rp + 2 * wave + pp) * WF_LW + seg * 32) * 16 + trl;
#pragma unroll
          for (int kx = 0; kx < WF_K; ++kx) {
            const bf8 xf = frag_of(lds_tr(px_ + kx * 16), lds_tr(px_ + (kx + 16) * 16));
            acc[pp][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf, acc[pp][kx], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- the two halves of every tap meet through LDS: red[pair][kx][ci][(rs, co)]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int kx = 0; kx < WF_K; ++kx)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(((2 * wave + pp) * WF_K + kx) * 16 + 4 * kq + r) * 16 + li] = acc[pp][kx][r];
  __syncthreads();
  float* out = a.ws + (int64_t)blockIdx.x * WF_K * WF_K * 8 * 16;
  for (int e = tid; e < WF_K * WF_K * 8 * 16; e += 256) {
    const int ci = e & 15, co = (e >> 4) & 7, tap = e >> 7;
    const int ky = tap / WF_K, kx = tap - ky * WF_K;
    out[e] = red[((ky * WF_K + kx) * 16 + ci) * 16 + co] + red[(((ky + 1) * WF_K + kx) * 16 + ci) * 16 + 8 + co];
  }
}

static bool wf_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy) {
  static const bool off = getenv("BP_BF16_NOFLATW") != nullptr;
  if (off || cv->transposed || cv->k != WF_K || cv->stride != 1 || cv->pad != 3 || X->c != 16 || Y->c != 8) return false;
  if (X->dtype != BP_BF16 || pwy.scale) return false;
  if (X->h != Y->h || X->w != Y->w || X->n != Y->n) return false;
  if ((X->cstride * 2) % 16 || (X->coff * 2) % 16 || reinterpret_cast<uintptr_t>(X->ptr) % 16) return false;
  const int yv = Y->dtype == BP_BF16 ? 8 : 4;         // channels per vector load of dY
  if (Y->cstride % yv || Y->coff % yv || reinterpret_cast<uintptr_t>(Y->ptr) % 16) return false;
  return true;
}

static int wf_launch(const bp_view* X, const PW& pwx, const bp_view* Y, float* ws, size_t ws_bytes, size_t* need,
                     int* nsplit_out, int* cxp, int* cyp, hipStream_t st, bool dry) {
  WfArgs a{};
  a.X = reinterpret_cast<const u16*>(X->ptr); a.h = X->h; a.w = X->w; a.xcs = X->cstride; a.xco = X->coff;
  a.Y = Y->ptr; a.ycs = Y->cstride; a.yco = Y->coff;
  a.n = X->n; a.pwx = pwx; a.ws = ws;
  a.tiles_x = bp_ceil_div(X->w, WF_TW); a.tiles_y = bp_ceil_div(X->h, WF_TH);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  const int ns = (int)(ntiles < 512 ? ntiles : 512);                 // two persistent workgroups per CU
  *need = (size_t)ns * WF_K * WF_K * 8 * 16 * sizeof(float);
  *nsplit_out = ns; *cxp = 16; *cyp = 8;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_flatb_k7_kernel<false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)WF_LDS);
  static const hipError_t optin_b = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_flatb_k7_kernel<true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)WF_LDS);
  if (optin != hipSuccess || optin_b != hipSuccess) return BP_ELAUNCH;
  if (Y->dtype == BP_BF16) hipLaunchKernelGGL(wgrad_flatb_k7_kernel<true>, dim3((unsigned)ns), dim3(256), WF_LDS, st, a);
  else hipLaunchKernelGGL(wgrad_flatb_k7_kernel<false>, dim3((unsigned)ns), dim3(256), WF_LDS, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- k5 stem, flattened
// Weight gradient of the generator's stem (unit-stride k5, 3(+1) fp32 channels in, 16 bf16 gradient channels): MFMA rows
// = the 16 gradient channels (dY image [row][pixel][16], transposing read), MFMA columns = (tap column kxs of a group of
// four, input channel ci): the X image is [row][pixel][4] bf16, so 16 consecutive elements ARE four neighbouring
// pixels x four channels and the transposing read takes its four 8-byte chunks from pixels P .. P + 3.  Ten
// accumulators (5 tap rows x 2 column groups) per wave; the waves split the tile's rows and meet through LDS.  The
// launch is HBM-bound (0.8 GB): 0.69 ms in wgrad_bf16_kernel<5,5,...> (one MFMA per tap, 12 of 16 rows padding).
struct WsfArgs {
  const float* X; int h, w, xcs, xco, cx;
  const u16* Y; int ycs, yco;
  int n;
  PW pwx;
  float* ws;
  int tiles_x, tiles_y;
};

constexpr int SF_K = 5, SF_TW = 64, SF_TH = 16, SF_LW = SF_TW + 8, SF_LH = SF_TH + SF_K - 1;   // (+8: the second column group reads on)
constexpr int SF_XE = SF_LH * SF_LW * 4 + 64, SF_YE = SF_TH * SF_TW * 16;
constexpr size_t SF_RED = (size_t)4 * SF_K * 2 * 256 * sizeof(float);
constexpr size_t SF_TILES = (size_t)(SF_XE + SF_YE) * 2 > SF_RED ? (size_t)(SF_XE + SF_YE) * 2 : SF_RED;
constexpr size_t SF_LDS = SF_TILES + 3 * 4 * sizeof(float);

__global__ __launch_bounds__(256, 2) void wgrad_flatb_stem_kernel(WsfArgs a) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* xs = smem;
  u16* ys = smem + SF_XE;
  float* lpw = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + SF_TILES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int trl = (4 * kq + (li >> 2)) * 16 + 4 * (li & 3);        // [pixel][16] image
  const int trx = ((4 * kq + (li >> 2)) + (li & 3)) * 4;           // [pixel][4] image: chunk c = pixel + c

  const bool on = a.pwx.scale != nullptr;
  if (tid < 4) {
    const bool ok = on && tid < a.cx;
    lpw[tid] = ok ? a.pwx.scale[tid] : 1.f; lpw[4 + tid] = ok ? a.pwx.shift[tid] : 0.f; lpw[8 + tid] = ok ? a.pwx.slope[tid] : 1.f;
  }
  for (int e = tid; e < SF_XE / 4; e += 256) *reinterpret_cast<uint2*>(xs + e * 4) = make_uint2(0u, 0u);   // (slack stays 0)
  __syncthreads();
  float sc[4], sf[4], sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sc[j] = lpw[j]; sf[j] = lpw[4 + j]; sl[j] = lpw[8 + j]; }

  v4f acc[SF_K][2];
#pragma unroll
  for (int ky = 0; ky < SF_K; ++ky) { acc[ky][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[ky][1] = v4f{0.f, 0.f, 0.f, 0.f}; }

  constexpr int XW = SF_TW + SF_K - 1;                                 // staged pixels per X row
  constexpr int NUX = SF_LH * XW, XS = (NUX + 255) / 256;              // pixels of the X halo tile (one float4 each)
  constexpr int NUY = SF_TH * SF_TW * 2, YS = NUY / 256;               // 8-channel units of the dY tile
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n = t / per_img, tr = t - n * per_img;
    const int ty0 = (tr / a.tiles_x) * SF_TH, tx0 = (tr % a.tiles_x) * SF_TW;
    float4 xr[XS];
    uint4 yr[YS];
    unsigned xin = 0, yin = 0;
    const int64_t ximg = (int64_t)n * a.h * a.w * a.xcs + a.xco;
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = tid + i * 256;
      const int row = e / XW, px = e - row * XW;
      const int gy = ty0 - 2 + row, gx = tx0 - 2 + px;
      if (e < NUX && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) xin |= 1u << i;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
      xr[i] = *reinterpret_cast<const float4*>(a.X + ximg + ((int64_t)cy * a.w + cx) * a.xcs);
    }
    const int64_t yimg = (int64_t)n * a.h * a.w * a.ycs + a.yco + (tid & 1) * 8;
#pragma unroll
    for (int i = 0; i < YS; ++i) {
      const int e = tid + i * 256, pi = e >> 1;
      const int row = pi / SF_TW, px = pi - row * SF_TW;
      const int gy = ty0 + row, gx = tx0 + px;
      if (gy < a.h && gx < a.w) yin |= 1u << i;
      const int cy = min(gy, a.h - 1), cx = min(gx, a.w - 1);
      yr[i] = *reinterpret_cast<const uint4*>(a.Y + yimg + ((int64_t)cy * a.w + cx) * a.ycs);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = tid + i * 256;
      if (e >= NUX) continue;
      const int row = e / XW, px = e - row * XW;
      const float r[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = r[j];
        if (on) { x = fmaf(x, sc[j], sf[j]); x = x > 0.f ? x : x * sl[j]; }
        v[j] = (((xin >> i) & 1u) && j < a.cx) ? x : 0.f;
      }
      lds_store_unit<4>(xs + (row * SF_LW + px) * 4, v);
    }
#pragma unroll
    for (int i = 0; i < YS; ++i)      // d_raw: no activation -- the raw words are the LDS image
      *reinterpret_cast<uint4*>(ys + (tid + i * 256) * 8) = ((yin >> i) & 1u) ? yr[i] : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
#pragma unroll 1
    for (int r = wave; r < SF_TH; r += 4) {
#pragma unroll
      for (int seg = 0; seg < SF_TW / 32; ++seg) {
        const u16* py = ys + (r * SF_TW + seg * 32) * 16 + trl;
        const bf8 yf = frag_of(lds_tr(py), lds_tr(py + 16 * 16));
#pragma unroll
        for (int ky = 0; ky < SF_K; ++ky)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const u16* px_ = xs + ((r + ky) * SF_LW + seg * 32 + 4 * g) * 4 + trx;
            const bf8 xf = frag_of(lds_tr(px_), lds_tr(px_ + 16 * 4));
            acc[ky][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf, acc[ky][g], 0, 0, 0);
          }
      }
    }
  }
  // ---- waves meet through LDS: red[wave][ky][g][co = 4 kq + r][n = li = (kxs, ci)]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int ky = 0; ky < SF_K; ++ky)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * SF_K + ky) * 2 + g) * 16 + 4 * kq + r) * 16 + li] = acc[ky][g][r];
  __syncthreads();
  float* out = a.ws + (int64_t)blockIdx.x * SF_K * SF_K * 16 * 4;             // [ky][kx][co][ci(4)]
  for (int e = tid; e < SF_K * SF_K * 16 * 4; e += 256) {
    const int ci = e & 3, co = (e >> 2) & 15, tap = e >> 6;
    const int ky = tap / SF_K, kx = tap - ky * SF_K;
    const int g = kx >> 2, kxs = kx & 3;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[(((w * SF_K + ky) * 2 + g) * 16 + co) * 16 + kxs * 4 + ci];
    out[e] = v;
  }
}

static bool sf_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy) {
  static const bool off = getenv("BP_BF16_NOFLATW") != nullptr;
  if (off || cv->transposed || cv->k != SF_K || cv->stride != 1 || cv->pad != 2 || X->c > 4 || Y->c != 16) return false;
  if (X->dtype != BP_F32 || Y->dtype != BP_BF16 || pwy.scale) return false;
  if (X->h != Y->h || X->w != Y->w || X->n != Y->n) return false;
  if (X->cstride % 4 || X->coff % 4 || X->coff + 4 > X->cstride || reinterpret_cast<uintptr_t>(X->ptr) % 16) return false;
  if ((Y->cstride * 2) % 16 || (Y->coff * 2) % 16 || reinterpret_cast<uintptr_t>(Y->ptr) % 16) return false;
  return true;
}

static int sf_launch(const bp_view* X, const PW& pwx, const bp_view* Y, float* ws, size_t ws_bytes, size_t* need,
                     int* nsplit_out, int* cxp, int* cyp, hipStream_t st, bool dry) {
  WsfArgs a{};
  a.X = reinterpret_cast<const float*>(X->ptr); a.h = X->h; a.w = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c;
  a.Y = reinterpret_cast<const u16*>(Y->ptr); a.ycs = Y->cstride; a.yco = Y->coff;
  a.n = X->n; a.pwx = pwx; a.ws = ws;
  a.tiles_x = bp_ceil_div(X->w, SF_TW); a.tiles_y = bp_ceil_div(X->h, SF_TH);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  const int ns = (int)(ntiles < 512 ? ntiles : 512);                 // two persistent workgroups per CU
  *need = (size_t)ns * SF_K * SF_K * 16 * 4 * sizeof(float);
  *nsplit_out = ns; *cxp = 4; *cyp = 16;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_flatb_stem_kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_LDS);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL(wgrad_flatb_stem_kernel, dim3((unsigned)ns), dim3(256), SF_LDS, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- k5 head tail, 8 -> 1
// Weight gradient of the heads' second layer (unit-stride k5, the head's 8-channel bf16 slot in, ONE fp32 gradient channel
// out; architecture_built.txt:104): a matrix-VECTOR product per tap -- on the vector ALUs (wgrad_cy1_kernel<5, 8, 16>) 0.26 ms
// for 0.3 GB.  Matrix form, K = 32 pixels of a row:
//   * MFMA rows = 16 consecutive bf16 of the staged X row [pixel][8] = (column select s = 0, 1; channel c) of pixels
//     x + kxb + s -- the transposing read starts at ANY pixel, so the three tap-column bases kxb = 0, 2, 4 cover kx = 0..5;
//   * MFMA columns = tap row ky: for X row i the gradient rows i - ky, plain 8-byte reads of the [row][pixel] bf16 image of
//     dy (one channel: K-contiguous as stored), zero rows around the tile; columns 5..15 repeat ky = 4 and are dropped.
// 3 accumulators per wave for all 25 taps; the waves split the X rows and meet through LDS; persistent workgroups, one
// partial each, fixed-order reduce (conv_wgrad.hip).
struct WhArgs {
  const u16* X; int h, w, xcs, xco;
  const float* Y; int ycs, yco;
  int n;
  PW pwx;
  float* ws;
  int tiles_x, tiles_y;
};

constexpr int WH_K = 5, WH_C = 8, WH_TW = 64, WH_TH = 16, WH_LW = WH_TW + WH_K - 1, WH_LH = WH_TH + WH_K - 1;
constexpr int WH_XE = WH_LH * WH_LW * WH_C + 64;          // (+ slack: the last tap-column base reads two pixels on)
constexpr int WH_YR = WH_TH + 2 * (WH_K - 1), WH_YE = WH_YR * WH_TW;
constexpr size_t WH_RED = (size_t)4 * 3 * 256 * sizeof(float);
constexpr size_t WH_TILES = (size_t)(WH_XE + WH_YE) * 2 > WH_RED ? (size_t)(WH_XE + WH_YE) * 2 : WH_RED;
constexpr size_t WH_LDS = WH_TILES + 3 * WH_C * sizeof(float);

__global__ __launch_bounds__(256, 4) void wgrad_head_kernel(WhArgs a) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* xs = smem;
  u16* ys = smem + WH_XE;
  float* lpw = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + WH_TILES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int trl = (4 * kq + (li >> 2)) * WH_C + 4 * (li & 3);      // [pixel][8] image: 16 elements = pixels P, P + 1

  const bool on = a.pwx.scale != nullptr;
  if (tid < WH_C) {
    lpw[tid] = on ? a.pwx.scale[tid] : 1.f; lpw[WH_C + tid] = on ? a.pwx.shift[tid] : 0.f; lpw[2 * WH_C + tid] = on ? a.pwx.slope[tid] : 1.f;
  }
  // the slack behind the X image and the zero rows of the dy image stay zero for every tile
  for (int e = tid; e < 64 / 8; e += 256) *reinterpret_cast<uint4*>(xs + WH_LH * WH_LW * WH_C + e * 8) = make_uint4(0u, 0u, 0u, 0u);
  for (int e = tid; e < WH_YE / 8; e += 256) *reinterpret_cast<uint4*>(ys + e * 8) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  float sc[8], sf[8], sl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = lpw[j]; sf[j] = lpw[WH_C + j]; sl[j] = lpw[2 * WH_C + j]; }

  v4f acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) acc[b] = v4f{0.f, 0.f, 0.f, 0.f};

  constexpr int NUX = WH_LH * WH_LW, XS = (NUX + 255) / 256;          // pixels of the X halo tile (16 bytes each)
  constexpr int YS = WH_TH * WH_TW / 256;                             // pixels of the dy tile per thread
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  const int kyl = min(li, WH_K - 1);                                  // this lane's tap row (MFMA column)
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n = t / per_img, tr = t - n * per_img;
    const int ty0 = (tr / a.tiles_x) * WH_TH, tx0 = (tr % a.tiles_x) * WH_TW;
    uint4 xr[XS];
    float yr[YS];
    unsigned xin = 0;
    const int64_t ximg = (int64_t)n * a.h * a.w * a.xcs + a.xco;
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = tid + i * 256;
      const int row = e / WH_LW, px = e - row * WH_LW;
      const int gy = ty0 - 2 + row, gx = tx0 - 2 + px;
      if (e < NUX && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) xin |= 1u << i;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
      xr[i] = *reinterpret_cast<const uint4*>(a.X + ximg + ((int64_t)cy * a.w + cx) * a.xcs);
    }
    const int64_t yimg = (int64_t)n * a.h * a.w * a.ycs + a.yco;
#pragma unroll
    for (int i = 0; i < YS; ++i) {
      const int e = tid + i * 256;
      const int row = e / WH_TW, px = e - row * WH_TW;
      const int gy = ty0 + row, gx = tx0 + px;
      const int cy = min(gy, a.h - 1), cx = min(gx, a.w - 1);
      const float v = a.Y[yimg + ((int64_t)cy * a.w + cx) * a.ycs];
      yr[i] = (gy < a.h && gx < a.w) ? v : 0.f;
    }
    __syncthreads();                            // the previous tile's readers are done with the images
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = tid + i * 256;
      if (e >= NUX) continue;
      const unsigned w[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = v[j];
        if (on) { x = fmaf(x, sc[j], sf[j]); x = x > 0.f ? x : x * sl[j]; }
        v[j] = ((xin >> i) & 1u) ? x : 0.f;
      }
      lds_store_unit<8>(xs + e * 8, v);
    }
#pragma unroll
    for (int i = 0; i < YS; ++i) {
      const int e = tid + i * 256;
      ys[(WH_K - 1) * WH_TW + e] = f2bf(yr[i]);           // tile row r lives in image row r + 4
    }
    __syncthreads();
    // ---- multiply: X rows i = wave, wave + 4, ... (image row ty0 - 2 + i); tap row ky pairs it with tile row i - ky
#pragma unroll 1
    for (int i = wave; i < WH_LH; i += 4) {
#pragma unroll
      for (int seg = 0; seg < WH_TW / 32; ++seg) {
        const u16* py = ys + (i - kyl + WH_K - 1) * WH_TW + seg * 32 + 4 * kq;
        const uint2 y0 = *reinterpret_cast<const uint2*>(py), y1 = *reinterpret_cast<const uint2*>(py + 16);
        const bf8 yf = __builtin_bit_cast(bf8, make_uint4(y0.x, y0.y, y1.x, y1.y));
        const u16* px_ = xs + (i * WH_LW + seg * 32) * WH_C + trl;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const bf8 xf = frag_of(lds_tr(px_ + 2 * b * WH_C), lds_tr(px_ + (2 * b + 16) * WH_C));
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf, acc[b], 0, 0, 0);
        }
      }
    }
  }
  // ---- waves meet through LDS: red[wave][b][row m = 4 kq + r = (s, c)][column li = ky]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * 3 + b) * 16 + 4 * kq + r) * 16 + li] = acc[b][r];
  __syncthreads();
  float* out = a.ws + (int64_t)blockIdx.x * WH_K * WH_K * WH_C;               // [ky][kx][1][ci]
  for (int e = tid; e < WH_K * WH_K * WH_C; e += 256) {
    const int ci = e & 7, tap = e >> 3;
    const int ky = tap / WH_K, kx = tap - ky * WH_K;
    const int b = kx >> 1, s_ = kx & 1;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((w * 3 + b) * 16 + s_ * 8 + ci) * 16 + ky];
    out[e] = v;
  }
}

static bool wh_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy) {
  static const bool off = getenv("BP_BF16_NOHEADW") != nullptr;
  if (off || cv->transposed || cv->k != WH_K || cv->stride != 1 || cv->pad != 2 || X->c != WH_C || Y->c != 1) return false;
  if (X->dtype != BP_BF16 || Y->dtype != BP_F32 || pwy.scale) return false;
  if (X->h != Y->h || X->w != Y->w || X->n != Y->n) return false;
  if ((X->cstride * 2) % 16 || (X->coff * 2) % 16 || reinterpret_cast<uintptr_t>(X->ptr) % 16) return false;
  return true;
}

static int wh_launch(const bp_view* X, const PW& pwx, const bp_view* Y, float* ws, size_t ws_bytes, size_t* need,
                     int* nsplit_out, int* cxp, int* cyp, hipStream_t st, bool dry) {
  WhArgs a{};
  a.X = reinterpret_cast<const u16*>(X->ptr); a.h = X->h; a.w = X->w; a.xcs = X->cstride; a.xco = X->coff;
  a.Y = reinterpret_cast<const float*>(Y->ptr); a.ycs = Y->cstride; a.yco = Y->coff;
  a.n = X->n; a.pwx = pwx; a.ws = ws;
  a.tiles_x = bp_ceil_div(X->w, WH_TW); a.tiles_y = bp_ceil_div(X->h, WH_TH);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  const int ns = (int)(ntiles < 1024 ? ntiles : 1024);               // four persistent workgroups per CU (27 KB of LDS each)
  *need = (size_t)ns * WH_K * WH_K * WH_C * sizeof(float);
  *nsplit_out = ns; *cxp = WH_C; *cyp = 1;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_head_kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)WH_LDS);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL(wgrad_head_kernel, dim3((unsigned)ns), dim3(256), WH_LDS, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// 16-byte loads of 8 channels need 16-byte aligned pixel rows; ragged channel counts take the scalar path
bool wb_view_ok(const bp_view* v) {
  const int esz = v->dtype == BP_BF16 ? 2 : 4;
  if (reinterpret_cast<uintptr_t>(v->ptr) % 16) return false;
  if (v->c % 8 == 0) return (v->cstride * esz) % 16 == 0 && (v->coff * esz) % 16 == 0;
  return true;
}

}  // namespace

// Same contract as bp_wgrad_tiles (conv_wgrad_tiles.hip): partial sums to ws, BP_EUNSUPPORTED if no variant fits.
int bp_wgrad_ws_bf16(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                     size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);

int bp_wgrad_bf16(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                  size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  if (!wb_view_ok(X) || !wb_view_ok(Y)) return BP_EUNSUPPORTED;
  if (wf_ok(cv, X, Y, pwy)) return wf_launch(X, pwx, Y, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  if (sf_ok(cv, X, Y, pwy)) return sf_launch(X, pwx, Y, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  if (wh_ok(cv, X, Y, pwy)) return wh_launch(X, pwx, Y, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  {      // the output-stationary kernel of the 128 <-> 128 k3 trunk layers (conv_wgrad_ws_bf16.hip)
    const int rc = bp_wgrad_ws_bf16(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
    if (rc == BP_OK && dry) {      // (size the workspace for either kernel: bp_set_option may switch later)
      size_t need2 = 0;
      int ns2, cx2, cy2;
      if (wb_launch<3, 3, 1, 2, 2, 2, 2, 4>(cv, X, pwx, Y, pwy, ws, ws_bytes, &need2, &ns2, &cx2, &cy2, st, true) == BP_OK &&
          need2 > *need)
        *need = need2;
    }
    if (rc != BP_EUNSUPPORTED) return rc;
  }
  const int k = cv->k, s = cv->stride, cx = X->c, cy = Y->c;
#define BP_WB_(...) return wb_launch<__VA_ARGS__>(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry)
  if (k == 3 && s == 1) {
    // 64 x 64 channel block, 4 x 32 pixel tiles (8-row tiles need 19 staging registers per thread on top of the 144
    // accumulators: 128 bytes of scratch, 118 -> 167 us)
    if (cx > 32 && cy > 32) BP_WB_(3, 3, 1, 2, 2, 2, 2, 4);
    if (cx > 16 && cy > 16) BP_WB_(3, 3, 1, 2, 2, 1, 1, 8);
    if (cy > 16) BP_WB_(3, 3, 1, 1, 2, 1, 1, 8);
    if (cx > 16) BP_WB_(3, 3, 1, 2, 1, 1, 1, 8);
    BP_WB_(3, 3, 1, 1, 1, 1, 1, 8);
  }
  if (k == 4 && s == 2) {
    if (cx > 32 && cy > 32) BP_WB_(2, 4, 2, 2, 2, 2, 2, 2);          // 64 x 64 channel block, two tap rows per group
    static const bool no_wt = getenv("BP_BF16_NOWT") != nullptr;
    if (cx > 16 && cx <= 32 && cy > 32 && cy <= 64 && !no_wt) BP_WB_(4, 4, 2, 2, 4, 1, 1, 4, true);    // 32 <-> 64: wave per tap row
    if (cx > 16 && cy > 16) BP_WB_(2, 4, 2, 2, 2, 1, 1, 4);
    if (cy > 16 && cy <= 32 && cx <= 16 && !no_wt) BP_WB_(4, 4, 2, 1, 2, 1, 1, 8, true);     // 16 <-> 32: 0.39 -> 0.34 ms
    if (cy > 16) BP_WB_(4, 4, 2, 1, 2, 1, 1, 4);
    if (cx > 16) BP_WB_(4, 4, 2, 2, 1, 1, 1, 4);
    BP_WB_(4, 4, 2, 1, 1, 1, 1, 4);
  }
  if (k == 5 && s == 1 && cx <= 16 && cy <= 16) BP_WB_(5, 5, 1, 1, 1, 1, 1, 8);
  if (k == 7 && s == 1 && cx <= 16 && cy <= 16) BP_WB_(4, 7, 1, 1, 1, 1, 1, 8);
#undef BP_WB_
  return BP_EUNSUPPORTED;
}

