// Unit-stride k7 convolutions gathering 8 channels into 16 (arch p_y_z_out: the data gradient of the head's first
// layer Conv2d 16 -> 8, k7; equally the forward of an 8 -> 16 k7 layer): 49 taps x 8 channels is a K of 392 with only
// 16 x 16 outputs per MFMA tile, so the general igemm -- one weight slab per tap staged through LDS -- spends its time
// between barriers.  Here, as in conv_stem.hip, the K dimension of the fp32 MFMA walks the FLATTENED (tap column,
// channel) index of one tap row, kf = 8*v + c: the window of pixel x in an NHWC row is 56 consecutive floats, i.e. 14
// K-groups of 4 per tap row, 98 MFMAs per 16 pixels with no padding, and all 98 weight fragments live in registers for
// the whole kernel (one VGPR each): no weight staging, no barrier inside a tile.
//   D[co][px] = sum_{u, g} W[u][g] (16 co x 4 kf)  x  Xflat[u][g] (4 kf x 16 px)
// The input tile is staged as two channel-quad planes [quad][row][pixel][4], so the 64 lanes of an operand read
// (pixel lm, channel kq of quad g % 2, tap column g / 2) touch 256 contiguous bytes.  Workgroups walk the tile
// sequence with a grid stride, the next tile in flight in registers (unconditional loads from clamped coordinates).
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int K = 7, CG = 8, CO = 16;
constexpr int NQ = CG / 4;                            // channel quads per pixel
constexpr int NG = K * NQ;                            // K-groups per tap row (14)
constexpr int TH = 8, TW = 64;                        // output tile of a workgroup: 4 waves x 2 rows x 64 columns
constexpr int IH = TH + K - 1, IWP = TW + K - 1;      // staged rows / pixels per row (14 x 70)
constexpr int PLANE = IH * IWP * 4;                   // floats of one quad plane
constexpr int NU = IH * IWP * NQ;                     // float4 units of the tile (1960)
constexpr int SL = (NU + 255) / 256;                  // per thread (8)

struct FlatArgs {
  const float* in; int h, w, in_cs, in_co;
  float* out; int out_cs, out_co;
  const float* bias;
  const float* wp;            // [u][g][kq][co]
  PW pw;
  int n, tiles_x, tiles_y, i0, in_vec;
  // mode-2 statistics (data gradient): the sums {sum g, sum g*raw} of bp_act_backward for the layer that produced the
  // tensor this kernel writes the gradient of, g = out * act'(spw(raw)); partial rows [workgroup][2][16]
  const float* raw; int raw_cs, raw_co;
  PW spw;
  double* stat;
};

template <bool OUT_VEC, bool STATS2>
__global__ __launch_bounds__(256, 2) void flat_k7_kernel(FlatArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[NQ * PLANE];
  __shared__ double lsum[STATS2 ? 8 * 256 : 1];      // per-lane running sums (registers hold the weights)
  __shared__ double red[STATS2 ? 4 : 1][2][CO];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;

  float wreg[K][NG];
#pragma unroll
  for (int u = 0; u < K; ++u)
#pragma unroll
    for (int g = 0; g < NG; ++g) wreg[u][g] = a.wp[((u * NG + g) * 4 + kq) * CO + lm];
  v4f b4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) b4 = v4f{a.bias[4 * kq], a.bias[4 * kq + 1], a.bias[4 * kq + 2], a.bias[4 * kq + 3]};

  if constexpr (STATS2) {
#pragma unroll
    for (int q = 0; q < 8; ++q) lsum[q * 256 + tid] = 0.0;
  }
  float ssc[STATS2 ? 4 : 1], ssf[STATS2 ? 4 : 1], ssl[STATS2 ? 4 : 1];   // activation of the lane's 4 channels
  if constexpr (STATS2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool on = a.spw.scale != nullptr;
      ssc[q] = on ? a.spw.scale[4 * kq + q] : 1.f;
      ssf[q] = on ? a.spw.shift[4 * kq + q] : 0.f;
      ssl[q] = on ? a.spw.slope[4 * kq + q] : 1.f;
    }
  }
  // staging: unit e = (pixel, quad); a thread always holds the same quad
  const int q4 = tid % NQ;
  const PW4 p4 = pw4_load(a.pw, q4 * 4, CG);
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float4 stage[SL];
  unsigned inside = 0;
  auto fetch = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH + a.i0, x0 = (r % a.tiles_x) * TW + a.i0;
    const float* in_n = a.in + (int64_t)n * a.h * a.w * a.in_cs + a.in_co + q4 * 4;
    unsigned in = 0;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + i * 256;
      const int pix = e / NQ, col = pix % IWP, row = pix / IWP;
      const int iy = y0 + row, ix = x0 + col;
      if (e < NU && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) in |= 1u << i;
      const int cy = min(max(iy, 0), a.h - 1), cx = min(max(ix, 0), a.w - 1);
      const float* p = in_n + ((int64_t)cy * a.w + cx) * a.in_cs;
      stage[i] = a.in_vec ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
    }
    inside = in;
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + i * 256;
      if (e < NU) {
        const float4 v = pw4_apply4(p4, stage[i]);
        const bool in = (inside >> i) & 1u;
        *reinterpret_cast<float4*>(tile + q4 * PLANE + (e / NQ) * 4) =
            make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
      }
    }
  };

  int t = blockIdx.x;
  if (t < ntiles) { fetch(t); commit(); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH, x0 = (r % a.tiles_x) * TW;
    float* out_n = a.out + (int64_t)n * a.h * a.w * a.out_cs + a.out_co;
    float ps[STATS2 ? 8 : 1] = {};                     // fp32 partial sums over the wave's 8 pixel groups of this tile
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll 1
      for (int ct = 0; ct < TW / 16; ++ct) {
        v4f acc = b4;
        const float* base = tile + (row * IWP + ct * 16 + lm) * 4 + kq;
        float4 rq = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (STATS2) {                        // (issued before the MFMA chain, used after it)
          const int Yc = min(y0 + row, a.h - 1), Xc = min(x0 + ct * 16 + lm, a.w - 1);
          rq = *reinterpret_cast<const float4*>(a.raw + (((int64_t)n * a.h + Yc) * a.w + Xc) * a.raw_cs + a.raw_co + 4 * kq);
          __builtin_amdgcn_sched_barrier(0);           // (keep the load in front of the MFMA chain)
        }
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
          for (int g = 0; g < NG; ++g)        // tap column g / NQ, channel quad g % NQ
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[u][g], base[(g % NQ) * PLANE + (u * IWP + g / NQ) * 4], acc, 0, 0,
                                                       0);
        const int Y = y0 + row, X = x0 + ct * 16 + lm;
        if (Y < a.h && X < a.w) {
          float* o = out_n + ((int64_t)Y * a.w + X) * a.out_cs + 4 * kq;
          if constexpr (OUT_VEC) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
          else { o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3]; }
          if constexpr (STATS2) {
            const float rv[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float tt = fmaf(rv[q], ssc[q], ssf[q]);
              const float g = tt > 0.f ? acc[q] : acc[q] * ssl[q];
              ps[q] += g;
              ps[4 + q] = fmaf(g, rv[q], ps[4 + q]);
            }
          }
        }
      }
    }
    if constexpr (STATS2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) lsum[q * 256 + tid] += (double)ps[q];
    }
    __syncthreads();                 // every wave is done reading the tile
    if (tn < ntiles) commit();
    __syncthreads();
  }
  if constexpr (STATS2) {
    double s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = lsum[q * 256 + tid]; s2[q] = lsum[(4 + q) * 256 + tid]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s1[q] += __shfl_xor(s1[q], off, 16);
        s2[q] += __shfl_xor(s2[q], off, 16);
      }
    if (lm == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { red[wave][0][4 * kq + q] = s1[q]; red[wave][1][4 * kq + q] = s2[q]; }
    }
    __syncthreads();
    if (tid < 2 * CO) {
      const int sidx = tid / CO, c = tid % CO;
      a.stat[(int64_t)blockIdx.x * 2 * CO + tid] = ((red[0][sidx][c] + red[1][sidx][c]) + red[2][sidx][c]) + red[3][sidx][c];
    }
  }
}

struct FlatPackArgs { const float* w; float* dst; int64_t sa, sb; int flip; };
__global__ void flat_pack_kernel(FlatPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // ((u*NG + g)*4 + kq)*16 + co
  if (i >= K * NG * 4 * CO) return;
  const int co = i % CO, kf = (i / CO) % (NG * 4), u = i / (CO * NG * 4);
  const int g = kf / 4, kq = kf % 4;
  const int v = g / NQ, c = (g % NQ) * 4 + kq;               // tap column, gathered channel
  const int ky = a.flip ? K - 1 - u : u, kx = a.flip ? K - 1 - v : v;
  a.dst[i] = a.w[c * a.sa + co * a.sb + ky * K + kx];
}

int flat_grid(int ntiles) {
  static const int cap = getenv("BP_FLAT_GRID") ? atoi(getenv("BP_FLAT_GRID")) : 512;
  return ntiles < cap ? ntiles : cap;
}

}  // namespace

bool bp_flat_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOFLAT") != nullptr;
  return !off && g.k == K && g.stride == 1 && g.pad == (K - 1) / 2 && g.cin_g == CG && g.cout_g == CO && g.nphase == 1;
}

int64_t bp_flat_packed_floats() { return (int64_t)K * NG * 4 * CO; }

int bp_flat_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  FlatPackArgs a{w_torch, packed, wm.sa, wm.sb, g.gather_transposed};
  hipLaunchKernelGGL(flat_pack_kernel, dim3((K * NG * 4 * CO + 255) / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

static int flat_k7_tiles(const bp_view* out) { return bp_ceil_div(out->w, TW) * bp_ceil_div(out->h, TH) * out->n; }

// mode-2 statistics are offered where the raw tensor (same grid as `out`) can be read 16 bytes at a time
size_t bp_flat_stats_workspace(const bp_view* out, int mode) {
  if (mode != 2 || !bp_view_vec4(out)) return 0;
  return (size_t)flat_grid(flat_k7_tiles(out)) * 2 * CO * sizeof(double);
}

int bp_flat_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  FlatArgs a{};
  if (sr) {
    if (sr->mode != 2 || bias || !sr->raw || !bp_view_vec4(sr->raw) || !bp_view_vec4(out) || sr->raw->n != out->n ||
        sr->raw->h != out->h || sr->raw->w != out->w || sr->raw->c != CO)
      return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < bp_flat_stats_workspace(out, 2) || !sr->sums) return BP_EWORKSPACE;
    a.raw = sr->raw->ptr; a.raw_cs = sr->raw->cstride; a.raw_co = sr->raw->coff; a.spw = sr->spw;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.in = in->ptr; a.h = in->h; a.w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.bias = bias; a.pw = pw; a.n = in->n;
  a.tiles_x = bp_ceil_div(out->w, TW); a.tiles_y = bp_ceil_div(out->h, TH);
  a.i0 = g.gather_transposed ? bp_t_i0(0, g.pad, 1, K) : -g.pad;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  if (ntiles > 0x7fffffff) return BP_EUNSUPPORTED;
  const int grid = flat_grid((int)ntiles);
  if (sr) hipLaunchKernelGGL((flat_k7_kernel<true, true>), dim3(grid), dim3(256), 0, st, a);
  else if (bp_view_vec4(out)) hipLaunchKernelGGL((flat_k7_kernel<true, false>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((flat_k7_kernel<false, false>), dim3(grid), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  if (sr) return bp_sum_partials(a.stat, grid, 2 * CO, sr->sums, st);
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 k4 transposed forms gathering 32 channels into 16 (ConvTranspose2d 32 -> 16 k4s2 forward, arch
// p_y_z_in.22, and the data gradient of Conv2d 16 -> 32 k4s2, p_y_z_in.3): four output phases of 2 x 2 taps each.
// The same flattened-K scheme per phase: the window of phase-grid pixel x in a tap row is two adjacent input pixels =
// 64 consecutive floats = 16 K-groups, 32 MFMAs per phase and 16 pixels with no padding; the 4 x 2 x 16 = 128 weight
// fragments of all four phases stay in registers, the input tile (staged once for all four phases) is eight channel-
// quad planes, and a wave writes the four phases of its pixels back to back.  Batch-norm sums (forward, mode 1) are
// kept per lane in double over all the tiles of the workgroup, as in conv_stem.hip.
namespace {

constexpr int TK = 4, TS = 2, TPAD = 1, TCG = 32, TCO = 16;
constexpr int TNQ = TCG / 4;                          // 8 quad planes
constexpr int TT = TK / TS;                           // taps per phase and dimension (2)
constexpr int TNG = TT * TNQ;                         // K-groups per tap row (16)
constexpr int TTH = 8, TTW = 16;                      // phase-grid tile of a workgroup: 4 waves x 2 rows x 16 columns
                                                      // (128 weight registers: the staging set has to stay small)
constexpr int TIH = TTH + TT, TIW = TTW + TT;         // staged input rows / pixels per row (10 x 34)
constexpr int TPLANE = TIH * TIW * 4;
constexpr int TNU = TIH * TIW * TNQ;                  // float4 units (2720)
constexpr int TSL = (TNU + 255) / 256;                // per thread (11)

struct FlatTArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_h, out_w, out_cs, out_co;
  const float* wp;            // [phase][t][g][kq][co]
  const float* bias;
  PW pw;
  int n, tiles_x, tiles_y, in_vec;
  double* stat;               // partial sums [workgroup][2][16] or nullptr
};

template <bool STATS>
__global__ __launch_bounds__(256, 2) void flat_t4_kernel(FlatTArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[TNQ * TPLANE];
  __shared__ double red[4][2][TCO];
  __shared__ double lsum[STATS ? 8 * 256 : 1];       // per-lane running sums (registers are full of weights)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;

  float wreg[4][TT][TNG];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int g = 0; g < TNG; ++g) wreg[ph][t][g] = a.wp[(((ph * TT + t) * TNG + g) * 4 + kq) * TCO + lm];

  v4f b4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) b4 = v4f{a.bias[4 * kq], a.bias[4 * kq + 1], a.bias[4 * kq + 2], a.bias[4 * kq + 3]};
  const int q8 = tid % TNQ;
  __shared__ float lpw[3][TCG];                     // pending activation of the gathered channels (read at commit time)
  if (tid < TCG) {
    const bool on = a.pw.scale != nullptr;
    lpw[0][tid] = on ? a.pw.scale[tid] : 1.f;
    lpw[1][tid] = on ? a.pw.shift[tid] : 0.f;
    lpw[2][tid] = on ? a.pw.slope[tid] : 1.f;
  }
  __syncthreads();
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float4 stage[TSL];
  unsigned inside = 0;
  // the tile covers phase-grid rows qy0 .. qy0+TTH-1; gathered rows qy0 - 1 .. qy0 + TTH (phase 0 starts at -1, phase 1 at 0)
  auto fetch = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TTH - 1, x0 = (r % a.tiles_x) * TTW - 1;
    const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co + q8 * 4;
    unsigned in = 0;
#pragma unroll
    for (int i = 0; i < TSL; ++i) {
      const int e = tid + i * 256;
      const int pix = e / TNQ, col = pix % TIW, row = pix / TIW;
      const int iy = y0 + row, ix = x0 + col;
      if (e < TNU && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) in |= 1u << i;
      const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
      const float* p = in_n + ((int64_t)cy * a.in_w + cx) * a.in_cs;
      stage[i] = a.in_vec ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
    }
    inside = in;
  };
  auto commit = [&]() {
    PW4 p4;
    p4.on = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p4.sc[j] = lpw[0][q8 * 4 + j]; p4.sf[j] = lpw[1][q8 * 4 + j]; p4.sl[j] = lpw[2][q8 * 4 + j]; }
#pragma unroll
    for (int i = 0; i < TSL; ++i) {
      const int e = tid + i * 256;
      if (e < TNU) {
        const float4 v = pw4_apply4(p4, stage[i]);
        const bool in = (inside >> i) & 1u;
        *reinterpret_cast<float4*>(tile + q8 * TPLANE + (e / TNQ) * 4) =
            make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
      }
    }
  };

  if constexpr (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) lsum[q * 256 + tid] = 0.0;
  }
  int t = blockIdx.x;
  if (t < ntiles) { fetch(t); commit(); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
    const int n = t / per_img, r = t % per_img;
    const int qy0 = (r / a.tiles_x) * TTH, qx0 = (r % a.tiles_x) * TTW;
    float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll 1
      for (int ct = 0; ct < TTW / 16; ++ct) {
        const float* base = tile + (row * TIW + ct * 16 + lm) * 4 + kq;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          const int py = ph >> 1, px = ph & 1;              // staged row of tap t: row + py + t, column: + px + s
          v4f acc = b4;
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int g = 0; g < TNG; ++g)                   // tap column s = g / TNQ, channel quad g % TNQ
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  wreg[ph][tt][g], base[(g % TNQ) * TPLANE + ((py + tt) * TIW + px + g / TNQ) * 4], acc, 0, 0, 0);
          const int Y = 2 * (qy0 + row) + py, X = 2 * (qx0 + ct * 16 + lm) + px;
          if (Y < a.out_h && X < a.out_w) {
            float* o = out_n + ((int64_t)Y * a.out_w + X) * a.out_cs + 4 * kq;
            *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if constexpr (STATS) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                lsum[q * 256 + tid] += (double)acc[q];
                lsum[(4 + q) * 256 + tid] = fma((double)acc[q], (double)acc[q], lsum[(4 + q) * 256 + tid]);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    if (tn < ntiles) commit();
    __syncthreads();
  }
  if constexpr (STATS) {
    double s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = lsum[q * 256 + tid]; s2[q] = lsum[(4 + q) * 256 + tid]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s1[q] += __shfl_xor(s1[q], off, 16);
        s2[q] += __shfl_xor(s2[q], off, 16);
      }
    if (lm == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { red[wave][0][4 * kq + q] = s1[q]; red[wave][1][4 * kq + q] = s2[q]; }
    }
    __syncthreads();
    if (tid < 2 * TCO) {
      const int s = tid / TCO, c = tid % TCO;
      a.stat[(int64_t)blockIdx.x * 2 * TCO + tid] = red[0][s][c] + red[1][s][c] + red[2][s][c] + red[3][s][c];
    }
  }
}

struct FlatTPackArgs { const float* w; float* dst; int64_t sa, sb; };
__global__ void flat_t4_pack_kernel(FlatTPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (((ph*TT + t)*TNG + g)*4 + kq)*16 + co
  if (i >= 4 * TT * TNG * 4 * TCO) return;
  const int co = i % TCO, kf = (i / TCO) % (TNG * 4), t = (i / (TCO * TNG * 4)) % TT, ph = i / (TCO * TNG * 4 * TT);
  const int g = kf / 4, kq = kf % 4;
  const int s = g / TNQ, c = (g % TNQ) * 4 + kq;            // tap column, gathered channel
  const int ky = bp_t_ky(ph >> 1, TPAD, TS, TT, t), kx = bp_t_ky(ph & 1, TPAD, TS, TT, s);
  a.dst[i] = a.w[c * a.sa + co * a.sb + ky * TK + kx];
}

}  // namespace

bool bp_flat_t4_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOFLAT") != nullptr;
  return !off && g.gather_transposed && g.k == TK && g.stride == TS && g.pad == TPAD && g.cin_g == TCG && g.cout_g == TCO &&
         g.nphase == 2;
}

int64_t bp_flat_t4_packed_floats() { return (int64_t)4 * TT * TNG * 4 * TCO; }

int bp_flat_t4_pack(const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  FlatTPackArgs a{w_torch, packed, wm.sa, wm.sb};
  hipLaunchKernelGGL(flat_t4_pack_kernel, dim3((4 * TT * TNG * 4 * TCO + 255) / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

static int flat_t4_tiles(const bp_view* out) {
  return bp_ceil_div(bp_ceil_div(out->w, 2), TTW) * bp_ceil_div(bp_ceil_div(out->h, 2), TTH) * out->n;
}

size_t bp_flat_t4_stats_workspace(const bp_view* out) {
  return (size_t)flat_grid(flat_t4_tiles(out)) * 2 * TCO * sizeof(double);
}

int bp_flat_t4_run(const bp_view* in, const PW& pw, const float* packed, const float* bias, const bp_view* out,
                   hipStream_t st, const IgemmStatsReq* sr) {
  if ((sr && (bias || sr->mode != 1)) || !bp_view_vec4(out)) return BP_EUNSUPPORTED;
  FlatTArgs a{};
  a.bias = bias;
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.pw = pw; a.n = in->n; a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.tiles_x = bp_ceil_div(bp_ceil_div(out->w, 2), TTW); a.tiles_y = bp_ceil_div(bp_ceil_div(out->h, 2), TTH);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  if (ntiles > 0x7fffffff) return BP_EUNSUPPORTED;
  const int grid = flat_grid((int)ntiles);
  if (sr) {
    if (!sr->ws || sr->ws_bytes < bp_flat_t4_stats_workspace(out) || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  if (sr) hipLaunchKernelGGL(flat_t4_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(flat_t4_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  if (sr) return bp_sum_partials_req(a.stat, grid, 2 * TCO, sr, st);
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 k4 conv form gathering 32 channels into 64 (Conv2d 32 -> 64 k4s2 forward, arch p_y_z_in.6, and the data
// gradient of ConvTranspose2d 64 -> 32 k4s2, p_y_z_in.19): K = 16 taps x 32 channels = 128 K-groups.  The weights of
// 16 produced channels are 128 fragments = the registers of one wave, so a workgroup is 8 waves: four blocks of 16
// produced channels x two row pairs of a 4 x 16 output tile; every wave walks all 128 K-groups of its two rows with one
// 4-byte LDS read per MFMA.  The gathered tile (10 x 34 pixels, eight channel-quad planes, columns split by parity so
// that the 16 pixels of a row read 256 contiguous bytes for any tap) is double buffered: the next tile is committed
// to the other buffer as soon as a wave is done with its MFMAs, one barrier per tile.  Batch-norm sums (forward,
// mode 1) are kept per lane in double over all the tiles of the workgroup.
namespace {

constexpr int GK = 4, GS = 2, GPAD = 1;
constexpr int GTW = 16;                               // output columns of a tile
constexpr int GIW = GS * GTW + 2;                     // gathered columns of a tile (34)
constexpr int GHALF = GIW / 2;                        // columns of one parity (17)
constexpr int GNT = 512;                              // threads

struct FlatGArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_h, out_w, out_cs, out_co;
  const float* wp;            // [co block][tap][quad][kq][co]
  const float* bias;
  PW pw;
  int n, tiles_x, tiles_y, in_vec;
  double* stat;               // partial sums [workgroup][2][64] or nullptr
};

// GCG gathered -> GCO produced channels: 32 -> 64 (four channel blocks x two row pairs, tile 4 x 16) or 16 -> 32 (two
// channel blocks x four row pairs, tile 8 x 16)
template <int GCG, int GCO, bool STATS>
__global__ __launch_bounds__(GNT) void flat_g4_kernel(FlatGArgs a) {
  constexpr int GNQ = GCG / 4;                          // quad planes
  constexpr int NCB = GCO / 16;                         // blocks of 16 produced channels (4 or 2)
  constexpr int GTH = 2 * (8 / NCB);                    // output rows of a tile (4 or 8)
  constexpr int GIH = GS * GTH + 2;                     // gathered rows of a tile (10 or 18)
  constexpr int GPLANE = GIH * GIW * 4;                 // floats of a quad plane
  constexpr int GTILE = GNQ * GPLANE;                   // floats of a tile
  constexpr int GNU = GIH * GIW * GNQ;                  // float4 units
  constexpr int GSL = (GNU + GNT - 1) / GNT;            // units per thread
  constexpr int GNW = GK * GK * GNQ;                    // weight fragments of a wave (128 or 64)
  __shared__ __attribute__((aligned(16))) float tile[2 * GTILE];
  __shared__ double red[STATS ? 8 : 1][2][16];
  __shared__ float lpw[3][GCG];                     // pending activation of the gathered channels (read at commit time)
  __shared__ double lsum[STATS ? 8 * GNT : 1];   // per-lane running sums {sum, sum of squares} x 4 channels (not registers)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave % NCB, rh = wave / NCB;         // block of 16 produced channels, row pair of the tile
  const int lm = lane & 15, kq = lane >> 4;

  float wreg[GNW];
#pragma unroll
  for (int i = 0; i < GNW; ++i) wreg[i] = a.wp[(((int64_t)cb * GNW + i) * 4 + kq) * 16 + lm];

  const int q8 = tid % GNQ;
  if (tid < GCG) {
    const bool on = a.pw.scale != nullptr;
    lpw[0][tid] = on ? a.pw.scale[tid] : 1.f;
    lpw[1][tid] = on ? a.pw.shift[tid] : 0.f;
    lpw[2][tid] = on ? a.pw.slope[tid] : 1.f;
  }
  __syncthreads();
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float4 stage[GSL];
  unsigned inside = 0;
  auto fetch = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    const int y0 = GS * (r / a.tiles_x) * GTH - GPAD, x0 = GS * (r % a.tiles_x) * GTW - GPAD;
    const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co + q8 * 4;
    unsigned in = 0;
#pragma unroll
    for (int i = 0; i < GSL; ++i) {
      const int e = tid + i * GNT;
      const int pix = e / GNQ, col = pix % GIW, row = pix / GIW;
      const int iy = y0 + row, ix = x0 + col;
      in |= ((unsigned)(e < GNU) & (unsigned)(iy >= 0) & (unsigned)(iy < a.in_h) & (unsigned)(ix >= 0) &
             (unsigned)(ix < a.in_w)) << i;
      const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
      const float* p = in_n + ((int64_t)cy * a.in_w + cx) * a.in_cs;
      stage[i] = a.in_vec ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
    }
    inside = in;
  };
  auto commit = [&](float* buf) {
    PW4 p4;                                            // (12 LDS reads per tile instead of 12 registers held)
    p4.on = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p4.sc[j] = lpw[0][q8 * 4 + j]; p4.sf[j] = lpw[1][q8 * 4 + j]; p4.sl[j] = lpw[2][q8 * 4 + j]; }
#pragma unroll
    for (int i = 0; i < GSL; ++i) {
      const int e = tid + i * GNT;
      if (e < GNU) {
        const int pix = e / GNQ, col = pix % GIW, row = pix / GIW;
        const float4 v = pw4_apply4(p4, stage[i]);
        const bool in = (inside >> i) & 1u;
        *reinterpret_cast<float4*>(buf + q8 * GPLANE + ((row * 2 + (col & 1)) * GHALF + (col >> 1)) * 4) =
            make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
      }
    }
  };

  if constexpr (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) lsum[q * GNT + tid] = 0.0;
  }
  int t = blockIdx.x, cur = 0;
  if (t < ntiles) { fetch(t); commit(tile); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
    const int n = t / per_img, r = t % per_img;
    const int oy0 = (r / a.tiles_x) * GTH + 2 * rh, ox = (r % a.tiles_x) * GTW + lm;
    // rows 2 rh and 2 rh + 1 of the tile: gathered rows 2 row + ky, column 2 lm + kx -> parity kx & 1, half lm + (kx >> 1)
    const float* base = tile + cur * GTILE + ((4 * rh) * 2 * GHALF + lm) * 4 + kq;
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // eight K-groups (both rows) per stage; the fragments of the next stage are read before the MFMAs of this one
    float f0[2][8], f1[2][8];
    auto frags = [&](int buf, int g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = g * 8 + i, tap = k / GNQ, j = k % GNQ, ky = tap / GK, kx = tap % GK;
        const int off = j * GPLANE + ((ky * 2 + (kx & 1)) * GHALF + (kx >> 1)) * 4;
        f0[buf][i] = base[off];
        f1[buf][i] = base[off + 2 * 2 * GHALF * 4];
      }
    };
    frags(0, 0);
#pragma unroll
    for (int g = 0; g < GNW / 8; ++g) {
      if (g + 1 < GNW / 8) frags((g + 1) & 1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g * 8 + i], f0[g & 1][i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g * 8 + i], f1[g & 1][i], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co + 16 * cb + 4 * kq;
    if (a.bias) {                                        // (uniform; read here, not held over the MFMA phase)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float b = a.bias[16 * cb + 4 * kq + q]; acc0[q] += b; acc1[q] += b; }
    }
    if (ox < a.out_w) {
      if (oy0 < a.out_h) {
        *reinterpret_cast<float4*>(out_n + ((int64_t)oy0 * a.out_w + ox) * a.out_cs) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            lsum[q * GNT + tid] += (double)acc0[q];
            lsum[(4 + q) * GNT + tid] = fma((double)acc0[q], (double)acc0[q], lsum[(4 + q) * GNT + tid]);
          }
        }
      }
      if (oy0 + 1 < a.out_h) {
        *reinterpret_cast<float4*>(out_n + ((int64_t)(oy0 + 1) * a.out_w + ox) * a.out_cs) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            lsum[q * GNT + tid] += (double)acc1[q];
            lsum[(4 + q) * GNT + tid] = fma((double)acc1[q], (double)acc1[q], lsum[(4 + q) * GNT + tid]);
          }
        }
      }
    }
    if (tn < ntiles) commit(tile + (cur ^ 1) * GTILE);     // (nobody reads that buffer in this iteration)
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (STATS) {
    double s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = lsum[q * GNT + tid]; s2[q] = lsum[(4 + q) * GNT + tid]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s1[q] += __shfl_xor(s1[q], off, 16);
        s2[q] += __shfl_xor(s2[q], off, 16);
      }
    if (lm == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { red[wave][0][4 * kq + q] = s1[q]; red[wave][1][4 * kq + q] = s2[q]; }
    }
    __syncthreads();
    if (tid < 2 * GCO) {
      const int sidx = tid / GCO, c = tid % GCO;
      double v = red[c / 16][sidx][c % 16];
#pragma unroll
      for (int r2 = 1; r2 < 8 / NCB; ++r2) v += red[r2 * NCB + c / 16][sidx][c % 16];
      a.stat[(int64_t)blockIdx.x * 2 * GCO + tid] = v;
    }
  }
}

struct FlatGPackArgs { const float* w; float* dst; int64_t sa, sb; int cg, co; };
__global__ void flat_g4_pack_kernel(FlatGPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (((cb*16 + tap)*GNQ + j)*4 + kq)*16 + lm
  if (i >= a.cg * a.co * GK * GK) return;
  const int GNQ = a.cg / 4;
  const int lm = i % 16, kq = (i / 16) % 4, j = (i / 64) % GNQ, tap = (i / (64 * GNQ)) % (GK * GK), cb = i / (64 * GNQ * GK * GK);
  const int c = 4 * j + kq, co = 16 * cb + lm;              // gathered channel, produced channel
  a.dst[i] = a.w[c * a.sa + co * a.sb + tap];
}

int flat_g4_grid(int ntiles) {
  static const int cap = getenv("BP_FLATG_GRID") ? atoi(getenv("BP_FLATG_GRID")) : 256;
  return ntiles < cap ? ntiles : cap;
}

}  // namespace

bool bp_flat_g4_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOFLAT") != nullptr || getenv("BP_NOFLATG") != nullptr;
  static const bool thin = getenv("BP_FLATG_THIN") != nullptr;     // (16 -> 32: no faster than the weights-resident igemm)
  return !off && !g.gather_transposed && g.k == GK && g.stride == GS && g.pad == GPAD && g.nphase == 1 && g.IS == GS &&
         ((g.cin_g == 32 && g.cout_g == 64) || (thin && g.cin_g == 16 && g.cout_g == 32));
}

int64_t bp_flat_g4_packed_floats(const ConvGeom& g) { return (int64_t)g.cin_g * g.cout_g * GK * GK; }

int bp_flat_g4_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  FlatGPackArgs a{w_torch, packed, wm.sa, wm.sb, g.cin_g, g.cout_g};
  hipLaunchKernelGGL(flat_g4_pack_kernel, dim3((g.cin_g * g.cout_g * GK * GK + 255) / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

static int flat_g4_th(const bp_view* out) { return out->c == 64 ? 4 : 8; }
static int flat_g4_tiles(const bp_view* out) {
  return bp_ceil_div(out->w, GTW) * bp_ceil_div(out->h, flat_g4_th(out)) * out->n;
}

size_t bp_flat_g4_stats_workspace(const bp_view* out) {
  return (size_t)flat_g4_grid(flat_g4_tiles(out)) * 2 * out->c * sizeof(double);
}

int bp_flat_g4_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                   const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  if ((sr && (bias || sr->mode != 1)) || !bp_view_vec4(out)) return BP_EUNSUPPORTED;
  FlatGArgs a{};
  a.bias = bias;
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.pw = pw; a.n = in->n; a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.tiles_x = bp_ceil_div(out->w, GTW); a.tiles_y = bp_ceil_div(out->h, flat_g4_th(out));
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  if (ntiles > 0x7fffffff) return BP_EUNSUPPORTED;
  const int grid = flat_g4_grid((int)ntiles);
  if (sr) {
    if (!sr->ws || sr->ws_bytes < bp_flat_g4_stats_workspace(out) || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  const dim3 gd(grid), bd(GNT);
  if (g.cin_g == 32) {
    if (sr) hipLaunchKernelGGL((flat_g4_kernel<32, 64, true>), gd, bd, 0, st, a);
    else hipLaunchKernelGGL((flat_g4_kernel<32, 64, false>), gd, bd, 0, st, a);
  } else {
    if (sr) hipLaunchKernelGGL((flat_g4_kernel<16, 32, true>), gd, bd, 0, st, a);
    else hipLaunchKernelGGL((flat_g4_kernel<16, 32, false>), gd, bd, 0, st, a);
  }
  BP_CHECK_LAUNCH();
  if (sr) return bp_sum_partials_req(a.stat, grid, 2 * g.cout_g, sr, st);
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 k4 transposed forms gathering 64 channels into 32 (ConvTranspose2d 64 -> 32 k4s2 forward, arch
// p_y_z_in.19, and the data gradient of Conv2d 32 -> 64 k4s2, p_y_z_in.6).  Eight waves = the four output phases x two
// blocks of 16 produced channels: a wave keeps the 2 x 2 taps x 16 channel quads = 64 weight fragments of ITS phase
// and channel block in registers and walks the 8 rows of the phase-grid tile (8 x 16 pixels; the gathered tile,
// 10 x 18 pixels in 16 quad planes, serves all four phases).  Double-buffered tile, one barrier per tile,
// fragments read one tap ahead, batch-norm sums per lane in LDS -- as in flat_g4_kernel.
namespace {

constexpr int WTH = 8, WTW = 16;                      // phase-grid tile
constexpr int WIH = WTH + 2, WIW = WTW + 2;           // gathered rows / columns (10 x 18)
constexpr int WPLANE = WIH * WIW * 4;

struct FlatWArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_h, out_w, out_cs, out_co;
  const float* wp;            // [co block][phase][t][s][quad][kq][co]
  const float* bias;
  PW pw;
  int n, tiles_x, tiles_y, in_vec;
  double* stat;               // partial sums [workgroup][2][32] or nullptr
};

// WCG gathered -> WCO produced channels: 64 -> 32 (two channel blocks x four phases) or 32 -> 16 (one channel block:
// the second group of four waves takes the lower half of the tile's rows)
template <int WCG, int WCO, bool STATS>
__global__ __launch_bounds__(GNT) void flat_t64_kernel(FlatWArgs a) {
  constexpr int WNQ = WCG / 4;                          // quad planes
  constexpr int WTILE = WNQ * WPLANE;                   // floats of a tile
  constexpr int WNU = WIH * WIW * WNQ;                  // float4 units
  constexpr int WSL = (WNU + GNT - 1) / GNT;            // per thread
  constexpr int WNW = 4 * WNQ;                          // weight fragments of a wave
  constexpr int NCB = WCO / 16;                         // blocks of 16 produced channels (1 or 2)
  constexpr int NRP = WTH / 2 / (2 / NCB);              // row pairs a wave walks
  constexpr int QG = WNQ / 8;                           // fragment groups per tap
  __shared__ __attribute__((aligned(16))) float tile[2 * WTILE];
  __shared__ double red[STATS ? 8 : 1][2][16];
  __shared__ double lsum[STATS ? 8 * GNT : 1];
  __shared__ float lpw[3][WCG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave & 3, cb = NCB == 2 ? wave >> 2 : 0;   // output phase, block of 16 produced channels
  const int rp0 = NCB == 2 ? 0 : (wave >> 2) * NRP;         // first row pair of this wave
  const int py = ph >> 1, px = ph & 1;
  const int lm = lane & 15, kq = lane >> 4;

  float wreg[WNW];
#pragma unroll
  for (int i = 0; i < WNW; ++i) wreg[i] = a.wp[((((int64_t)cb * 4 + ph) * WNW + i) * 4 + kq) * 16 + lm];

  const int q16 = tid % WNQ;
  if (tid < WCG) {
    const bool on = a.pw.scale != nullptr;
    lpw[0][tid] = on ? a.pw.scale[tid] : 1.f;
    lpw[1][tid] = on ? a.pw.shift[tid] : 0.f;
    lpw[2][tid] = on ? a.pw.slope[tid] : 1.f;
  }
  if constexpr (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) lsum[q * GNT + tid] = 0.0;
  }
  __syncthreads();
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float4 stage[WSL];
  unsigned inside = 0;
  auto fetch = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * WTH - 1, x0 = (r % a.tiles_x) * WTW - 1;
    const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co + q16 * 4;
    unsigned in = 0;
#pragma unroll
    for (int i = 0; i < WSL; ++i) {
      const int e = tid + i * GNT;
      const int pix = e / WNQ, col = pix % WIW, row = pix / WIW;
      const int iy = y0 + row, ix = x0 + col;
      in |= ((unsigned)(e < WNU) & (unsigned)(iy >= 0) & (unsigned)(iy < a.in_h) & (unsigned)(ix >= 0) &
             (unsigned)(ix < a.in_w)) << i;
      const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
      const float* p = in_n + ((int64_t)cy * a.in_w + cx) * a.in_cs;
      stage[i] = a.in_vec ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
    }
    inside = in;
  };
  auto commit = [&](float* buf) {
    PW4 p4;
    p4.on = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p4.sc[j] = lpw[0][q16 * 4 + j]; p4.sf[j] = lpw[1][q16 * 4 + j]; p4.sl[j] = lpw[2][q16 * 4 + j]; }
#pragma unroll
    for (int i = 0; i < WSL; ++i) {
      const int e = tid + i * GNT;
      if (e < WNU) {
        const float4 v = pw4_apply4(p4, stage[i]);
        const bool in = (inside >> i) & 1u;
        *reinterpret_cast<float4*>(buf + q16 * WPLANE + (e / WNQ) * 4) =
            make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
      }
    }
  };

  int t = blockIdx.x, cur = 0;
  if (t < ntiles) { fetch(t); commit(tile); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
    const int n = t / per_img, r = t % per_img;
    const int qy0 = (r / a.tiles_x) * WTH, qx0 = (r % a.tiles_x) * WTW;
    float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co + 16 * cb + 4 * kq;
    const int X = 2 * (qx0 + lm) + px;
    v4f b4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) b4 = v4f{a.bias[16 * cb + 4 * kq], a.bias[16 * cb + 4 * kq + 1], a.bias[16 * cb + 4 * kq + 2], a.bias[16 * cb + 4 * kq + 3]};
    // staged row of tap t: row + py + t, column: lm + px + s
    const float* base = tile + cur * WTILE + ((py * WIW) + lm + px) * 4 + kq;
#pragma unroll 1
    for (int rp = rp0; rp < rp0 + NRP; ++rp) {         // two rows at a time (two accumulator chains)
      const float* b0 = base + (2 * rp) * WIW * 4;
      v4f acc0 = b4, acc1 = b4;
      float f0[2][8], f1[2][8];
      auto frags = [&](int buf, int g) {               // group g = (tap ts, half of the quads)
        const int ts = g / QG, tt = ts >> 1, ss = ts & 1, j0 = (g % QG) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int off = (j0 + j) * WPLANE + (tt * WIW + ss) * 4;
          f0[buf][j] = b0[off];
          f1[buf][j] = b0[off + WIW * 4];
        }
      };
      frags(0, 0);
#pragma unroll
      for (int g = 0; g < 4 * QG; ++g) {
        if (g + 1 < 4 * QG) frags((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g * 8 + j], f0[g & 1][j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g * 8 + j], f1[g & 1][j], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int Y0 = 2 * (qy0 + 2 * rp) + py;
      if (X < a.out_w) {
        if (Y0 < a.out_h) {
          *reinterpret_cast<float4*>(out_n + ((int64_t)Y0 * a.out_w + X) * a.out_cs) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
          if constexpr (STATS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              lsum[q * GNT + tid] += (double)acc0[q];
              lsum[(4 + q) * GNT + tid] = fma((double)acc0[q], (double)acc0[q], lsum[(4 + q) * GNT + tid]);
            }
          }
        }
        if (Y0 + 2 < a.out_h) {
          *reinterpret_cast<float4*>(out_n + ((int64_t)(Y0 + 2) * a.out_w + X) * a.out_cs) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
          if constexpr (STATS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              lsum[q * GNT + tid] += (double)acc1[q];
              lsum[(4 + q) * GNT + tid] = fma((double)acc1[q], (double)acc1[q], lsum[(4 + q) * GNT + tid]);
            }
          }
        }
      }
    }
    if (tn < ntiles) commit(tile + (cur ^ 1) * WTILE);
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (STATS) {
    double s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = lsum[q * GNT + tid]; s2[q] = lsum[(4 + q) * GNT + tid]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s1[q] += __shfl_xor(s1[q], off, 16);
        s2[q] += __shfl_xor(s2[q], off, 16);
      }
    if (lm == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { red[wave][0][4 * kq + q] = s1[q]; red[wave][1][4 * kq + q] = s2[q]; }
    }
    __syncthreads();
    if (tid < 2 * WCO) {
      const int sidx = tid / WCO, c = tid % WCO, w0 = NCB == 2 ? (c / 16) * 4 : 0;
      double v = ((red[w0][sidx][c % 16] + red[w0 + 1][sidx][c % 16]) + red[w0 + 2][sidx][c % 16]) + red[w0 + 3][sidx][c % 16];
      if (NCB == 1)
        v += ((red[4][sidx][c % 16] + red[5][sidx][c % 16]) + red[6][sidx][c % 16]) + red[7][sidx][c % 16];
      a.stat[(int64_t)blockIdx.x * 2 * WCO + tid] = v;
    }
  }
}

struct FlatWPackArgs { const float* w; float* dst; int64_t sa, sb; int cg, co; };
__global__ void flat_t64_pack_kernel(FlatWPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // ((((cb*4 + ph)*4 + ts)*WNQ + j)*4 + kq)*16 + lm
  if (i >= a.cg * a.co * 16) return;
  const int WNQ = a.cg / 4;
  const int lm = i % 16, kq = (i / 16) % 4, j = (i / 64) % WNQ, ts = (i / (64 * WNQ)) % 4, ph = (i / (64 * WNQ * 4)) % 4,
            cb = i / (64 * WNQ * 16);
  const int c = 4 * j + kq, co = 16 * cb + lm;              // gathered channel, produced channel
  const int ky = bp_t_ky(ph >> 1, TPAD, TS, TT, ts >> 1), kx = bp_t_ky(ph & 1, TPAD, TS, TT, ts & 1);
  a.dst[i] = a.w[c * a.sa + co * a.sb + ky * TK + kx];
}

}  // namespace

static bool flat_t64_wide(const ConvGeom& g) { return g.cin_g == 64; }

bool bp_flat_t64_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOFLAT") != nullptr || getenv("BP_NOFLATW") != nullptr;
  static const bool thin = getenv("BP_FLATW_THIN") != nullptr;       // (32 -> 16 through this kernel instead of flat_t4)
  return !off && g.gather_transposed && g.k == TK && g.stride == TS && g.pad == TPAD && g.nphase == 2 &&
         ((g.cin_g == 64 && g.cout_g == 32) || (thin && g.cin_g == 32 && g.cout_g == 16));
}

int64_t bp_flat_t64_packed_floats(const ConvGeom& g) { return (int64_t)g.cin_g * g.cout_g * 16; }

int bp_flat_t64_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  FlatWPackArgs a{w_torch, packed, wm.sa, wm.sb, g.cin_g, g.cout_g};
  hipLaunchKernelGGL(flat_t64_pack_kernel, dim3((g.cin_g * g.cout_g * 16 + 255) / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

static int flat_t64_tiles(const bp_view* out) {
  return bp_ceil_div(bp_ceil_div(out->w, 2), WTW) * bp_ceil_div(bp_ceil_div(out->h, 2), WTH) * out->n;
}

size_t bp_flat_t64_stats_workspace(const bp_view* out) {
  return (size_t)flat_g4_grid(flat_t64_tiles(out)) * 2 * out->c * sizeof(double);
}

int bp_flat_t64_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                    const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  if ((sr && (bias || sr->mode != 1)) || !bp_view_vec4(out)) return BP_EUNSUPPORTED;
  FlatWArgs a{};
  a.bias = bias;
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.pw = pw; a.n = in->n; a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.tiles_x = bp_ceil_div(bp_ceil_div(out->w, 2), WTW); a.tiles_y = bp_ceil_div(bp_ceil_div(out->h, 2), WTH);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  if (ntiles > 0x7fffffff) return BP_EUNSUPPORTED;
  const int grid = flat_g4_grid((int)ntiles);
  if (sr) {
    if (!sr->ws || sr->ws_bytes < bp_flat_t64_stats_workspace(out) || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  const dim3 gd(grid), bd(GNT);
  if (flat_t64_wide(g)) {
    if (sr) hipLaunchKernelGGL((flat_t64_kernel<64, 32, true>), gd, bd, 0, st, a);
    else hipLaunchKernelGGL((flat_t64_kernel<64, 32, false>), gd, bd, 0, st, a);
  } else {
    if (sr) hipLaunchKernelGGL((flat_t64_kernel<32, 16, true>), gd, bd, 0, st, a);
    else hipLaunchKernelGGL((flat_t64_kernel<32, 16, false>), gd, bd, 0, st, a);
  }
  BP_CHECK_LAUNCH();
  if (sr) return bp_sum_partials_req(a.stat, grid, 2 * g.cout_g, sr, st);
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Unit-stride k7 gathering 16 channels into 8 (Conv2d 16 -> 8 k7 forward, the generator head's first layer, arch
// p_y_z_out).  8 produced channels fill half an MFMA tile, so the 16 MFMA rows are (pixel offset d in {0, 1}) x 8
// channels and the 16 columns are pixel PAIRS: row (d, co) of pair p is output pixel 2p + d, K walks a window of 8
// columns x 16 channels per tap row (tap column tx = tx' - d; the two out-of-range taps are zero weights: 12.5 %
// padding), 7 x 32 = 224 K-groups.  224 weight fragments do not fit one wave, so K is split over TWO waves (two of the
// four channel quads = 112 fragments each) whose accumulators are added through LDS; eight waves = two K halves x four row pairs of an
// 8 x 32 tile.  Gathered tile: four channel-quad planes, columns split by parity (pair p, column 2p + tx' -> parity
// tx' & 1, half p + (tx' >> 1)): one 4-byte read per lane and MFMA, 256 contiguous bytes per wave.
namespace {

constexpr int HK = 7, HPAD = 3, HCG = 16, HCO = 8;
constexpr int HNQ = HCG / 4;                          // 4 quad planes
constexpr int HTH = 8, HTW = 32;                      // output tile
constexpr int HIH = HTH + HK - 1;                     // gathered rows (14)
constexpr int HIW = HTW + HK + 1;                     // gathered columns, even (40: 20 per parity; 38 are read)
constexpr int HHALF = HIW / 2;
constexpr int HPLANE = HIH * HIW * 4;
constexpr int HTILE = HNQ * HPLANE;                   // floats of a tile (8960)
constexpr int HNU = HIH * HIW * HNQ;                  // float4 units (2240)
constexpr int HSL = (HNU + GNT - 1) / GNT;            // per thread (5)
constexpr int HKS = HK * 8 * HNQ;                     // K-groups (224)
constexpr int HNW = HKS / 2;                          // weight fragments of a wave (112)

struct FlatHArgs {
  const float* in; int h, w, in_cs, in_co;
  float* out; int out_cs, out_co;
  const float* wp;            // [K half][K-group][kq][(d, co)]
  const float* bias;
  PW pw;
  int n, tiles_x, tiles_y, in_vec;
};

__global__ __launch_bounds__(GNT) void flat_h7_kernel(FlatHArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[2 * HTILE];
  __shared__ __attribute__((aligned(16))) float xch[2 * 4 * 2 * 64 * 4];   // accumulators of the upper-K waves (two tiles deep)
  __shared__ float lpw[3][HCG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >> 2, rg = wave & 3;            // K half, row pair of the tile (a SIMD hosts waves w and w + 4:
                                                      // one of each half, so the epilogue of one overlaps MFMAs of the other)
  const int lm = lane & 15, kq = lane >> 4;

  float wreg[HNW];
#pragma unroll
  for (int i = 0; i < HNW; ++i) wreg[i] = a.wp[(((int64_t)kh * HNW + i) * 4 + kq) * 16 + lm];

  const int q4 = tid % HNQ;
  if (tid < HCG) {
    const bool on = a.pw.scale != nullptr;
    lpw[0][tid] = on ? a.pw.scale[tid] : 1.f;
    lpw[1][tid] = on ? a.pw.shift[tid] : 0.f;
    lpw[2][tid] = on ? a.pw.slope[tid] : 1.f;
  }
  __syncthreads();
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float4 stage[HSL];
  unsigned inside = 0;
  // Staging of the next tile, one unit at a time and branch-free, so that it can be issued between the MFMAs of the
  // current tile (the eight waves run in lockstep from barrier to barrier: vector-ALU work outside the MFMA stream
  // is not hidden by another wave): fetch_unit = clamped address + load + validity bit, finish_unit = activation and
  // zero padding in registers, commit = the LDS stores.
  int fy0 = 0, fx0 = 0;
  const float* fin_n = nullptr;
  auto tile_setup = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    fy0 = (r / a.tiles_x) * HTH - HPAD; fx0 = (r % a.tiles_x) * HTW - HPAD;
    fin_n = a.in + (int64_t)n * a.h * a.w * a.in_cs + a.in_co + q4 * 4;
    inside = 0;
  };
  auto fetch_unit = [&](int i) {
    if (i >= HSL) return;
    const int e = tid + i * GNT;
    const int pix = e / HNQ, col = pix % HIW, row = pix / HIW;
    const int iy = fy0 + row, ix = fx0 + col;
    inside |= ((unsigned)(e < HNU) & (unsigned)(iy >= 0) & (unsigned)(iy < a.h) & (unsigned)(ix >= 0) & (unsigned)(ix < a.w)) << i;
    const int cy = min(max(iy, 0), a.h - 1), cx = min(max(ix, 0), a.w - 1);
    const float* p = fin_n + (cy * a.w + cx) * a.in_cs;
    stage[i] = a.in_vec ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
  };
  float psc[4], psf[4], psl[4];
  auto finish_unit = [&](int i) {
    if (i >= HSL) return;
    const unsigned m = 0u - ((inside >> i) & 1u);
    float v[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tt = fmaf(v[j], psc[j], psf[j]);
      v[j] = __uint_as_float(__float_as_uint(tt > 0.f ? tt : tt * psl[j]) & m);
    }
    stage[i] = make_float4(v[0], v[1], v[2], v[3]);
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < HSL; ++i) {
      const int e = tid + i * GNT;
      if (e < HNU) {
        const int pix = e / HNQ, col = pix % HIW, row = pix / HIW;
        *reinterpret_cast<float4*>(buf + q4 * HPLANE + ((row * 2 + (col & 1)) * HHALF + (col >> 1)) * 4) = stage[i];
      }
    }
  };
  auto load_pw = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) { psc[j] = lpw[0][q4 * 4 + j]; psf[j] = lpw[1][q4 * 4 + j]; psl[j] = lpw[2][q4 * 4 + j]; }
  };

  int t = blockIdx.x, cur = 0;
  if (t < ntiles) {
    tile_setup(t);
    load_pw();
#pragma unroll
    for (int i = 0; i < HSL; ++i) fetch_unit(i);
#pragma unroll
    for (int i = 0; i < HSL; ++i) finish_unit(i);
    commit(tile);
  }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    tile_setup(tn < ntiles ? tn : t);              // (past the end: this tile again, not committed)
    const int n = t / per_img, r = t % per_img;
    const int oy0 = (r / a.tiles_x) * HTH + 2 * rg, ox0 = (r % a.tiles_x) * HTW;
    // output rows 2 rg, 2 rg + 1 of the tile: gathered rows (2 rg [+ 1]) + ty; pair lm, window column tx'
    const float* base = tile + cur * HTILE + kh * 2 * HPLANE + ((2 * rg) * 2 * HHALF + lm) * 4 + kq;
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float f0[2][8], f1[2][8];
    auto frags = [&](int buf, int g) {                 // K-groups 8g .. 8g + 7 of this half: i = (ty*8 + tx')*2 + j2,
#pragma unroll                                         // channel quad 2 kh + j2 (the half is folded into `base`)
      for (int ii = 0; ii < 8; ++ii) {
        const int i = g * 8 + ii, j2 = i % 2, txp = (i / 2) % 8, ty = i / 16;
        const int off = j2 * HPLANE + ((ty * 2 + (txp & 1)) * HHALF + (txp >> 1)) * 4;
        f0[buf][ii] = base[off];
        f1[buf][ii] = base[off + 2 * HHALF * 4];
      }
    };
    frags(0, 0);
#pragma unroll
    for (int g = 0; g < HNW / 8; ++g) {                // 14 stages of 16 MFMAs
      if (g + 1 < HNW / 8) frags((g + 1) & 1, g + 1);
      if (g == 8) load_pw();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g * 8 + i], f0[g & 1][i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g * 8 + i], f1[g & 1][i], acc1, 0, 0, 0);
      }
      if (g < HSL) fetch_unit(g);                      // stages 0..4: the loads of the next tile
      if (g >= 9) finish_unit(g - 9);                  // stages 9..13: its activation, in the MFMA shadow
      if (g < HSL || g >= 9) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the upper-K wave hands its accumulators to the lower-K wave of the same rows
    float4* xw = reinterpret_cast<float4*>(xch) + ((cur * 4 + rg) * 2) * 64 + lane;
    if (kh == 1) {
      xw[0] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
      xw[64] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    }
    if (tn < ntiles) commit(tile + (cur ^ 1) * HTILE);
    __syncthreads();
    if (kh == 0) {
      const float4 u0 = xw[0], u1 = xw[64];
      acc0[0] += u0.x; acc0[1] += u0.y; acc0[2] += u0.z; acc0[3] += u0.w;
      acc1[0] += u1.x; acc1[1] += u1.y; acc1[2] += u1.z; acc1[3] += u1.w;
      const int c0 = 4 * (kq & 1);                     // lane: pixel 2 lm + (kq >> 1), channels c0 .. c0 + 3
      if (a.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float b = a.bias[c0 + q]; acc0[q] += b; acc1[q] += b; }
      }
      const int ox = ox0 + 2 * lm + (kq >> 1);
      float* out_n = a.out + (int64_t)n * a.h * a.w * a.out_cs + a.out_co + c0;
      if (ox < a.w) {
        if (oy0 < a.h)
          *reinterpret_cast<float4*>(out_n + ((int64_t)oy0 * a.w + ox) * a.out_cs) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        if (oy0 + 1 < a.h)
          *reinterpret_cast<float4*>(out_n + ((int64_t)(oy0 + 1) * a.w + ox) * a.out_cs) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
      }
    }
    cur ^= 1;
  }
}

struct FlatHPackArgs { const float* w; float* dst; int64_t sa, sb; int flip; };
__global__ void flat_h7_pack_kernel(FlatHPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // ((kh*112 + ii)*4 + kq)*16 + m,  ii = (ty*8 + tx')*2 + j2,
  if (i >= HKS * 64) return;                                // channel quad j = 2 kh + j2, m = d*8 + co
  const int m = i % 16, kq = (i / 16) % 4, k = i / 64;
  const int kh = k / HNW, ii = k % HNW;
  const int j = 2 * kh + ii % 2, txp = (ii / 2) % 8, ty = ii / 16;
  const int d = m / 8, co = m % 8, tx = txp - d, c = 4 * j + kq;
  float v = 0.f;
  if (tx >= 0 && tx < HK) {
    const int ky = a.flip ? HK - 1 - ty : ty, kx = a.flip ? HK - 1 - tx : tx;
    v = a.w[c * a.sa + co * a.sb + ky * HK + kx];
  }
  a.dst[i] = v;
}

}  // namespace

bool bp_flat_h7_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOFLAT") != nullptr || getenv("BP_NOFLATH") != nullptr;
  return !off && g.k == HK && g.stride == 1 && g.pad == HPAD && g.cin_g == HCG && g.cout_g == HCO && g.nphase == 1;
}

int64_t bp_flat_h7_packed_floats() { return (int64_t)HKS * 64; }

int bp_flat_h7_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  FlatHPackArgs a{w_torch, packed, wm.sa, wm.sb, g.gather_transposed};
  hipLaunchKernelGGL(flat_h7_pack_kernel, dim3((HKS * 64 + 255) / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_flat_h7_run(const bp_view* in, const PW& pw, const float* packed, const float* bias, const bp_view* out,
                   hipStream_t st) {
  if (!bp_view_vec4(out)) return BP_EUNSUPPORTED;
  FlatHArgs a{};
  a.bias = bias;
  a.in = in->ptr; a.h = in->h; a.w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.pw = pw; a.n = in->n; a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.tiles_x = bp_ceil_div(out->w, HTW); a.tiles_y = bp_ceil_div(out->h, HTH);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  if (ntiles > 0x7fffffff) return BP_EUNSUPPORTED;
  hipLaunchKernelGGL(flat_h7_kernel, dim3(flat_g4_grid((int)ntiles)), dim3(GNT), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
