// The k8 / stride-4 layer 8 -> 16 of the recognition and prior networks (q_x_in.3, q_y_in.3, prior_z_y.3: 256^2 -> 64^2,
// utils.py:96 conv_down(8, 16, scale=4)), fp32 MFMA 16x16x4: forward (F form) and data gradient (T form).
// 4.3 GFLOP and 150 MB per pass at batch 64 -- 27 us of matrix-core time, 30 us of HBM time; the generic kernels
// took 0.12 ms (forward, igemm_kernel<8,1,1,1>) and 0.18 ms (data gradient, igemm_dmaf_kernel<16,1,1,4>), three
// networks per step.
//
// Forward.  GEMM K = (ky, kx, ci) = 8 x 64: the 64 floats (kx, ci) of a tap row are CONSECUTIVE in the NHWC input row,
// starting at pixel 4*ox - 2.  Rows of the MFMA tile are the 16 produced channels (weights, in registers), columns 16
// neighbouring output pixels.  A lane reads 16 bytes = four consecutive k of its pixel and feeds four MFMAs with
// them (the k order inside a block of 16 is permuted to match: MFMA m takes float m of every lane).  Wave w owns tap
// rows 2w, 2w+1 for all 64 pixels of the 4 x 16 tile (32 weight registers instead of 128), the four partial tiles
// meet in LDS.  The input tile is stored in groups of 4 pixels (32 floats) padded to 36: the 16 lanes of a quarter
// wave, 128 bytes apart in the plain layout (8-way bank conflict), then hit 16 distinct bank quads.  The next tile's
// loads are issued before the MFMAs of the current one.
//
// Data gradient.  dX pixel (4q + p) takes taps ky = (p+2)%4 + 4j of dY rows q + (p+2)/4 - j, j = 0, 1 (common.hpp).
// Wave w owns row phase p = w; two x-phases share their dY columns, so MFMA rows are (x-phase of a pair, ci) = 16,
// K = (jy, jx, co) = 64, columns 16 neighbouring coarse positions q.  The 2 x 3 dY pixels a coarse position needs
// are read once (16-byte reads, pixel pitch 20 floats: conflict-free) and feed both pairs.  No cross-wave sums.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct EncArgs {
  const float* in; int ih, iw, ics, ico;
  float* out; int oh, ow, ocs, oco;
  const float* wp;
  const float* bias;
  double* stat;       // forward: per-workgroup rows [2 x 16] of {sum y, sum y^2} (nullptr: none)
  PW pw;
  int n, tiles_x, tiles_y, ntiles;
  int in_vec, out_vec;
};

// ------------------------------------------------------------------------------------------------ forward
constexpr int EF_R = 4;                        // output rows per tile (x 16 output pixels)
constexpr int EF_IR = 4 * EF_R + 4;            // input rows staged
constexpr int EF_IPX = 68;                     // input pixels per staged row
constexpr int EF_RP = (EF_IPX / 4) * 36;       // row pitch in floats (17 groups of 4 pixels, 32 floats + 4 pad)
constexpr int EF_NU = EF_IR * EF_IPX * 2;      // float4 units of a tile
constexpr int EF_UX = (EF_NU + 255) / 256;
constexpr size_t EF_LDS = (size_t)EF_IR * EF_RP * 4;
static_assert(EF_LDS >= 4 * 64 * 16 * 4, "the partial tiles reuse the input tile's LDS");

__global__ __launch_bounds__(256) void enc_fwd_kernel(EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;

  float wr[2][4][4];
#pragma unroll
  for (int kyi = 0; kyi < 2; ++kyi)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) wr[kyi][t][m] = a.wp[((((2 * wk + kyi) * 4 + t) * 4 + m) << 6) + lane];

  // staging units: (row, pixel) of unit tid + 256 j; channel quad c4 = tid & 1 for all of them
  const int c4 = tid & 1;
  int pos[EF_UX], xoff[EF_UX];
#pragma unroll
  for (int j = 0; j < EF_UX; ++j) {
    const int e = min(tid + j * 256, EF_NU - 1);
    const int p = e >> 1;
    const int r = p / EF_IPX, c = p - r * EF_IPX;
    pos[j] = (r << 8) | c;
    xoff[j] = (r * a.iw + c) * a.ics + 4 * c4;
  }
  const PW4 pw4 = pw4_load(a.pw, 4 * c4, 8);
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  float4 xv[EF_UX];
  unsigned okm = 0;

  auto issue = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int iy0 = 4 * EF_R * ty_ - 2, ix0 = 64 * tx_ - 2;
    if (a.in_vec && iy0 >= 0 && ix0 >= 0 && iy0 + EF_IR <= a.ih && ix0 + EF_IPX <= a.iw) {
      const float* t0 = a.in + (((int64_t)n * a.ih + iy0) * a.iw + ix0) * a.ics + a.ico;
#pragma unroll
      for (int j = 0; j < EF_UX; ++j) xv[j] = *reinterpret_cast<const float4*>(t0 + xoff[j]);
      okm = ~0u;
      return;
    }
    const float* base = a.in + (int64_t)n * a.ih * a.iw * a.ics + a.ico + 4 * c4;
    okm = 0;
#pragma unroll
    for (int j = 0; j < EF_UX; ++j) {
      const int iy = iy0 + (pos[j] >> 8), ix = ix0 + (pos[j] & 255);
      if (iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw) okm |= 1u << j;
      const float* p = base + ((int64_t)min(max(iy, 0), a.ih - 1) * a.iw + min(max(ix, 0), a.iw - 1)) * a.ics;
      if (a.in_vec) xv[j] = *reinterpret_cast<const float4*>(p);
      else xv[j] = make_float4(p[0], p[1], p[2], p[3]);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < EF_UX; ++j) {
      if (tid + j * 256 < EF_NU) {
        const int r = pos[j] >> 8, c = pos[j] & 255;
        float4 w = pw4_apply4(pw4, xv[j]);
        if (!((okm >> j) & 1)) w = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(smem + r * EF_RP + (c >> 2) * 36 + (c & 3) * 8 + 4 * c4) = w;
      }
    }
  };

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};     // this thread's pixels, its 4 channels
  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                           // the previous tile's partial sums are consumed
    commit();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) issue(next);

    v4f acc[EF_R];
#pragma unroll
    for (int nt = 0; nt < EF_R; ++nt) acc[nt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kyi = 0; kyi < 2; ++kyi) {
      const float* row0 = smem + (2 * wk + kyi) * EF_RP + li * 36 + 4 * kq;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 b[EF_R];
#pragma unroll
        for (int nt = 0; nt < EF_R; ++nt)
          b[nt] = *reinterpret_cast<const float4*>(row0 + 4 * nt * EF_RP + (t >> 1) * 36 + (t & 1) * 16);
#pragma unroll
        for (int nt = 0; nt < EF_R; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kyi][t][0], b[nt].x, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < EF_R; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kyi][t][1], b[nt].y, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < EF_R; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kyi][t][2], b[nt].z, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < EF_R; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kyi][t][3], b[nt].w, acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();                           // every wave is done reading the input tile
    // D[co = 4*(lane>>4) + r][pixel = lane & 15] -> part[wave][pixel 0..63][16 co]
#pragma unroll
    for (int nt = 0; nt < EF_R; ++nt)
      *reinterpret_cast<float4*>(smem + ((wk * 64 + nt * 16 + li) << 4) + 4 * kq) =
          make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    __syncthreads();
    {
      const int px = tid >> 2, cq = tid & 3;
      const float4 p0 = *reinterpret_cast<const float4*>(smem + ((0 * 64 + px) << 4) + 4 * cq);
      const float4 p1 = *reinterpret_cast<const float4*>(smem + ((1 * 64 + px) << 4) + 4 * cq);
      const float4 p2 = *reinterpret_cast<const float4*>(smem + ((2 * 64 + px) << 4) + 4 * cq);
      const float4 p3 = *reinterpret_cast<const float4*>(smem + ((3 * 64 + px) << 4) + 4 * cq);
      const float4 s0 = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                   (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
      float4 s = s0;
      if (a.bias) { s.x += a.bias[4 * cq]; s.y += a.bias[4 * cq + 1]; s.z += a.bias[4 * cq + 2]; s.w += a.bias[4 * cq + 3]; }
      const int n = tile / tiles_per_img;
      const int trem = tile - n * tiles_per_img;
      const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
      const int oy = EF_R * ty_ + (px >> 4), ox = 16 * tx_ + (px & 15);
      if (oy < a.oh && ox < a.ow) {
        float* o = a.out + (((int64_t)n * a.oh + oy) * a.ow + ox) * a.ocs + a.oco + 4 * cq;
        if (a.out_vec) *reinterpret_cast<float4*>(o) = s;
        else { o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w; }
        ssum[0] += s.x; ssum[1] += s.y; ssum[2] += s.z; ssum[3] += s.w;
        ssq[0] = fmaf(s.x, s.x, ssq[0]); ssq[1] = fmaf(s.y, s.y, ssq[1]);
        ssq[2] = fmaf(s.z, s.z, ssq[2]); ssq[3] = fmaf(s.w, s.w, ssq[3]);
      }
    }
  }
  // batch-norm statistics of what was stored: a thread's few pixels in fp32, everything beyond in double, fixed order
  if (a.stat) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (double)ssum[j]; v[4 + j] = (double)ssq[j]; }
#pragma unroll
    for (int sh = 4; sh < 64; sh <<= 1)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += __shfl_xor(v[j], sh, 64);
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);          // [wave][which 0..1][16 channels]
    if (lane < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(wk * 2 + 0) * 16 + 4 * lane + j] = v[j]; red[(wk * 2 + 1) * 16 + 4 * lane + j] = v[4 + j]; }
    }
    __syncthreads();
    if (tid < 32)
      a.stat[(int64_t)blockIdx.x * 32 + tid] = (red[0 * 32 + tid] + red[1 * 32 + tid]) + (red[2 * 32 + tid] + red[3 * 32 + tid]);
  }
}

// wp[ky][t][m][kq][co]: the weight of k = 16 t + 4 kq + m (kx = k / 8, ci = k % 8) of tap row ky
__global__ void enc_fwd_pack_kernel(const float* w, int64_t sa, int64_t sb, float* wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 8192) return;
  const int co = idx & 15, kq = (idx >> 4) & 3, m = (idx >> 6) & 3, t = (idx >> 8) & 3, ky = idx >> 10;
  const int k = 16 * t + 4 * kq + m;
  const int kx = k >> 3, ci = k & 7;
  wp[idx] = w[ci * sa + co * sb + ky * 8 + kx];
}

// ------------------------------------------------------------------------------------------------ data gradient
constexpr int ED_RQ = 4;                       // coarse rows per tile (x 16 coarse columns): 16 x 64 produced pixels
constexpr int ED_YR = ED_RQ + 2, ED_YC = 18, ED_PP = 20;
constexpr int ED_NU = ED_YR * ED_YC * 4;
constexpr int ED_UX = (ED_NU + 255) / 256;

__global__ __launch_bounds__(256) void enc_dgrad_kernel(EncArgs a) {
  __shared__ __attribute__((aligned(16))) float ys[ED_YR * ED_YC * ED_PP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int py = __builtin_amdgcn_readfirstlane(tid >> 6);       // this wave's row phase
  const int li = lane & 15, kq = lane >> 4;
  const int cy = (py + 2) >> 2;

  float wa[2][4][4];
#pragma unroll
  for (int P = 0; P < 2; ++P)
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int m = 0; m < 4; ++m) wa[P][blk][m] = a.wp[(((((py * 2 + P) * 4 + blk) * 4 + m)) << 6) + lane];

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  float4 yv[ED_UX];
  unsigned okm = 0;
  const PW4 pw4 = pw4_load(a.pw, 4 * (tid & 3), 16);       // (unit tid + 256 j: channel quad tid & 3 for every j)
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = a.bias[4 * (kq & 1) + r];
  }
  auto issue = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const float* base = a.in + (int64_t)n * a.ih * a.iw * a.ics + a.ico;
#pragma unroll
    for (int j = 0; j < ED_UX; ++j) {
      const int e = min(tid + j * 256, ED_NU - 1);
      const int c4 = e & 3, p = e >> 2;
      const int r = p / ED_YC, c = p - r * ED_YC;
      const int gy = ED_RQ * ty_ - 1 + r, gx = 16 * tx_ - 1 + c;
      const bool ok = gy >= 0 && gy < a.ih && gx >= 0 && gx < a.iw;
      const float* q = base + ((int64_t)min(max(gy, 0), a.ih - 1) * a.iw + min(max(gx, 0), a.iw - 1)) * a.ics + 4 * c4;
      float4 v;
      if (a.in_vec) v = *reinterpret_cast<const float4*>(q);
      else v = make_float4(q[0], q[1], q[2], q[3]);
      yv[j] = v;
      okm = ok ? (okm | (1u << j)) : (okm & ~(1u << j));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < ED_UX; ++j) {
      const int e = tid + j * 256;
      if (e < ED_NU)
        *reinterpret_cast<float4*>(ys + (e >> 2) * ED_PP + 4 * (e & 3)) =
            ((okm >> j) & 1) ? pw4_apply4(pw4, yv[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) issue(next);
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
#pragma unroll
    for (int ql = 0; ql < ED_RQ; ql += 2) {
      // staged rows ql + cy + {0, 1, 2}: coarse row ql reads rows {1 (jy = 0), 0 (jy = 1)}, row ql + 1 reads {2, 1}
      float4 b[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          b[r][d] = *reinterpret_cast<const float4*>(ys + ((ql + cy + r) * ED_YC + li + d) * ED_PP + 4 * kq);
      v4f acc[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int P = 0; P < 2; ++P) acc[u][P] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int jy = blk >> 1, jx = blk & 1;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int P = 0; P < 2; ++P) {
              const float4 bv = b[u + 1 - jy][1 + P - jx];
              const float bm = m == 0 ? bv.x : (m == 1 ? bv.y : (m == 2 ? bv.z : bv.w));
              acc[u][P] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[P][blk][m], bm, acc[u][P], 0, 0, 0);
            }
      }
      // D[(pxl, ci) = 4*(lane>>4) + r][coarse column = lane & 15]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int oy = 4 * (ED_RQ * ty_ + ql + u) + py;
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          const int ox = 4 * (16 * tx_ + li) + 2 * P + (kq >> 1);
          if (oy < a.oh && ox < a.ow) {
            float* o = a.out + (((int64_t)n * a.oh + oy) * a.ow + ox) * a.ocs + a.oco + 4 * (kq & 1);
            const float4 v = make_float4(acc[u][P][0] + bias4[0], acc[u][P][1] + bias4[1], acc[u][P][2] + bias4[2],
                                         acc[u][P][3] + bias4[3]);
            if (a.out_vec) *reinterpret_cast<float4*>(o) = v;
            else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
          }
        }
      }
    }
  }
}

// wp[py][P][blk = (jy, jx)][m][kq][(pxl, ci)]: W[gathered co = 4 kq + m][produced ci][ky][kx],
// ky = (py + 2) % 4 + 4 jy, kx = (2 P + pxl + 2) % 4 + 4 jx
__global__ void enc_dgrad_pack_kernel(const float* w, int64_t sa, int64_t sb, float* wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 8192) return;
  const int li = idx & 15, kq = (idx >> 4) & 3, m = (idx >> 6) & 3, blk = (idx >> 8) & 3, P = (idx >> 10) & 1, py = idx >> 11;
  const int pxl = li >> 3, ci = li & 7, co = 4 * kq + m;
  const int ky = ((py + 2) & 3) + 4 * (blk >> 1), kx = ((2 * P + pxl + 2) & 3) + 4 * (blk & 1);
  wp[idx] = w[co * sa + ci * sb + ky * 8 + kx];
}

// ------------------------------------------------------------------------------------------------ k4 s2 {1,2} -> 8
// The first layer of the recognition / prior networks (512^2 -> 256^2, utils.py:96 conv_down(c, 8, scale=2)), forward.
// The per-pixel vector-ALU kernel (conv_small.hip) is bound by its 128 / 256 weights, which do not fit the scalar
// registers and arrive in sixteen waited-for batches: 0.15 ms for the 2-channel layer against 0.05 of HBM time.  As a
// GEMM it is K = 16 taps x CI = 4 or 8 k-steps of the 16x16x4 MFMA with the weights in 4 or 8 registers: the (tx, ci)
// floats of a tap row are consecutive in the NHWC input row starting at pixel 2x - 1, so lane (pixel, kq) reads float
// 4j + kq of its run -- 64 consecutive-ish addresses per wave, no conflicts.  Rows 8..15 of the tile are padding (8
// produced channels): the matrix work is 27 us either way.  A wave owns two output rows of the 8 x 32 tile; outputs
// go straight from the accumulators to memory; batch-norm sums per lane over its tiles, one row per workgroup.
constexpr int E0_R = 8, E0_W = 32;                   // output tile
constexpr int E0_IR = 2 * E0_R + 2, E0_IC = 2 * E0_W + 2;

template <int CI>
__global__ __launch_bounds__(256) void enc0_fwd_kernel(EncArgs a) {
  constexpr int KS = 4 * CI;                         // k-steps: 4 tap rows x CI
  constexpr int RP = E0_IC * CI;                     // LDS row pitch in floats
  constexpr int NU = E0_IR * E0_IC, UX = (NU + 255) / 256;      // staging units: one input pixel each
  __shared__ __attribute__((aligned(16))) float xs[E0_IR * RP + 4];
  __shared__ double red[4][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;

  float wr[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wr[j] = a.wp[j * 64 + lane];
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias && kq < 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = a.bias[4 * kq + r];
  }
  const bool on = a.pw.scale != nullptr;
  float psc[CI], psf[CI], psl[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) { psc[c] = on ? a.pw.scale[c] : 1.f; psf[c] = on ? a.pw.shift[c] : 0.f; psl[c] = on ? a.pw.slope[c] : 1.f; }

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  float xv[UX][CI];
  unsigned okm = 0;
  auto issue = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int iy0 = 2 * E0_R * ty_ - 1, ix0 = 2 * E0_W * tx_ - 1;
    const float* base = a.in + (int64_t)n * a.ih * a.iw * a.ics + a.ico;
    okm = 0;
#pragma unroll
    for (int j = 0; j < UX; ++j) {
      const int e = min(tid + j * 256, NU - 1);
      const int r = e / E0_IC, c = e - r * E0_IC;
      const int iy = iy0 + r, ix = ix0 + c;
      if (iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw) okm |= 1u << j;
      const float* q = base + ((int64_t)min(max(iy, 0), a.ih - 1) * a.iw + min(max(ix, 0), a.iw - 1)) * a.ics;
      if (CI == 2 && a.in_vec) {                    // (in_vec here: 8-byte aligned channel pairs)
        const float2 t = *reinterpret_cast<const float2*>(q);
        xv[j][0] = t.x; xv[j][CI - 1] = t.y;
      } else {
#pragma unroll
        for (int c2 = 0; c2 < CI; ++c2) xv[j][c2] = q[c2];
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < UX; ++j) {
      const int e = tid + j * 256;
      if (e < NU) {
#pragma unroll
        for (int c2 = 0; c2 < CI; ++c2) {
          float v = xv[j][c2];
          if (on) { v = fmaf(v, psc[c2], psf[c2]); v = v > 0.f ? v : v * psl[c2]; }
          xs[e * CI + c2] = ((okm >> j) & 1) ? v : 0.f;
        }
      }
    }
  };

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) issue(next);
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    v4f acc[2][2];                               // [output row of this wave][16-pixel half]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hx = 0; hx < 2; ++hx) acc[u][hx] = v4f{bias4[0], bias4[1], bias4[2], bias4[3]};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int ty = j / CI, jj = j % CI;        // tap row, 4-float step inside its run of 4 CI floats
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          const float bf = xs[(2 * (2 * wk + u) + ty) * RP + (2 * (16 * hx + li)) * CI + 4 * jj + kq];
          acc[u][hx] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j], bf, acc[u][hx], 0, 0, 0);
        }
    }
    if (kq < 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          const int oy = E0_R * ty_ + 2 * wk + u, ox = E0_W * tx_ + 16 * hx + li;
          if (oy < a.oh && ox < a.ow) {
            float* o = a.out + (((int64_t)n * a.oh + oy) * a.ow + ox) * a.ocs + a.oco + 4 * kq;
            const v4f v = acc[u][hx];
            if (a.out_vec) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[r] += v[r]; ssq[r] = fmaf(v[r], v[r], ssq[r]); }
          }
        }
    }
  }
  if (a.stat) {
    double v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = (double)ssum[r]; v[4 + r] = (double)ssq[r]; }
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += __shfl_xor(v[r], sh, 64);
    if (li == 0 && kq < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[wk][4 * kq + r] = v[r]; red[wk][8 + 4 * kq + r] = v[4 + r]; }
    }
    __syncthreads();
    if (tid < 16) a.stat[(int64_t)blockIdx.x * 16 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// wp[j][kq][co]: k-step j = (ty, jj): float f = 4 jj + kq of tap row ty's run -> (tx, ci) = (f / CI, f % CI); rows 8..15: 0
__global__ void enc0_fwd_pack_kernel(const float* w, int64_t sa, int64_t sb, int ci_n, float* wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * ci_n * 64) return;
  const int co = idx & 15, kq = (idx >> 4) & 3, j = idx >> 6;
  const int ty = j / ci_n, jj = j % ci_n;
  const int f = 4 * jj + kq;
  const int tx = f / ci_n, ci = f % ci_n;
  wp[idx] = co < 8 ? w[ci * sa + co * sb + ty * 4 + tx] : 0.f;
}

// ------------------------------------------------------------------------------------------------ weight gradient
//   dW[co][ci][ky][kx] = sum_{n,oy,ox} act(X)[n, 4 oy - 2 + ky, 4 ox - 2 + kx, ci] * dY[n, oy, ox, co]
// GEMM K = output pixels.  MFMA rows = the 16 channels of dY, columns = 16 consecutive floats (kx, ci) of X's tap row
// (four column tiles per ky, 32 in all).  Wave w owns ky = 2w, 2w+1 -- eight accumulator tiles, outputs no other
// wave touches -- and walks ALL 32 pixels of the 2 x 16 tile; X is staged in groups of 4 pixels (one group per
// output pixel) at a pitch of 48 floats: the four pixels of a k-step then read bank quarters 0, 3, 2, 1.
// (conv_wgrad_small.hip held all 32 tiles in every wave: 128 accumulator registers, one wave per SIMD, 0.10 ms.)
constexpr int EW_R = 2;
constexpr int EW_XR = 4 * EW_R + 4, EW_XP = 68, EW_RP = 17 * 48;
constexpr int EW_NUX = EW_XR * EW_XP * 2, EW_UX = (EW_NUX + 255) / 256;
constexpr int EW_XF = EW_XR * EW_RP, EW_YF = EW_R * 16 * 16;
constexpr size_t EW_LDS = (size_t)(EW_XF + EW_YF) * 4;

struct EncWArgs {
  const float* X; int xh, xw, xcs, xco;
  const float* Y; int yh, yw, ycs, yco;
  PW pwx, pwy;
  float* ws;
  int n, tiles_x, tiles_y, ntiles, xvec, yvec;
};

__global__ __launch_bounds__(256) void enc_wgrad_kernel(EncWArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + EW_XF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int c4 = tid & 1;
  const PW4 px4 = pw4_load(a.pwx, 4 * c4, 8);
  const PW4 py4 = pw4_load(a.pwy, 4 * (tid & 3), 16);
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  float4 xv[EW_UX], yv = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned okx = 0;
  bool oky = false;

  auto issue = [&](int tile) {
    const int n = tile / tiles_per_img;
    const int trem = tile - n * tiles_per_img;
    const int ty_ = trem / a.tiles_x, tx_ = trem - ty_ * a.tiles_x;
    const int iy0 = 4 * EW_R * ty_ - 2, ix0 = 64 * tx_ - 2;
    const float* base = a.X + (int64_t)n * a.xh * a.xw * a.xcs + a.xco + 4 * c4;
    okx = 0;
#pragma unroll
    for (int j = 0; j < EW_UX; ++j) {
      const int e = min(tid + j * 256, EW_NUX - 1);
      const int p = e >> 1;
      const int r = p / EW_XP, c = p - r * EW_XP;
      const int iy = iy0 + r, ix = ix0 + c;
      if (iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw) okx |= 1u << j;
      const float* q = base + ((int64_t)min(max(iy, 0), a.xh - 1) * a.xw + min(max(ix, 0), a.xw - 1)) * a.xcs;
      if (a.xvec) xv[j] = *reinterpret_cast<const float4*>(q);
      else xv[j] = make_float4(q[0], q[1], q[2], q[3]);
    }
    if (tid < EW_R * 16 * 4) {
      const int p = tid >> 2;
      const int oy = EW_R * ty_ + (p >> 4), ox = 16 * tx_ + (p & 15);
      oky = oy < a.yh && ox < a.yw;
      const float* q = a.Y + (((int64_t)n * a.yh + min(oy, a.yh - 1)) * a.yw + min(ox, a.yw - 1)) * a.ycs + a.yco + 4 * (tid & 3);
      if (a.yvec) yv = *reinterpret_cast<const float4*>(q);
      else yv = make_float4(q[0], q[1], q[2], q[3]);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < EW_UX; ++j) {
      const int e = tid + j * 256;
      if (e < EW_NUX) {
        const int p = e >> 1;
        const int r = p / EW_XP, c = p - r * EW_XP;
        float4 w = pw4_apply4(px4, xv[j]);
        if (!((okx >> j) & 1)) w = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xs + r * EW_RP + (c >> 2) * 48 + (c & 3) * 8 + 4 * c4) = w;
      }
    }
    if (tid < EW_R * 16 * 4)
      *reinterpret_cast<float4*>(ys + 4 * tid) = oky ? pw4_apply4(py4, yv) : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  v4f acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = v4f{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) issue(next);
#pragma unroll
    for (int s = 0; s < EW_R * 4; ++s) {
      const float af = ys[(4 * s + kq) * 16 + li];
      const float* xp = xs + (4 * (s >> 2) + 2 * wk) * EW_RP + (4 * (s & 3) + kq) * 48 + li;
#pragma unroll
      for (int kyi = 0; kyi < 2; ++kyi)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float bf = xp[kyi * EW_RP + (t >> 1) * 48 + (t & 1) * 16];
          acc[kyi][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[kyi][t], 0, 0, 0);
        }
    }
  }
  // D[co = 4*(lane>>4) + r][n = lane & 15], n-tile (ky, t): tap-row float 16 t + n = (kx, ci) -> ws[split][ky][kx][co][ci]
#pragma unroll
  for (int kyi = 0; kyi < 2; ++kyi)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int nf = 16 * t + li;
      const int ky = 2 * wk + kyi, kx = nf >> 3, ci = nf & 7;
      float* o = a.ws + (((int64_t)blockIdx.x * 64 + ky * 8 + kx) * 16 + 4 * kq) * 8 + ci;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[8 * r] = acc[kyi][t][r];
    }
}

bool enc_off() {
  static const bool off = getenv("BP_NOENC") != nullptr;
  return off;
}

}  // namespace

bool bp_enc_fwd_ok(const ConvGeom& g) {
  return !enc_off() && !g.gather_transposed && g.k == 8 && g.stride == 4 && g.pad == 2 && g.cin_g == 8 && g.cout_g == 16;
}
bool bp_enc_dgrad_ok(const ConvGeom& g) {
  return !enc_off() && g.gather_transposed && g.k == 8 && g.stride == 4 && g.pad == 2 && g.cin_g == 16 && g.cout_g == 8;
}
bool bp_enc0_fwd_ok(const ConvGeom& g) {
  static const bool off0 = getenv("BP_NOENC0") != nullptr;
  return !enc_off() && !off0 && !g.gather_transposed && g.k == 4 && g.stride == 2 && g.pad == 1 &&
         (g.cin_g == 1 || g.cin_g == 2) && g.cout_g == 8;
}
bool bp_enc_ok(const ConvGeom& g) { return bp_enc_fwd_ok(g) || bp_enc_dgrad_ok(g) || bp_enc0_fwd_ok(g); }
int bp_enc_kernel_id(const ConvGeom& g) { return bp_enc0_fwd_ok(g) ? 780000 + g.cin_g : (g.gather_transposed ? 770000 : 760000); }
int64_t bp_enc_packed_floats(const ConvGeom& g) { return bp_enc0_fwd_ok(g) ? 4 * g.cin_g * 64 : 8192; }

int bp_enc_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  if (bp_enc0_fwd_ok(g)) hipLaunchKernelGGL(enc0_fwd_pack_kernel, dim3(2), dim3(256), 0, st, w_torch, wm.sa, wm.sb, g.cin_g, packed);
  else if (bp_enc_fwd_ok(g)) hipLaunchKernelGGL(enc_fwd_pack_kernel, dim3(32), dim3(256), 0, st, w_torch, wm.sa, wm.sb, packed);
  else hipLaunchKernelGGL(enc_dgrad_pack_kernel, dim3(32), dim3(256), 0, st, w_torch, wm.sa, wm.sb, packed);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

size_t bp_stats_rows_bytes(int64_t rows, int C);
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st);

static int enc_fwd_grid(const bp_view* in, const bp_view* out, int* tiles_x, int* tiles_y, int* ntiles) {
  static const int per_cu = getenv("BP_ENC_WGS") ? atoi(getenv("BP_ENC_WGS")) : 2;   // (188 VGPRs: two waves per SIMD)
  *tiles_x = bp_ceil_div(out->w, 16);
  *tiles_y = bp_ceil_div(out->h, EF_R);
  const int64_t nt = (int64_t)in->n * *tiles_x * *tiles_y;
  if (nt > 0x7fffffff) return -1;
  *ntiles = (int)nt;
  return *ntiles < 256 * per_cu ? *ntiles : 256 * per_cu;
}

static int enc0_fwd_grid(const bp_view* in, const bp_view* out, int* tiles_x, int* tiles_y, int* ntiles) {
  static const int cap = getenv("BP_ENC0_WGS") ? atoi(getenv("BP_ENC0_WGS")) : 2048;
  *tiles_x = bp_ceil_div(out->w, E0_W);
  *tiles_y = bp_ceil_div(out->h, E0_R);
  const int64_t nt = (int64_t)in->n * *tiles_x * *tiles_y;
  if (nt > 0x7fffffff) return -1;
  *ntiles = (int)nt;
  return *ntiles < cap ? *ntiles : cap;
}

// forward only: {sum y, sum y^2} per produced channel (mode 1), one row per workgroup
size_t bp_enc_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode) {
  if (bp_enc0_fwd_ok(g) && mode == 1) {
    int tx, ty, nt;
    const int grid = enc0_fwd_grid(in, out, &tx, &ty, &nt);
    return grid > 0 ? bp_stats_rows_bytes(grid, 8) : 0;
  }
  if (!bp_enc_fwd_ok(g) || mode != 1) return 0;
  int tx, ty, nt;
  const int grid = enc_fwd_grid(in, out, &tx, &ty, &nt);
  return grid > 0 ? bp_stats_rows_bytes(grid, 16) : 0;
}

int bp_enc_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
               const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  EncArgs a{};
  a.bias = bias;
  a.in = in->ptr; a.ih = in->h; a.iw = in->w; a.ics = in->cstride; a.ico = in->coff;
  a.out = out->ptr; a.oh = out->h; a.ow = out->w; a.ocs = out->cstride; a.oco = out->coff;
  a.wp = packed; a.pw = pw; a.n = in->n;
  a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.out_vec = bp_view_vec4(out) ? 1 : 0;
  if (bp_enc0_fwd_ok(g)) {
    const int grid = enc0_fwd_grid(in, out, &a.tiles_x, &a.tiles_y, &a.ntiles);
    if (grid <= 0) return BP_EUNSUPPORTED;
    if (sr) {
      if (bias || sr->mode != 1) return BP_EUNSUPPORTED;
      if (!sr->ws || sr->ws_bytes < bp_stats_rows_bytes(grid, 8) || !sr->sums) return BP_EWORKSPACE;
      a.stat = reinterpret_cast<double*>(sr->ws);
    }
    // (in_vec for this kernel: channel pairs on 8-byte boundaries)
    a.in_vec = (g.cin_g == 2 && in->cstride % 2 == 0 && in->coff % 2 == 0 && reinterpret_cast<uintptr_t>(in->ptr) % 8 == 0) ? 1 : 0;
    if (g.cin_g == 1) hipLaunchKernelGGL(enc0_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(enc0_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    BP_CHECK_LAUNCH();
    return sr ? bp_stats_rows_finish(a.stat, grid, 8, sr, st) : BP_OK;
  }
  if (bp_enc_fwd_ok(g)) {
    const int grid = enc_fwd_grid(in, out, &a.tiles_x, &a.tiles_y, &a.ntiles);
    if (grid <= 0) return BP_EUNSUPPORTED;
    if (sr) {
      if (bias || sr->mode != 1) return BP_EUNSUPPORTED;
      if (!sr->ws || sr->ws_bytes < bp_stats_rows_bytes(grid, 16) || !sr->sums) return BP_EWORKSPACE;
      a.stat = reinterpret_cast<double*>(sr->ws);
    }
    hipLaunchKernelGGL(enc_fwd_kernel, dim3(grid), dim3(256), EF_LDS, st, a);
    BP_CHECK_LAUNCH();
    return sr ? bp_stats_rows_finish(a.stat, grid, 16, sr, st) : BP_OK;
  } else {
    if (sr) return BP_EUNSUPPORTED;
    a.tiles_x = bp_ceil_div(bp_ceil_div(out->w, 4), 16);
    a.tiles_y = bp_ceil_div(bp_ceil_div(out->h, 4), ED_RQ);
    const int64_t nt = (int64_t)in->n * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffff) return BP_EUNSUPPORTED;
    a.ntiles = (int)nt;
    const int grid = a.ntiles < 256 * 4 ? a.ntiles : 256 * 4;
    hipLaunchKernelGGL(enc_dgrad_kernel, dim3(grid), dim3(256), 0, st, a);
  }
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// Weight gradient of the same layer (either orientation: X is the 8-channel tensor at 4x the resolution of the
// 16-channel Y).  Same contract as bp_wgrad_small (conv_wgrad.hip reduces the partials).
int bp_wgrad_enc(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                 size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  if (enc_off() || cv->k != 8 || cv->stride != 4 || cv->pad != 2 || X->c != 8 || Y->c != 16) return BP_EUNSUPPORTED;
  EncWArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff;
  a.pwx = pwx; a.pwy = pwy; a.n = X->n;
  a.tiles_x = bp_ceil_div(Y->w, 16);
  a.tiles_y = bp_ceil_div(Y->h, EW_R);
  const int64_t nt = (int64_t)X->n * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffff) return BP_EUNSUPPORTED;
  a.ntiles = (int)nt;
  static const int cap = getenv("BP_ENC_WSPLIT") ? atoi(getenv("BP_ENC_WSPLIT")) : 768;      // (41 KB of LDS: three workgroups per CU)
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  *need = (size_t)grid * 8192 * sizeof(float);
  *nsplit = grid; *cxp = 8; *cyp = 16;
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.ws = ws;
  a.xvec = bp_view_vec4(X) ? 1 : 0;
  a.yvec = bp_view_vec4(Y) ? 1 : 0;
  hipLaunchKernelGGL(enc_wgrad_kernel, dim3(grid), dim3(256), EW_LDS, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
