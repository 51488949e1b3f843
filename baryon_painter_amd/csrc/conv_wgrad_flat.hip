// Weight gradient of the generator head's first layer, Conv2d 16 -> 8, k7, stride 1 (arch p_y_z_out, cvae.py:26-45):
//     dW[co][ci][ty][tx] = sum_{n,y,x} dY[n,y,x,co] * act(X)[n, y+ty-3, x+tx-3, ci]
// 49 taps x 128 channel pairs over full-resolution tensors.  As a GEMM per tap (16 ci x 8 co) half of every fp32
// MFMA tile would be padding, and the general kernels (conv_wgrad_small.hip) re-stage X once per tap row.  Here
//   * M = the 16 input channels, K = 4 consecutive X columns x', N = (tap column pair j, co): tx = 2*nt + j, so the
//     8 output channels of TWO tap columns fill the 16 MFMA columns (7 tap columns -> 4 N tiles, 12.5 % padding);
//     shifting X by tx is the same as shifting dY by -tx:  B[x'][(j,co)] = dY[y][x' - tx + 3][co];
//   * for one dY row and one K step the wave reads 7 A fragments (the X rows y+ty-3) and 4 B fragments (one dY row,
//     shifted) and issues 7 x 4 = 28 MFMAs into 28 accumulators it keeps for the whole kernel: 11 LDS reads of 4
//     bytes per lane per 28 MFMAs, all of them 256 contiguous bytes per wave (no bank conflicts);
//   * workgroups walk the (image, tile) sequence with a grid stride, the next tile (X with its 6 halo rows, activated
//     and zero padded; dY with its 6 halo columns) in flight in registers; waves, then workgroups, are folded in a
//     fixed order (deterministic).
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int K = 7, PAD = 3, CI = 16, CO = 8;
constexpr int TH = 8, TW = 32;                        // dY tile of a workgroup: 4 waves x 2 rows x 32 columns
constexpr int XH = TH + K - 1;                        // X rows staged (14)
constexpr int YW = TW + K - 1;                        // dY columns staged (38)
constexpr int XQ = XH * TW * (CI / 4);                // float4 units of the X tile (1792)
constexpr int YQ = TH * YW * (CO / 4);                // float4 units of the dY tile (608)
constexpr int XS = XQ / 256;                          // 7 per thread
constexpr int YS = (YQ + 255) / 256;                  // 3 per thread
constexpr int NTX = (K + 1) / 2;                      // N tiles: tap-column pairs (4)
constexpr int PART = K * NTX * 16 * 16;               // accumulator elements of a wave / workgroup (7168)
static_assert(XQ % 256 == 0, "X tile units divide among the threads");

struct FlatArgs {
  const float* x; int h, w, x_cs, x_co;
  const float* dy; int dy_cs, dy_co;
  PW pw;
  int n, tiles_x, tiles_y;
  float* partial;             // [workgroup][ty][nt][ci][n]
};

__global__ __launch_bounds__(256, 2) void wgrad_flat_kernel(FlatArgs a) {
  __shared__ __attribute__((aligned(16))) float xt[XH * TW * CI];
  __shared__ __attribute__((aligned(16))) float yt[TH * YW * CO + 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;

  // staging: thread owns channel quad c4 of XS X pixels and YS dY units
  const int c4 = tid & 3;
  const PW4 p4 = pw4_load(a.pw, c4 * 4, CI);
  float4 sx[XS], sy[YS];
  unsigned xin = 0, yin = 0;
  auto fetch = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH, x0 = (r % a.tiles_x) * TW;
    const float* xn = a.x + (int64_t)n * a.h * a.w * a.x_cs + a.x_co + c4 * 4;
    const float* yn = a.dy + (int64_t)n * a.h * a.w * a.dy_cs + a.dy_co;
    xin = 0; yin = 0;
#pragma unroll
    for (int i = 0; i < XS; ++i) {                 // unit e = (row, col, quad): X[y0 - 3 + row][x0 + col]
      const int e = tid + i * 256;
      const int pix = e >> 2, col = pix % TW, row = pix / TW;
      const int iy = y0 - PAD + row, ix = x0 + col;
      if (iy >= 0 && iy < a.h && ix < a.w) xin |= 1u << i;
      const int cy = min(max(iy, 0), a.h - 1), cx = min(ix, a.w - 1);
      sx[i] = *reinterpret_cast<const float4*>(xn + ((int64_t)cy * a.w + cx) * a.x_cs);
    }
#pragma unroll
    for (int i = 0; i < YS; ++i) {                 // unit e = (row, col, half): dY[y0 + row][x0 - 3 + col]
      const int e = tid + i * 256;
      const int hq = e & 1, pix = e >> 1, col = pix % YW, row = pix / YW;
      const int iy = y0 + row, ix = x0 - PAD + col;
      if (e < YQ && iy < a.h && ix >= 0 && ix < a.w) yin |= 1u << i;
      const int cy = min(iy, a.h - 1), cx = min(max(ix, 0), a.w - 1);
      sy[i] = *reinterpret_cast<const float4*>(yn + ((int64_t)cy * a.w + cx) * a.dy_cs + hq * 4);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const float4 v = ((xin >> i) & 1u) ? pw4_apply4(p4, sx[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(xt + (tid + i * 256) * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < YS; ++i) {
      const int e = tid + i * 256;
      if (e < YQ) *reinterpret_cast<float4*>(yt + e * 4) = ((yin >> i) & 1u) ? sy[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  v4f acc[K][NTX];
#pragma unroll
  for (int ty = 0; ty < K; ++ty)
#pragma unroll
    for (int nt = 0; nt < NTX; ++nt) acc[ty][nt] = v4f{0.f, 0.f, 0.f, 0.f};
  if (tid < 16) yt[TH * YW * CO + tid] = 0.f;
  const int boff = lm < 8 ? lm : lm - 16;          // (co - 8*j) of column n = lm = 8*j + co

  int t = blockIdx.x;
  if (t < ntiles) { fetch(t); commit(); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int y = wave * 2 + rr;                 // dY row of the tile; X rows y .. y+6 of the staged tile
      const float* xa = xt + (y * TW + kq) * CI + lm;
      const float* yb = yt + (y * YW + kq + 2 * PAD) * CO + boff;
#pragma unroll 2
      for (int s = 0; s < TW / 4; ++s) {
        float af[K], bf[NTX];
#pragma unroll
        for (int ty = 0; ty < K; ++ty) af[ty] = xa[(ty * TW + 4 * s) * CI];
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt) bf[nt] = yb[(4 * s - 2 * nt) * CO];
#pragma unroll
        for (int ty = 0; ty < K; ++ty)
#pragma unroll
          for (int nt = 0; nt < NTX; ++nt)
            acc[ty][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ty], bf[nt], acc[ty][nt], 0, 0, 0);
      }
    }
    __syncthreads();
    if (tn < ntiles) commit();
    __syncthreads();
  }

  // fold the four waves through LDS (waves 1..3 in turn), then one row per workgroup
  float* red = xt;                                 // PART floats (28 KiB) fit the X tile
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ty = 0; ty < K; ++ty)
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) red[((ty * NTX + nt) * 16 + 4 * kq + q) * 16 + lm] = acc[ty][nt][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int ty = 0; ty < K; ++ty)
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[ty][nt][q] += red[((ty * NTX + nt) * 16 + 4 * kq + q) * 16 + lm];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* dst = a.partial + (int64_t)blockIdx.x * PART;
#pragma unroll
    for (int ty = 0; ty < K; ++ty)
#pragma unroll
      for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[((ty * NTX + nt) * 16 + 4 * kq + q) * 16 + lm] = acc[ty][nt][q];
  }
}

// partial[rows][PART] -> dW in torch layout [co][ci][ty][tx]: thread per element, rows in order, in double
__global__ __launch_bounds__(256) void wgrad_flat_fold_kernel(const float* partial, int rows, int rows_per_block,
                                                              double* part2, float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;           // element ((ty*NTX + nt)*16 + ci)*16 + n
  if (i >= PART) return;
  if (part2) {
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    double t = 0.0;
    for (int r = r0; r < r1; ++r) t += (double)partial[(int64_t)r * PART + i];
    part2[(int64_t)blockIdx.y * PART + i] = t;
  } else {
    const double* p2 = reinterpret_cast<const double*>(partial);
    double t = 0.0;
    for (int r = 0; r < rows; ++r) t += p2[(int64_t)r * PART + i];
    const int n = i % 16, ci = (i / 16) % 16, nt = (i / 256) % NTX, ty = i / (256 * NTX);
    const int tx = 2 * nt + n / CO, co = n % CO;
    if (tx < K) dst[((co * CI + ci) * K + ty) * K + tx] = (float)t;
  }
}

int flat_grid(int ntiles) {
  static const int cap = getenv("BP_WFLAT_GRID") ? atoi(getenv("BP_WFLAT_GRID")) : 512;
  return ntiles < cap ? ntiles : cap;
}
constexpr int FOLD_ROWS = 32;

int flat_tiles(const bp_view* X) { return bp_ceil_div(X->w, TW) * bp_ceil_div(X->h, TH) * X->n; }

}  // namespace

bool bp_wgrad_flat_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy) {
  static const bool off = getenv("BP_NOWFLAT") != nullptr;
  return !off && !cv->transposed && cv->k == K && cv->stride == 1 && cv->pad == PAD && cv->cin == CI && cv->cout == CO &&
         X->c == CI && Y->c == CO && X->h == Y->h && X->w == Y->w && pwy.scale == nullptr && X->dtype == BP_F32 &&
         Y->dtype == BP_F32 && bp_view_vec4(X) && bp_view_vec4(Y);
}

size_t bp_wgrad_flat_workspace(const bp_view* X) {
  const int grid = flat_grid(flat_tiles(X));
  const int chunks = bp_ceil_div(grid, FOLD_ROWS);
  return (size_t)grid * PART * sizeof(float) + (size_t)chunks * PART * sizeof(double);
}

int bp_wgrad_flat(const bp_view* X, const PW& pwx, const bp_view* Y, float* dst, void* workspace, size_t workspace_bytes,
                  hipStream_t st) {
  if (!workspace || workspace_bytes < bp_wgrad_flat_workspace(X)) return BP_EWORKSPACE;
  FlatArgs a{};
  a.x = X->ptr; a.h = X->h; a.w = X->w; a.x_cs = X->cstride; a.x_co = X->coff;
  a.dy = Y->ptr; a.dy_cs = Y->cstride; a.dy_co = Y->coff; a.pw = pwx; a.n = X->n;
  a.tiles_x = bp_ceil_div(X->w, TW); a.tiles_y = bp_ceil_div(X->h, TH);
  const int grid = flat_grid(flat_tiles(X));
  const int chunks = bp_ceil_div(grid, FOLD_ROWS);
  a.partial = reinterpret_cast<float*>(workspace);
  double* part2 = reinterpret_cast<double*>(a.partial + (size_t)grid * PART);
  hipLaunchKernelGGL(wgrad_flat_kernel, dim3(grid), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(wgrad_flat_fold_kernel, dim3(PART / 256, chunks), dim3(256), 0, st, a.partial, grid, FOLD_ROWS, part2,
                     (float*)nullptr);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(wgrad_flat_fold_kernel, dim3(PART / 256, 1), dim3(256), 0, st, reinterpret_cast<const float*>(part2),
                     chunks, 0, (double*)nullptr, dst);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
