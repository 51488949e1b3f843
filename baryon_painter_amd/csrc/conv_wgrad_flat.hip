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

// one workgroup per CU when the launch shares the GPU with another stream (BP_IMPL_SHARED): in the training step
// that leaves room for the data-gradient chain's workgroups (45.6 -> 45.2 ms per step); two per CU alone
int shared_grid(int grid, bool shared) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
  }();
  static const int cap = getenv("BP_WFLAT_SHARED_GRID") ? atoi(getenv("BP_WFLAT_SHARED_GRID")) : cus;
  return shared && grid > cap ? cap : grid;
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
                  hipStream_t st, bool shared) {
  if (!workspace || workspace_bytes < bp_wgrad_flat_workspace(X)) return BP_EWORKSPACE;
  FlatArgs a{};
  a.x = X->ptr; a.h = X->h; a.w = X->w; a.x_cs = X->cstride; a.x_co = X->coff;
  a.dy = Y->ptr; a.dy_cs = Y->cstride; a.dy_co = Y->coff; a.pw = pwx; a.n = X->n;
  a.tiles_x = bp_ceil_div(X->w, TW); a.tiles_y = bp_ceil_div(X->h, TH);
  const int grid = shared_grid(flat_grid(flat_tiles(X)), shared);
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

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the thin stride-2 k4 layers: Conv2d 16 -> 32 (p_y_z_in.3) and ConvTranspose2d 32 -> 16
// (p_y_z_in.22), arch cvae.py:26-45.  Both are the same sum once the full-resolution 16-channel tensor is called B
// ("big": the convolution's input / the transposed convolution's output gradient) and the half-resolution 32-channel
// one S ("small"):
//     dW[s][b][ky][kx] = sum_{n,r,c} S[n][r][c][s] * B[n][2r + ky - 1][2c + kx - 1][b]
//   * M = the 16 channels of B, N = 16 of the 32 channels of S, K = 4 consecutive S columns of one row;
//   * wave ky of a workgroup owns tap row ky: 4 tap columns x 2 N tiles = 8 accumulators kept for the whole kernel,
//     so no wave shares an output with another (no fold inside the workgroup);
//   * B is staged with its columns split by parity, [row][parity][half][16], and S as [row][N tile][col][16]: every
//     operand read is one 4-byte LDS read per lane, 256 contiguous bytes per wave: 6 reads per 8 MFMAs;
//   * tile = 4 x 32 pixels of S (10 x 66 of B), next tile in flight in registers, two workgroups per CU.
namespace {
namespace s2 {

constexpr int CB = 16, CS = 32;
constexpr int SH = 4, SW = 32;                        // S tile
constexpr int BH = 2 * SH + 2, BW = 2 * SW + 2;       // B tile (10 x 66)
constexpr int BHALF = BW / 2;                         // 33 columns of each parity
constexpr int BQ = BH * BW * (CB / 4);                // float4 units of the B tile (2640)
constexpr int SQ = SH * SW * (CS / 4);                // float4 units of the S tile (1024)
constexpr int BS = (BQ + 255) / 256;                  // 11 per thread
constexpr int SS = SQ / 256;                          // 4 per thread
constexpr int PART = 4 * 4 * 2 * 16 * 16;             // [ky][kx][nt][b][n] (8192)
static_assert(SQ % 256 == 0, "S tile units divide among the threads");

struct Args {
  const float* b; int bh, bw, b_cs, b_co;
  const float* s; int sh, sw, s_cs, s_co;
  PW pw;
  int n, tiles_x, tiles_y;
  float* partial;
};

template <bool PW_ON_S>
__global__ __launch_bounds__(256, 2) void wgrad_flat_s2_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) float bt[BH * BW * CB];
  __shared__ __attribute__((aligned(16))) float stl[SH * SW * CS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ky = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;

  const int cq = PW_ON_S ? (tid & 7) : (tid & 3);
  const PW4 p4 = pw4_load(a.pw, cq * 4, PW_ON_S ? CS : CB);
  float4 sb[BS], ss[SS];
  unsigned bin = 0, sin = 0;
  // Fetch of the next tile, one unit at a time and branch-free (loads from clamped coordinates, validity as mask bits):
  // the units are spread over the first MFMA groups of the current tile -- a burst of 15 loads per wave at the start
  // of the tile stalls the wave at issue for about a tenth of the tile's MFMA time.
  int r0 = 0, c0 = 0;
  const float* bn = nullptr;
  const float* sn = nullptr;
  auto tile_setup = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    r0 = (r / a.tiles_x) * SH; c0 = (r % a.tiles_x) * SW;
    bn = a.b + (int64_t)n * a.bh * a.bw * a.b_cs + a.b_co + (tid & 3) * 4;
    sn = a.s + (int64_t)n * a.sh * a.sw * a.s_cs + a.s_co + (tid & 7) * 4;
    bin = 0; sin = 0;
  };
  auto fetch_unit = [&](int u) {
    if (u < BS) {                                  // unit e = (row, col, quad): B[2 r0 - 1 + row][2 c0 - 1 + col]
      const int e = tid + u * 256;
      const int pix = e >> 2, col = pix % BW, row = pix / BW;
      const int iy = 2 * r0 - 1 + row, ix = 2 * c0 - 1 + col;
      const unsigned ok = (unsigned)(e < BQ) & (unsigned)(iy >= 0) & (unsigned)(iy < a.bh) & (unsigned)(ix >= 0) &
                          (unsigned)(ix < a.bw);
      bin |= ok << u;
      const int cy = min(max(iy, 0), a.bh - 1), cx = min(max(ix, 0), a.bw - 1);
      sb[u] = *reinterpret_cast<const float4*>(bn + (cy * a.bw + cx) * a.b_cs);
    } else if (u < BS + SS) {                      // unit e = (row, col, quad of 8): S[r0 + row][c0 + col]
      const int i = u - BS, e = tid + i * 256;
      const int pix = e >> 3, col = pix % SW, row = pix / SW;
      const int iy = r0 + row, ix = c0 + col;
      sin |= ((unsigned)(iy < a.sh) & (unsigned)(ix < a.sw)) << i;
      const int cy = min(iy, a.sh - 1), cx = min(ix, a.sw - 1);
      ss[i] = *reinterpret_cast<const float4*>(sn + (cy * a.sw + cx) * a.s_cs);
    }
  };
  // activation and zero padding of staged unit u (B units first), in registers: issued in the shadow of the last
  // MFMAs of the tile before, so that commit() is stores only
  auto finish = [&](int u) {                     // branch-free: the padding is an AND with 0 / ~0
    auto masked = [](float4 v, unsigned keep) {
      const unsigned m = 0u - keep;
      return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                         __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
    };
    auto act = [&](float4 v) {                   // pw4_apply4 without its uniform branch (an absent activation has
      return make_float4(pw4_apply(p4, 0, v.x), pw4_apply(p4, 1, v.y), pw4_apply(p4, 2, v.z), pw4_apply(p4, 3, v.w));
    };                                           // scale 1, shift 0, slope 1: the identity)
    if (u < BS) sb[u] = masked(PW_ON_S ? sb[u] : act(sb[u]), (bin >> u) & 1u);
    else if (u < BS + SS) ss[u - BS] = masked(PW_ON_S ? act(ss[u - BS]) : ss[u - BS], (sin >> (u - BS)) & 1u);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      const int e = tid + i * 256;
      const int pix = e >> 2, col = pix % BW, row = pix / BW;
      if (e < BQ)
        *reinterpret_cast<float4*>(bt + (((row * 2 + (col & 1)) * BHALF + (col >> 1)) * CB + (tid & 3) * 4)) = sb[i];
    }
#pragma unroll
    for (int i = 0; i < SS; ++i) {
      const int e = tid + i * 256;
      const int q8 = e & 7, pix = e >> 3, col = pix % SW, row = pix / SW;
      *reinterpret_cast<float4*>(stl + (((row * 2 + (q8 >> 2)) * SW + col) * 16 + (q8 & 3) * 4)) = ss[i];
    }
  };

  v4f acc[4][2];
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[kx][nt] = v4f{0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  if (t < ntiles) {
    tile_setup(t);
#pragma unroll
    for (int u = 0; u < BS + SS; ++u) fetch_unit(u);
#pragma unroll
    for (int u = 0; u < BS + SS; ++u) finish(u);
    commit();
  }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    tile_setup(tn < ntiles ? tn : t);              // (past the end: this tile again, not committed)
    {
      // 32 K steps (row, 4 S columns) in groups of 2; the fragments of the next group are read before the MFMAs of
      // the current one are issued (left to itself the compiler reads each fragment right before its use)
      constexpr int G = 2, NGRP = SH * (SW / 4) / G, FIN = 4, FET = 8, UPF = (BS + SS + FET - 1) / FET;
      float fa[2][G][4], fb[2][G][2];
      auto frags = [&](int buf, int grp) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int step = grp * G + j, row = step / (SW / 4), s = step % (SW / 4);
          const float* xa = bt + ((2 * row + ky) * 2 * BHALF + kq) * CB + lm;   // B row 2 row + ky of the tile
          const float* yb = stl + (row * 2 * SW + kq) * 16 + lm;
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) fa[buf][j][kx] = xa[((kx & 1) * BHALF + 4 * s + (kx >> 1)) * CB];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) fb[buf][j][nt] = yb[(nt * SW + 4 * s) * 16];
        }
      };
      frags(0, 0);
#pragma unroll
      for (int grp = 0; grp < NGRP; ++grp) {
        if (grp + 1 < NGRP) frags((grp + 1) & 1, grp + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[kx][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[grp & 1][j][kx], fb[grp & 1][j][nt], acc[kx][nt], 0, 0, 0);
        if (grp < FET) {                             // two units of the next tile per group: address arithmetic and loads
#pragma unroll
          for (int u = grp * UPF; u < (grp + 1) * UPF; ++u) fetch_unit(u);
#pragma unroll
          for (int m = 0; m < G * 8; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        }
        if (grp >= NGRP - FIN) {                     // a quarter of the staged units of the next tile per group
          constexpr int UPG = (BS + SS + FIN - 1) / FIN;
#pragma unroll
          for (int u = (grp - (NGRP - FIN)) * UPG; u < (grp - (NGRP - FIN) + 1) * UPG; ++u) finish(u);
#pragma unroll
          for (int m = 0; m < G * 8; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // ... and six vector-ALU instructions in its shadow
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (tn < ntiles) commit();
    __syncthreads();
  }

  float* dst = a.partial + (int64_t)blockIdx.x * PART + ky * (PART / 4);
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[((kx * 2 + nt) * 16 + 4 * kq + q) * 16 + lm] = acc[kx][nt][q];
}

// partial[rows][PART] -> dW[s][b][ky][kx] (the layout of both torch weights: Conv2d [co][ci], ConvTranspose2d [ci][co])
__global__ __launch_bounds__(256) void fold_kernel(const float* partial, int rows, int rows_per_block, double* part2,
                                                   float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;           // element (((ky*4 + kx)*2 + nt)*16 + b)*16 + n
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
    const int n = i % 16, b = (i / 16) % 16, nt = (i / 256) % 2, tap = i / 512;
    dst[((16 * nt + n) * CB + b) * 16 + tap] = (float)t;
  }
}

int grid_of(int ntiles) {
  static const int cap = getenv("BP_WFLAT_S2_GRID") ? atoi(getenv("BP_WFLAT_S2_GRID")) : 512;
  return ntiles < cap ? ntiles : cap;
}
int tiles_of(const bp_view* S) { return bp_ceil_div(S->w, SW) * bp_ceil_div(S->h, SH) * S->n; }

}  // namespace s2
}  // namespace

// X = the full-resolution side, Y = the half-resolution side (conv_wgrad.hip's convention for both directions)
bool bp_wgrad_flat_s2_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwx, const PW& pwy) {
  static const bool off = getenv("BP_NOWFLAT") != nullptr || getenv("BP_NOWFLAT_S2") != nullptr;
  return !off && cv->k == 4 && cv->stride == 2 && cv->pad == 1 && X->c == s2::CB && Y->c == s2::CS &&
         X->h == 2 * Y->h && X->w == 2 * Y->w && X->n == Y->n && !(pwx.scale && pwy.scale) && X->dtype == BP_F32 &&
         Y->dtype == BP_F32 && bp_view_vec4(X) && bp_view_vec4(Y);
}

size_t bp_wgrad_flat_s2_workspace(const bp_view* Y) {
  const int grid = s2::grid_of(s2::tiles_of(Y));
  const int chunks = bp_ceil_div(grid, FOLD_ROWS);
  return (size_t)grid * s2::PART * sizeof(float) + (size_t)chunks * s2::PART * sizeof(double);
}

int bp_wgrad_flat_s2(const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst, void* workspace,
                     size_t workspace_bytes, hipStream_t st, bool shared) {
  if (!workspace || workspace_bytes < bp_wgrad_flat_s2_workspace(Y)) return BP_EWORKSPACE;
  s2::Args a{};
  a.b = X->ptr; a.bh = X->h; a.bw = X->w; a.b_cs = X->cstride; a.b_co = X->coff;
  a.s = Y->ptr; a.sh = Y->h; a.sw = Y->w; a.s_cs = Y->cstride; a.s_co = Y->coff;
  a.n = X->n; a.tiles_x = bp_ceil_div(Y->w, s2::SW); a.tiles_y = bp_ceil_div(Y->h, s2::SH);
  const int grid = shared_grid(s2::grid_of(s2::tiles_of(Y)), shared);
  const int chunks = bp_ceil_div(grid, FOLD_ROWS);
  a.partial = reinterpret_cast<float*>(workspace);
  double* part2 = reinterpret_cast<double*>(a.partial + (size_t)grid * s2::PART);
  if (pwy.scale) {
    a.pw = pwy;
    hipLaunchKernelGGL((s2::wgrad_flat_s2_kernel<true>), dim3(grid), dim3(256), 0, st, a);
  } else {
    a.pw = pwx;
    hipLaunchKernelGGL((s2::wgrad_flat_s2_kernel<false>), dim3(grid), dim3(256), 0, st, a);
  }
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(s2::fold_kernel, dim3(s2::PART / 256, chunks), dim3(256), 0, st, a.partial, grid, FOLD_ROWS, part2,
                     (float*)nullptr);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(s2::fold_kernel, dim3(s2::PART / 256, 1), dim3(256), 0, st, reinterpret_cast<const float*>(part2),
                     chunks, 0, (double*)nullptr, dst);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
