// The generator's stem: Conv2d 3 -> 16, k5, stride 1, pad 2 on full-resolution tiles (arch p_y_z_in.0, cvae.py:26-45),
// forward and weight gradient.  Its tensors are the largest of the network (16 channels at 512^2) and its arithmetic
// the smallest, so the general igemm (K padded to 4 channels x 25 taps, one weight slab staged per tap row) spends its
// time on staging and barriers.  Here the K dimension of the fp32 MFMA is the FLATTENED (tap column, channel) index of
// one tap row, kf = 3*tx + ci in [0, 15): in an NHWC row with the 3 channels packed, the operand of pixel x is the
// window row[3*x + kf] -- 16 consecutive floats -- so the four K-groups of a tap row are plain strided LDS reads of the
// same flat row, 20 MFMAs (5 tap rows x 4 groups) per 16 pixels instead of 25, and all 20 weight fragments live in
// registers for the whole kernel (no weight staging, no barrier inside a tile).
//   forward   D[co][px] = sum_{ty, g} W[ty][g] (16 co x 4 kf)  x  Xflat[ty][g] (4 kf x 16 px)
//   wgrad     D_ty[co][kf] = sum_{px}  dY (16 co x 4 px)  x  Xflat[ty] (4 px x 16 kf): operands straight from global
//             memory (the dY fragment of 4 pixels x 16 channels is 256 contiguous bytes)
// Workgroups walk the tile sequence with a grid stride; the next tile is fetched into registers before the current
// one's MFMAs and written to the other LDS buffer after them: one barrier per tile.  Training-mode batch-norm sums
// ({sum y, sum y^2} per channel, in double) are taken from the accumulators and reduced once per workgroup.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int K5 = 5, CI3 = 3, CO16 = 16;
constexpr int TH = 8, TW = 64;                         // output tile of a workgroup: 4 waves x 2 rows x 64 columns
constexpr int IH = TH + K5 - 1, IWP = TW + K5 - 1;     // staged rows / pixels per row
constexpr int ROWF = IWP * CI3;                        // floats per staged row (204)
constexpr int NPIX = IH * IWP;                         // staged pixels (816)
constexpr int PPT = (NPIX + 255) / 256;                // pixels per thread (4)

struct StemArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_cs, out_co;
  const float* bias;
  const float* wp;            // [ty][g][kq][co] = W[co][ci][ty][tx], kf = 4*g + kq = 3*tx + ci (kf = 15: zero)
  PW pw;
  int n, tiles_x, tiles_y, in_vec, out_vec;
  double* stat;               // partial sums [workgroup][2][16] or nullptr
};

template <bool OUT_VEC>
__global__ __launch_bounds__(256) void stem_forward_kernel(StemArgs a) {
  // (+16 zeroed floats: the K-group of kf = 12..15 of a row's last pixel reads one float past the row, with weight 0 --
  // past the LAST row that float must not be a NaN bit pattern left in LDS)
  __shared__ float tile[2][IH * ROWF + 16];
  __shared__ double red[4][2][CO16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;

  float wreg[K5][4];
#pragma unroll
  for (int ty = 0; ty < K5; ++ty)
#pragma unroll
    for (int g = 0; g < 4; ++g) wreg[ty][g] = a.wp[((ty * 4 + g) * 4 + kq) * CO16 + lm];
  float sc[CI3], sf[CI3], sl[CI3];
  const bool pw_on = a.pw.scale != nullptr;
#pragma unroll
  for (int c = 0; c < CI3; ++c) {
    sc[c] = pw_on ? a.pw.scale[c] : 1.f; sf[c] = pw_on ? a.pw.shift[c] : 0.f; sl[c] = pw_on ? a.pw.slope[c] : 1.f;
  }

  v4f b4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) b4 = v4f{a.bias[4 * kq], a.bias[4 * kq + 1], a.bias[4 * kq + 2], a.bias[4 * kq + 3]};
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float stage[PPT][CI3];
  auto fetch = [&](int t) {                     // tile t -> registers (activation applied, zero outside the image)
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH - 2, x0 = (r % a.tiles_x) * TW - 2;
    const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = tid + i * 256;
      const int row = e / IWP, col = e % IWP;
      const int iy = y0 + row, ix = x0 + col;
      float v[CI3] = {0.f, 0.f, 0.f};
      if (e < NPIX && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) {
        const float* p = in_n + ((int64_t)iy * a.in_w + ix) * a.in_cs;
        if (a.in_vec) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          v[0] = q.x; v[1] = q.y; v[2] = q.z;
        } else {
          v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        }
        if (pw_on) {
#pragma unroll
          for (int c = 0; c < CI3; ++c) { const float tt = fmaf(v[c], sc[c], sf[c]); v[c] = tt > 0.f ? tt : tt * sl[c]; }
        }
      }
#pragma unroll
      for (int c = 0; c < CI3; ++c) stage[i][c] = v[c];
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = tid + i * 256;
      if (e < NPIX) {
#pragma unroll
        for (int c = 0; c < CI3; ++c) tile[buf][e * CI3 + c] = stage[i][c];
      }
    }
  };

  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < 16) { tile[0][IH * ROWF + tid] = 0.f; tile[1][IH * ROWF + tid] = 0.f; }
  int t = blockIdx.x, buf = 0;
  if (t < ntiles) { fetch(t); commit(0); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH, x0 = (r % a.tiles_x) * TW;
    float* out_n = a.out + (int64_t)n * a.in_h * a.in_w * a.out_cs + a.out_co;
    const float* lt = tile[buf];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll
      for (int ct = 0; ct < TW / 16; ++ct) {
        v4f acc = b4;
        const float* base = lt + row * ROWF + CI3 * (ct * 16 + lm) + kq;
#pragma unroll
        for (int ty = 0; ty < K5; ++ty)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ty][g], base[ty * ROWF + 4 * g], acc, 0, 0, 0);
        const int Y = y0 + row, X = x0 + ct * 16 + lm;
        if (Y < a.in_h && X < a.in_w) {
          float* o = out_n + ((int64_t)Y * a.in_w + X) * a.out_cs + 4 * kq;
          // (a template parameter, not a run-time branch: the compiler merges two branches that store the same values
          // into the four-dword form)
          if constexpr (OUT_VEC) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
          else { o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3]; }
          if (a.stat) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += (double)acc[q]; s2[q] = fma((double)acc[q], (double)acc[q], s2[q]); }
          }
        }
      }
    }
    if (tn < ntiles) commit(buf ^ 1);
    buf ^= 1;
    __syncthreads();
  }
  if (a.stat) {
    // lanes lm of a wave share channels 4*kq .. 4*kq+3
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
    if (tid < 2 * CO16) {
      const int s = tid / CO16, c = tid % CO16;
      a.stat[(int64_t)blockIdx.x * 2 * CO16 + tid] = red[0][s][c] + red[1][s][c] + red[2][s][c] + red[3][s][c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the stem: dW[co][ci][ty][tx] = sum_{n,y,x} dY[n,y,x,co] * act(X)[n, y+ty-2, x+tx-2, ci].
// Per 4 pixels of a row one MFMA per tap row: D_ty (16 co x 16 kf) += dY (16 co x 4 px) x Xflat (4 px x 16 kf), with
// Xflat[px][kf] = tile_row[3*px + kf] out of the same staged (activated, zero-padded) tile as the forward pass and the
// dY fragment (4 pixels x 16 channels = 256 contiguous bytes) straight from global memory, read exactly once.  A wave
// keeps its 5 accumulators over all the tiles it visits; waves, then workgroups, are folded in a fixed order.
struct StemWgradArgs {
  const float* x; int h, w, x_cs, x_co;
  const float* dy; int dy_cs, dy_co;
  PW pw;
  int n, tiles_x, tiles_y, x_vec;
  float* partial;             // [workgroup][ty][co][kf]
};

constexpr int WG_ELEMS = K5 * CO16 * 16;      // 1280

__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemWgradArgs a) {
  __shared__ float tile[2][IH * ROWF + 16];   // (+16: the kf = 15 column of the last pixels reads past the row)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  float sc[CI3], sf[CI3], sl[CI3];
  const bool pw_on = a.pw.scale != nullptr;
#pragma unroll
  for (int c = 0; c < CI3; ++c) {
    sc[c] = pw_on ? a.pw.scale[c] : 1.f; sf[c] = pw_on ? a.pw.shift[c] : 0.f; sl[c] = pw_on ? a.pw.slope[c] : 1.f;
  }
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntiles = per_img * a.n;
  float stage[PPT][CI3];
  auto fetch = [&](int t) {
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH - 2, x0 = (r % a.tiles_x) * TW - 2;
    const float* in_n = a.x + (int64_t)n * a.h * a.w * a.x_cs + a.x_co;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = tid + i * 256;
      const int row = e / IWP, col = e % IWP;
      const int iy = y0 + row, ix = x0 + col;
      float v[CI3] = {0.f, 0.f, 0.f};
      if (e < NPIX && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) {
        const float* p = in_n + ((int64_t)iy * a.w + ix) * a.x_cs;
        if (a.x_vec) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          v[0] = q.x; v[1] = q.y; v[2] = q.z;
        } else {
          v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        }
        if (pw_on) {
#pragma unroll
          for (int c = 0; c < CI3; ++c) { const float tt = fmaf(v[c], sc[c], sf[c]); v[c] = tt > 0.f ? tt : tt * sl[c]; }
        }
      }
#pragma unroll
      for (int c = 0; c < CI3; ++c) stage[i][c] = v[c];
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = tid + i * 256;
      if (e < NPIX) {
#pragma unroll
        for (int c = 0; c < CI3; ++c) tile[buf][e * CI3 + c] = stage[i][c];
      }
    }
  };

  v4f acc[K5];
#pragma unroll
  for (int ty = 0; ty < K5; ++ty) acc[ty] = v4f{0.f, 0.f, 0.f, 0.f};
  if (tid < 16) { tile[0][IH * ROWF + tid] = 0.f; tile[1][IH * ROWF + tid] = 0.f; }
  int t = blockIdx.x, buf = 0;
  if (t < ntiles) { fetch(t); commit(0); }
  __syncthreads();
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn);
    const int n = t / per_img, r = t % per_img;
    const int y0 = (r / a.tiles_x) * TH, x0 = (r % a.tiles_x) * TW;
    const float* dy_n = a.dy + (int64_t)n * a.h * a.w * a.dy_cs + a.dy_co;
    const float* lt = tile[buf];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
      const int Y = y0 + row;
      float av[TW / 4];
#pragma unroll
      for (int q = 0; q < TW / 4; ++q) {
        const int X = x0 + 4 * q + kq;
        av[q] = (Y < a.h && X < a.w) ? dy_n[((int64_t)Y * a.w + X) * a.dy_cs + lm] : 0.f;
      }
      const float* base = lt + row * ROWF + CI3 * kq + lm;
#pragma unroll
      for (int q = 0; q < TW / 4; ++q)
#pragma unroll
        for (int ty = 0; ty < K5; ++ty)
          acc[ty] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], base[ty * ROWF + CI3 * 4 * q], acc[ty], 0, 0, 0);
    }
    if (tn < ntiles) commit(buf ^ 1);
    buf ^= 1;
    __syncthreads();
  }
  // fold the four waves (waves 1..3 through the tile memory, in wave order), one row per workgroup
  float* red = &tile[0][0];
  if (wave > 0) {
#pragma unroll
    for (int ty = 0; ty < K5; ++ty)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[(wave - 1) * WG_ELEMS + (ty * CO16 + 4 * kq + q) * 16 + lm] = acc[ty][q];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = a.partial + (int64_t)blockIdx.x * WG_ELEMS;
#pragma unroll
    for (int ty = 0; ty < K5; ++ty)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (ty * CO16 + 4 * kq + q) * 16 + lm;
        dst[i] = ((acc[ty][q] + red[i]) + red[WG_ELEMS + i]) + red[2 * WG_ELEMS + i];
      }
  }
}

// partial[rows][1280] -> dW in torch layout [co][ci][ty][tx]: thread per element, rows in order, in double
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* partial, int rows, int rows_per_block,
                                                              double* part2, float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;           // element (ty, co, kf)
  if (i >= WG_ELEMS) return;
  if (part2) {                                            // stage 1: a chunk of rows -> part2[chunk][1280]
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    double t = 0.0;
    for (int r = r0; r < r1; ++r) t += (double)partial[(int64_t)r * WG_ELEMS + i];
    part2[(int64_t)blockIdx.y * WG_ELEMS + i] = t;
  } else {                                                // stage 2: chunks -> dW
    const double* p2 = reinterpret_cast<const double*>(partial);
    double t = 0.0;
    for (int r = 0; r < rows; ++r) t += p2[(int64_t)r * WG_ELEMS + i];
    const int kf = i % 16, co = (i / 16) % CO16, ty = i / (16 * CO16);
    if (kf < K5 * CI3) dst[(co * CI3 + kf % CI3) * (K5 * K5) + ty * K5 + kf / CI3] = (float)t;
  }
}

int stem_wgrad_grid(int ntiles) { return ntiles < 1024 ? ntiles : 1024; }
constexpr int FOLD_ROWS = 32;

struct StemPackArgs { const float* w; float* dst; int64_t sa, sb; };
__global__ void stem_pack_kernel(StemPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // ((ty*4 + g)*4 + kq)*16 + co
  if (i >= K5 * 16 * CO16) return;
  const int co = i % CO16, kf = (i / CO16) % 16, ty = i / (CO16 * 16);
  const int tx = kf / CI3, ci = kf % CI3;
  a.dst[i] = kf < K5 * CI3 ? a.w[ci * a.sa + co * a.sb + ty * K5 + tx] : 0.f;
}

int stem_grid(int ntiles) {
  static const int cap = getenv("BP_STEM_GRID") ? atoi(getenv("BP_STEM_GRID")) : 256 * 8;
  return ntiles < cap ? ntiles : cap;
}

}  // namespace

bool bp_stem_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOSTEM") != nullptr;
  return !off && !g.gather_transposed && g.k == K5 && g.stride == 1 && g.pad == 2 && g.cin_g == CI3 && g.cout_g == CO16;
}

int64_t bp_stem_packed_floats() { return K5 * 16 * CO16; }

int bp_stem_pack(const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  StemPackArgs a{w_torch, packed, wm.sa, wm.sb};
  hipLaunchKernelGGL(stem_pack_kernel, dim3((K5 * 16 * CO16 + 255) / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

size_t bp_stem_stats_workspace(const bp_view* out) {
  const int ntiles = bp_ceil_div(out->w, TW) * bp_ceil_div(out->h, TH) * out->n;
  return (size_t)stem_grid(ntiles) * 2 * CO16 * sizeof(double);
}

int bp_stem_run(const bp_view* in, const PW& pw, const float* packed, const float* bias, const bp_view* out,
                hipStream_t st, const IgemmStatsReq* sr) {
  if (sr && (bias || sr->mode != 1)) return BP_EUNSUPPORTED;
  StemArgs a{};
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.bias = bias; a.pw = pw; a.n = in->n;
  a.tiles_x = bp_ceil_div(out->w, TW); a.tiles_y = bp_ceil_div(out->h, TH);
  a.in_vec = (in->cstride >= 4 && bp_view_vec4(in)) ? 1 : 0;      // (reads the 4th float of a pixel: it must exist)
  a.out_vec = bp_view_vec4(out) ? 1 : 0;
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.n;
  if (ntiles > 0x7fffffff) return BP_EUNSUPPORTED;
  const int grid = stem_grid((int)ntiles);
  if (sr) {
    if (!sr->ws || sr->ws_bytes < bp_stem_stats_workspace(out) || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  if (a.out_vec) hipLaunchKernelGGL(stem_forward_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(stem_forward_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  if (sr) return bp_sum_partials_req(a.stat, grid, 2 * CO16, sr, st);
  return BP_OK;
}

// ---- weight gradient
bool bp_stem_wgrad_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy, const float* dbias) {
  static const bool off = getenv("BP_NOSTEM") != nullptr;
  return !off && !cv->transposed && cv->k == K5 && cv->stride == 1 && cv->pad == 2 && cv->cin == CI3 &&
         cv->cout == CO16 && X->c == CI3 && Y->c == CO16 && X->h == Y->h && X->w == Y->w && pwy.scale == nullptr &&
         dbias == nullptr && X->dtype == BP_F32 && Y->dtype == BP_F32;
}

static int stem_wgrad_tiles(const bp_view* X) { return bp_ceil_div(X->w, TW) * bp_ceil_div(X->h, TH) * X->n; }

size_t bp_stem_wgrad_workspace(const bp_view* X) {
  const int grid = stem_wgrad_grid(stem_wgrad_tiles(X));
  const int chunks = bp_ceil_div(grid, FOLD_ROWS);
  return (size_t)grid * WG_ELEMS * sizeof(float) + (size_t)chunks * WG_ELEMS * sizeof(double);
}

int bp_stem_wgrad(const bp_view* X, const PW& pwx, const bp_view* Y, float* dst, void* workspace, size_t workspace_bytes,
                  hipStream_t st) {
  if (!workspace || workspace_bytes < bp_stem_wgrad_workspace(X)) return BP_EWORKSPACE;
  StemWgradArgs a{};
  a.x = X->ptr; a.h = X->h; a.w = X->w; a.x_cs = X->cstride; a.x_co = X->coff;
  a.dy = Y->ptr; a.dy_cs = Y->cstride; a.dy_co = Y->coff; a.pw = pwx; a.n = X->n;
  a.tiles_x = bp_ceil_div(X->w, TW); a.tiles_y = bp_ceil_div(X->h, TH);
  a.x_vec = (X->cstride >= 4 && bp_view_vec4(X)) ? 1 : 0;
  const int ntiles = stem_wgrad_tiles(X);
  const int grid = stem_wgrad_grid(ntiles);
  const int chunks = bp_ceil_div(grid, FOLD_ROWS);
  a.partial = reinterpret_cast<float*>(workspace);
  double* part2 = reinterpret_cast<double*>(a.partial + (size_t)grid * WG_ELEMS);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3(WG_ELEMS / 256, chunks), dim3(256), 0, st, a.partial, grid, FOLD_ROWS,
                     part2, (float*)nullptr);
  BP_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3(WG_ELEMS / 256, 1), dim3(256), 0, st,
                     reinterpret_cast<const float*>(part2), chunks, 0, (double*)nullptr, dst);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
