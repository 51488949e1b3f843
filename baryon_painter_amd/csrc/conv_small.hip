// Unit-stride convolutions with a handful of channel pairs (8->1 k5, 1->8 k5, 1->1 k3: the last layers of
// the generator heads and their data gradients) on the vector ALUs, NHWC.
//
// With cin*cout <= 8 a 16x16x4 matrix tile is >= 87 % padding and the layer is bound by the bytes of its
// full-resolution tensors, so this kernel is built around HBM and LDS traffic instead:
//   * a workgroup stages a (TH+K-1) x (64+K-1) input tile once (TH = 16 rows, 8 where LDS demands), applying the producer's pending activation and
//     the zero padding, as channel-quad planes [row][quad][x][4] (adjacent lanes -> adjacent 16 bytes);
//   * a thread owns ONE column and TH/4 consecutive output rows: it walks the input rows of its column
//     window once, and every value it reads feeds up to TH/4 rows x CO accumulators (vertical sliding window),
//     so all lanes of a wave read consecutive LDS addresses: conflict-free b128/b32 reads;
//   * the K*K*CI*CO weights are pre-packed in correlation order [ty][tx][ci][co]; their addresses are
//     compile-time offsets from a kernel argument, i.e. scalar loads: FMA operands come from SGPRs.
// The same kernel serves the forward of a Conv2d (taps ascending, origin -pad) and the data gradient of a
// unit-stride Conv2d (taps reversed by the packing, origin pad-(K-1)).
#include "common.hpp"
#include <cstdlib>

// conv_igemm.hip: fold of per-workgroup rows of n doubles into sr->sums
size_t bp_stats_rows_bytes(int64_t rows, int C);
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st);
int bp_stats_row_stride(int n);
size_t bp_stats_rows_bytes_n(int64_t rows, int n);
int bp_stats_rows_finish_n(double* ws, int64_t rows, int n, const IgemmStatsReq* sr, hipStream_t st);

namespace {

struct SmallArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_h, out_w, out_cs, out_co;
  const float* wp; const float* bias;
  PW pw;
  int i0;            // first gathered row/column relative to the output position
  int ci_total;      // gathered channels of the layer (a multiple of the kernel's CI: it loops over chunks)
  int tiles_x, tiles_y;
  int in_vec, out_vec;
  // ACT epilogue (data gradient into a layer WITHOUT batch-norm): what is stored is g = d * act'(spw(raw)) -- the
  // producer's activation backward -- and stat gets one row {sum g, sum g*raw, sum_{t<=0} d*t} per workgroup
  const float* raw; int raw_cs, raw_co, raw_vec;
  PW spw;
  double* stat; int stat_stride;
};

template <int K, int CI, int CO, int TH, bool CHUNKED, bool ACT = false>
__global__ __launch_bounds__(256) void small_conv_kernel(SmallArgs a) {
  const int CT = CHUNKED ? a.ci_total : CI;     // gathered channels of the layer (compile-time when one tile holds them)
  constexpr int TW = 64, PXR = TH / 4;   // 4 waves x PXR rows
  constexpr int IW = TW + K - 1, IH = TH + K - 1;
  constexpr int CQ = CI >= 4 ? CI / 4 : 1;      // channel quads
  constexpr int CV = CI >= 4 ? 4 : CI;          // floats per LDS element
  __shared__ __attribute__((aligned(16))) float tile[IH * CQ * IW * CV];

  const int tid = threadIdx.x;
  const int tx_ = blockIdx.x % a.tiles_x, ty_ = blockIdx.x / a.tiles_x;
  const int n = blockIdx.y;
  const int x0 = tx_ * TW, y0 = ty_ * TH;
  const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;

  // ---- thread = column `col`, output rows r0 .. r0+PXR-1
  const int col = tid & 63, r0 = (tid >> 6) * PXR;
  // accumulators as PAIRS (v_pk_fma_f32: two fp32 FMAs per lane and issue): over produced channels where CO is even,
  // otherwise over the parity of the gathered channel (two partial sums per output, added at the end)
  constexpr bool PCO = CO % 2 == 0, PCI = !PCO && CI % 2 == 0;
  constexpr int NA = PCO ? CO / 2 : CO;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc2[PXR][NA];
#pragma unroll
  for (int j = 0; j < PXR; ++j)
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      if constexpr (PCO) acc2[j][q] = f2{a.bias ? a.bias[2 * q] : 0.f, a.bias ? a.bias[2 * q + 1] : 0.f};
      else acc2[j][q] = f2{a.bias ? a.bias[q] : 0.f, 0.f};
    }

  // layers with more gathered channels than fit one tile (k9 32->1) run in chunks of CI channels
  for (int c0 = 0; c0 < CT; c0 += CI) {
    if (c0) __syncthreads();     // the previous chunk's tile is read out
    // ---- stage the tile (activation applied, zero outside the image)
    if (CI >= 4) {
      const int c4 = tid % CQ;
      const PW4 p4 = pw4_load(a.pw, c0 + c4 * 4, CT);
      // four units per trip, loaded unconditionally from clamped coordinates before the first is used (a load under
      // a per-lane condition is waited for on the spot: the staging loop was one HBM round trip per unit)
      constexpr int NU = IH * IW * CQ;
      for (int e0 = tid; e0 < NU; e0 += 4 * 256) {
        float4 v[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = min(e0 + j * 256, NU - 1);
          const int pix = e / CQ;
          const int c = pix % IW, r = pix / IW;
          const int iy = y0 + a.i0 + r, ix = x0 + a.i0 + c;
          ok[j] = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
          const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
          const float* p = in_n + ((int64_t)cy * a.in_w + cx) * a.in_cs + c0 + c4 * 4;
          if (a.in_vec) v[j] = *reinterpret_cast<const float4*>(p);
          else v[j] = make_float4(p[0], p[1], p[2], p[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + j * 256;
          if (e < NU) {
            const int pix = e / CQ;
            const int c = pix % IW, r = pix / IW;
            const float4 w = pw4_apply4(p4, v[j]);
            *reinterpret_cast<float4*>(tile + ((r * CQ + c4) * IW + c) * 4) =
                ok[j] ? w : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    } else {
      constexpr int NE = IH * IW * CI;
      for (int e0 = tid; e0 < NE; e0 += 4 * 256) {      // (four loads in flight, as above)
        float v[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = min(e0 + j * 256, NE - 1);
          const int ch = e % CI;
          const int pix = e / CI;
          const int c = pix % IW, r = pix / IW;
          const int iy = y0 + a.i0 + r, ix = x0 + a.i0 + c;
          ok[j] = iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
          const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
          v[j] = in_n[((int64_t)cy * a.in_w + cx) * a.in_cs + c0 + ch];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = e0 + j * 256;
          if (e < NE) {
            const int ch = e % CI;
            const int pix = e / CI;
            const int c = pix % IW, r = pix / IW;
            tile[(r * IW + c) * CI + ch] = ok[j] ? pw_apply(a.pw, c0 + ch, v[j]) : 0.f;
          }
        }
      }
    }
    __syncthreads();

    // kx outermost and NOT unrolled: only the K*CI*CO weights of one tap column are live in SGPRs at a time
    // (all K*K*CI*CO would spill).
#pragma unroll 1
    for (int kx = 0; kx < K; ++kx) {
      const float* wk = a.wp + (kx * CT + c0) * CO;
      const int wrow = K * CT * CO;                   // floats between tap rows
#pragma unroll
      for (int dy = 0; dy < PXR + K - 1; ++dy) {
        float v[CI];
        if constexpr (CI >= 4) {
#pragma unroll
          for (int q = 0; q < CQ; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(tile + (((r0 + dy) * CQ + q) * IW + col + kx) * 4);
            v[q * 4 + 0] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) v[ci] = tile[((r0 + dy) * IW + col + kx) * CI + ci];
        }
#pragma unroll
        for (int j = 0; j < PXR; ++j) {
          const int ky = dy - j;               // compile-time after unrolling
          if (ky >= 0 && ky < K) {
            const float* wt = wk + ky * wrow;
            if constexpr (PCO) {
#pragma unroll
              for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int q = 0; q < NA; ++q)
                  acc2[j][q] = f2{v[ci], v[ci]} * f2{wt[ci * CO + 2 * q], wt[ci * CO + 2 * q + 1]} + acc2[j][q];
            } else if constexpr (PCI) {
#pragma unroll
              for (int ci = 0; ci < CI; ci += 2)
#pragma unroll
                for (int co = 0; co < CO; ++co)
                  acc2[j][co] = f2{v[ci], v[ci + 1]} * f2{wt[ci * CO + co], wt[(ci + 1) * CO + co]} + acc2[j][co];
            } else {
#pragma unroll
              for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int co = 0; co < CO; ++co)
                  acc2[j][co][0] = fmaf(v[ci], wt[ci * CO + co], acc2[j][co][0]);
            }
          }
        }
      }
    }
  }
  float acc[PXR][CO];
#pragma unroll
  for (int j = 0; j < PXR; ++j)
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[j][co] = PCO ? acc2[j][co / 2][co & 1] : acc2[j][co][0] + acc2[j][co][1];

  float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
  const int X = x0 + col;
  float es[ACT ? 3 : 1][ACT ? CO : 1];
  if constexpr (ACT) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int co = 0; co < CO; ++co) es[q][co] = 0.f;
  }
  if (X < a.out_w) {
#pragma unroll
    for (int j = 0; j < PXR; ++j) {
      const int Y = y0 + r0 + j;
      if (Y < a.out_h) {
        if constexpr (ACT) {
          // (the activation parameters are wave-uniform: scalar loads)
          const float* rp = a.raw + (((int64_t)n * a.out_h + Y) * a.out_w + X) * a.raw_cs + a.raw_co;
          float r[CO];
          if (CO % 4 == 0 && a.raw_vec) {
#pragma unroll
            for (int q = 0; q < CO / 4; ++q) {
              const float4 t4 = *reinterpret_cast<const float4*>(rp + 4 * q);
              r[4 * q] = t4.x; r[4 * q + 1] = t4.y; r[4 * q + 2] = t4.z; r[4 * q + 3] = t4.w;
            }
          } else {
#pragma unroll
            for (int co = 0; co < CO; ++co) r[co] = rp[co];
          }
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float d = acc[j][co];
            const float t = a.spw.scale ? fmaf(r[co], a.spw.scale[co], a.spw.shift[co]) : r[co];
            const bool pos = t > 0.f;
            const float g = pos ? d : d * (a.spw.scale ? a.spw.slope[co] : 1.f);
            acc[j][co] = g;
            es[0][co] += g;
            es[1][co] = fmaf(g, r[co], es[1][co]);
            if (!pos) es[2][co] = fmaf(d, t, es[2][co]);
          }
        }
        float* o = out_n + ((int64_t)Y * a.out_w + X) * a.out_cs;
        if (CO % 4 == 0 && a.out_vec) {
#pragma unroll
          for (int q = 0; q < CO / 4; ++q)
            *reinterpret_cast<float4*>(o + q * 4) =
                make_float4(acc[j][q * 4], acc[j][q * 4 + 1], acc[j][q * 4 + 2], acc[j][q * 4 + 3]);
        } else {
#pragma unroll
          for (int co = 0; co < CO; ++co) o[co] = acc[j][co];
        }
      }
    }
  }
  if constexpr (ACT) {
    // a thread's PXR pixels and a wave's 64 columns in fp32 (256 terms), waves and workgroups in double, fixed order
    __shared__ float ered[4][3 * CO];
    const int lane = tid & 63, wk = tid >> 6;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        float v = es[q][co];
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_down(v, sh, 64);
        if (lane == 0) ered[wk][q * CO + co] = v;
      }
    __syncthreads();
    if (tid < a.stat_stride) {                 // (rows of a power-of-two width for the fold: zero beyond 3 CO)
      const int64_t row = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
      a.stat[row * a.stat_stride + tid] =
          tid < 3 * CO ? ((double)ered[0][tid] + (double)ered[1][tid]) + ((double)ered[2][tid] + (double)ered[3][tid]) : 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Strided layers with one or two channels on one side (the 512^2 ends of the recognition / prior networks:
// Conv2d k4s2 {1,2}->8; the latent up-sampler ConvTranspose2d k8s4 / k4s2 1->1 and its data gradient).  A matrix
// tile would be > 90 % padding and the layers are a few hundred MB of HBM traffic with almost no arithmetic, so:
// one thread per produced pixel, all of its channels in registers, operands straight from global memory (every
// input value is used by (K/S)^2 neighbouring threads: the vector L1 serves the re-reads), weights from scalar
// loads (gather form) or a 1-KiB table in L1 (transposed form, where the tap set depends on the pixel's phase).
struct TinyArgs {
  const float* in; int in_h, in_w, in_cs, in_co;
  float* out; int out_h, out_w, out_cs, out_co;
  const float* wp; const float* bias;
  PW pw;
  int pad;
  int64_t total;   // produced pixels per image
  int n;
  double* stat;    // transposed form, one produced channel: rows {sum y, sum y^2} per workgroup (nullptr: none)
};

// gather form: out[y, x, :] = sum_{ty,tx,ci} act(in[S*y - pad + ty, S*x - pad + tx, ci]) * wp[ty][tx][ci][:]
template <int K, int S, int CI, int CO>
__global__ __launch_bounds__(256) void tiny_gather_kernel(TinyArgs a) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)a.total) return;
  const int x = i % (unsigned)a.out_w;
  const int y = i / (unsigned)a.out_w;
  const int n = blockIdx.y;
  const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = a.bias ? a.bias[co] : 0.f;
  float sc[CI], sf[CI], sl[CI];
  const bool pw_on = a.pw.scale != nullptr;
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    sc[ci] = pw_on ? a.pw.scale[ci] : 1.f; sf[ci] = pw_on ? a.pw.shift[ci] : 0.f; sl[ci] = pw_on ? a.pw.slope[ci] : 1.f;
  }
  const int iy0 = S * y - a.pad, ix0 = S * x - a.pad;
  const bool vec2 = CI == 2 && a.in_cs % 2 == 0 && a.in_co % 2 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 7) == 0;
#pragma unroll
  for (int ty = 0; ty < K; ++ty) {
    const int iy = iy0 + ty;
    const bool yin = iy >= 0 && iy < a.in_h;
    const float* row = in_n + (int64_t)iy * a.in_w * a.in_cs;
#pragma unroll
    for (int tx = 0; tx < K; ++tx) {
      const int ix = ix0 + tx;
      const bool in = yin && ix >= 0 && ix < a.in_w;
      float vin[CI];
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) vin[ci] = 0.f;
      if (in) {
        if (CI == 2 && vec2) {
          const float2 q = *reinterpret_cast<const float2*>(row + ix * a.in_cs);
          vin[0] = q.x; vin[CI - 1] = q.y;
        } else {
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) vin[ci] = row[ix * a.in_cs + ci];
        }
      }
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        float v = vin[ci];
        if (in && pw_on) { const float t = fmaf(v, sc[ci], sf[ci]); v = t > 0.f ? t : t * sl[ci]; }
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, a.wp[((ty * K + tx) * CI + ci) * CO + co], acc[co]);
      }
    }
  }
  float* o = a.out + (((int64_t)n * a.out_h + y) * a.out_w + x) * a.out_cs + a.out_co;
  if (CO % 4 == 0 && a.out_cs % 4 == 0 && a.out_co % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
#pragma unroll
    for (int q = 0; q < CO / 4; ++q)
      *reinterpret_cast<float4*>(o + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  } else {
#pragma unroll
    for (int co = 0; co < CO; ++co) o[co] = acc[co];
  }
}

// transposed form: out[Y, X, :] = sum_{j,i,ci} act(in[(Y+pad)/S - j, (X+pad)/S - i, ci]) * wp[ry + S*j][rx + S*i][ci][:],
// r = (Y+pad) % S: K/S taps per dimension.
template <int K, int S, int CI, int CO>
__global__ __launch_bounds__(256) void tiny_transposed_kernel(TinyArgs a) {
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const bool live = i0 < (unsigned)a.total;
  if (!live && !a.stat) return;
  const unsigned i = live ? i0 : 0u;             // (idle lanes of the last workgroup: compute pixel 0, store nothing)
  constexpr int T = K / S;
  const int X = i % (unsigned)a.out_w;
  const int Y = i / (unsigned)a.out_w;
  const int n = blockIdx.y;
  const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = a.bias ? a.bias[co] : 0.f;
  const bool pw_on = a.pw.scale != nullptr;
  const int qy = (Y + a.pad) / S, ry = (Y + a.pad) % S;
  const int qx = (X + a.pad) / S, rx = (X + a.pad) % S;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int iy = qy - j;
    const bool yin = iy >= 0 && iy < a.in_h;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int ix = qx - t;
      if (!(yin && ix >= 0 && ix < a.in_w)) continue;
      const float* p = in_n + ((int64_t)iy * a.in_w + ix) * a.in_cs;
      const float* w = a.wp + (((ry + S * j) * K + rx + S * t) * CI) * CO;
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        float v = p[ci];
        if (pw_on) { const float tt = fmaf(v, a.pw.scale[ci], a.pw.shift[ci]); v = tt > 0.f ? tt : tt * a.pw.slope[ci]; }
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, w[ci * CO + co], acc[co]);
      }
    }
  }
  float* o = a.out + (((int64_t)n * a.out_h + Y) * a.out_w + X) * a.out_cs + a.out_co;
  if (live) {
#pragma unroll
    for (int co = 0; co < CO; ++co) o[co] = acc[co];
  }
  if constexpr (CO == 1) {
    if (a.stat) {          // batch-norm sums of what was stored (the latent up-sampler's strided one-channel output:
      //                      the separate pass reads a 4-channel slot for one channel, 0.08 ms at 512^2)
      __shared__ double red[4][2];
      double d1 = live ? (double)acc[0] : 0.0, d2 = d1 * d1;
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) { d1 += __shfl_down(d1, sh, 64); d2 += __shfl_down(d2, sh, 64); }
      if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = d1; red[threadIdx.x >> 6][1] = d2; }
      __syncthreads();
      if (threadIdx.x < 2) {
        const int64_t row = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        a.stat[row * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      }
    }
  }
}

template <int K, int S, int CI, int CO>
int launch_tiny(const TinyArgs& a, bool transposed, hipStream_t st) {
  const dim3 grid((unsigned)((a.total + 255) / 256), (unsigned)a.n);
  if (transposed) hipLaunchKernelGGL((tiny_transposed_kernel<K, S, CI, CO>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((tiny_gather_kernel<K, S, CI, CO>), grid, dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

bool tiny_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOTINY") != nullptr;
  if (off || g.stride < 2) return false;
  const bool ks = (g.k == 4 && g.stride == 2) || (g.k == 8 && g.stride == 4);
  if (!ks) return false;
  if (g.gather_transposed) return g.cin_g == 1 && g.cout_g == 1;
  return (g.cin_g == 1 && g.cout_g == 1) || (g.k == 4 && g.cin_g <= 2 && g.cout_g == 8);
}

bool tiny_stats_geom(const ConvGeom& g) { return tiny_ok(g) && g.gather_transposed && g.cin_g == 1 && g.cout_g == 1; }
int64_t tiny_rows(const bp_view* out) { return (((int64_t)out->h * out->w + 255) / 256) * out->n; }

int tiny_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
             const bp_view* out, hipStream_t st, const IgemmStatsReq* sr = nullptr) {
  TinyArgs a{};
  if (sr) {
    const int64_t rows = tiny_rows(out);
    if (sr->mode != 1 || !tiny_stats_geom(g) || bias) return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < bp_stats_rows_bytes(rows, 1) || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.bias = bias; a.pw = pw; a.pad = g.pad;
  a.total = (int64_t)out->h * out->w; a.n = out->n;
  if (a.total > 0x7fffff00LL || out->n > 65535) return BP_EUNSUPPORTED;
  const bool tr = g.gather_transposed != 0;
  if (sr) {
    const int rc = g.k == 4 ? launch_tiny<4, 2, 1, 1>(a, tr, st) : launch_tiny<8, 4, 1, 1>(a, tr, st);
    return rc != BP_OK ? rc : bp_stats_rows_finish(a.stat, tiny_rows(out), 1, sr, st);
  }
  if (g.k == 4 && g.cin_g == 1 && g.cout_g == 1) return launch_tiny<4, 2, 1, 1>(a, tr, st);
  if (g.k == 8 && g.cin_g == 1 && g.cout_g == 1) return launch_tiny<8, 4, 1, 1>(a, tr, st);
  if (g.k == 4 && g.cin_g == 1 && g.cout_g == 8 && !tr) return launch_tiny<4, 2, 1, 8>(a, tr, st);
  if (g.k == 4 && g.cin_g == 2 && g.cout_g == 8 && !tr) return launch_tiny<4, 2, 2, 8>(a, tr, st);
  return BP_EUNSUPPORTED;
}

struct SmallPackArgs {
  const float* w; float* dst;
  int64_t sa, sb;
  int k, ci, co, flip, total;
};

__global__ void small_pack_kernel(SmallPackArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.total) return;
  int r = i;
  const int co = r % a.co; r /= a.co;
  const int ci = r % a.ci; r /= a.ci;
  const int tx = r % a.k, ty = r / a.k;
  const int ky = a.flip ? a.k - 1 - ty : ty, kx = a.flip ? a.k - 1 - tx : tx;
  a.dst[i] = a.w[ci * a.sa + co * a.sb + ky * a.k + kx];
}

template <int K, int CI, int CO, int TH, bool CHUNKED = false, bool ACT = false>
int launch(SmallArgs a, const bp_view* out, int n, hipStream_t st) {
  a.tiles_x = bp_ceil_div(out->w, 64);
  a.tiles_y = bp_ceil_div(out->h, TH);
  hipLaunchKernelGGL((small_conv_kernel<K, CI, CO, TH, CHUNKED, ACT>), dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)n),
                     dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// the ACT epilogue exists for the data gradient of the heads' 8 -> 1 k5 layer (gathers 1 channel, produces 8)
bool act_geom(const ConvGeom& g) {
  static const bool off = getenv("BP_NOACTEPI") != nullptr;
  return !off && g.IS == 1 && g.OS == 1 && g.nphase == 1 && g.stride == 1 && g.cin_g == 1 &&
         ((g.k == 5 && g.cout_g == 8) || (g.k == 3 && g.cout_g == 1));      // (... and of the 1 -> 1 k3 layer behind it)
}
int64_t act_rows(const bp_view* out) { return (int64_t)bp_ceil_div(out->w, 64) * bp_ceil_div(out->h, 16) * out->n; }

}  // namespace

// Does (k, gathered channels, produced channels) of this unit-stride correlation have an instantiation?
bool bp_small_ok(const ConvGeom& g) {
  static const bool off = getenv("BP_NOSMALL") != nullptr;
  if (!off && tiny_ok(g)) return true;
  if (off || g.IS != 1 || g.OS != 1 || g.nphase != 1 || g.stride != 1) return false;
  return (g.k == 5 && g.cin_g == 8 && g.cout_g == 1) || (g.k == 5 && g.cin_g == 1 && g.cout_g == 8) ||
         (g.k == 5 && g.cin_g == 16 && g.cout_g == 1) || (g.k == 3 && g.cin_g == 1 && g.cout_g == 1) ||
         (g.k == 9 && g.cin_g % 8 == 0 && g.cin_g <= 64 && g.cout_g <= 2) ||    // CGAN head and stem gradient
         (g.k == 4 && g.cin_g % 8 == 0 && g.cin_g <= 1024 && g.cout_g == 1);   // PatchGAN logits (512 -> 1)
}

int64_t bp_small_packed_floats(const ConvGeom& g) { return (int64_t)g.k * g.k * g.cin_g * g.cout_g; }

int bp_small_kernel_id(const ConvGeom& g) {
  if (tiny_ok(g)) return 800000 + g.k * 10000 + g.stride * 1000 + g.cin_g * 100 + g.cout_g * 10 + g.gather_transposed;
  return 900000 + g.k * 1000 + g.cin_g * 10 + g.cout_g;
}

int bp_small_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st) {
  SmallPackArgs a{};
  a.w = w_torch; a.dst = packed; a.sa = wm.sa; a.sb = wm.sb;
  a.k = g.k; a.ci = g.cin_g; a.co = g.cout_g;
  a.flip = tiny_ok(g) ? 0 : g.gather_transposed;       // (the strided kernels index taps by the torch (ky, kx))
  a.total = (int)bp_small_packed_floats(g);
  hipLaunchKernelGGL(small_pack_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// mode 3 (IgemmStatsReq): activation backward of the produced slot in the epilogue; 0 = this layer has none
size_t bp_small_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode) {
  if (mode == 1 && tiny_stats_geom(g)) return bp_stats_rows_bytes(tiny_rows(out), 1);
  if (mode != 3 || tiny_ok(g) || !act_geom(g)) return 0;
  return bp_stats_rows_bytes_n(act_rows(out), 3 * g.cout_g);
}

int bp_small_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                 const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  if (tiny_ok(g)) return tiny_run(g, in, pw, packed, bias, out, st, sr);
  if (sr && (sr->mode != 3 || !act_geom(g) || bias)) return BP_EUNSUPPORTED;
  SmallArgs a{};
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed; a.bias = bias; a.pw = pw; a.ci_total = g.cin_g;
  a.i0 = g.gather_transposed ? bp_t_i0(0, g.pad, 1, g.k) : -g.pad;
  a.in_vec = bp_view_vec4(in) ? 1 : 0;
  a.out_vec = bp_view_vec4(out) ? 1 : 0;
  if (in->n > 65535) return BP_EUNSUPPORTED;
  if (g.k == 5 && g.cin_g == 8 && g.cout_g == 1) return launch<5, 8, 1, 16>(a, out, in->n, st);
  if (sr) {
    const bp_view* r = sr->raw;
    const int64_t rows = act_rows(out);
    if (!r || r->n != out->n || r->h != out->h || r->w != out->w || r->c != out->c || !sr->sums) return BP_EINVAL;
    const int ns = 3 * g.cout_g;
    if (!sr->ws || sr->ws_bytes < bp_stats_rows_bytes_n(rows, ns)) return BP_EWORKSPACE;
    a.raw = r->ptr; a.raw_cs = r->cstride; a.raw_co = r->coff; a.raw_vec = bp_view_vec4(r) ? 1 : 0;
    a.spw = sr->spw; a.stat = reinterpret_cast<double*>(sr->ws); a.stat_stride = bp_stats_row_stride(ns);
    const int rc = g.k == 5 ? launch<5, 1, 8, 16, false, true>(a, out, in->n, st)
                            : launch<3, 1, 1, 16, false, true>(a, out, in->n, st);
    return rc != BP_OK ? rc : bp_stats_rows_finish_n(a.stat, rows, ns, sr, st);
  }
  if (g.k == 5 && g.cin_g == 1 && g.cout_g == 8) return launch<5, 1, 8, 16>(a, out, in->n, st);
  if (g.k == 5 && g.cin_g == 16 && g.cout_g == 1) return launch<5, 16, 1, 8>(a, out, in->n, st);
  if (g.k == 3 && g.cin_g == 1 && g.cout_g == 1) return launch<3, 1, 1, 16>(a, out, in->n, st);
  if (g.k == 4 && g.cout_g == 1) return launch<4, 8, 1, 16, true>(a, out, in->n, st);
  if (g.k == 9 && g.cout_g == 1) return launch<9, 8, 1, 8, true>(a, out, in->n, st);
  if (g.k == 9 && g.cout_g == 2) return launch<9, 8, 2, 8, true>(a, out, in->n, st);
  return BP_EUNSUPPORTED;
}
