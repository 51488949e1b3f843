// bf16 matrix-core convolutions (v_mfma_f32_16x16x32_bf16): BASELINE.json configs[3] -- bf16 activations and
// gradients in HBM, fp32 accumulation, fp32 master weights.  At bf16 the step is HBM-bound (ridge
// 310 FLOP/B vs an arithmetic intensity of ~200), so these kernels are built around bytes, not MFMA issue:
//   * activations / gradients are read as 16-byte vectors (8 bf16 channels) and written as 16-byte vectors;
//   * the pending batch-norm affine + (leaky) ReLU of the producer is applied in fp32 registers on the way into LDS,
//     exactly like the fp32 kernels ("lazy activation"), and rounded to bf16 once;
//   * either side of a layer may still be fp32 (the few-channel edges of the network stay fp32: a 1-channel tensor
//     at 512^2 is 1/16 of the bytes of its 16-channel neighbour), so input and output element types are template
//     parameters of one kernel, not separate code paths.
//
// Forward / data gradient: the stride-IS correlation over output phases of conv_igemm.hip (ConvGeom), with the GEMM
// K dimension = 32 consecutive bf16 of the LDS halo image  [row][x % IS][x / IS][CC]:
//     CC = 32  one tap, 32 channels of a channel chunk         (cin 32, 64, 128, ...)
//     CC = 16  two x-adjacent taps of a 16-channel tensor;  CC = 8: four;  CC = 4: eight (the 3(+1)-channel stem)
// i.e. a lane's 16-byte fragment is ALWAYS 8 consecutive bf16 of the image, whatever the channel count, and the
// packed weights carry zeros for the taps of a run that do not exist.  D = W-tile x X-tile, so a lane ends up with
// 4 channels of one pixel per N tile; the packing interleaves the channels of an N-tile pair so that those are 8
// consecutive channels = one 16-byte bf16 store.
//
// Weight gradient: M = 16 coarse-grid (Y) channels, N = 16 fine-grid (X) channels, K = 32 pixels of a row; both
// fragments are K-major while the tensors are channel-major, so both come out of LDS through the transposing read
// ds_read_b64_tr_b16 (4 pixels x 16 channels per 16 lanes) -- no transposed copy of anything is ever stored.
//
// Compile-time split: this header holds the kernels and their launch templates; conv_bf16_cc{4,8,16,32}.hip instantiate
// them for one channel-chunk width each (one translation unit per width: they build in parallel), conv_bf16.hip holds
// the configuration, the weight packing and the entry points, conv_wgrad_bf16.hip the weight gradient.
#pragma once
#include "common.hpp"
#include <cstdlib>

namespace bpbf16 {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }   // RNE, NaN stays NaN
// two floats -> two bf16 in one word (a in the low half): ONE v_cvt_pk_bf16_f32 (RNE, a NaN stays a NaN).  Written as
// two scalar casts + shift + or, the compiler pairs the casts of DIFFERENT words and shuffles halves afterwards.
typedef float pk_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const pk_f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_b2));
}

// ---------------------------------------------------------------------------------------------- forward / dgrad
struct BArgs {
  const void* in; int in_h, in_w, in_cs, in_co, cin;
  void* out; int out_h, out_w, out_cs, out_co, cout;
  const u16* wp; const float* bias;
  PW pw;
  int tapsy, ISy, ISx, OS, nphase, transposed, stride, pad;
  int nrun;                 // K-steps per tap row and channel chunk
  int run_off[16];          // LDS pixel offset of run s inside a tap row: xm * IWq + xq
  int tiles_x, tiles_y, TPR, BH;
  int nchunk, cout_padP;
  int IH, IWq;
  int npixp;                // pixels of the LDS halo image, padded to 16 (plane stride of the CC = 32 layout)
  double* stat; int stat_c; // batch-norm sums from the epilogue: rows [tile x image x phase][2][stat_c], or nullptr
};

__device__ __forceinline__ void b_tile_of_block(int* tile, int* by) {
  const int gx = gridDim.x, n = gx * gridDim.y;
  const int L = blockIdx.y * gx + blockIdx.x;
  const int q = n >> 3, r = n & 7;
  const int xcd = L & 7, idx = L >> 3;
  const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  *tile = Lp % gx;
  *by = Lp / gx;
}

// U channels of one pixel: global (fp32 or bf16) -> fp32 registers
template <int U, bool IN_BF16>
__device__ __forceinline__ void load_unit(const void* base, int64_t elem_off, float (&v)[U]) {
  if constexpr (IN_BF16) {
    const u16* p = reinterpret_cast<const u16*>(base) + elem_off;
    if constexpr (U == 8) {
      const uint4 t = *reinterpret_cast<const uint4*>(p);
      const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
    } else {
      const uint2 t = *reinterpret_cast<const uint2*>(p);
      v[0] = bf2f((u16)(t.x & 0xffffu)); v[1] = bf2f((u16)(t.x >> 16));
      v[2] = bf2f((u16)(t.y & 0xffffu)); v[3] = bf2f((u16)(t.y >> 16));
    }
  } else {
    const float* p = reinterpret_cast<const float*>(base) + elem_off;
#pragma unroll
    for (int j = 0; j < U; j += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + j);
      v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
    }
  }
}

// The same unit as raw 16-byte words (conversion deferred): a prefetch must not touch the loaded registers, or the
// wait for the data lands right behind the load instead of after the MFMAs the load was meant to overlap.
template <int U, bool IN_BF16> struct RawUnit { uint4 q[IN_BF16 ? 1 : U / 4]; };
template <int U, bool IN_BF16>
__device__ __forceinline__ void load_unit_raw(const void* base, int64_t elem_off, RawUnit<U, IN_BF16>& r) {
  if constexpr (IN_BF16) {
    const u16* p = reinterpret_cast<const u16*>(base) + elem_off;
    if constexpr (U == 8) r.q[0] = *reinterpret_cast<const uint4*>(p);
    else { const uint2 t = *reinterpret_cast<const uint2*>(p); r.q[0] = make_uint4(t.x, t.y, 0u, 0u); }
  } else {
    const float* p = reinterpret_cast<const float*>(base) + elem_off;
#pragma unroll
    for (int j = 0; j < U / 4; ++j) r.q[j] = *reinterpret_cast<const uint4*>(p + 4 * j);
  }
}
template <int U, bool IN_BF16>
__device__ __forceinline__ void unpack_unit(const RawUnit<U, IN_BF16>& r, float (&v)[U]) {
  if constexpr (IN_BF16) {
    const unsigned w[4] = {r.q[0].x, r.q[0].y, r.q[0].z, r.q[0].w};
#pragma unroll
    for (int j = 0; j < U / 2; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
  } else {
#pragma unroll
    for (int j = 0; j < U / 4; ++j) {
      v[4 * j] = __builtin_bit_cast(float, r.q[j].x); v[4 * j + 1] = __builtin_bit_cast(float, r.q[j].y);
      v[4 * j + 2] = __builtin_bit_cast(float, r.q[j].z); v[4 * j + 3] = __builtin_bit_cast(float, r.q[j].w);
    }
  }
}

template <int U>
__device__ __forceinline__ void lds_store_unit(u16* dst, const float (&v)[U]) {
  if constexpr (U == 8) {
    *reinterpret_cast<uint4*>(dst) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
  } else {
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
  }
}

// 8 consecutive bf16 of LDS as an MFMA fragment (16-byte aligned unless CC == 4: two 8-byte reads)
template <int CC>
__device__ __forceinline__ bf8 lds_frag(const u16* p) {
  if constexpr (CC == 4) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    const uint2 b = *reinterpret_cast<const uint2*>(p + 4);
    return __builtin_bit_cast(bf8, make_uint4(a.x, a.y, b.x, b.y));
  } else {
    return __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(p));
  }
}

// Epilogue of one workgroup tile: D[row = 4*kq + r][col = lm] of N tile nt = produced channel (see b_channel_of) of
// pixel lm.  Channel order inside a COB block (set by the packing): tile pair (2t, 2t+1), rows 4*kq..4*kq+3 of the
// even tile then of the odd tile = channels 32*t + 8*kq .. + 7, i.e. one 16-byte bf16 store per lane and pair.
template <int NT, int MT, bool OUT_BF16>
__device__ __forceinline__ void b_store_tile(const BArgs& a, const v4f (&acc)[MT][NT], int n, int py, int px, int qy0,
                                             int qx0, int qh, int qw, int co0, int wm, int wn, int lm, int kq) {
  const int64_t out_img = (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    const int qy = qy0 + tr, qx = qx0 + tc * 16 + lm;
    if (qy >= qh || qx >= qw) continue;
    const int Y = py + a.OS * qy, X = px + a.OS * qx;
    const int64_t o = out_img + ((int64_t)Y * a.out_w + X) * a.out_cs;
    if constexpr (NT == 1) {
      const int j0 = co0 + wn * 16 + kq * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mt][0][r] + ((a.bias && j0 + r < a.cout) ? a.bias[j0 + r] : 0.f);
      if constexpr (OUT_BF16) {
        u16* q = reinterpret_cast<u16*>(a.out) + o + j0;
        if (j0 + 3 < a.cout) *reinterpret_cast<uint2*>(q) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (j0 + r < a.cout) q[r] = f2bf(v[r]);
      } else {
        float* q = reinterpret_cast<float*>(a.out) + o + j0;
        if (j0 + 3 < a.cout && (a.out_cs & 3) == 0 && (a.out_co & 3) == 0) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (j0 + r < a.cout) q[r] = v[r];
      }
    } else {
#pragma unroll
      for (int np = 0; np < NT / 2; ++np) {
        const int j0 = co0 + (wn * NT / 2 + np) * 32 + kq * 8;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * np][r]; v[4 + r] = acc[mt][2 * np + 1][r]; }
        if (a.bias) {
#pragma unroll
          for (int r = 0; r < 8; ++r) if (j0 + r < a.cout) v[r] += a.bias[j0 + r];
        }
        if (j0 >= a.cout) continue;
        if constexpr (OUT_BF16) {
          u16* q = reinterpret_cast<u16*>(a.out) + o + j0;
          if (j0 + 7 < a.cout) *reinterpret_cast<uint4*>(q) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
          else
#pragma unroll
            for (int r = 0; r < 8; ++r) if (j0 + r < a.cout) q[r] = f2bf(v[r]);
        } else {
          float* q = reinterpret_cast<float*>(a.out) + o + j0;
          if (j0 + 7 < a.cout && (a.out_cs & 3) == 0 && (a.out_co & 3) == 0) {
            *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else
#pragma unroll
            for (int r = 0; r < 8; ++r) if (j0 + r < a.cout) q[r] = v[r];
        }
      }
    }
  }
}

// Training-mode batch-norm sums {sum y, sum y^2} of the tile just produced, from the accumulators (y = the value as
// STORED: rounded to bf16 where the output is bf16, so the statistics are those of the tensor the next layer reads).
// One N tile at a time: a lane's 4 channels over its MT pixels in double, the 16 lanes that share them (lm) by a
// halving exchange, the waves that share them (wm) through LDS; one row per (tile, image, phase), fixed order.
template <int NT, int MT, bool OUT_BF16, int WMN>
__device__ __forceinline__ void b_stats_tile(const BArgs& a, const v4f (&acc)[MT][NT], double* red, int64_t row, int qy0,
                                             int qx0, int qh, int qw, int co0, int wm, int wn, int lm, int kq, int COB) {
  const int tid = threadIdx.x;
  __syncthreads();                           // the LDS image is read out
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};     // <= 256 terms of 8-bit-mantissa values
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int t = wm * MT + mt;
      const int qy = qy0 + t / a.TPR, qx = qx0 + (t % a.TPR) * 16 + lm;
      if (qy >= qh || qx >= qw) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[mt][nt][r];
        if constexpr (OUT_BF16) v = bf2f(f2bf(v));
        s1[r] += v; s2[r] = fmaf(v, v, s2[r]);
      }
    }
    float h4[4], h2[2], h1;
    {
      const bool up = lm & 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float send = up ? s1[i] : s2[i], keep = up ? s2[i] : s1[i];
        h4[i] = keep + __shfl_xor(send, 8, 16);
      }
    }
    {
      const bool up = lm & 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float send = up ? h4[i] : h4[2 + i], keep = up ? h4[2 + i] : h4[i];
        h2[i] = keep + __shfl_xor(send, 4, 16);
      }
    }
    {
      const bool up = lm & 2;
      const float send = up ? h2[0] : h2[1], keep = up ? h2[1] : h2[0];
      h1 = keep + __shfl_xor(send, 2, 16);
    }
    h1 += __shfl_xor(h1, 1, 16);
    if ((lm & 1) == 0) {
      const int idx = lm >> 1, r = idx & 3;   // value index = 4 * s + r
      const int jl = NT == 1 ? wn * 16 + kq * 4 + r : (wn * NT / 2 + nt / 2) * 32 + kq * 8 + (nt & 1) * 4 + r;
      red[(wm * COB + jl) * 2 + (idx >> 2)] = (double)h1;
    }
  }
  __syncthreads();
  if (tid < COB && co0 + tid < a.cout) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < WMN; ++w) { t1 += red[(w * COB + tid) * 2]; t2 += red[(w * COB + tid) * 2 + 1]; }
    a.stat[(row * 2) * a.stat_c + co0 + tid] = t1;
    a.stat[(row * 2 + 1) * a.stat_c + co0 + tid] = t2;
  }
}

// NW waves per workgroup: 4, or 8 for the 128-channel trunk -- at bf16 MFMA speed a 128-pixel tile is only ~2 us of
// matrix work per 295 KB of weights, and streaming the weights from L2 once per tile is what bounds the kernel; eight
// waves (256 pixels) halve that traffic per pixel.
template <int CC, int NT, int WN, int MT, int SLOTS, bool IN_BF16, bool OUT_BF16, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void igemm_bf16_kernel(BArgs a) {
  constexpr int U = CC < 8 ? CC : 8;            // channels per staging unit
  constexpr int UPP = CC / U;                   // units per pixel
  constexpr int WM = NW / WN;
  constexpr int NTH = 64 * NW;
  constexpr int COB = 16 * NT * WN;
  // LDS images (bf16).  Input halo tile: CC = 32 as four k-group planes [channel octet][pixel][8] -- the 16 pixels
  // of an MFMA fragment read are then 256 contiguous bytes per lane quarter (the plain [pixel][32] image has a
  // 64-byte pixel stride: pixels p and p+4 share banks, measured 39 % conflict cycles); CC < 32: [pixel][CC], where a
  // fragment is 8 consecutive bf16 across x-adjacent pixels.  Weights: [run][k octet][row][8], two slabs deep.
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* lds_in = smem;
  const int in_elems = a.npixp * CC;
  const int w_off = (in_elems + 511) & ~511;          // 1 KiB aligned: LDS-DMA pieces
  u16* lds_w = smem + w_off;
  const int slab_t = a.nrun * COB * 32;               // bf16 of the slabs of one (chunk, tap row)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lm = lane & 15, kq = lane >> 4;

  int tile, by;
  b_tile_of_block(&tile, &by);
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int co0 = blockIdx.z * COB;
  const int ph = by % (a.nphase * a.nphase);
  const int n = by / (a.nphase * a.nphase);
  const int py = ph / a.nphase, px = ph % a.nphase;

  const int BW = 16 * a.TPR;
  const int qy0 = tile_y * a.BH, qx0 = tile_x * BW;
  const int qh = (a.out_h - py + a.OS - 1) / a.OS;
  const int qw = (a.out_w - px + a.OS - 1) / a.OS;
  const int64_t stat_row = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (qy0 >= qh || qx0 >= qw) {        // uniform per block
    if (a.stat && tid < COB && co0 + tid < a.cout) {
      a.stat[(stat_row * 2) * a.stat_c + co0 + tid] = 0.0;
      a.stat[(stat_row * 2 + 1) * a.stat_c + co0 + tid] = 0.0;
    }
    return;
  }

  int iy0, ix0;
  if (a.transposed) {
    iy0 = bp_t_i0(py, a.pad, a.stride, a.tapsy);
    ix0 = bp_t_i0(px, a.pad, a.stride, a.tapsy);
  } else {
    iy0 = -a.pad; ix0 = -a.pad;
  }
  const int gy0 = a.ISy * qy0 + iy0, gx0 = a.ISx * qx0 + ix0;

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    if constexpr (CC == 32) abase[mt] = (kq * a.npixp + tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * 8;
    else abase[mt] = (tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * CC + kq * 8;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (kq * COB + (wn * NT + nt) * 16 + lm) * 8;

  v4f acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};

  const int64_t in_img = (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
  const int E = a.IH * a.ISx * a.IWq * UPP;     // staging units of the LDS image

  // this thread's units: (row, column) are the same for every channel chunk
  const int cu = tid % UPP;                     // NTH % UPP == 0: fixed channel group per thread
  int s_g[SLOTS];                               // element offset inside the image, -1 outside / unused slot
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * NTH;
    s_g[i] = -2;
    if (e < E) {
      const int pi = e / UPP;
      const int xq = pi % a.IWq;
      const int t = pi / a.IWq;
      const int xm = t % a.ISx, r = t / a.ISx;
      const int iy = gy0 + r, ix = gx0 + xq * a.ISx + xm;
      s_g[i] = (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) ? (iy * a.in_w + ix) * a.in_cs + cu * U : -1;
    }
  }

  RawUnit<U, IN_BF16> stage[SLOTS];
  // Every slot loads (slots outside the image / unused fetch the image's first pixel and are zeroed when stored):
  // a load under a per-lane condition is followed by a full vmcnt(0) wait by the compiler, which serialises the
  // slots of a chunk -- one HBM round trip each -- instead of having them all in flight.
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
      load_unit_raw<U, IN_BF16>(a.in, in_img + (s_g[i] >= 0 ? s_g[i] : 0) + chunk * CC, stage[i]);
  };
  // the pending activation's parameters live in LDS ([scale | shift | slope][nchunk * CC], identity past cin): read
  // per chunk from global memory they are 3 * U dependent loads, each waited for, in front of every chunk
  float* lpw = reinterpret_cast<float*>(lds_w + 2 * slab_t);
  const int cpad = a.nchunk * CC;
  const bool on = a.pw.scale != nullptr;
  if (on) {
    for (int i = tid; i < cpad; i += NTH) {
      const bool ok = i < a.cin;
      lpw[i] = ok ? a.pw.scale[i] : 1.f; lpw[cpad + i] = ok ? a.pw.shift[i] : 0.f; lpw[2 * cpad + i] = ok ? a.pw.slope[i] : 1.f;
    }
  }
  auto store_chunk = [&](int chunk) {
    const int ch = chunk * CC + cu * U;
    float sc[U], sf[U], sl[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      sc[j] = on ? lpw[ch + j] : 1.f;
      sf[j] = on ? lpw[cpad + ch + j] : 0.f;
      sl[j] = on ? lpw[2 * cpad + ch + j] : 1.f;
    }
    bool relu = on && ch + U <= a.cin;
#pragma unroll
    for (int j = 0; j < U; ++j) relu = relu && sl[j] == 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      if (s_g[i] == -2) continue;
      if constexpr (IN_BF16 && U == 8) {
        if (!on && chunk * CC + cu * U + U <= a.cin) {
          // bf16 in, no pending activation (every data gradient): the raw words ARE the LDS image
          const int e = tid + i * NTH;
          u16* dst = CC == 32 ? lds_in + ((e % UPP) * a.npixp + e / UPP) * 8 : lds_in + e * U;
          *reinterpret_cast<uint4*>(dst) = s_g[i] >= 0 ? stage[i].q[0] : make_uint4(0u, 0u, 0u, 0u);
          continue;
        }
      }
      float v[U], raw[U];
      unpack_unit<U, IN_BF16>(stage[i], raw);
      if (relu) {        // batch-norm + ReLU (slopes 0, whole unit inside the tensor's channels): fma + compare + select per element,
                         // NaN-propagating
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = s_g[i] >= 0 ? bp_relu_nan(fmaf(raw[j], sc[j], sf[j])) : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
          float t = 0.f;
          if (s_g[i] >= 0 && ch + j < a.cin) {
            t = raw[j];
            if (on) { t = fmaf(t, sc[j], sf[j]); t = t > 0.f ? t : t * sl[j]; }
          }
          v[j] = t;
        }
      }
      const int e = tid + i * NTH;
      if constexpr (CC == 32) lds_store_unit<U>(lds_in + ((e % UPP) * a.npixp + e / UPP) * 8, v);
      else lds_store_unit<U>(lds_in + e * U, v);
    }
  };
  // weights of (phase, ty, all runs, chunk) by LDS-DMA into slab buffer `slot`: nrun x COB/16 pieces of 1 KiB
  // dealt to the four waves; the packed image holds each workgroup's [k octet][row][8] block contiguously
  auto issue_w = [&](int chunk, int ty, int slot) {
    constexpr int PPR = COB / 16;                    // pieces per run
    for (int k = wave; k < a.nrun * PPR; k += NW) {
      const int s = k / PPR, part = k - s * PPR;
      const u16* src = a.wp + ((((int64_t)(ph * a.tapsy + ty) * a.nrun + s) * a.nchunk + chunk) * a.cout_padP + co0) * 32 +
                       part * 512;
      bp_glds16(reinterpret_cast<const float*>(src), (unsigned)lane * 16u, (w_off + slot * slab_t + k * 512) / 2);
    }
  };

  // Pipeline: the slab of step q+1 is in flight (DMA, no registers) while step q is multiplied, the next channel
  // chunk of the input is in flight in registers; one barrier per tap row plus one per chunk.
  int slot = 0;
  issue_w(0, 0, 0);
  load_chunk(0);
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();                 // the previous chunk's readers are done with lds_in
    store_chunk(chunk);
    for (int ty = 0; ty < a.tapsy; ++ty) {
      bp_wait_dma_barrier();         // this step's slab has landed (and ty == 0: the chunk is stored); the previous
      //                                step's readers are done with the other slab buffer
      {
        int nch = chunk, nty = ty + 1;
        if (nty == a.tapsy) { nty = 0; ++nch; }
        if (nch < a.nchunk) issue_w(nch, nty, slot ^ 1);
      }
      if (ty == 0 && chunk + 1 < a.nchunk) load_chunk(chunk + 1);      // in flight while this chunk is computed
      const u16* lw = lds_w + slot * slab_t;
      slot ^= 1;
      for (int s = 0; s < a.nrun; ++s) {
        const int tapoff = (ty * a.ISx * a.IWq + a.run_off[s]) * (CC == 32 ? 8 : CC);
        bf8 xf[MT], wf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = lds_frag<CC>(lds_in + abase[mt] + tapoff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<32>(lw + s * COB * 32 + bbase[nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  b_store_tile<NT, MT, OUT_BF16>(a, acc, n, py, px, qy0, qx0, qh, qw, co0, wm, wn, lm, kq);
  if (a.stat)
    b_stats_tile<NT, MT, OUT_BF16, WM>(a, acc, reinterpret_cast<double*>(smem), stat_row, qy0, qx0, qh, qw, co0, wm, wn, lm,
                                      kq, COB);
}

// ------------------------------------------------------------------------------------------------
// Persistent form for layers with ONE channel chunk (cin <= 32) whose whole weight image fits in LDS -- the
// full-resolution few-channel layers, where a 256-pixel tile is only a few hundred MFMA cycles of work and the
// per-tap-row weight staging (two barriers per row) of the kernel above is most of the time:
//   * the weights of all tap rows are staged ONCE per workgroup (per phase and channel block);
//   * a workgroup walks a contiguous range of (image, tile) pairs (neighbouring tiles share halo rows / columns:
//     they meet in one XCD's L2) with the NEXT tile's halo in flight in registers while the current one is
//     multiplied: two barriers per tile, no exposed global latency.
struct BPArgs {
  BArgs b;
  int ntiles_total;     // images x tiles
  int per_block;        // tiles per workgroup (contiguous)
};

// FUSE (transposed forms with stride^2 = 4 output phases): one workgroup computes all four phases of a tile from ONE
// staged input tile (the union of the phases' halos) with all four phases' weights resident -- the halo is fetched,
// activated and written to LDS once instead of four times, and the four phases' pixels of an output row are written
// by the same workgroup back to back.
template <int CC, int NT, int WN, int MT, int SLOTS, bool IN_BF16, bool OUT_BF16, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void igemm_bf16_p_kernel(BPArgs pa) {
  const BArgs& a = pa.b;
  constexpr int U = CC < 8 ? CC : 8;
  constexpr int UPP = CC / U;
  constexpr int WM = 4 / WN;
  constexpr int COB = 16 * NT * WN;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* lds_in = smem;
  const int in_elems = a.npixp * CC;
  u16* lds_w = smem + ((in_elems + 511) & ~511);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lm = lane & 15, kq = lane >> 4;

  constexpr int NPH = FUSE ? 4 : 1;
  const int ph0 = FUSE ? 0 : blockIdx.y;
  const int co0 = blockIdx.z * COB;
  const int t_begin = blockIdx.x * pa.per_block;
  int t_end = t_begin + pa.per_block;
  if (t_end > pa.ntiles_total) t_end = pa.ntiles_total;
  if (t_begin >= t_end) return;          // uniform per block
  const int tiles_per_img = a.tiles_x * a.tiles_y;

  // origin of the staged tile: the first gathered row / column of phase ph0 (fused: phase 0, the smallest origin)
  int iy0, ix0;
  if (a.transposed) {
    iy0 = bp_t_i0(ph0 / a.nphase, a.pad, a.stride, a.tapsy);
    ix0 = bp_t_i0(ph0 % a.nphase, a.pad, a.stride, a.tapsy);
  } else {
    iy0 = -a.pad; ix0 = -a.pad;
  }

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    if constexpr (CC == 32) abase[mt] = (kq * a.npixp + tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * 8;
    else abase[mt] = (tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * CC + kq * 8;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (kq * COB + (wn * NT + nt) * 16 + lm) * 8;

  // this thread's staging units: (row, column) inside the halo image are the same for every tile
  const int E = a.IH * a.ISx * a.IWq * UPP;
  const int cu = tid % UPP;
  int s_r[SLOTS], s_c[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256;
    s_r[i] = -1; s_c[i] = 0;
    if (e < E) {
      const int pi = e / UPP;
      const int xq = pi % a.IWq;
      const int t = pi / a.IWq;
      s_r[i] = t / a.ISx;
      s_c[i] = xq * a.ISx + t % a.ISx;
    }
  }
  // pending activation of this thread's channel group
  float sc[U], sf[U], sl[U];
  const bool on = a.pw.scale != nullptr;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const bool ok = on && cu * U + j < a.cin;
    sc[j] = ok ? a.pw.scale[cu * U + j] : 1.f;
    sf[j] = ok ? a.pw.shift[cu * U + j] : 0.f;
    sl[j] = ok ? a.pw.slope[cu * U + j] : 1.f;
  }

  RawUnit<U, IN_BF16> stage[SLOTS];
  unsigned inside = 0;
  auto tile_coords = [&](int t, int* n, int* qy0, int* qx0) {
    *n = t / tiles_per_img;
    const int r = t - *n * tiles_per_img;
    *qy0 = (r / a.tiles_x) * a.BH;
    *qx0 = (r % a.tiles_x) * 16 * a.TPR;
  };
  auto load_tile = [&](int t) {
    int n, qy0, qx0;
    tile_coords(t, &n, &qy0, &qx0);
    const int gy0 = a.ISy * qy0 + iy0, gx0 = a.ISx * qx0 + ix0;
    const int64_t img = (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co + cu * U;
    // every slot loads, coordinates clamped into the image (see igemm_bf16_kernel::load_chunk)
    inside = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int iy = gy0 + s_r[i], ix = gx0 + s_c[i];
      if (s_r[i] >= 0 && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) inside |= 1u << i;
      const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
      load_unit_raw<U, IN_BF16>(a.in, img + ((int64_t)cy * a.in_w + cx) * a.in_cs, stage[i]);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      if (s_r[i] < 0) continue;
      if constexpr (IN_BF16 && U == 8) {
        if (!on && cu * U + U <= a.cin) {      // bf16 in, no pending activation: the raw words ARE the LDS image
          const int e = tid + i * 256;
          u16* dst = CC == 32 ? lds_in + ((e % UPP) * a.npixp + e / UPP) * 8 : lds_in + e * U;
          *reinterpret_cast<uint4*>(dst) = ((inside >> i) & 1u) ? stage[i].q[0] : make_uint4(0u, 0u, 0u, 0u);
          continue;
        }
      }
      float v[U], raw[U];
      unpack_unit<U, IN_BF16>(stage[i], raw);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        float t = 0.f;
        if (((inside >> i) & 1u) && cu * U + j < a.cin) {
          t = raw[j];
          if (on) { t = fmaf(t, sc[j], sf[j]); t = t > 0.f ? t : t * sl[j]; }
        }
        v[j] = t;
      }
      const int e = tid + i * 256;
      if constexpr (CC == 32) lds_store_unit<U>(lds_in + ((e % UPP) * a.npixp + e / UPP) * 8, v);
      else lds_store_unit<U>(lds_in + e * U, v);
    }
  };

  // weights of every tap row of this phase and channel block: [ty][run][k octet][COB][8]
  {
    constexpr int slab8 = COB * 32 / 8;
    const int total = NPH * a.tapsy * a.nrun * slab8;
    for (int e = tid; e < total; e += 256) {
      const int sl_ = e / slab8, o = e % slab8;          // sl_ = (phase * tapsy + ty) * nrun + s
      const u16* src = a.wp + (((int64_t)ph0 * a.tapsy * a.nrun + sl_) * a.cout_padP + co0) * 32;
      *reinterpret_cast<uint4*>(lds_w + (size_t)sl_ * COB * 32 + o * 8) = *reinterpret_cast<const uint4*>(src + o * 8);
    }
  }

  load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();                 // the previous tile's readers are done with lds_in
    store_tile();
    __syncthreads();
    if (t + 1 < t_end) load_tile(t + 1);      // in flight while this tile is multiplied

    int n, qy0, qx0;
    tile_coords(t, &n, &qy0, &qx0);
#pragma unroll 1
    for (int p = 0; p < NPH; ++p) {
      const int ph = FUSE ? p : ph0;
      const int py = ph / a.nphase, px = ph % a.nphase;
      const int qh = (a.out_h - py + a.OS - 1) / a.OS;
      const int qw = (a.out_w - px + a.OS - 1) / a.OS;
      // this phase's first gathered row / column relative to the tile origin
      const int dy = FUSE ? bp_t_i0(py, a.pad, a.stride, a.tapsy) - iy0 : 0;
      const int dx = FUSE ? bp_t_i0(px, a.pad, a.stride, a.tapsy) - ix0 : 0;
      v4f acc[MT][NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};
      for (int ty = 0; ty < a.tapsy; ++ty) {
        for (int s = 0; s < a.nrun; ++s) {
          const int tapoff = ((ty + dy) * a.ISx * a.IWq + a.run_off[s] + dx) * (CC == 32 ? 8 : CC);
          const u16* lw = lds_w + ((p * a.tapsy + ty) * a.nrun + s) * COB * 32;
          bf8 xf[MT], wf[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) xf[mt] = lds_frag<CC>(lds_in + abase[mt] + tapoff);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<32>(lw + bbase[nt]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
        }
      }
      b_store_tile<NT, MT, OUT_BF16>(a, acc, n, py, px, qy0, qx0, qh, qw, co0, wm, wn, lm, kq);
      if (a.stat)
        b_stats_tile<NT, MT, OUT_BF16, WM>(a, acc, reinterpret_cast<double*>(lds_w + NPH * a.tapsy * a.nrun * COB * 32),
                                          (int64_t)ph * pa.ntiles_total + t, qy0, qx0, qh, qw, co0, wm, wn, lm, kq, COB);
    }
  }
}

// produced channel <-> (N tile, MFMA row) inside a COB block; NT == 1 blocks keep the natural order
__host__ __device__ __forceinline__ int b_channel_of(int NT, int jb /* index inside the COB block: tile*16 + row */) {
  if (NT == 1) return jb;
  const int tile = jb >> 4, row = jb & 15;
  return 32 * (tile >> 1) + 8 * (row >> 2) + 4 * (tile & 1) + (row & 3);
}

struct BConfig {
  int CC, R, NT, WN, MT, COB, nchunk, cout_padP, nrun, run_xm[16], run_xq[16];
  int TPR, BH, IH, IWq, slots, npixp;
  int NW;                   // waves per workgroup (4, or 8: igemm_bf16_kernel<..., 8>)
  size_t lds_bytes;
  bool ok;
  bool persistent;          // igemm_bf16_p_kernel: one chunk, all tap rows' weights resident
  bool fuse;                // ... all four phases of a stride-2 transposed form in one workgroup
  size_t lds_p;
};

static inline BConfig b_config_for(const ConvGeom& g, int NT, int WN, int NW = 4) {
  BConfig c{};
  c.NW = NW;
  const int cin = g.cin_g;
  if (cin % 32 == 0) c.CC = 32;
  else if (cin == 16) c.CC = 16;
  else if (cin == 8) c.CC = 8;
  else if (cin <= 4) c.CC = 4;
  else return c;
  c.R = 32 / c.CC;
  c.nchunk = c.CC == 32 ? cin / 32 : 1;
  c.NT = NT; c.WN = WN;
  static const bool no_mt8 = getenv("BP_BF16_NOMT8") != nullptr;
  c.MT = (NW == 8 && !no_mt8) ? 8 : 4;          // eight waves: 128 pixels x 64 channels per wave (12 fragment reads per 32 MFMAs)
  c.COB = 16 * c.NT * c.WN;
  c.cout_padP = bp_round_up(g.cout_g, c.COB);
  // K-steps of one tap row: per parity plane xm (x % IS), runs of R plane-adjacent taps
  int rmax = 0;
  for (int xm = 0; xm < g.IS && xm < g.taps; ++xm) {
    const int tp = bp_ceil_div(g.taps - xm, g.IS);         // taps of this plane
    const int nr = bp_ceil_div(tp, c.R);
    for (int r = 0; r < nr; ++r) {
      if (c.nrun >= 16) return c;
      c.run_xm[c.nrun] = xm; c.run_xq[c.nrun] = r * c.R; ++c.nrun;
    }
    if (nr * c.R > rmax) rmax = nr * c.R;
  }
  const int TM = (NW / c.WN) * c.MT;                       // M tiles per workgroup
  c.TPR = 2; c.BH = TM / 2;
  c.IH = (c.BH - 1) * g.IS + g.taps;
  c.IWq = 16 * c.TPR + rmax - 1;
  const int U = c.CC < 8 ? c.CC : 8;
  const int E = c.IH * g.IS * c.IWq * (c.CC / U);
  c.slots = bp_ceil_div(E, 64 * NW);
  c.npixp = bp_round_up(c.IH * g.IS * c.IWq, 16);
  const size_t in_b = (((size_t)c.npixp * c.CC + 511) & ~(size_t)511) * 2;
  c.lds_bytes = in_b + (size_t)2 * c.nrun * c.COB * 32 * 2          // two weight slabs
                + (size_t)3 * c.nchunk * c.CC * sizeof(float);       // + the pending activation's parameters
  c.ok = c.lds_bytes <= (size_t)(NW == 8 ? 100 : 80) * 1024 && c.slots <= 12;       // two workgroups per CU (one of eight waves)
  constexpr size_t RED = 4 * 128 * 2 * sizeof(double);              // cross-wave fold of the epilogue statistics
  c.lds_p = in_b + (size_t)g.taps * c.nrun * c.COB * 32 * 2 + RED;
  static const bool no_p = getenv("BP_BF16_NOPERSIST") != nullptr;
  static const bool no_fuse = getenv("BP_BF16_NOFUSE") != nullptr;
  // (measured on the fiducial layers: the persistent form wins for the strided gathers -- 16->32 k4s2 forward
  //  0.61 -> 0.28 ms, 32->16 transposed data gradient 0.54 -> 0.27 ms -- whose halo tiles are four times the
  //  output tile, and for the four-phase transposed forms -- 32->16 forward 0.71 -> 0.61 ms; the unit-stride k7 head
  //  is the same either way)
  static const bool p_all = getenv("BP_BF16_PALL") != nullptr;
  c.persistent = c.ok && !no_p && NW == 4 && c.nchunk == 1 && c.lds_p <= 64 * 1024 && (g.IS == 2 || g.nphase > 1 || p_all);
  if (c.persistent && !no_fuse && g.gather_transposed && g.nphase == 2) {
    // all four phases from one staged tile: the union of their halos is `spread` rows / columns larger
    const int spread = bp_t_i0(g.nphase - 1, g.pad, g.stride, g.taps) - bp_t_i0(0, g.pad, g.stride, g.taps);
    BConfig f = c;
    f.IH += spread; f.IWq += spread;
    const int Ef = f.IH * g.IS * f.IWq * (c.CC / U);
    f.slots = bp_ceil_div(Ef, 256);
    f.npixp = bp_round_up(f.IH * g.IS * f.IWq, 16);
    const size_t in_f = (((size_t)f.npixp * c.CC + 511) & ~(size_t)511) * 2;
    f.lds_p = in_f + (size_t)4 * g.taps * c.nrun * c.COB * 32 * 2 + RED;
    if (f.lds_p <= 64 * 1024 && f.slots <= 12) { f.fuse = true; c = f; }
  }
  return c;
}

// Waves split the pixels of a 256-pixel tile (WN = 1) unless the produced-channel block is 128 wide or the halo
// of such a tile does not fit (strided gathers of 32-channel chunks): then two waves share each half tile.
inline BConfig b_config(const ConvGeom& g) {
  const int nT = bp_ceil_div(g.cout_g, 16);
  static const int cand[6][2] = {{4, 2}, {4, 1}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const int first = nT >= 5 ? 0 : (nT >= 3 ? 1 : (nT == 2 ? 3 : 5));
  BConfig c{};
  static const bool no_nw8 = getenv("BP_BF16_NONW8") != nullptr;
  if (first == 0 && g.cin_g % 32 == 0 && !no_nw8) {          // 128-wide channel block, 32-channel chunks: eight waves
    c = b_config_for(g, 4, 2, 8);
    if (c.ok && c.slots <= 6) return c;
  }
  for (int i = first; i < 6; ++i) {            // widest channel block whose tile + two weight slabs fit
    if (16 * cand[i][0] * cand[i][1] > 16 * nT && i != first) continue;      // (never wider than the layer)
    c = b_config_for(g, cand[i][0], cand[i][1]);
    if (c.ok) break;
  }
  return c;
}

// weights: torch layout (fp32) -> [phase][ty][run][chunk][channel block][k octet][row][8] bf16 (the LDS image of a
// workgroup's slab, contiguous: one linear copy / LDS-DMA), k = 8*octet + i = j*CC + cc <-> tap xm + IS*(xq + j)
template <int CC, int NT, int WN, int SLOTS, bool IB, bool OB, int NW = 4, int MT = 4>
int b_launch(const BArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&igemm_bf16_kernel<CC, NT, WN, MT, SLOTS, IB, OB, NW>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (NW == 8 ? 100 : 80) * 1024);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((igemm_bf16_kernel<CC, NT, WN, MT, SLOTS, IB, OB, NW>), grid, dim3(64 * NW), lds, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int NT, int WN, int SLOTS, bool IB, bool OB, bool FUSE>
int b_launch_p(const BPArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((igemm_bf16_p_kernel<CC, NT, WN, 4, SLOTS, IB, OB, FUSE>), grid, dim3(256), lds, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int NT, int WN, int SLOTS, bool FUSE = false>
int b_launch_p_io(const BPArgs& a, bool ib, bool ob, dim3 grid, size_t lds, hipStream_t st) {
  if (ib && ob) return b_launch_p<CC, NT, WN, SLOTS, true, true, FUSE>(a, grid, lds, st);
  if (ib) return b_launch_p<CC, NT, WN, SLOTS, true, false, FUSE>(a, grid, lds, st);
  if (ob) return b_launch_p<CC, NT, WN, SLOTS, false, true, FUSE>(a, grid, lds, st);
  return b_launch_p<CC, NT, WN, SLOTS, false, false, FUSE>(a, grid, lds, st);
}

template <int CC, int NT, int WN, int SLOTS, int NW = 4, int MT = 4>
int b_launch_io(const BArgs& a, bool ib, bool ob, dim3 grid, size_t lds, hipStream_t st) {
  if (ib && ob) return b_launch<CC, NT, WN, SLOTS, true, true, NW, MT>(a, grid, lds, st);
  if (ib) return b_launch<CC, NT, WN, SLOTS, true, false, NW, MT>(a, grid, lds, st);
  if (ob) return b_launch<CC, NT, WN, SLOTS, false, true, NW, MT>(a, grid, lds, st);
  return b_launch<CC, NT, WN, SLOTS, false, false, NW, MT>(a, grid, lds, st);
}

template <int CC, int NT, int WN>
int b_launch_slots(const BConfig& c, const BArgs& a, bool ib, bool ob, dim3 grid, hipStream_t st) {
  if constexpr (NT <= 2) {          // (persistent form: the layers with <= 32 produced channels per block)
    if (c.persistent) {
      BPArgs pa{};
      pa.b = a;
      const int nimg = (int)grid.y / (a.nphase * a.nphase);
      pa.ntiles_total = nimg * a.tiles_x * a.tiles_y;
      // ~2 resident workgroups per CU and phase/channel block, each with a contiguous run of >= 4 tiles
      int nb = 512 / ((int)grid.z * a.nphase * a.nphase);
      if (nb < 64) nb = 64;
      int per = bp_ceil_div(pa.ntiles_total, nb);
      if (per < 4) per = 4;
      pa.per_block = per;
      if (c.fuse) {          // every workgroup does all four phases of its tiles
        nb = 512 / (int)grid.z;
        per = bp_ceil_div(pa.ntiles_total, nb);
        if (per < 2) per = 2;
        pa.per_block = per;
        dim3 fg((unsigned)bp_ceil_div(pa.ntiles_total, per), 1, grid.z);
        if (c.slots <= 6) return b_launch_p_io<CC, NT, WN, 6, true>(pa, ib, ob, fg, c.lds_p, st);
        return b_launch_p_io<CC, NT, WN, 12, true>(pa, ib, ob, fg, c.lds_p, st);
      }
      dim3 pg((unsigned)bp_ceil_div(pa.ntiles_total, per), (unsigned)(a.nphase * a.nphase), grid.z);
      if (c.slots <= 6) return b_launch_p_io<CC, NT, WN, 6>(pa, ib, ob, pg, c.lds_p, st);
      return b_launch_p_io<CC, NT, WN, 12>(pa, ib, ob, pg, c.lds_p, st);
    }
  }
  if (c.slots <= 6) return b_launch_io<CC, NT, WN, 6>(a, ib, ob, grid, c.lds_bytes, st);
  return b_launch_io<CC, NT, WN, 12>(a, ib, ob, grid, c.lds_bytes, st);
}

template <int CC>
int b_launch_cc(const BConfig& c, const BArgs& a, bool ib, bool ob, dim3 grid, hipStream_t st) {
  if constexpr (CC == 32) {
    if (c.NW == 8 && c.MT == 8) return b_launch_io<CC, 4, 2, 6, 8, 8>(a, ib, ob, grid, c.lds_bytes, st);
    if (c.NW == 8) return b_launch_io<CC, 4, 2, 6, 8>(a, ib, ob, grid, c.lds_bytes, st);
  }
  if (c.NT == 4 && c.WN == 2) return b_launch_slots<CC, 4, 2>(c, a, ib, ob, grid, st);
  if (c.NT == 4 && c.WN == 1) return b_launch_slots<CC, 4, 1>(c, a, ib, ob, grid, st);
  if (c.NT == 2 && c.WN == 2) return b_launch_slots<CC, 2, 2>(c, a, ib, ob, grid, st);
  if (c.NT == 2) return b_launch_slots<CC, 2, 1>(c, a, ib, ob, grid, st);
  if (c.NT == 1 && c.WN == 2) return b_launch_slots<CC, 1, 2>(c, a, ib, ob, grid, st);
  return b_launch_slots<CC, 1, 1>(c, a, ib, ob, grid, st);
}

}  // namespace bpbf16
