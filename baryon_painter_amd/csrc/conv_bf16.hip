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
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }   // RNE, NaN stays NaN
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

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

template <int CC, int NT, int WN, int MT, int SLOTS, bool IN_BF16, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(BArgs a) {
  constexpr int U = CC < 8 ? CC : 8;            // channels per staging unit
  constexpr int UPP = CC / U;                   // units per pixel
  constexpr int WM = 4 / WN;
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
  if (qy0 >= qh || qx0 >= qw) return;  // uniform per block

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
  const int cu = tid % UPP;                     // 256 % UPP == 0: fixed channel group per thread
  int s_g[SLOTS];                               // element offset inside the image, -1 outside / unused slot
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256;
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

  float stage[SLOTS][U];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
      if (s_g[i] >= 0) load_unit<U, IN_BF16>(a.in, in_img + s_g[i] + chunk * CC, stage[i]);
  };
  auto store_chunk = [&](int chunk) {
    const int ch = chunk * CC + cu * U;
    float sc[U], sf[U], sl[U];
    const bool on = a.pw.scale != nullptr;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const bool ok = on && ch + j < a.cin;
      sc[j] = ok ? a.pw.scale[ch + j] : 1.f;
      sf[j] = ok ? a.pw.shift[ch + j] : 0.f;
      sl[j] = ok ? a.pw.slope[ch + j] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      if (s_g[i] == -2) continue;
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        float t = 0.f;
        if (s_g[i] >= 0 && ch + j < a.cin) {
          t = stage[i][j];
          if (on) { t = fmaf(t, sc[j], sf[j]); t = t > 0.f ? t : t * sl[j]; }
        }
        v[j] = t;
      }
      const int e = tid + i * 256;
      if constexpr (CC == 32) lds_store_unit<U>(lds_in + ((e % UPP) * a.npixp + e / UPP) * 8, v);
      else lds_store_unit<U>(lds_in + e * U, v);
    }
  };
  // weights of (phase, ty, all runs, chunk) by LDS-DMA into slab buffer `slot`: nrun x COB/16 pieces of 1 KiB
  // dealt to the four waves; the packed image holds each workgroup's [k octet][row][8] block contiguously
  auto issue_w = [&](int chunk, int ty, int slot) {
    constexpr int PPR = COB / 16;                    // pieces per run
    for (int k = wave; k < a.nrun * PPR; k += 4) {
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

template <int CC, int NT, int WN, int MT, int SLOTS, bool IN_BF16, bool OUT_BF16>
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
  (void)WM;

  const int ph = blockIdx.y;
  const int py = ph / a.nphase, px = ph % a.nphase;
  const int co0 = blockIdx.z * COB;
  const int t_begin = blockIdx.x * pa.per_block;
  int t_end = t_begin + pa.per_block;
  if (t_end > pa.ntiles_total) t_end = pa.ntiles_total;
  if (t_begin >= t_end) return;          // uniform per block
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int qh = (a.out_h - py + a.OS - 1) / a.OS;
  const int qw = (a.out_w - px + a.OS - 1) / a.OS;

  int iy0, ix0;
  if (a.transposed) {
    iy0 = bp_t_i0(py, a.pad, a.stride, a.tapsy);
    ix0 = bp_t_i0(px, a.pad, a.stride, a.tapsy);
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

  float stage[SLOTS][U];
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
    inside = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int iy = gy0 + s_r[i], ix = gx0 + s_c[i];
      if (s_r[i] >= 0 && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) {
        inside |= 1u << i;
        load_unit<U, IN_BF16>(a.in, img + ((int64_t)iy * a.in_w + ix) * a.in_cs, stage[i]);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      if (s_r[i] < 0) continue;
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        float t = 0.f;
        if (((inside >> i) & 1u) && cu * U + j < a.cin) {
          t = stage[i][j];
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
    const int total = a.tapsy * a.nrun * slab8;
    for (int e = tid; e < total; e += 256) {
      const int sl_ = e / slab8, o = e % slab8;          // sl_ = ty * nrun + s
      const u16* src = a.wp + (((int64_t)ph * a.tapsy * a.nrun + sl_) * a.cout_padP + co0) * 32;
      *reinterpret_cast<uint4*>(lds_w + (size_t)sl_ * COB * 32 + o * 8) = *reinterpret_cast<const uint4*>(src + o * 8);
    }
  }

  load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();                 // the previous tile's readers are done with lds_in
    store_tile();
    __syncthreads();
    if (t + 1 < t_end) load_tile(t + 1);      // in flight while this tile is multiplied

    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int ty = 0; ty < a.tapsy; ++ty) {
      for (int s = 0; s < a.nrun; ++s) {
        const int tapoff = (ty * a.ISx * a.IWq + a.run_off[s]) * (CC == 32 ? 8 : CC);
        const u16* lw = lds_w + (ty * a.nrun + s) * COB * 32;
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
    int n, qy0, qx0;
    tile_coords(t, &n, &qy0, &qx0);
    b_store_tile<NT, MT, OUT_BF16>(a, acc, n, py, px, qy0, qx0, qh, qw, co0, wm, wn, lm, kq);
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
  size_t lds_bytes;
  bool ok;
  bool persistent;          // igemm_bf16_p_kernel: one chunk, all tap rows' weights resident
  size_t lds_p;
};

static BConfig b_config_for(const ConvGeom& g, int NT, int WN) {
  BConfig c{};
  const int cin = g.cin_g;
  if (cin % 32 == 0) c.CC = 32;
  else if (cin == 16) c.CC = 16;
  else if (cin == 8) c.CC = 8;
  else if (cin <= 4) c.CC = 4;
  else return c;
  c.R = 32 / c.CC;
  c.nchunk = c.CC == 32 ? cin / 32 : 1;
  c.NT = NT; c.WN = WN;
  c.MT = 4;
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
  const int TM = (4 / c.WN) * c.MT;                        // M tiles per workgroup
  c.TPR = 2; c.BH = TM / 2;
  c.IH = (c.BH - 1) * g.IS + g.taps;
  c.IWq = 16 * c.TPR + rmax - 1;
  const int U = c.CC < 8 ? c.CC : 8;
  const int E = c.IH * g.IS * c.IWq * (c.CC / U);
  c.slots = bp_ceil_div(E, 256);
  c.npixp = bp_round_up(c.IH * g.IS * c.IWq, 16);
  const size_t in_b = (((size_t)c.npixp * c.CC + 511) & ~(size_t)511) * 2;
  c.lds_bytes = in_b + (size_t)2 * c.nrun * c.COB * 32 * 2;        // two weight slabs
  c.ok = c.lds_bytes <= 80 * 1024 && c.slots <= 12;       // two workgroups per CU
  c.lds_p = in_b + (size_t)g.taps * c.nrun * c.COB * 32 * 2;
  static const bool no_p = getenv("BP_BF16_NOPERSIST") != nullptr;
  // (measured on the fiducial layers: the persistent form wins for the strided gathers -- 16->32 k4s2 forward
  //  0.61 -> 0.46 ms, 32->16 transposed data gradient 0.54 -> 0.41 ms -- whose halo tiles are four times the
  //  output tile, and loses 10-20 % on the unit-stride forms, which are bound by LDS fragment reads, not by staging)
  c.persistent = c.ok && !no_p && c.nchunk == 1 && c.lds_p <= 64 * 1024 && g.IS == 2;
  return c;
}

// Waves split the pixels of a 256-pixel tile (WN = 1) unless the produced-channel block is 128 wide or the halo
// of such a tile does not fit (strided gathers of 32-channel chunks): then two waves share each half tile.
BConfig b_config(const ConvGeom& g) {
  const int nT = bp_ceil_div(g.cout_g, 16);
  static const int cand[6][2] = {{4, 2}, {4, 1}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const int first = nT >= 5 ? 0 : (nT >= 3 ? 1 : (nT == 2 ? 3 : 5));
  BConfig c{};
  for (int i = first; i < 6; ++i) {            // widest channel block whose tile + two weight slabs fit
    if (16 * cand[i][0] * cand[i][1] > 16 * nT && i != first) continue;      // (never wider than the layer)
    c = b_config_for(g, cand[i][0], cand[i][1]);
    if (c.ok) break;
  }
  return c;
}

// weights: torch layout (fp32) -> [phase][ty][run][chunk][channel block][k octet][row][8] bf16 (the LDS image of a
// workgroup's slab, contiguous: one linear copy / LDS-DMA), k = 8*octet + i = j*CC + cc <-> tap xm + IS*(xq + j)
struct BPackArgs {
  const float* w; u16* dst;
  int64_t sa, sb;
  int k, stride, pad, tapsy, nphase, transposed, IS;
  int cin_g, cout_g, CC, nchunk, cout_padP, COB, NT, nrun;
  int run_xm[16], run_xq[16];
  int64_t total;
};

__global__ __launch_bounds__(256) void pack_bf16_kernel(BPackArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.total) return;
  int64_t r = i;
  const int k8 = r % 8; r /= 8;
  const int row = r % a.COB; r /= a.COB;
  const int kq = r % 4; r /= 4;
  const int cb = r % (a.cout_padP / a.COB); r /= (a.cout_padP / a.COB);
  const int kk = kq * 8 + k8;
  const int jb_abs = cb * a.COB + row;
  const int chunk = r % a.nchunk; r /= a.nchunk;
  const int s = r % a.nrun; r /= a.nrun;
  const int ty = r % a.tapsy; r /= a.tapsy;
  const int ph = (int)r;
  const int py = ph / a.nphase, px = ph % a.nphase;
  const int blk = jb_abs / a.COB, jb = jb_abs % a.COB;
  const int co = blk * a.COB + b_channel_of(a.NT, jb);
  const int j = kk / a.CC, cc = kk % a.CC;
  const int tx = a.run_xm[s] + a.IS * (a.run_xq[s] + j);
  const int ci = chunk * a.CC + cc;
  int ky, kx;
  if (a.transposed) {
    ky = bp_t_ky(py, a.pad, a.stride, a.tapsy, ty);
    kx = tx < a.tapsy ? bp_t_ky(px, a.pad, a.stride, a.tapsy, tx) : -1;
  } else {
    ky = ty; kx = tx < a.tapsy ? tx : -1;
  }
  float v = 0.f;
  if (ci < a.cin_g && co < a.cout_g && ky < a.k && kx >= 0 && kx < a.k) v = a.w[ci * a.sa + co * a.sb + ky * a.k + kx];
  a.dst[i] = f2bf(v);
}

template <int CC, int NT, int WN, int SLOTS, bool IB, bool OB>
int b_launch(const BArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&igemm_bf16_kernel<CC, NT, WN, 4, SLOTS, IB, OB>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((igemm_bf16_kernel<CC, NT, WN, 4, SLOTS, IB, OB>), grid, dim3(256), lds, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int NT, int WN, int SLOTS, bool IB, bool OB>
int b_launch_p(const BPArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((igemm_bf16_p_kernel<CC, NT, WN, 4, SLOTS, IB, OB>), grid, dim3(256), lds, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int NT, int WN, int SLOTS>
int b_launch_p_io(const BPArgs& a, bool ib, bool ob, dim3 grid, size_t lds, hipStream_t st) {
  if (ib && ob) return b_launch_p<CC, NT, WN, SLOTS, true, true>(a, grid, lds, st);
  if (ib) return b_launch_p<CC, NT, WN, SLOTS, true, false>(a, grid, lds, st);
  if (ob) return b_launch_p<CC, NT, WN, SLOTS, false, true>(a, grid, lds, st);
  return b_launch_p<CC, NT, WN, SLOTS, false, false>(a, grid, lds, st);
}

template <int CC, int NT, int WN, int SLOTS>
int b_launch_io(const BArgs& a, bool ib, bool ob, dim3 grid, size_t lds, hipStream_t st) {
  if (ib && ob) return b_launch<CC, NT, WN, SLOTS, true, true>(a, grid, lds, st);
  if (ib) return b_launch<CC, NT, WN, SLOTS, true, false>(a, grid, lds, st);
  if (ob) return b_launch<CC, NT, WN, SLOTS, false, true>(a, grid, lds, st);
  return b_launch<CC, NT, WN, SLOTS, false, false>(a, grid, lds, st);
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
      dim3 pg((unsigned)bp_ceil_div(pa.ntiles_total, per), (unsigned)(a.nphase * a.nphase), grid.z);
      if (c.slots <= 3) return b_launch_p_io<CC, NT, WN, 3>(pa, ib, ob, pg, c.lds_p, st);
      if (c.slots <= 6) return b_launch_p_io<CC, NT, WN, 6>(pa, ib, ob, pg, c.lds_p, st);
      return b_launch_p_io<CC, NT, WN, 12>(pa, ib, ob, pg, c.lds_p, st);
    }
  }
  if (c.slots <= 3) return b_launch_io<CC, NT, WN, 3>(a, ib, ob, grid, c.lds_bytes, st);
  if (c.slots <= 6) return b_launch_io<CC, NT, WN, 6>(a, ib, ob, grid, c.lds_bytes, st);
  return b_launch_io<CC, NT, WN, 12>(a, ib, ob, grid, c.lds_bytes, st);
}

template <int CC>
int b_launch_cc(const BConfig& c, const BArgs& a, bool ib, bool ob, dim3 grid, hipStream_t st) {
  if (c.NT == 4 && c.WN == 2) return b_launch_slots<CC, 4, 2>(c, a, ib, ob, grid, st);
  if (c.NT == 4 && c.WN == 1) return b_launch_slots<CC, 4, 1>(c, a, ib, ob, grid, st);
  if (c.NT == 2 && c.WN == 2) return b_launch_slots<CC, 2, 2>(c, a, ib, ob, grid, st);
  if (c.NT == 2) return b_launch_slots<CC, 2, 1>(c, a, ib, ob, grid, st);
  if (c.NT == 1 && c.WN == 2) return b_launch_slots<CC, 1, 2>(c, a, ib, ob, grid, st);
  return b_launch_slots<CC, 1, 1>(c, a, ib, ob, grid, st);
}

}  // namespace

// ---- entry points used by capi.hip
bool bp_bf16_igemm_ok(const ConvGeom& g, const bp_view* in, const bp_view* out) {
  const BConfig c = b_config(g);
  if (!c.ok) return false;
  if (in) {
    const int U = c.CC < 8 ? c.CC : 8;
    const int esz = in->dtype == BP_BF16 ? 2 : 4;
    const int bytes = U * esz < 16 ? U * esz : 16;        // widest single load of a unit
    if ((in->cstride * esz) % bytes || (in->coff * esz) % bytes || reinterpret_cast<uintptr_t>(in->ptr) % 16) return false;
    if (in->c != g.cin_g && !(c.CC == 4 && in->coff + 4 <= in->cstride)) return false;
  }
  if (out && out->dtype == BP_BF16) {
    const int need = c.NT == 1 ? 4 : 8;
    if (out->cstride % need || out->coff % need || reinterpret_cast<uintptr_t>(out->ptr) % 16) return false;
  }
  return true;
}

int64_t bp_bf16_packed_elems(const ConvGeom& g) {
  const BConfig c = b_config(g);
  if (!c.ok) return -1;
  return (int64_t)g.nphase * g.nphase * g.taps * c.nrun * c.nchunk * c.cout_padP * 32;
}

int bp_bf16_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, void* packed, hipStream_t st) {
  const BConfig c = b_config(g);
  if (!c.ok) return BP_EUNSUPPORTED;
  BPackArgs a{};
  a.w = w_torch; a.dst = reinterpret_cast<u16*>(packed); a.sa = wm.sa; a.sb = wm.sb;
  a.k = g.k; a.stride = g.stride; a.pad = g.pad; a.tapsy = g.taps; a.nphase = g.nphase;
  a.transposed = g.gather_transposed; a.IS = g.IS; a.cin_g = g.cin_g; a.cout_g = g.cout_g;
  a.CC = c.CC; a.nchunk = c.nchunk; a.cout_padP = c.cout_padP; a.COB = c.COB; a.NT = c.NT; a.nrun = c.nrun;
  for (int s = 0; s < c.nrun; ++s) { a.run_xm[s] = c.run_xm[s]; a.run_xq[s] = c.run_xq[s]; }
  a.total = bp_bf16_packed_elems(g);
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_bf16_igemm_run(const ConvGeom& g, const bp_view* in, const PW& pw, const void* packed, const float* bias,
                      const bp_view* out, hipStream_t st) {
  const BConfig c = b_config(g);
  if (!c.ok || !bp_bf16_igemm_ok(g, in, out)) return BP_EUNSUPPORTED;
  BArgs a{};
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff; a.cin = g.cin_g;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.cout = g.cout_g; a.wp = reinterpret_cast<const u16*>(packed); a.bias = bias; a.pw = pw;
  a.tapsy = g.taps; a.ISy = g.IS; a.ISx = g.IS; a.OS = g.OS; a.nphase = g.nphase;
  a.transposed = g.gather_transposed; a.stride = g.stride; a.pad = g.pad;
  a.nrun = c.nrun;
  for (int s = 0; s < c.nrun; ++s) a.run_off[s] = c.run_xm[s] * c.IWq + c.run_xq[s];
  a.TPR = c.TPR; a.BH = c.BH; a.IH = c.IH; a.IWq = c.IWq; a.npixp = c.npixp;
  a.nchunk = c.nchunk; a.cout_padP = c.cout_padP;
  const int qh = bp_ceil_div(out->h, g.OS), qw = bp_ceil_div(out->w, g.OS);
  a.tiles_x = bp_ceil_div(qw, 16 * c.TPR);
  a.tiles_y = bp_ceil_div(qh, c.BH);
  const int64_t gz = (int64_t)in->n * g.nphase * g.nphase;
  if (gz > 65535 || c.cout_padP / c.COB > 65535) return BP_EUNSUPPORTED;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)gz, (unsigned)(c.cout_padP / c.COB));
  const bool ib = in->dtype == BP_BF16, ob = out->dtype == BP_BF16;
  switch (c.CC) {
    case 32: return b_launch_cc<32>(c, a, ib, ob, grid, st);
    case 16: return b_launch_cc<16>(c, a, ib, ob, grid, st);
    case 8: return b_launch_cc<8>(c, a, ib, ob, grid, st);
    case 4: return b_launch_cc<4>(c, a, ib, ob, grid, st);
  }
  return BP_EUNSUPPORTED;
}

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
  static constexpr size_t LDS = LDS_MAIN > LDS_RED ? LDS_MAIN : LDS_RED;
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
template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int BH, bool XB, bool YB>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WbArgs a) {
  using Cfg = WbCfg<KHB, KW, S, NTX, NTY, WX, WY, BH>;
  constexpr int CXC = Cfg::CXC, CYC = Cfg::CYC, XR = Cfg::XR, IWq = Cfg::IWq, XT = Cfg::XT, YT = Cfg::YT;
  constexpr int TAPS = Cfg::TAPS, WK = Cfg::WK;
  static_assert(BH % WK == 0, "rows are dealt to the waves of a channel sub-block");
  static_assert(WK == 1 || (WX == 1 && WY == 1), "row-split variants own the whole channel block");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* xs = smem;
  u16* ys = smem + CXC / 16 * XT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave / (WY * WK), wy = (wave / WK) % WY, wk = wave % WK;
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
    // ---- this wave's rows of the tile
#pragma unroll 1
    for (int r = wk; r < BH; r += WK) {
      bf8 yf[NTY];
#pragma unroll
      for (int j = 0; j < NTY; ++j) {
        const u16* p = ys + (wy * NTY + j) * YT + r * 32 * 16 + trl;
        yf[j] = frag_of(lds_tr(p), lds_tr(p + 16 * 16));
      }
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

  // ---- partial tiles of this split: D[row = 4*kq + r : X channel][col = li : Y channel]
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

template <int KHB, int KW, int S, int NTX, int NTY, int WX, int WY, int BH>
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
  int64_t ns = (1024 + base - 1) / base;     // ~4 workgroups per CU
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
        reinterpret_cast<const void*>(&wgrad_bf16_kernel<KHB, KW, S, NTX, NTY, WX, WY, BH, XB_, YB_>),               \
        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                                                     \
    if (optin != hipSuccess) return BP_ELAUNCH;                                                                     \
    hipLaunchKernelGGL((wgrad_bf16_kernel<KHB, KW, S, NTX, NTY, WX, WY, BH, XB_, YB_>), grid, dim3(256), Cfg::LDS, st, a); \
  } while (0)
  if (xb && yb) BP_WB(true, true);
  else if (xb) BP_WB(true, false);
  else if (yb) BP_WB(false, true);
  else BP_WB(false, false);
#undef BP_WB
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
int bp_wgrad_bf16(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                  size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  if (!wb_view_ok(X) || !wb_view_ok(Y)) return BP_EUNSUPPORTED;
  const int k = cv->k, s = cv->stride, cx = X->c, cy = Y->c;
#define BP_WB_(...) return wb_launch<__VA_ARGS__>(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry)
  if (k == 3 && s == 1) {
    if (cx > 32 && cy > 32) BP_WB_(3, 3, 1, 2, 2, 2, 2, 4);          // 64 x 64 channel block
    if (cx > 16 && cy > 16) BP_WB_(3, 3, 1, 2, 2, 1, 1, 8);
    if (cy > 16) BP_WB_(3, 3, 1, 1, 2, 1, 1, 8);
    if (cx > 16) BP_WB_(3, 3, 1, 2, 1, 1, 1, 8);
    BP_WB_(3, 3, 1, 1, 1, 1, 1, 8);
  }
  if (k == 4 && s == 2) {
    if (cx > 32 && cy > 32) BP_WB_(2, 4, 2, 2, 2, 2, 2, 2);          // 64 x 64 channel block, two tap rows per group
    if (cx > 16 && cy > 16) BP_WB_(2, 4, 2, 2, 2, 1, 1, 4);
    if (cy > 16) BP_WB_(4, 4, 2, 1, 2, 1, 1, 4);
    if (cx > 16) BP_WB_(4, 4, 2, 2, 1, 1, 1, 4);
    BP_WB_(4, 4, 2, 1, 1, 1, 1, 4);
  }
  if (k == 5 && s == 1 && cx <= 16 && cy <= 16) BP_WB_(5, 5, 1, 1, 1, 1, 1, 8);
  if (k == 7 && s == 1 && cx <= 16 && cy <= 16) BP_WB_(4, 7, 1, 1, 1, 1, 1, 8);
#undef BP_WB_
  return BP_EUNSUPPORTED;
}
