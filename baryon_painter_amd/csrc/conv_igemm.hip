// Implicit-GEMM direct convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC.
//
// Conv2d forward, ConvTranspose2d forward and both data gradients are ONE operation here: a stride-IS
// correlation over an output sub-grid ("phase", see ConvGeom in common.hpp).
//   GEMM view:  16 consecutive output pixels of one row  x  16 produced channels per MFMA,
//               K = (tap, gathered channel), 4 channels per MFMA; issued as D = W-tile x X-tile so that a lane
//               ends up with 4 consecutive channels of one pixel (16-byte NHWC stores).
// LDS images:  input  [row][x % IS][x / IS][CC]   (x de-interleaved by the stride: the 16 pixels of an M tile
//                                                  are contiguous for every tap -> conflict-free reads)
//              weight [tap][COB][CC]
// Lane l supplies k-slice (l>>4): it reads CC/4 consecutive channels (b32/b64/b128) of its pixel and of its
// produced channel and issues CC/4 MFMAs from them.
//
// Three kernels share that formulation (igemm_config picks per layer):
//   igemm_kernel       stages the halo tile through registers (pending activation applied on the way, zero
//                      padding AFTER it, as torch does) and the weights per tap row; takes few-channel layers,
//                      pixel packing (<= 8 produced channels) and strided gathers whose halo does not fit twice
//   igemm_dma_kernel   both operands by LDS-DMA one step ahead (2-deep weight-slab ring, two input buffers,
//                      in-place rewrite for activation / padding); 4 or 8 waves; the 64-512-channel layers
//   igemm_dmaf_kernel  LDS-DMA with a tap row per barrier and all stride^2 phases of a transposed form from one
//                      staged tile; layers with <= 32 produced channels
// Workgroups are mapped to tiles XCD-aware (igemm_tile_of_block); the channel block is the slowest grid dimension.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct IgemmArgs {
  const float* in; int in_h, in_w, in_cs, in_co, cin;
  float* out; int out_h, out_w, out_cs, out_co, cout;
  const float* wp; const float* bias;
  PW pw;
  int tapsy, tapsx, ISy, ISx, OS, nphase, transposed, stride, pad;
  int PP, COP;   // pixel packing: N column j = (pp = j / COP, co = j % COP), pixel x = PP*group + pp
  int tiles_x, tiles_y, TPR, BH;
  int nchunk, cout_padP;
  int IH, IW, IWq;
  int vec_ok;
  int out_vec;   // 16-byte stores allowed
  int in_pad4;   // DMA variants: float4 groups per input buffer, padded to whole 1-KiB pieces
  int ts, in_bufs;   // igemm_dmaf_kernel: taps per step, input buffers (1 or 2)
  // Per-channel sums taken from the accumulators in the epilogue (igemm_stats): partial rows
  // stat[workgroup (x, y)][2][stat_c], folded afterwards in a fixed order.  mode 1: {sum y, sum y^2} of the produced
  // tensor (training-mode batch-norm statistics); mode 2: the produced tensor is the gradient d of an activated
  // slot whose raw values are `raw` and pending activation `spw`: {sum g, sum g*raw}, g = d * act'(spw(raw)) --
  // the two sums the batch-norm backward of that slot's producer needs.
  double* stat; int stat_c, stat_mode;
  const float* raw; int raw_cs, raw_co, raw_vec;
  PW spw;
};

template <int VW> struct Frag;
template <> struct Frag<1> { float v[1]; };
template <> struct Frag<2> { float v[2]; };
template <> struct Frag<4> { float v[4]; };

template <int VW>
__device__ __forceinline__ void lds_read(const float* p, float (&v)[VW]) {
  if constexpr (VW == 1) {
    v[0] = p[0];
  } else if constexpr (VW == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
}

template <int NT, int MT>
__device__ __forceinline__ void igemm_store(const IgemmArgs& a, const v4f (&acc)[MT][NT], float* out_n, int wm, int wn,
                                            int lm, int kq, int co0, int qy0, int qx0, int qh, int qw, int py,
                                            int px) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    const int qy = qy0 + tr;
    const int qx = qx0 + tc * 16 + lm;
    const int Y = py + a.OS * qy;
    if (qy >= qh || qx >= qw) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j0 = co0 + (wn * NT + nt) * 16 + kq * 4;
      if (a.PP == 1) {
        const int X = px + a.OS * qx;
        float* o = out_n + ((int64_t)Y * a.out_w + X) * a.out_cs + j0;
        if (a.out_vec && j0 + 3 < a.cout) {
          float4 v = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
          if (a.bias) { v.x += a.bias[j0]; v.y += a.bias[j0 + 1]; v.z += a.bias[j0 + 2]; v.w += a.bias[j0 + 3]; }
          *reinterpret_cast<float4*>(o) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j0 + r < a.cout) o[r] = acc[mt][nt][r] + (a.bias ? a.bias[j0 + r] : 0.f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + r;
          const int co = j % a.COP, pp = j / a.COP;
          const int X = a.PP * qx + pp;
          if (X < a.out_w && co < a.cout)
            out_n[((int64_t)Y * a.out_w + X) * a.out_cs + co] = acc[mt][nt][r] + (a.bias ? a.bias[co] : 0.f);
        }
      }
    }
  }
}

// Channel sums of this workgroup's output (IgemmArgs::stat).  A lane owns channels j0..j0+3 of MT (x NPH) pixels:
// it adds them up in double (exact products, as the streaming passes do: the batch-norm backward these sums feed is
// cancellation-dominated), the 16 lanes that share the channels (lm) are folded by shuffles, the waves that share
// them (wm) through LDS, and one thread per channel writes the workgroup's row.  `accf(p, mt,
// nt)` hands out the accumulators.  Fixed order throughout: the sums do not depend on scheduling.
template <int NT, int MT, int NPH, typename AccF>
__device__ __forceinline__ void igemm_stats(const IgemmArgs& a, AccF accf, float* smem, int n, int WM, int COB,
                                            int wm, int wn, int lm, int kq, int co0, int qy0, int qx0, int ph0) {
  double* red = reinterpret_cast<double*>(smem);   // [WM][COB][2]
  const int tid = threadIdx.x;
  __syncthreads();                                  // the main loop's LDS reads are over
  const bool m2 = a.stat_mode == 2;
  const float* raw_n = m2 ? a.raw + (int64_t)n * a.out_h * a.out_w * a.raw_cs + a.raw_co : nullptr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int jl = (wn * NT + nt) * 16 + kq * 4;
    const int j0 = co0 + jl;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sf[4] = {0.f, 0.f, 0.f, 0.f}, sl[4] = {1.f, 1.f, 1.f, 1.f};
    if (m2 && a.spw.scale) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = min(j0 + r, a.cout - 1);
        sc[r] = a.spw.scale[ch]; sf[r] = a.spw.shift[ch]; sl[r] = a.spw.slope[ch];
      }
    }
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      const int ph = NPH == 1 ? ph0 : p;
      const int py = ph / a.nphase, px = ph % a.nphase;
      const int qh = (a.out_h - py + a.OS - 1) / a.OS;
      const int qw = (a.out_w - px + a.OS - 1) / a.OS;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int t = wm * MT + mt;
        const int tr = t / a.TPR, tc = t % a.TPR;
        const int qy = qy0 + tr;
        const int qx = qx0 + tc * 16 + lm;
        if (qy >= qh || qx >= qw) continue;
        const v4f v = accf(p, mt, nt);              // (channels past cout: zero weights, zero sums)
        if (!m2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[r] += (double)v[r]; s2[r] = fma((double)v[r], (double)v[r], s2[r]); }
        } else {
          const int Y = py + a.OS * qy, X = px + a.OS * qx;
          const float* rp = raw_n + ((int64_t)Y * a.out_w + X) * a.raw_cs + j0;
          float rv[4];
          if (a.raw_vec && j0 + 3 < a.cout) {
            const float4 q = *reinterpret_cast<const float4*>(rp);
            rv[0] = q.x; rv[1] = q.y; rv[2] = q.z; rv[3] = q.w;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = j0 + r < a.cout ? rp[r] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float tt = fmaf(rv[r], sc[r], sf[r]);
            const float g = tt > 0.f ? v[r] : v[r] * sl[r];
            s1[r] += (double)g; s2[r] = fma((double)g, (double)rv[r], s2[r]);
          }
        }
      }
    }
    // fold over the 16 lanes (lm) that hold the same channels: halve the values a lane carries at every step
    // (8 -> 4 -> 2 -> 1), so 8 exchanges of a double instead of 32; lane lm ends with value index lm >> 1
    // (index = 4 * s + r) summed over all 16 lanes, even lanes write it.
    double h4[4], h2[2], h1;
    {
      const bool up = lm & 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double lo = s1[i], hi = s2[i];                        // values i (s1) and 4 + i (s2)
        const double send = up ? lo : hi, keep = up ? hi : lo;
        h4[i] = keep + __shfl_xor(send, 8, 16);
      }
    }
    {
      const bool up = lm & 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double send = up ? h4[i] : h4[2 + i], keep = up ? h4[2 + i] : h4[i];
        h2[i] = keep + __shfl_xor(send, 4, 16);
      }
    }
    {
      const bool up = lm & 2;
      const double send = up ? h2[0] : h2[1], keep = up ? h2[1] : h2[0];
      h1 = keep + __shfl_xor(send, 2, 16);
    }
    h1 += __shfl_xor(h1, 1, 16);
    if ((lm & 1) == 0) {
      const int idx = lm >> 1;                     // = 4 * s + r
      red[(wm * COB + jl + (idx & 3)) * 2 + (idx >> 2)] = h1;
    }
  }
  __syncthreads();
  if (tid < COB && co0 + tid < a.cout) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < WM; ++w) { t1 += red[(w * COB + tid) * 2]; t2 += red[(w * COB + tid) * 2 + 1]; }
    const int64_t L = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    a.stat[(L * 2) * a.stat_c + co0 + tid] = t1;
    a.stat[(L * 2 + 1) * a.stat_c + co0 + tid] = t2;
  }
}

// a workgroup with nothing to produce still owns a row of the partial sums
__device__ __forceinline__ void igemm_stats_zero(const IgemmArgs& a, int COB, int co0) {
  const int tid = threadIdx.x;
  if (tid < COB && co0 + tid < a.cout) {
    const int64_t L = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    a.stat[(L * 2) * a.stat_c + co0 + tid] = 0.0;
    a.stat[(L * 2 + 1) * a.stat_c + co0 + tid] = 0.0;
  }
}

// XCD-aware workgroup -> (tile, image x phase) mapping.  Workgroups are dealt to the 8 XCDs round-robin by
// linear id, and each XCD has its own L2: with the natural order the tiles of one image (which share halo rows
// and columns) land on 8 different L2s and every halo is fetched from HBM once per XCD.  Here XCD x works through
// a contiguous range of the (image, tile) sequence instead, so neighbouring tiles meet in one L2
// (bijective for any count: the first n%8 XCDs take one more).
__device__ __forceinline__ void igemm_tile_of_block(int* tile, int* by) {
  const int gx = gridDim.x, n = gx * gridDim.y;
  const int L = blockIdx.y * gx + blockIdx.x;
  const int q = n >> 3, r = n & 7;
  const int xcd = L & 7, idx = L >> 3;
  const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  *tile = Lp % gx;
  *by = Lp / gx;
}

template <int CC, int NT, int WN, int MT>
__global__ __launch_bounds__(256, (MT == 4 && NT == 4) ? 4 : 1) void igemm_kernel(IgemmArgs a) {
  constexpr int VW = CC / 4;
  constexpr int WM = 4 / WN;
  constexpr int COB = 16 * NT * WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lds_in = smem;
  float* lds_w = smem + (size_t)a.IH * a.ISx * a.IWq * CC;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lm = lane & 15, kq = lane >> 4;

  // grid = (tiles, images x phases, channel blocks): workgroups that run together read the same weight slabs
  int tile, by;
  igemm_tile_of_block(&tile, &by);
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int co0 = blockIdx.z * COB;
  const int ph = by % (a.nphase * a.nphase);
  const int n = by / (a.nphase * a.nphase);
  const int py = ph / a.nphase, px = ph % a.nphase;

  const int BW = 16 * a.TPR;
  const int qy0 = tile_y * a.BH, qx0 = tile_x * BW;
  // phase grid extents
  const int qh = (a.out_h - py + a.OS - 1) / a.OS;
  const int qw = a.PP > 1 ? (a.out_w + a.PP - 1) / a.PP : (a.out_w - px + a.OS - 1) / a.OS;   // groups
  if (qy0 >= qh || qx0 >= qw) {        // uniform per block
    if (a.stat) igemm_stats_zero(a, COB, co0);
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

  // per-lane LDS bases of this wave's M tiles
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    abase[mt] = (tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * CC + kq * VW;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = ((wn * NT + nt) * 16 + lm) * CC + kq * VW;

  v4f acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};

  const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
  const int in_elems4 = a.IH * a.IW * VW;  // float4 groups per chunk

  // Staging slots of this thread: the (row, column) of each halo element it copies is the same for every
  // channel chunk, so the global offset (or -1 outside the image) and the LDS offset are computed once.
  constexpr int SLOTS = NT == 4 ? 3 : (NT == 2 ? 8 : 6);   // (NT 1 used to take the element-by-element path: every load waited for)
  constexpr int SG = SLOTS < 4 ? SLOTS : 4;                // loads in flight per thread
  const int c4 = tid % VW;                                 // 256 % VW == 0: fixed channel quad per thread
  const bool slots_ok = in_elems4 <= SLOTS * 256;
  int s_g[SLOTS], s_l[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256;
    s_g[i] = -1; s_l[i] = -1;
    if (slots_ok && e < in_elems4) {
      const int pix = e / VW;
      const int c = pix % a.IW;
      const int r = pix / a.IW;
      const int iy = gy0 + r, ix = gx0 + c;
      s_l[i] = ((r * a.ISx + c % a.ISx) * a.IWq + c / a.ISx) * CC + c4 * 4;
      if (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) s_g[i] = (iy * a.in_w + ix) * a.in_cs;
    }
  }

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();  // previous chunk's readers are done with lds_in / lds_w
    // ---- stage the input halo tile for channels [chunk*CC, chunk*CC+CC)
    {
      const int ch = chunk * CC + c4 * 4;
      const PW4 p4 = pw4_load(a.pw, ch, a.cin);
      if (slots_ok && a.vec_ok && ch + 3 < a.cin) {
#pragma unroll
        for (int i0 = 0; i0 < SLOTS; i0 += SG) {
          float4 v[SG];
#pragma unroll
          for (int i = 0; i < SG; ++i)       // (unconditional: a load under a per-lane condition may be waited for on the spot)
            v[i] = *reinterpret_cast<const float4*>(in_n + (s_g[i0 + i] >= 0 ? s_g[i0 + i] : 0) + ch);
#pragma unroll
          for (int i = 0; i < SG; ++i)
            if (s_l[i0 + i] >= 0)
              *reinterpret_cast<float4*>(lds_in + s_l[i0 + i]) =
                  s_g[i0 + i] >= 0 ? pw4_apply4(p4, v[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        for (int e = tid; e < in_elems4; e += 256) {
          const int pix = e / VW;
          const int c = pix % a.IW;
          const int r = pix / a.IW;
          const int iy = gy0 + r, ix = gx0 + c;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w && ch < a.cin) {
            const float* p = in_n + ((int64_t)iy * a.in_w + ix) * a.in_cs + ch;
            if (a.vec_ok) {
              v = *reinterpret_cast<const float4*>(p);
            } else {
              v.x = p[0];
              if (ch + 1 < a.cin) v.y = p[1];
              if (ch + 2 < a.cin) v.z = p[2];
              if (ch + 3 < a.cin) v.w = p[3];
            }
            v = pw4_apply4(p4, v);
            if (ch + 1 >= a.cin) v.y = 0.f;
            if (ch + 2 >= a.cin) v.z = 0.f;
            if (ch + 3 >= a.cin) v.w = 0.f;
          }
          const int li = ((r * a.ISx + c % a.ISx) * a.IWq + c / a.ISx) * CC + c4 * 4;
          *reinterpret_cast<float4*>(lds_in + li) = v;
        }
      }
    }
    for (int ty = 0; ty < a.tapsy; ++ty) {
      if (ty) __syncthreads();  // readers of the previous tap row's weights are done
      // ---- stage weights for (phase, ty, all tx, chunk): taps slabs of COB*CC floats
      {
        const int slab4 = COB * CC / 4;
        for (int e = tid; e < a.tapsx * slab4; e += 256) {
          const int tx = e / slab4, o = e % slab4;
          const float* src = a.wp + ((((int64_t)(ph * a.tapsy + ty) * a.tapsx + tx) * a.nchunk + chunk) *
                                         a.cout_padP + co0) * CC;
          *reinterpret_cast<float4*>(lds_w + (size_t)tx * COB * CC + o * 4) =
              *reinterpret_cast<const float4*>(src + o * 4);
        }
      }
      __syncthreads();
      for (int tx = 0; tx < a.tapsx; ++tx) {
        const int tapoff = ((ty * a.ISx + tx % a.ISx) * a.IWq + tx / a.ISx) * CC;
        float af[MT][VW], bf[NT][VW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) lds_read<VW>(lds_in + abase[mt] + tapoff, af[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds_read<VW>(lds_w + tx * COB * CC + bbase[nt], bf[nt]);
#pragma unroll
        for (int s = 0; s < VW; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue.  The MFMA is issued as D = W^T-tile x X-tile, so D[row = 4*(lane>>4)+r][col = lane&15] holds
  // produced channel (or (pp, channel)) `row` of pixel (or pixel group) `col`: a lane owns 4 consecutive
  // channels of one pixel, which is one 16-byte store in NHWC.
  float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
  igemm_store<NT, MT>(a, acc, out_n, wm, wn, lm, kq, co0, qy0, qx0, qh, qw, py, px);
  if (a.stat)
    igemm_stats<NT, MT, 1>(a, [&](int, int mt, int nt) { return acc[mt][nt]; }, smem, n, WM, COB, wm, wn, lm, kq, co0,
                           qy0, qx0, ph);
}

// ------------------------------------------------------------------------------------------------
// The same contraction with both operands brought in by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write) one step ahead of their use:
//   * weights: one tap slab [COB][CC] per step into a 2-deep ring; the slab of step s+1 is issued right
//     after the barrier that opens step s, so it has the whole of step s's MFMAs to land;
//   * input halo tile: the tile of chunk c+1 is issued at the first tap of chunk c into the other of two
//     buffers.  A DMA cannot apply the pending activation or the zero padding, so in the last tap of
//     chunk c every thread rewrites in place exactly the elements its own lanes fetched (they were
//     retired by the vmcnt(0) of an earlier barrier; nobody reads that buffer before the next barrier).
// One barrier per tap; its vmcnt(0) retires this wave's DMAs, the barrier publishes them.
// LDS image of the input is the one of igemm_kernel, enumerated in LDS order so that the 64 lanes of a
// wave-instruction land on 1 KiB of consecutive LDS (destination = wave-uniform base + lane * 16).
template <int CC, int NT, int WN, int SLOTS, int NW>
__global__ __launch_bounds__(64 * NW, NT == 4 ? ((WN == 2 && (SLOTS == 3 || NW == 8)) ? 4 : 3) : 2) void igemm_dma_kernel(IgemmArgs a) {
  constexpr int MT = 4;
  constexpr int VW = CC / 4;
  constexpr int WM = NW / WN;
  constexpr int NTH = 64 * NW;
  constexpr int RING = 2;                 // weight slabs in LDS (a ring of three measured slower)
  constexpr int COB = 16 * NT * WN;
  constexpr int SLAB = COB * CC;          // floats per tap slab
  constexpr int SLAB_I = SLAB / 256;      // 1-KiB pieces per slab
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int in_f = a.in_pad4 * 4;
  float* lds_w = smem + 2 * in_f;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lm = lane & 15, kq = lane >> 4;

  // grid = (tiles, images x phases, channel blocks): workgroups that run together read the same weight slabs
  int tile, by;
  igemm_tile_of_block(&tile, &by);
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int co0 = blockIdx.z * COB;
  const int ph = by % (a.nphase * a.nphase);
  const int n = by / (a.nphase * a.nphase);
  const int py = ph / a.nphase, px = ph % a.nphase;

  const int BW = 16 * a.TPR;
  const int qy0 = tile_y * a.BH, qx0 = tile_x * BW;
  const int qh = (a.out_h - py + a.OS - 1) / a.OS;
  const int qw = (a.out_w - px + a.OS - 1) / a.OS;
  if (qy0 >= qh || qx0 >= qw) {        // uniform per block
    if (a.stat) igemm_stats_zero(a, COB, co0);
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
    abase[mt] = (tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * CC + kq * VW;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = ((wn * NT + nt) * 16 + lm) * CC + kq * VW;

  v4f acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};

  const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
  const int E = a.IH * a.ISx * a.IWq * VW;   // float4 groups of the LDS image

  // This thread's elements of the image: e = i * NTH + tid (LDS order).  Source offset with the
  // coordinates clamped into the image (every lane of a DMA fetches something; what lies outside is
  // zeroed by the rewrite), and one bit per slot for "outside".
  const int c4 = tid % VW;
  int s_g[SLOTS];
  unsigned outside = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = i * NTH + tid;
    const int p = e / VW;
    const int xq = p % a.IWq;
    const int t = p / a.IWq;
    const int xm = t % a.ISx;
    const int r = t / a.ISx;
    const int c = xq * a.ISx + xm;
    const int iy = gy0 + r, ix = gx0 + c;
    const bool in = e < E && c < a.IW && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
    if (!in) outside |= 1u << i;
    const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
    s_g[i] = ((cy * a.in_w + cx) * a.in_cs + c4 * 4) * 4;   // bytes
  }

  auto issue_input = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int ebase = i * NTH + wave * 64;   // wave-uniform
      if (ebase < a.in_pad4) bp_glds16(in_n + chunk * CC, (unsigned)s_g[i], buf * in_f + ebase * 4);
    }
  };
  // The pending activation's per-channel parameters live in LDS ([scale | shift | slope][cin], after the ring):
  // an ordinary global load inside the loop would make the compiler wait vmcnt(0) -- and with it every DMA in
  // flight -- at the next reuse of its destination registers.
  float* lpw = smem + 2 * in_f + RING * SLAB;
  const bool pw_on = a.pw.scale != nullptr;
  if (pw_on) {
    for (int i = tid; i < a.cin; i += NTH) {
      lpw[i] = a.pw.scale[i]; lpw[a.cin + i] = a.pw.shift[i]; lpw[2 * a.cin + i] = a.pw.slope[i];
    }
    __syncthreads();
  }
  auto rewrite_input = [&](int chunk, int buf) {
    PW4 p4;
    p4.on = pw_on;
    if (pw_on) {
      const int ch = chunk * CC + c4 * 4;
      const float4 sc = *reinterpret_cast<const float4*>(lpw + ch);
      const float4 sf = *reinterpret_cast<const float4*>(lpw + a.cin + ch);
      const float4 sl = *reinterpret_cast<const float4*>(lpw + 2 * a.cin + ch);
      p4.sc[0] = sc.x; p4.sc[1] = sc.y; p4.sc[2] = sc.z; p4.sc[3] = sc.w;
      p4.sf[0] = sf.x; p4.sf[1] = sf.y; p4.sf[2] = sf.z; p4.sf[3] = sf.w;
      p4.sl[0] = sl.x; p4.sl[1] = sl.y; p4.sl[2] = sl.z; p4.sl[3] = sl.w;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = i * NTH + tid;
      if (e < E) {
        float4* q = reinterpret_cast<float4*>(smem + buf * in_f + e * 4);
        if ((outside >> i) & 1u) *q = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (p4.on) *q = pw4_apply4(p4, *q);
      }
    }
  };
  auto issue_slab = [&](int chunk, int ty, int tx, int slot) {
    const float* src = a.wp + ((((int64_t)(ph * a.tapsy + ty) * a.tapsx + tx) * a.nchunk + chunk) *
                                   a.cout_padP + co0) * CC;
    for (int k = wave; k < SLAB_I; k += NW) bp_glds16(src + k * 256, (unsigned)lane * 16u, 2 * in_f + slot * SLAB + k * 256);
  };

  auto advance = [&](int& ch, int& ty, int& tx) {
    if (++tx == a.tapsx) { tx = 0; ++ty; }
    if (ty == a.tapsy) { ty = 0; ++ch; }
  };
  // prologue: first input tile and first slab
  issue_input(0, 0);
  issue_slab(0, 0, 0, 0);
  bp_wait_dma();
  rewrite_input(0, 0);

  int slot = 0;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    const float* lin = smem + (chunk & 1) * in_f;
    for (int ty = 0; ty < a.tapsy; ++ty) {
      for (int tx = 0; tx < a.tapsx; ++tx) {
        // slab `step` (and a pending input tile) has landed, step-1 is read out
        bp_wait_dma_barrier();
        const bool first = (ty == 0 && tx == 0), last = (ty == a.tapsy - 1 && tx == a.tapsx - 1);
        if (first && chunk + 1 < a.nchunk) issue_input(chunk + 1, (chunk + 1) & 1);
        {
          int nch = chunk, nty = ty, ntx = tx;
          advance(nch, nty, ntx);
          if (nch < a.nchunk) issue_slab(nch, nty, ntx, slot ^ 1);
        }

        const float* lw = lds_w + slot * SLAB;
        slot ^= 1;
        const int tapoff = ((ty * a.ISx + tx % a.ISx) * a.IWq + tx / a.ISx) * CC;
        float af[MT][VW], bf[NT][VW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) lds_read<VW>(lin + abase[mt] + tapoff, af[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds_read<VW>(lw + bbase[nt], bf[nt]);
#pragma unroll
        for (int s = 0; s < VW; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nt][s], af[mt][s], acc[mt][nt], 0, 0, 0);

        // the next chunk's tile was retired by an earlier barrier's vmcnt(0) (taps >= 2): activation + padding
        if (last && chunk + 1 < a.nchunk) rewrite_input(chunk + 1, (chunk + 1) & 1);
      }
    }
  }

  float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
  igemm_store<NT, MT>(a, acc, out_n, wm, wn, lm, kq, co0, qy0, qx0, qh, qw, py, px);
  if (a.stat)
    igemm_stats<NT, MT, 1>(a, [&](int, int mt, int nt) { return acc[mt][nt]; }, smem, n, WM, COB, wm, wn, lm, kq, co0,
                           qy0, qx0, ph);
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA pipeline for layers with at most 32 produced channels (NT <= 2), where one tap is too little work
// per barrier and the input tile, not the weights, is the traffic:
//   * a step covers TS consecutive taps (a tap row, or all taps) -> >= 64 MFMAs per wave between barriers;
//   * transposed forms (ConvTranspose2d forward, data gradient of a strided Conv2d) compute all stride^2
//     output phases in one workgroup from ONE staged input tile (the union of the phases' halos), with one
//     accumulator set per phase, instead of staging the tile once per phase;
//   * a single input buffer when the layer has one channel chunk (nothing to prefetch).
// Four waves, each 4 M tiles (64 pixels of the phase grid) x all NT channel tiles.
template <int CC, int NT, int SLOTS, int NPH>
__global__ __launch_bounds__(256, 2) void igemm_dmaf_kernel(IgemmArgs a) {
  constexpr int MT = 4;
  constexpr int VW = CC / 4;
  constexpr int COB = 16 * NT;
  constexpr int SLAB = COB * CC;          // floats per tap slab
  constexpr int SLAB_I = SLAB / 256;      // 1-KiB pieces per slab
  static_assert(SLAB % 256 == 0, "a tap slab is whole 1-KiB pieces");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int in_f = a.in_pad4 * 4;
  const int ring0 = a.in_bufs * in_f;               // float offset of the slab ring
  const int grp = a.ts * SLAB;                      // floats per ring slot
  float* lpw = smem + ring0 + 2 * grp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave;
  const int lm = lane & 15, kq = lane >> 4;

  int tile, by;
  igemm_tile_of_block(&tile, &by);
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int co0 = blockIdx.z * COB;
  const int nph2 = a.nphase * a.nphase;
  const int ph0 = NPH == 1 ? by % nph2 : 0;
  const int n = NPH == 1 ? by / nph2 : by;
  const int py0 = ph0 / a.nphase, px0 = ph0 % a.nphase;

  const int BW = 16 * a.TPR;
  const int qy0 = tile_y * a.BH, qx0 = tile_x * BW;
  {
    // extents of the (largest) phase grid this workgroup works on
    const int qh = (a.out_h - py0 + a.OS - 1) / a.OS;
    const int qw = a.PP > 1 ? (a.out_w + a.PP - 1) / a.PP : (a.out_w - px0 + a.OS - 1) / a.OS;
    if (qy0 >= qh || qx0 >= qw) {        // uniform per block
    if (a.stat) igemm_stats_zero(a, COB, co0);
    return;
  }
  }

  // origin of the staged tile: the first gathered row/column of phase (py0, px0) -- with fused phases that is
  // phase 0, whose origin is the smallest; phase p starts dy(p) rows / dx(p) columns further in
  int iy0, ix0;
  if (a.transposed) {
    iy0 = bp_t_i0(py0, a.pad, a.stride, a.tapsy);
    ix0 = bp_t_i0(px0, a.pad, a.stride, a.tapsy);
  } else {
    iy0 = -a.pad; ix0 = -a.pad;
  }
  const int gy0 = a.ISy * qy0 + iy0, gx0 = a.ISx * qx0 + ix0;

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    abase[mt] = (tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * CC + kq * VW;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (nt * 16 + lm) * CC + kq * VW;

  v4f acc[NPH][MT][NT];
#pragma unroll
  for (int p = 0; p < NPH; ++p)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[p][mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};

  const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co;
  const int E = a.IH * a.ISx * a.IWq * VW;   // float4 groups of the LDS image

  const int c4 = tid % VW;
  int s_g[SLOTS];
  unsigned outside = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = i * 256 + tid;
    const int p = e / VW;
    const int xq = p % a.IWq;
    const int t = p / a.IWq;
    const int xm = t % a.ISx;
    const int r = t / a.ISx;
    const int c = xq * a.ISx + xm;
    const int iy = gy0 + r, ix = gx0 + c;
    const bool in = e < E && c < a.IW && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w;
    if (!in) outside |= 1u << i;
    const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
    s_g[i] = ((cy * a.in_w + cx) * a.in_cs + c4 * 4) * 4;   // bytes
  }

  auto issue_input = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int ebase = i * 256 + wave * 64;   // wave-uniform
      if (ebase < a.in_pad4) bp_glds16(in_n + chunk * CC, (unsigned)s_g[i], buf * in_f + ebase * 4);
    }
  };
  const bool pw_on = a.pw.scale != nullptr;
  if (pw_on) {
    for (int i = tid; i < a.cin; i += 256) {
      lpw[i] = a.pw.scale[i]; lpw[a.cin + i] = a.pw.shift[i]; lpw[2 * a.cin + i] = a.pw.slope[i];
    }
    __syncthreads();
  }
  auto rewrite_input = [&](int chunk, int buf) {
    PW4 p4;
    p4.on = pw_on;
    if (pw_on) {
      const int ch = chunk * CC + c4 * 4;
      const float4 sc = *reinterpret_cast<const float4*>(lpw + ch);
      const float4 sf = *reinterpret_cast<const float4*>(lpw + a.cin + ch);
      const float4 sl = *reinterpret_cast<const float4*>(lpw + 2 * a.cin + ch);
      p4.sc[0] = sc.x; p4.sc[1] = sc.y; p4.sc[2] = sc.z; p4.sc[3] = sc.w;
      p4.sf[0] = sf.x; p4.sf[1] = sf.y; p4.sf[2] = sf.z; p4.sf[3] = sf.w;
      p4.sl[0] = sl.x; p4.sl[1] = sl.y; p4.sl[2] = sl.z; p4.sl[3] = sl.w;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = i * 256 + tid;
      if (e < E) {
        float4* q = reinterpret_cast<float4*>(smem + buf * in_f + e * 4);
        if ((outside >> i) & 1u) *q = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (p4.on) *q = pw4_apply4(p4, *q);
      }
    }
  };
  // the TS tap slabs of group g of phase ph, channel chunk `chunk` -> ring slot
  auto issue_group = [&](int chunk, int ph, int g, int slot) {
    for (int k = wave; k < a.ts * SLAB_I; k += 4) {
      const int tap = g * a.ts + k / SLAB_I, part = k % SLAB_I;
      const int ty = tap / a.tapsx, tx = tap - ty * a.tapsx;
      const float* src = a.wp + ((((int64_t)(ph * a.tapsy + ty) * a.tapsx + tx) * a.nchunk + chunk) *
                                     a.cout_padP + co0) * CC + part * 256;
      bp_glds16(src, (unsigned)lane * 16u, ring0 + slot * grp + k * 256);
    }
  };

  const int G = a.tapsy * a.tapsx / a.ts;          // steps per (chunk, phase)
  const bool single_step = NPH * G == 1;            // the next tile is issued and rewritten in the same step

  issue_input(0, 0);
  issue_group(0, NPH == 1 ? ph0 : 0, 0, 0);
  bp_wait_dma();
  rewrite_input(0, 0);

  int slot = 0;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    const float* lin = smem + (a.in_bufs == 2 ? (chunk & 1) * in_f : 0);
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      const int ph = NPH == 1 ? ph0 : p;
      const int ppy = ph / a.nphase, ppx = ph % a.nphase;
      // this phase's first gathered row / column relative to the tile origin
      const int dy = (NPH > 1 && a.transposed) ? bp_t_i0(ppy, a.pad, a.stride, a.tapsy) - iy0 : 0;
      const int dx = (NPH > 1 && a.transposed) ? bp_t_i0(ppx, a.pad, a.stride, a.tapsy) - ix0 : 0;
      for (int g = 0; g < G; ++g) {
        bp_wait_dma_barrier();   // group (chunk, p, g) and a pending input tile have landed; the previous step is read out
        const bool first = (p == 0 && g == 0), last = (p == NPH - 1 && g == G - 1);
        if (first && chunk + 1 < a.nchunk) issue_input(chunk + 1, (chunk + 1) & 1);
        {
          int ng = g + 1, nch = chunk;
          int nph = ph;
          if (ng == G) {
            ng = 0;
            if (NPH == 1) ++nch;
            else if (p + 1 < NPH) nph = p + 1;
            else { nph = 0; ++nch; }
          }
          if (nch < a.nchunk) issue_group(nch, nph, ng, slot ^ 1);
        }
        const float* lw = smem + ring0 + slot * grp;
        slot ^= 1;
        for (int t = 0; t < a.ts; ++t) {
          const int tap = g * a.ts + t;
          const int ty = tap / a.tapsx, tx = tap - ty * a.tapsx;
          const int ey = dy + ty, ex = dx + tx;
          const int tapoff = ((ey * a.ISx + ex % a.ISx) * a.IWq + ex / a.ISx) * CC;
          float af[MT][VW], bf[NT][VW];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) lds_read<VW>(lin + abase[mt] + tapoff, af[mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) lds_read<VW>(lw + t * SLAB + bbase[nt], bf[nt]);
#pragma unroll
          for (int s = 0; s < VW; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nt][s], af[mt][s], acc[p][mt][nt], 0, 0, 0);
        }
        if (last && chunk + 1 < a.nchunk) {
          if (single_step) bp_wait_dma();
          rewrite_input(chunk + 1, (chunk + 1) & 1);
        }
      }
    }
  }

  float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
    const int ph = NPH == 1 ? ph0 : p;
    const int py = ph / a.nphase, px = ph % a.nphase;
    const int qh = (a.out_h - py + a.OS - 1) / a.OS;
    const int qw = a.PP > 1 ? (a.out_w + a.PP - 1) / a.PP : (a.out_w - px + a.OS - 1) / a.OS;
    igemm_store<NT, MT>(a, acc[p], out_n, wm, 0, lm, kq, co0, qy0, qx0, qh, qw, py, px);
  }
  if (a.stat)
    igemm_stats<NT, MT, NPH>(a, [&](int p, int mt, int nt) { return acc[p][mt][nt]; }, smem, n, 4, COB, wm, 0, lm, kq,
                             co0, qy0, qx0, ph0);
}

// ------------------------------------------------------------------------------------------------
// Weights-resident persistent form for the thin full-resolution gathers (<= 32 produced channels, <= 16 gathered
// channels in one chunk: the k4s2 16->32 layers and the k7 16<->8 head), whose halo tile is 4-5x the output tile
// (stride 2) or whose 49 taps make the per-tap-row weight staging of igemm_kernel most of the time:
//   * ALL tap slabs of the workgroup's channel block are copied to LDS once; the workgroup then walks a contiguous
//     range of (image, tile) pairs (neighbouring tiles share halo rows / columns: they meet in one XCD's L2);
//   * the NEXT tile's halo is in flight in registers while the current one is multiplied, then written (pending
//     activation and zero padding applied) into the single input buffer: two barriers per tile, none inside it;
//   * eight waves share the tile (MT M-tiles each), so a thread carries ~10 staging registers, not ~20.
// LDS image, fragments and epilogue are those of igemm_kernel; batch-norm sums (mode 1) are kept per lane over all
// the tiles of the workgroup and folded once.
template <int CC, int NT, int MT, int NW, int SLOTS, bool STATS>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void igemm_wres_kernel(IgemmArgs a, int ntiles_total, int per_block) {
  constexpr int VW = CC / 4;
  constexpr int NTH = 64 * NW;
  constexpr int COB = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lds_in = smem;
  float* lds_w = smem + a.in_pad4 * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave;
  const int lm = lane & 15, kq = lane >> 4;
  const int co0 = blockIdx.z * COB;
  const int t_begin = blockIdx.x * per_block;
  int t_end = t_begin + per_block;
  if (t_end > ntiles_total) t_end = ntiles_total;
  const bool idle = t_begin >= t_end;          // (still writes its zero row of the statistics)
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int BW = 16 * a.TPR;
  const int qh = a.out_h;                       // gather form: one phase, OS = 1
  const int qw = a.PP > 1 ? (a.out_w + a.PP - 1) / a.PP : a.out_w;

  // ---- weights: taps x [COB][CC], contiguous per tap in the packed image (one chunk)
  {
    const int slab4 = COB * CC / 4, ntap = a.tapsy * a.tapsx;
    for (int e = tid; e < ntap * slab4 && !idle; e += NTH) {
      const int tap = e / slab4, o = e - tap * slab4;
      const float* src = a.wp + ((int64_t)tap * a.cout_padP + co0) * CC;
      *reinterpret_cast<float4*>(lds_w + (size_t)tap * COB * CC + o * 4) = *reinterpret_cast<const float4*>(src + o * 4);
    }
  }

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    abase[mt] = (tr * a.ISy * a.ISx * a.IWq + tc * 16 + lm) * CC + kq * VW;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (nt * 16 + lm) * CC + kq * VW;

  // staging slots of this thread: (row, column, channel quad) of the halo element, the same for every tile
  const int in_elems4 = a.IH * a.IW * VW;
  const int c4 = tid % VW;                      // NTH % VW == 0
  // (row << 16 | column) of the halo element, its LDS offset, its offset from the tile's first element in the tensor.
  // Unused slots (past the end of the tile) write to a spare 16 bytes behind the weights: no branch in the commit.
  int s_rc[SLOTS], s_l[SLOTS], s_off[SLOTS];
  unsigned valid = 0;
  const int dummy = a.in_pad4 * 4 + a.tapsy * a.tapsx * COB * CC;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * NTH;
    s_l[i] = dummy; s_rc[i] = 0; s_off[i] = 0;
    if (e < in_elems4) {
      const int pix = e / VW;
      const int c = pix % a.IW, r = pix / a.IW;
      s_rc[i] = (r << 16) | c;
      s_l[i] = ((r * a.ISx + c % a.ISx) * a.IWq + c / a.ISx) * CC + c4 * 4;
      s_off[i] = (r * a.in_w + c) * a.in_cs;
      valid |= 1u << i;
    }
  }
  const PW4 p4 = pw4_load(a.pw, c4 * 4, a.cin);
  // first gathered row / column relative to the output position (unit-stride data gradients arrive in the transposed form)
  const int i0 = a.transposed ? bp_t_i0(0, a.pad, a.stride, a.tapsy) : -a.pad;
  float4 stage[SLOTS];
  unsigned inside = 0;
  auto tile_origin = [&](int t, int* n, int* qy0, int* qx0) {
    *n = t / tiles_per_img;
    const int r = t - *n * tiles_per_img;
    *qy0 = (r / a.tiles_x) * a.BH; *qx0 = (r % a.tiles_x) * BW;
  };
  // Tiles whose halo lies inside the image (all but the border ring) take one pointer add per slot; the others clamp
  // their coordinates (every slot loads either way: a load under a per-lane condition is waited for on the spot).
  auto fetch = [&](int t) {
    int n, qy0, qx0;
    tile_origin(t, &n, &qy0, &qx0);
    const int gy0 = a.ISy * qy0 + i0, gx0 = a.ISx * qx0 + i0;
    const float* in_n = a.in + (int64_t)n * a.in_h * a.in_w * a.in_cs + a.in_co + c4 * 4;
    if (gy0 >= 0 && gx0 >= 0 && gy0 + a.IH <= a.in_h && gx0 + a.IW <= a.in_w) {       // uniform
      const float* base = in_n + ((int64_t)gy0 * a.in_w + gx0) * a.in_cs;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) stage[i] = *reinterpret_cast<const float4*>(base + s_off[i]);
      inside = valid;
    } else {
      unsigned in = 0;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int iy = gy0 + (s_rc[i] >> 16), ix = gx0 + (s_rc[i] & 0xffff);
        if (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) in |= 1u << i;
        const int cy = min(max(iy, 0), a.in_h - 1), cx = min(max(ix, 0), a.in_w - 1);
        stage[i] = *reinterpret_cast<const float4*>(in_n + (cy * a.in_w + cx) * a.in_cs);
      }
      inside = in & valid;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const float4 v = pw4_apply4(p4, stage[i]);
      const bool in = (inside >> i) & 1u;
      *reinterpret_cast<float4*>(lds_in + s_l[i]) = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
    }
  };

  constexpr int SN = STATS ? NT : 1;            // (the sums cost 16 * NT registers: only the batch-norm layers' variant has them)
  double s1[SN][4], s2[SN][4];
#pragma unroll
  for (int nt = 0; nt < SN; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.0; s2[nt][r] = 0.0; }

  if (!idle) { fetch(t_begin); commit(); }
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) fetch(t + 1);
    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};
    // taps in (ty, tx) order; the fragments of tap k+1 are read while tap k is multiplied (one tap is only 8-16
    // MFMAs per wave: an exposed LDS round trip per tap would be a fifth of the loop)
    {
      const int ntap = a.tapsy * a.tapsx;
      int ty = 0, xm = 0, xq = 0;            // tx = xq * ISx + xm
      float af[2][MT][VW], bf[2][NT][VW];
      auto read_tap = [&](int k, int buf) {
        const int tapoff = ((ty * a.ISx + xm) * a.IWq + xq) * CC;
        const float* lw = lds_w + (size_t)k * COB * CC;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) lds_read<VW>(lds_in + abase[mt] + tapoff, af[buf][mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lds_read<VW>(lw + bbase[nt], bf[buf][nt]);
        if (++xm == a.ISx) { xm = 0; ++xq; }
        if (xq * a.ISx + xm >= a.tapsx) { xm = 0; xq = 0; ++ty; }
      };
      auto mul_tap = [&](int buf) {
#pragma unroll
        for (int q = 0; q < VW; ++q)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[buf][nt][q], af[buf][mt][q], acc[mt][nt], 0, 0, 0);
      };
      read_tap(0, 0);
      int k = 0;
      for (; k + 2 <= ntap - 1; k += 2) {     // two taps per trip: the buffer index stays a compile-time constant
        read_tap(k + 1, 1);
        mul_tap(0);
        read_tap(k + 2, 0);
        mul_tap(1);
      }
      // k taps done, buffer 0 holds tap k; 1 or 2 taps remain
      if (k + 1 < ntap) {
        read_tap(k + 1, 1);
        mul_tap(0);
        mul_tap(1);
      } else {
        mul_tap(0);
      }
    }
    int n, qy0, qx0;
    tile_origin(t, &n, &qy0, &qx0);
    float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
    igemm_store<NT, MT>(a, acc, out_n, wm, 0, lm, kq, co0, qy0, qx0, qh, qw, 0, 0);
    if constexpr (STATS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int tt = wm * MT + mt;
        const int qy = qy0 + tt / a.TPR, qx = qx0 + (tt % a.TPR) * 16 + lm;
        if (qy >= qh || qx >= qw) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = (double)acc[mt][nt][r];
            s1[nt][r] += v; s2[nt][r] = fma(v, v, s2[nt][r]);
          }
      }
    }
    __syncthreads();                       // every wave is done reading the tile
    if (t + 1 < t_end) commit();
    __syncthreads();
  }
  if constexpr (STATS) {
    double* red = reinterpret_cast<double*>(lds_in);        // [NW][COB][2]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
          s1[nt][r] += __shfl_xor(s1[nt][r], off, 16);
          s2[nt][r] += __shfl_xor(s2[nt][r], off, 16);
        }
        if (lm == 0) {
          red[(wm * COB + nt * 16 + kq * 4 + r) * 2] = s1[nt][r];
          red[(wm * COB + nt * 16 + kq * 4 + r) * 2 + 1] = s2[nt][r];
        }
      }
    __syncthreads();
    if (tid < COB && co0 + tid < a.cout) {
      double t1 = 0.0, t2 = 0.0;
      for (int w = 0; w < NW; ++w) { t1 += red[(w * COB + tid) * 2]; t2 += red[(w * COB + tid) * 2 + 1]; }
      const int64_t L = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
      a.stat[(L * 2) * a.stat_c + co0 + tid] = t1;
      a.stat[(L * 2 + 1) * a.stat_c + co0 + tid] = t2;
    }
  }
}

// weights: torch layout -> [phase][ty][tx][chunk][cout_padP][CC]
struct PackArgs {
  const float* w; float* dst;
  int64_t sa, sb;
  int k, stride, pad, tapsy, tapsx, nphase, transposed;
  int cin_g, cout_g, CC, nchunk, cout_padP, PP, COP;
  int64_t total;
};

__device__ __forceinline__ void pack_element(const PackArgs& a, int64_t i) {
  if (i >= a.total) return;
  int64_t r = i;
  const int cl = r % a.CC; r /= a.CC;
  const int co = r % a.cout_padP; r /= a.cout_padP;
  const int chunk = r % a.nchunk; r /= a.nchunk;
  const int tx = r % a.tapsx; r /= a.tapsx;
  const int ty = r % a.tapsy; r /= a.tapsy;
  const int ph = (int)r;
  const int py = ph / a.nphase, px = ph % a.nphase;
  int ky, kx, cch = co;
  if (a.transposed) {
    ky = bp_t_ky(py, a.pad, a.stride, a.tapsy, ty);
    kx = bp_t_ky(px, a.pad, a.stride, a.tapsy, tx);
  } else {
    ky = ty; kx = tx;
  }
  if (a.PP > 1) {            // column = (pp, channel): pixel pp of the group sees tap tx - pp
    const int pp = co / a.COP;
    cch = co % a.COP;
    kx = (a.transposed ? bp_t_ky(px, a.pad, a.stride, a.tapsy, tx - pp) : tx - pp);
    if (tx - pp < 0 || tx - pp >= a.tapsy) kx = -1;
  }
  const int ci = chunk * a.CC + cl;
  float v = 0.f;
  if (ci < a.cin_g && cch < a.cout_g && ky < a.k && kx >= 0 && kx < a.k)
    v = a.w[ci * a.sa + cch * a.sb + ky * a.k + kx];
  a.dst[i] = v;
}

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
  pack_element(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Every layer's weights in ONE launch: `jobs` is a device table of PackArgs, `first_block[j]` the first workgroup
// of job j (first_block[njobs] = grid size).  A workgroup finds its job by bisection.
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackArgs* jobs, const int64_t* first_block, int njobs) {
  const int64_t b = blockIdx.x;
  int lo = 0, hi = njobs;              // invariant: first_block[lo] <= b < first_block[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first_block[mid] <= b) lo = mid; else hi = mid;
  }
  const PackArgs a = jobs[lo];
  pack_element(a, (b - first_block[lo]) * 256 + threadIdx.x);
}

// First fold of the epilogue statistics: rows [b*R, (b+1)*R) of partial[rows][n] -> out[b][n].  A workgroup's rows
// are one contiguous range, read with unit stride; n is a power of two (<= 1024), so a thread meets a fixed set of
// n/256 (or one) columns.  bp_sum_partials folds the <= 256 rows that remain.
__global__ __launch_bounds__(256) void stats_fold_kernel(const double* partial, int64_t rows, int64_t R, int n, double* out) {
  __shared__ double sh[256];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  int64_t r1 = r0 + R;
  if (r1 > rows) r1 = rows;
  const double* src = partial + r0 * n;
  const int64_t total = (r1 - r0) * n;
  const int NA = n > 256 ? n / 256 : 1;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int64_t e = tid;
  for (int64_t k = 0; e < total; ++k, e += 256) acc[k & (NA - 1)] += src[e];
  if (n >= 256) {
    for (int q = 0; q < NA; ++q) out[(int64_t)blockIdx.x * n + q * 256 + tid] = acc[q];
    return;
  }
  sh[tid] = acc[0];
  __syncthreads();
  if (tid < n) {
    double t = 0.0;
    for (int i = tid; i < 256; i += n) t += sh[i];
    out[(int64_t)blockIdx.x * n + tid] = t;
  }
}

// Spatial tile of one workgroup and the halo it gathers.
struct TileGeom {
  int TPR, BH, IH, IW, IWq;
};
static TileGeom tile_geom(int TM, int IS, int ISx, int taps, int tapsx) {
  TileGeom t{};
  t.TPR = (TM >= 16) ? 2 : 1;
  t.BH = TM / t.TPR;
  const int BW = 16 * t.TPR;
  t.IH = (t.BH - 1) * IS + taps;
  t.IW = (BW - 1) * ISx + tapsx;
  t.IWq = bp_ceil_div(t.IW, ISx);
  return t;
}

struct IgemmConfig {
  int CC, NT, WN, MT, COB, nchunk, cout_padP;
  TileGeom t;
  int PP, COP, tapsx, ISx;
  size_t lds_bytes;
  bool ok;
  // LDS-DMA variant (igemm_dma_kernel), taken when the view is 16-byte addressable at run time; it shares
  // CC (hence the packed weight image) with the plain kernel above, which stays the fallback.
  bool dma;
  int NW, in_pad4, slots;
  TileGeom td;
  size_t lds_dma;
  // igemm_dmaf_kernel (NT <= 2): taps per step, fused output phases (1 or 4), input buffers
  bool dmaf;
  int ts, nph, in_bufs;
  // igemm_wres_kernel (NT <= 2, one channel chunk, gather form): all tap slabs resident, eight waves x 2 M tiles
  bool wres;
  int w_slots, w_in_pad4, w_NW;
  TileGeom tw;
  size_t lds_wres;
};

IgemmConfig igemm_config(const ConvGeom& g) {
  IgemmConfig c{};
  const int nT = bp_ceil_div(g.cout_g, 16);
  if (nT >= 5) { c.NT = 4; c.WN = 2; }
  else if (nT >= 3) { c.NT = 4; c.WN = 1; }
  else if (nT == 2) { c.NT = 2; c.WN = 1; }
  else { c.NT = 1; c.WN = 1; }
  c.COB = 16 * c.NT * c.WN;
  c.cout_padP = bp_round_up(g.cout_g, c.COB);
  // Few produced channels on a unit-stride grid: pack PP neighbouring pixels into the 16 MFMA columns
  // (column = (pixel-in-group, channel)); costs PP-1 extra taps along x, saves a factor PP of M tiles.
  c.PP = 1; c.COP = 16; c.tapsx = g.taps; c.ISx = g.IS;
  // (not with many gathered channels: the PP-1 extra taps multiply a long K, and the halo forces narrow chunks --
  // the 64->2 and 32->2 k5 latent heads at 16x16 ran 4x slower packed)
  static const int pp_max_cin = getenv("BP_IGEMM_PPCIN") ? atoi(getenv("BP_IGEMM_PPCIN")) : 16;
  if (g.cout_g <= 8 && g.IS == 1 && g.OS == 1 && g.nphase == 1 && g.cin_g <= pp_max_cin) {
    c.COP = g.cout_g <= 1 ? 1 : (g.cout_g <= 2 ? 2 : (g.cout_g <= 4 ? 4 : 8));
    c.PP = 16 / c.COP;
    c.tapsx = g.taps + c.PP - 1;
    c.ISx = c.PP;
  }
  const int cin4 = bp_round_up(g.cin_g, 4);
  const int cc_first = cin4 >= 16 ? 16 : (cin4 >= 8 ? 8 : 4);
  auto plain_lds = [&](const TileGeom& t, int CC) {
    return ((size_t)t.IH * c.ISx * t.IWq * CC + (size_t)c.tapsx * c.COB * CC) * sizeof(float);
  };
  // First choice: the DMA-pipelined kernel (MT 4).  Eight waves per workgroup where the produced-channel
  // block is 128 wide (each weight slab then serves 256 pixels), else four; the chunk width that keeps at
  // least two (eight waves) or three (four waves) workgroups per CU.
  // Thin strided / many-tap gathers: weights resident, persistent (igemm_wres_kernel).
  static const bool no_wres = getenv("BP_IGEMM_NOWRES") != nullptr;
  static const bool wres_all = getenv("BP_IGEMM_WRESALL") != nullptr;
  if (!no_wres && c.NT <= 2 && c.PP <= 2 && g.nphase == 1 && g.OS == 1 && g.cin_g == 16 &&
      (g.IS == 2 || g.taps >= 5 || wres_all)) {
    const int CC = g.cin_g;
    static const bool no_w4 = getenv("BP_WRES_NO4") != nullptr;
    // four waves / 128 pixels where two such workgroups fit a CU (they fill each other's commit phases), else eight
    // waves / 256 pixels with one workgroup per CU
    for (int NW = no_w4 ? 8 : 4; NW <= 8 && !c.wres; NW += 4) {
      const TileGeom tw = tile_geom(2 * NW, g.IS, c.ISx, g.taps, c.tapsx);
      const int E = tw.IH * tw.IW * (CC / 4);
      const int slots = bp_ceil_div(E, 64 * NW);
      const int in_pad4 = bp_round_up(tw.IH * c.ISx * tw.IWq * (CC / 4), 64);
      const size_t lds = (size_t)in_pad4 * 16 + (size_t)g.taps * c.tapsx * c.COB * CC * sizeof(float) + 16;
      const bool plain_ok = plain_lds(tile_geom(16, g.IS, c.ISx, g.taps, c.tapsx), CC) <= 64 * 1024 ||
                            plain_lds(tile_geom(4, g.IS, c.ISx, g.taps, c.tapsx), CC) <= 64 * 1024;
      if (lds <= (size_t)(NW == 4 ? 80 : 150) * 1024 && slots <= 12 && plain_ok) {
        c.wres = true; c.CC = CC; c.tw = tw; c.w_slots = slots <= 8 ? 8 : (slots <= 10 ? 10 : 12); c.w_in_pad4 = in_pad4; c.lds_wres = lds;
        c.w_NW = NW;
      }
    }
  }
  static const bool no_dma = getenv("BP_IGEMM_NODMA") != nullptr;
  static const bool no_nw8 = getenv("BP_IGEMM_NONW8") != nullptr;
  static const bool no_dmaf = getenv("BP_IGEMM_NODMAF") != nullptr;
  static const int pp_dma = getenv("BP_IGEMM_PPDMA") ? atoi(getenv("BP_IGEMM_PPDMA")) : 1;
  if (!c.wres && c.NT <= 2 && c.PP <= pp_dma && !no_dma && !no_dmaf) {   // (pixel-packed heads measured slower here)
    const int T = g.taps * c.tapsx;
    for (int CC = cc_first; CC >= 8 && !c.dma; CC /= 2) {
      if (g.cin_g % CC != 0 || (c.COB * CC) % 256 != 0) continue;
      // all stride^2 phases of a transposed form in one workgroup (one accumulator set per phase)
      const int nph = (g.gather_transposed && g.nphase == 2) ? 4 : 1;
      const int need = bp_ceil_div(64, c.NT * CC);              // taps per step for >= 64 MFMAs per wave
      const int ts = need <= 1 ? 1 : (need <= c.tapsx ? c.tapsx : T);
      TileGeom td = tile_geom(16, g.IS, c.ISx, g.taps, c.tapsx);
      if (nph > 1) {
        const int spread = bp_t_i0(g.nphase - 1, g.pad, g.stride, g.taps) - bp_t_i0(0, g.pad, g.stride, g.taps);
        td.IH += spread; td.IW += spread;
        td.IWq = bp_ceil_div(td.IW, c.ISx);
      }
      const int nchunk = g.cin_g / CC;
      const int in_bufs = nchunk > 1 ? 2 : 1;
      const int E = td.IH * c.ISx * td.IWq * (CC / 4);
      const int in_pad4 = bp_round_up(E, 64);
      const int per_thread = bp_ceil_div(in_pad4, 256);
      const size_t lds_dma = ((size_t)in_bufs * in_pad4 * 4 + (size_t)2 * ts * c.COB * CC + (size_t)3 * g.cin_g) * sizeof(float);
      const TileGeom t4 = tile_geom(16, g.IS, c.ISx, g.taps, c.tapsx);
      const bool plain_ok = plain_lds(t4, CC) <= 64 * 1024 ||
                            plain_lds(tile_geom(4, g.IS, c.ISx, g.taps, c.tapsx), CC) <= 64 * 1024;
      if (lds_dma <= 80 * 1024 && per_thread <= 16 && plain_ok) {
        c.dma = true; c.dmaf = true; c.NW = 4; c.td = td; c.in_pad4 = in_pad4;
        c.slots = per_thread <= 3 ? 3 : (per_thread <= 6 ? 6 : 16);
        c.lds_dma = lds_dma; c.ts = ts; c.nph = nph; c.in_bufs = in_bufs;
        c.CC = CC;
      }
    }
  }
  if (!c.wres && !c.dma && c.PP == 1 && g.taps * c.tapsx >= 2 && !no_dma) {
    for (int NW = (c.NT == 4 && c.WN == 2 && !no_nw8) ? 8 : 4; NW >= 4 && !c.dma; NW -= 4) {
      for (int CC = cc_first; CC >= 8 && !c.dma; CC /= 2) {
        if (g.cin_g % CC != 0 || (c.COB * CC) % 256 != 0) continue;
        const TileGeom td = tile_geom((NW / c.WN) * 4, g.IS, c.ISx, g.taps, c.tapsx);
        const int E = td.IH * c.ISx * td.IWq * (CC / 4);
        const int in_pad4 = bp_round_up(E, 64);
        const size_t lds_in2 = (size_t)2 * in_pad4 * 4 * sizeof(float) + (size_t)3 * g.cin_g * sizeof(float);   // + activation parameters
        const size_t lds_slab = (size_t)c.COB * CC * sizeof(float);
        const size_t lds_cap = (size_t)(NW == 8 ? 80 : 53) * 1024;   // two / three workgroups per CU
        const size_t lds_dma = lds_in2 + 2 * lds_slab;
        const int per_thread = bp_ceil_div(in_pad4, 64 * NW);
        // the plain fallback with this CC must exist too
        const TileGeom t4 = tile_geom((4 / c.WN) * 4, g.IS, c.ISx, g.taps, c.tapsx);
        if (lds_dma <= lds_cap && per_thread <= 6 && plain_lds(t4, CC) <= 64 * 1024) {
          c.dma = true; c.NW = NW; c.td = td; c.in_pad4 = in_pad4; c.slots = per_thread <= 3 ? 3 : 6;
          c.lds_dma = lds_dma;
          c.CC = CC;
        }
      }
    }
  }
  const int mts[2] = {4, 1};
  for (int mi = 0; mi < 2 && !c.ok; ++mi) {
    for (int CC = cc_first; CC >= 4 && !c.ok; CC /= 2) {
      if ((c.dma || c.wres) && CC != c.CC) continue;
      const int MT = mts[mi];
      const TileGeom t = tile_geom((4 / c.WN) * MT, g.IS, c.ISx, g.taps, c.tapsx);
      const size_t lds = plain_lds(t, CC);
      if (lds <= 64 * 1024) {
        c.CC = CC; c.MT = MT; c.t = t;
        c.nchunk = bp_ceil_div(g.cin_g, CC);
        c.lds_bytes = lds; c.ok = true;
      }
    }
  }
  return c;
}

template <int CC, int NT, int WN, int MT>
int launch_one(const IgemmArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((igemm_kernel<CC, NT, WN, MT>), grid, dim3(256), lds, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int MT>
int launch_cc(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  if (c.NT == 4 && c.WN == 2) return launch_one<CC, 4, 2, MT>(a, grid, c.lds_bytes, st);
  if (c.NT == 4 && c.WN == 1) return launch_one<CC, 4, 1, MT>(a, grid, c.lds_bytes, st);
  if (c.NT == 2 && c.WN == 1) return launch_one<CC, 2, 1, MT>(a, grid, c.lds_bytes, st);
  if (c.NT == 1 && c.WN == 1) return launch_one<CC, 1, 1, MT>(a, grid, c.lds_bytes, st);
  return BP_EUNSUPPORTED;
}

template <int CC, int NT, int WN, int SLOTS, int NW>
int launch_dma_one(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&igemm_dma_kernel<CC, NT, WN, SLOTS, NW>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((igemm_dma_kernel<CC, NT, WN, SLOTS, NW>), grid, dim3(64 * NW), c.lds_dma, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int SLOTS>
int launch_dma_cc(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  if (c.NT == 4 && c.WN == 2 && c.NW == 8) return launch_dma_one<CC, 4, 2, SLOTS, 8>(c, a, grid, st);
  if (c.NW != 4) return BP_EUNSUPPORTED;
  if (c.NT == 4 && c.WN == 2) return launch_dma_one<CC, 4, 2, SLOTS, 4>(c, a, grid, st);
  if (c.NT == 4 && c.WN == 1) return launch_dma_one<CC, 4, 1, SLOTS, 4>(c, a, grid, st);
  if (c.NT == 2 && c.WN == 1) return launch_dma_one<CC, 2, 1, SLOTS, 4>(c, a, grid, st);
  if (c.NT == 1 && c.WN == 1) return launch_dma_one<CC, 1, 1, SLOTS, 4>(c, a, grid, st);
  return BP_EUNSUPPORTED;
}

template <int CC, int NT, int SLOTS, int NPH>
int launch_dmaf_one(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&igemm_dmaf_kernel<CC, NT, SLOTS, NPH>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((igemm_dmaf_kernel<CC, NT, SLOTS, NPH>), grid, dim3(256), c.lds_dma, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int NT>
int launch_dmaf_nt(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  if (c.nph == 4) {
    if (c.slots == 3) return launch_dmaf_one<CC, NT, 3, 4>(c, a, grid, st);
    if (c.slots == 6) return launch_dmaf_one<CC, NT, 6, 4>(c, a, grid, st);
    return launch_dmaf_one<CC, NT, 16, 4>(c, a, grid, st);
  }
  if (c.slots == 3) return launch_dmaf_one<CC, NT, 3, 1>(c, a, grid, st);
  if (c.slots == 6) return launch_dmaf_one<CC, NT, 6, 1>(c, a, grid, st);
  return launch_dmaf_one<CC, NT, 16, 1>(c, a, grid, st);
}

int launch_dmaf(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  if (c.CC == 16 && c.NT == 1) return launch_dmaf_nt<16, 1>(c, a, grid, st);
  if (c.CC == 16 && c.NT == 2) return launch_dmaf_nt<16, 2>(c, a, grid, st);
  if (c.CC == 8 && c.NT == 2) return launch_dmaf_nt<8, 2>(c, a, grid, st);
  return BP_EUNSUPPORTED;
}

int launch_dma(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  if (c.CC == 16) return c.slots == 3 ? launch_dma_cc<16, 3>(c, a, grid, st) : launch_dma_cc<16, 6>(c, a, grid, st);
  if (c.CC == 8) return c.slots == 3 ? launch_dma_cc<8, 3>(c, a, grid, st) : launch_dma_cc<8, 6>(c, a, grid, st);
  return BP_EUNSUPPORTED;
}

template <int CC, int NT, int SLOTS, int NW, bool STATS>
int launch_wres_st(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, int ntiles, int per_block, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&igemm_wres_kernel<CC, NT, 2, NW, SLOTS, STATS>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (NW == 4 ? 80 : 150) * 1024);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((igemm_wres_kernel<CC, NT, 2, NW, SLOTS, STATS>), grid, dim3(64 * NW), c.lds_wres, st, a, ntiles,
                     per_block);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int CC, int NT, int SLOTS, int NW>
int launch_wres_one(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, int ntiles, int per_block, hipStream_t st) {
  if (a.stat) return launch_wres_st<CC, NT, SLOTS, NW, true>(c, a, grid, ntiles, per_block, st);
  return launch_wres_st<CC, NT, SLOTS, NW, false>(c, a, grid, ntiles, per_block, st);
}

int launch_wres(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, int ntiles, int per_block, hipStream_t st) {
#define BP_WRES(CCv, NTv) \
  if (c.CC == CCv && c.NT == NTv) { \
    if (c.w_NW == 4) { \
      if (c.w_slots == 8) return launch_wres_one<CCv, NTv, 8, 4>(c, a, grid, ntiles, per_block, st); \
      if (c.w_slots == 10) return launch_wres_one<CCv, NTv, 10, 4>(c, a, grid, ntiles, per_block, st); \
      return launch_wres_one<CCv, NTv, 12, 4>(c, a, grid, ntiles, per_block, st); \
    } \
    if (c.w_slots == 8) return launch_wres_one<CCv, NTv, 8, 8>(c, a, grid, ntiles, per_block, st); \
    if (c.w_slots == 10) return launch_wres_one<CCv, NTv, 10, 8>(c, a, grid, ntiles, per_block, st); \
    return launch_wres_one<CCv, NTv, 12, 8>(c, a, grid, ntiles, per_block, st); \
  }
  BP_WRES(16, 1) BP_WRES(16, 2)
#undef BP_WRES
  return BP_EUNSUPPORTED;
}

template <int MT>
int launch_mt(const IgemmConfig& c, const IgemmArgs& a, dim3 grid, hipStream_t st) {
  switch (c.CC) {
    case 16: return launch_cc<16, MT>(c, a, grid, st);
    case 8: return launch_cc<8, MT>(c, a, grid, st);
    case 4: return launch_cc<4, MT>(c, a, grid, st);
  }
  return BP_EUNSUPPORTED;
}

}  // namespace

// conv_small.hip: vector-ALU kernel for unit-stride layers with cin*cout <= 8
bool bp_small_ok(const ConvGeom& g);
int64_t bp_small_packed_floats(const ConvGeom& g);
int bp_small_kernel_id(const ConvGeom& g);
int bp_small_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_small_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode);
int bp_small_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                 const bp_view* out, hipStream_t st, const IgemmStatsReq* sr);

// conv_stem.hip: the 3 -> 16 k5 stem (flattened (tap column, channel) K, weights in registers)
bool bp_stem_ok(const ConvGeom& g);
int64_t bp_stem_packed_floats();
int bp_stem_pack(const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_stem_stats_workspace(const bp_view* out);
int bp_stem_run(const bp_view* in, const PW& pw, const float* packed, const float* bias, const bp_view* out,
                hipStream_t st, const IgemmStatsReq* sr);

// conv_flat.hip: unit-stride k7, 8 gathered -> 16 produced channels (weights in registers, flattened K)
bool bp_flat_ok(const ConvGeom& g);
int64_t bp_flat_packed_floats();
int bp_flat_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_flat_stats_workspace(const bp_view* out, int mode);
int bp_flat_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                const bp_view* out, hipStream_t st, const IgemmStatsReq* sr);
// ... and the stride-2 k4 transposed form 32 -> 16 (four phases, all weights in registers)
bool bp_flat_t4_ok(const ConvGeom& g);
int64_t bp_flat_t4_packed_floats();
int bp_flat_t4_pack(const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_flat_t4_stats_workspace(const bp_view* out);
int bp_flat_t4_run(const bp_view* in, const PW& pw, const float* packed, const float* bias, const bp_view* out,
                   hipStream_t st, const IgemmStatsReq* sr);
// ... and the stride-2 k4 conv form 32 -> 64 (eight waves: four blocks of 16 produced channels, weights in registers)
bool bp_flat_g4_ok(const ConvGeom& g);
int64_t bp_flat_g4_packed_floats(const ConvGeom& g);
int bp_flat_g4_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_flat_g4_stats_workspace(const bp_view* out);
int bp_flat_g4_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                   const bp_view* out, hipStream_t st, const IgemmStatsReq* sr);

// ... and the stride-2 k4 transposed form 64 -> 32 (eight waves: four phases x two blocks of 16 produced channels)
bool bp_flat_t64_ok(const ConvGeom& g);
int64_t bp_flat_t64_packed_floats(const ConvGeom& g);
int bp_flat_t64_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_flat_t64_stats_workspace(const bp_view* out);
int bp_flat_t64_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                    const bp_view* out, hipStream_t st, const IgemmStatsReq* sr);

// ... and unit-stride k7 16 -> 8 (the head's first layer forward: K split over two waves, pixel pairs per MFMA column)
bool bp_flat_h7_ok(const ConvGeom& g);
int64_t bp_flat_h7_packed_floats();
int bp_flat_h7_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
int bp_flat_h7_run(const bp_view* in, const PW& pw, const float* packed, const float* bias, const bp_view* out,
                   hipStream_t st);

// ... and the k8 stride-4 layer 8 -> 16 of the recognition / prior networks, forward and data gradient (conv_enc.hip)
bool bp_enc_ok(const ConvGeom& g);
int bp_enc_kernel_id(const ConvGeom& g);
int64_t bp_enc_packed_floats(const ConvGeom& g);
int bp_enc_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_enc_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode);
int bp_enc_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
               const bp_view* out, hipStream_t st, const IgemmStatsReq* sr);

int bp_igemm_kernel_id(const ConvGeom& g) {
  if (bp_enc_ok(g)) return bp_enc_kernel_id(g);
  if (bp_stem_ok(g)) return 700000;
  if (bp_flat_ok(g)) return 710000;
  if (bp_flat_t4_ok(g) && !bp_flat_t64_ok(g)) return 720000;
  if (bp_flat_g4_ok(g)) return 730000;
  if (bp_flat_t64_ok(g)) return 740000;
  if (bp_flat_h7_ok(g)) return 750000;
  if (bp_small_ok(g)) return bp_small_kernel_id(g);
  const IgemmConfig c = igemm_config(g);
  if (c.ok && c.wres) return 400000 + c.CC * 1000 + c.NT * 100 + (c.w_NW / 4) * 10 + 2;
  return c.ok ? (c.dma ? 100000 * (c.dmaf ? 3 : c.NW / 4) : 0) + c.CC * 1000 + c.NT * 100 + c.WN * 10 + c.MT : -1;
}

int64_t bp_igemm_packed_floats(const ConvGeom& g) {
  if (bp_enc_ok(g)) return bp_enc_packed_floats(g);
  if (bp_stem_ok(g)) return bp_stem_packed_floats();
  if (bp_flat_ok(g)) return bp_flat_packed_floats();
  if (bp_flat_t4_ok(g) && !bp_flat_t64_ok(g)) return bp_flat_t4_packed_floats();
  if (bp_flat_g4_ok(g)) return bp_flat_g4_packed_floats(g);
  if (bp_flat_t64_ok(g)) return bp_flat_t64_packed_floats(g);
  if (bp_flat_h7_ok(g)) return bp_flat_h7_packed_floats();
  if (bp_small_ok(g)) return bp_small_packed_floats(g);
  const IgemmConfig c = igemm_config(g);
  if (!c.ok) return -1;
  return (int64_t)g.nphase * g.nphase * g.taps * c.tapsx * c.nchunk * c.cout_padP * c.CC;
}

static bool igemm_pack_args(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, PackArgs& a) {
  const IgemmConfig c = igemm_config(g);
  if (!c.ok) return false;
  a = PackArgs{};
  a.w = w_torch; a.dst = packed; a.sa = wm.sa; a.sb = wm.sb;
  a.k = g.k; a.stride = g.stride; a.pad = g.pad; a.tapsy = g.taps; a.tapsx = c.tapsx; a.nphase = g.nphase;
  a.transposed = g.gather_transposed; a.cin_g = g.cin_g; a.cout_g = g.cout_g;
  a.CC = c.CC; a.nchunk = c.nchunk; a.cout_padP = c.cout_padP; a.PP = c.PP; a.COP = c.COP;
  a.total = bp_igemm_packed_floats(g);
  return true;
}

int bp_igemm_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed,
                  hipStream_t st) {
  if (bp_enc_ok(g)) return bp_enc_pack(g, wm, w_torch, packed, st);
  if (bp_stem_ok(g)) return bp_stem_pack(wm, w_torch, packed, st);
  if (bp_flat_ok(g)) return bp_flat_pack(g, wm, w_torch, packed, st);
  if (bp_flat_t4_ok(g) && !bp_flat_t64_ok(g)) return bp_flat_t4_pack(wm, w_torch, packed, st);
  if (bp_flat_g4_ok(g)) return bp_flat_g4_pack(g, wm, w_torch, packed, st);
  if (bp_flat_t64_ok(g)) return bp_flat_t64_pack(g, wm, w_torch, packed, st);
  if (bp_flat_h7_ok(g)) return bp_flat_h7_pack(g, wm, w_torch, packed, st);
  if (bp_small_ok(g)) return bp_small_pack(g, wm, w_torch, packed, st);
  PackArgs a;
  if (!igemm_pack_args(g, wm, w_torch, packed, a)) return BP_EUNSUPPORTED;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// Batched packing (bp_conv_pack_job / bp_conv_pack_jobs): the job record is the kernel's own argument block.
size_t bp_igemm_pack_job_bytes() { return sizeof(PackArgs); }

int bp_igemm_pack_job(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, void* job,
                      int64_t* nblocks) {
  if (bp_enc_ok(g) || bp_stem_ok(g) || bp_flat_ok(g) || bp_flat_t4_ok(g) || bp_flat_g4_ok(g) || bp_flat_t64_ok(g) || bp_flat_h7_ok(g) || bp_small_ok(g)) return BP_EUNSUPPORTED;       // (their own tiny pack kernels: packed by bp_conv_pack)
  PackArgs a;
  if (!igemm_pack_args(g, wm, w_torch, packed, a)) return BP_EUNSUPPORTED;
  *reinterpret_cast<PackArgs*>(job) = a;
  *nblocks = (a.total + 255) / 256;
  return BP_OK;
}

int bp_igemm_pack_jobs(const void* jobs_dev, const int64_t* first_block_dev, int njobs, int64_t total_blocks,
                       hipStream_t st) {
  if (njobs <= 0 || total_blocks <= 0) return BP_OK;
  hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st,
                     reinterpret_cast<const PackArgs*>(jobs_dev), first_block_dev, njobs);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// Launch shape of a layer on these views: which kernel family, its tile and its grid.
struct IgemmLaunch { bool dma, wres; TileGeom t; dim3 grid; int vec_ok, out_vec; size_t lds; int WM; int ntiles, per_block; };
static bool igemm_launch_of(const ConvGeom& g, const IgemmConfig& c, const bp_view* in, const bp_view* out, IgemmLaunch& l) {
  l.vec_ok = (in->cstride % 4 == 0 && in->coff % 4 == 0 && (reinterpret_cast<uintptr_t>(in->ptr) % 16 == 0)) ? 1 : 0;
  l.out_vec = (out->cstride % 4 == 0 && out->coff % 4 == 0 && reinterpret_cast<uintptr_t>(out->ptr) % 16 == 0) ? 1 : 0;
  l.dma = c.dma && l.vec_ok;
  l.wres = c.wres && l.vec_ok;
  l.t = l.wres ? c.tw : (l.dma ? c.td : c.t);
  l.ntiles = l.per_block = 0;
  const int qh = bp_ceil_div(out->h, g.OS), qw = c.PP > 1 ? bp_ceil_div(out->w, c.PP) : bp_ceil_div(out->w, g.OS);
  const int tiles_x = bp_ceil_div(qw, 16 * l.t.TPR), tiles_y = bp_ceil_div(qh, l.t.BH);
  const int64_t gz = (int64_t)in->n * ((l.dma && c.dmaf && c.nph > 1) ? 1 : g.nphase * g.nphase);
  if (gz > 65535 || c.cout_padP / c.COB > 65535) return false;
  l.grid = dim3((unsigned)(tiles_x * tiles_y), (unsigned)gz, (unsigned)(c.cout_padP / c.COB));
  l.lds = l.dma ? c.lds_dma : c.lds_bytes;
  l.WM = l.dma ? (c.dmaf ? 4 : c.NW / c.WN) : 4 / c.WN;
  if (l.wres) {      // persistent: ~one workgroup per CU and channel block, each with a contiguous run of tiles
    const int64_t nt = (int64_t)tiles_x * tiles_y * in->n;
    if (nt > 0x7fffffff) return false;
    static const int cus = getenv("BP_WRES_GRID") ? atoi(getenv("BP_WRES_GRID")) : 256;
    int nb = cus * (c.w_NW == 4 ? 2 : 1) / (int)l.grid.z;
    if (nb < 1) nb = 1;
    if (nb > nt) nb = (int)nt;
    l.ntiles = (int)nt;
    l.per_block = (int)((nt + nb - 1) / nb);
    l.grid = dim3((unsigned)((nt + l.per_block - 1) / l.per_block), 1, l.grid.z);
    l.lds = c.lds_wres; l.WM = c.w_NW;
  }
  return true;
}

// Epilogue statistics (IgemmArgs::stat): partial rows, their first fold, and whether this layer's kernel has them.
struct StatsPlan { int64_t rows; int n; int nfold; int64_t R; size_t bytes; };
static void stats_fold_plan(StatsPlan& p) {
  // up to this many rows the last stage adds them itself (a wave per column, 32 loads per lane): one launch less per
  // batch-norm layer than fold + sum -- 123 -> 116 us for a trunk layer's forward, generator forward -0.1 ms; beyond
  // (the full-resolution layers' 8 ... 16 k rows) the fold pays
  static const int64_t direct = getenv("BP_STATS_DIRECT_ROWS") ? atoll(getenv("BP_STATS_DIRECT_ROWS")) : 2048;
  p.nfold = p.rows <= direct ? 0 : (int)(p.rows / 32 < 256 ? (p.rows + 31) / 32 : 256);
  p.R = p.nfold ? (p.rows + p.nfold - 1) / p.nfold : 0;
  if (p.nfold) p.nfold = (int)((p.rows + p.R - 1) / p.R);
  p.bytes = (size_t)(p.rows + p.nfold) * p.n * sizeof(double);
}
static bool stats_plan(const ConvGeom& g, const IgemmConfig& c, const IgemmLaunch& l, StatsPlan& p) {
  const int C = g.cout_g;
  if (c.PP != 1 || C <= 0 || (C & (C - 1)) != 0 || C > 512) return false;
  if ((size_t)l.WM * c.COB * 2 * sizeof(double) > l.lds) return false;
  p.rows = (int64_t)l.grid.x * l.grid.y;
  p.n = 2 * C;
  stats_fold_plan(p);
  return true;
}
static int stats_finish(const StatsPlan& sp, double* ws, const IgemmStatsReq* sr, hipStream_t st) {
  const double* rows = ws;
  int64_t nrows = sp.rows;
  if (sp.nfold) {
    double* folded = ws + sp.rows * sp.n;
    hipLaunchKernelGGL(stats_fold_kernel, dim3((unsigned)sp.nfold), dim3(256), 0, st, rows, sp.rows, sp.R, sp.n, folded);
    BP_CHECK_LAUNCH();
    rows = folded; nrows = sp.nfold;
  }
  return bp_sum_partials_req(rows, (int)nrows, sp.n, sr, st);
}

// Partial rows [rows][2*C] written by some kernel's epilogue -> sums[2*C] (used by the bf16 kernels too):
// bytes of workspace for the rows and their first fold, and the fold itself.
size_t bp_stats_rows_bytes(int64_t rows, int C) {
  if (C <= 0 || (C & (C - 1)) != 0 || C > 512) return 0;
  StatsPlan p{};
  p.rows = rows; p.n = 2 * C;
  stats_fold_plan(p);
  return p.bytes;
}
// ... the same for rows of which n doubles are wanted (the three sums of an activation backward: n = 3 C).  The fold
// needs a power-of-two row: the kernel writes rows of bp_stats_row_stride(n) doubles, zero beyond n.
int bp_stats_row_stride(int n) {
  int s = 1;
  while (s < n) s <<= 1;
  return s;
}
size_t bp_stats_rows_bytes_n(int64_t rows, int n) {
  StatsPlan p{};
  p.rows = rows; p.n = bp_stats_row_stride(n);
  stats_fold_plan(p);
  return p.bytes;
}
int bp_stats_rows_finish_n(double* ws, int64_t rows, int n, const IgemmStatsReq* sr, hipStream_t st) {
  StatsPlan p{};
  p.rows = rows; p.n = bp_stats_row_stride(n);
  stats_fold_plan(p);
  const double* src = ws;
  int64_t nrows = p.rows;
  if (p.nfold) {
    double* folded = ws + p.rows * p.n;
    hipLaunchKernelGGL(stats_fold_kernel, dim3((unsigned)p.nfold), dim3(256), 0, st, src, p.rows, p.R, p.n, folded);
    BP_CHECK_LAUNCH();
    src = folded; nrows = p.nfold;
  }
  return bp_sum_partials_strided(src, (int)nrows, p.n, n, sr->sums, st);
}
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st) {
  StatsPlan p{};
  p.rows = rows; p.n = 2 * C;
  stats_fold_plan(p);
  return stats_finish(p, ws, sr, st);
}

// conv_ws_f32.hip: weights-stationary kernel of the 128 -> 128 k3 trunk layers (reads the tiled image packed above)
bool bp_f32_ws_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int stats_mode);
size_t bp_f32_ws_stats_workspace(const ConvGeom& g, const bp_view* out);
int bp_f32_ws_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed_tiled, const bp_view* out,
                  hipStream_t st, const IgemmStatsReq* sr);

size_t bp_igemm_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode) {
  if (bp_f32_ws_ok(g, in, out, nullptr, mode)) {        // (the larger of the two: bp_set_option may switch kernels later)
    const IgemmConfig c = igemm_config(g);
    IgemmLaunch l;
    StatsPlan p;
    const size_t tiled = (c.ok && igemm_launch_of(g, c, in, out, l) && stats_plan(g, c, l, p)) ? p.bytes : 0;
    const size_t ws = bp_f32_ws_stats_workspace(g, out);
    return ws > tiled ? ws : tiled;
  }
  if (bp_enc_ok(g)) return bp_enc_stats_workspace(g, in, out, mode);
  if (bp_stem_ok(g)) return mode == 1 ? bp_stem_stats_workspace(out) : 0;
  if (bp_flat_ok(g)) return bp_flat_stats_workspace(out, mode);
  if (bp_flat_t4_ok(g) && !bp_flat_t64_ok(g)) return mode == 1 ? bp_flat_t4_stats_workspace(out) : 0;
  if (bp_flat_g4_ok(g)) return mode == 1 ? bp_flat_g4_stats_workspace(out) : 0;
  if (bp_flat_t64_ok(g)) return mode == 1 ? bp_flat_t64_stats_workspace(out) : 0;
  if (bp_flat_h7_ok(g)) return 0;
  if (bp_small_ok(g)) return bp_small_stats_workspace(g, in, out, mode);
  if (mode != 1 && mode != 2) return 0;
  const IgemmConfig c = igemm_config(g);
  IgemmLaunch l;
  StatsPlan p;
  if (!c.ok || !igemm_launch_of(g, c, in, out, l) || !stats_plan(g, c, l, p)) return 0;
  if (l.wres && mode != 1) return 0;
  return p.bytes;
}

int bp_igemm_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed,
                 const float* bias, const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  if (bp_enc_ok(g)) return bp_enc_run(g, in, pw, packed, bias, out, st, sr);
  if (bp_stem_ok(g)) return bp_stem_run(in, pw, packed, bias, out, st, sr);
  if (bp_flat_ok(g)) return bp_flat_run(g, in, pw, packed, bias, out, st, sr);
  if (bp_flat_t4_ok(g) && !bp_flat_t64_ok(g)) return bp_flat_t4_run(in, pw, packed, bias, out, st, sr);
  if (bp_flat_g4_ok(g)) return bp_flat_g4_run(g, in, pw, packed, bias, out, st, sr);
  if (bp_flat_t64_ok(g)) return bp_flat_t64_run(g, in, pw, packed, bias, out, st, sr);
  if (bp_flat_h7_ok(g)) return sr ? BP_EUNSUPPORTED : bp_flat_h7_run(in, pw, packed, bias, out, st);
  if (bp_small_ok(g)) return bp_small_run(g, in, pw, packed, bias, out, st, sr);
  const IgemmConfig c = igemm_config(g);
  if (!c.ok) return BP_EUNSUPPORTED;
  if (c.CC == 16 && c.PP == 1 && c.cout_padP == 128 && bp_f32_ws_ok(g, in, out, bias, sr ? sr->mode : 0))
    return bp_f32_ws_run(g, in, pw, packed, out, st, sr);
  IgemmLaunch l;
  if (!igemm_launch_of(g, c, in, out, l)) return BP_EUNSUPPORTED;
  IgemmArgs a{};
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff; a.cin = g.cin_g;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.cout = g.cout_g; a.wp = packed; a.bias = bias; a.pw = pw;
  a.tapsy = g.taps; a.tapsx = c.tapsx; a.ISy = g.IS; a.ISx = c.ISx; a.OS = g.OS; a.nphase = g.nphase;
  a.transposed = g.gather_transposed; a.PP = c.PP; a.COP = c.COP;
  a.stride = g.stride; a.pad = g.pad;
  a.nchunk = c.nchunk; a.cout_padP = c.cout_padP;
  a.vec_ok = l.vec_ok; a.out_vec = l.out_vec;
  const bool dma = l.dma;
  const TileGeom& t = l.t;
  a.TPR = t.TPR; a.BH = t.BH; a.IH = t.IH; a.IW = t.IW; a.IWq = t.IWq;
  const int qh = bp_ceil_div(out->h, g.OS), qw = c.PP > 1 ? bp_ceil_div(out->w, c.PP) : bp_ceil_div(out->w, g.OS);
  a.tiles_x = bp_ceil_div(qw, 16 * t.TPR);
  a.tiles_y = bp_ceil_div(qh, t.BH);
  const dim3 grid = l.grid;
  a.in_pad4 = l.wres ? c.w_in_pad4 : c.in_pad4; a.ts = c.ts; a.in_bufs = c.in_bufs;
  StatsPlan sp{};
  if (sr) {
    if (bias || (sr->mode != 1 && sr->mode != 2) || !stats_plan(g, c, l, sp) || (l.wres && sr->mode != 1)) return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < sp.bytes || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws); a.stat_c = g.cout_g; a.stat_mode = sr->mode;
    if (sr->mode == 2) {
      const bp_view* r = sr->raw;
      a.raw = r->ptr; a.raw_cs = r->cstride; a.raw_co = r->coff; a.spw = sr->spw;
      a.raw_vec = (r->cstride % 4 == 0 && r->coff % 4 == 0 && reinterpret_cast<uintptr_t>(r->ptr) % 16 == 0) ? 1 : 0;
    }
  }
  int rc;
  if (l.wres) rc = launch_wres(c, a, grid, l.ntiles, l.per_block, st);
  else if (dma && c.dmaf) rc = launch_dmaf(c, a, grid, st);
  else if (dma) rc = launch_dma(c, a, grid, st);
  else if (c.MT == 4) rc = launch_mt<4>(c, a, grid, st);
  else rc = launch_mt<1>(c, a, grid, st);
  if (rc != BP_OK || !sr) return rc;
  return stats_finish(sp, a.stat, sr, st);
}
