// Implicit-GEMM direct convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC.
//
// One kernel serves Conv2d forward, ConvTranspose2d forward and both data gradients: each is a
// stride-IS correlation over an output sub-grid ("phase", see ConvGeom in common.hpp).
//   GEMM view:  M = 16 consecutive output pixels of one row,  N = 16 produced channels,
//               K = (tap, gathered channel), 4 channels per MFMA.
// A workgroup (4 waves) owns BH x BW output pixels x COB produced channels.  Per channel chunk CC
// it stages the input halo tile through registers into LDS -- applying the producer's pending
// batch-norm affine + leaky-ReLU on the way, zero padding AFTER the activation as torch does --
// and per tap row the matching slab of pre-packed weights.  LDS images:
//   input  [row][x % IS][x / IS][CC]    (x de-interleaved by the stride so that the 16 pixels of
//                                        an M tile are contiguous for every tap: conflict-free)
//   weight [tx][COB][CC]
// Lane l supplies k-slice (l>>4): it reads CC/4 consecutive channels (b32/b64/b128) of its pixel
// (A) and of its produced channel (B) and issues CC/4 MFMAs from them.
#include "common.hpp"

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

template <int CC, int NT, int WN, int MT>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs a) {
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

  const int tile = blockIdx.x;
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int co0 = blockIdx.y * COB;
  const int ph = blockIdx.z % (a.nphase * a.nphase);
  const int n = blockIdx.z / (a.nphase * a.nphase);
  const int py = ph / a.nphase, px = ph % a.nphase;

  const int BW = 16 * a.TPR;
  const int qy0 = tile_y * a.BH, qx0 = tile_x * BW;
  // phase grid extents
  const int qh = (a.out_h - py + a.OS - 1) / a.OS;
  const int qw = a.PP > 1 ? (a.out_w + a.PP - 1) / a.PP : (a.out_w - px + a.OS - 1) / a.OS;   // groups
  if (qy0 >= qh || qx0 >= qw) return;  // uniform per block

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

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();  // previous chunk's readers are done with lds_in / lds_w
    // ---- stage the input halo tile for channels [chunk*CC, chunk*CC+CC)
    {
      const int c4 = tid % VW;                   // 256 % VW == 0: fixed channel quad per thread
      const int ch = chunk * CC + c4 * 4;
      const PW4 p4 = pw4_load(a.pw, ch, a.cin);
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
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][s], bf[nt][s], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: D[row = 4*(lane>>4)+r (pixel or pixel group)][col = lane&15 (channel, or (pp,channel))]
  float* out_n = a.out + (int64_t)n * a.out_h * a.out_w * a.out_cs + a.out_co;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = wm * MT + mt;
    const int tr = t / a.TPR, tc = t % a.TPR;
    const int qy = qy0 + tr;
    const int Y = py + a.OS * qy;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = co0 + (wn * NT + nt) * 16 + lm;
      const int co = a.PP > 1 ? j % a.COP : j;
      const int pp = a.PP > 1 ? j / a.COP : 0;
      const float b = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qx = qx0 + tc * 16 + kq * 4 + r;
        const int X = a.PP > 1 ? a.PP * qx + pp : px + a.OS * qx;
        if (qy < qh && qx < qw && X < a.out_w && co < a.cout)
          out_n[((int64_t)Y * a.out_w + X) * a.out_cs + co] = acc[mt][nt][r] + b;
      }
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

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
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

struct IgemmConfig {
  int CC, NT, WN, MT, TPR, BH, COB, nchunk, cout_padP, IH, IW, IWq;
  int PP, COP, tapsx, ISx;
  size_t lds_bytes;
  bool ok;
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
  if (g.cout_g <= 8 && g.IS == 1 && g.OS == 1 && g.nphase == 1) {
    c.COP = g.cout_g <= 1 ? 1 : (g.cout_g <= 2 ? 2 : (g.cout_g <= 4 ? 4 : 8));
    c.PP = 16 / c.COP;
    c.tapsx = g.taps + c.PP - 1;
    c.ISx = c.PP;
  }
  const int cin4 = bp_round_up(g.cin_g, 4);
  const int cc_first = cin4 >= 16 ? 16 : (cin4 >= 8 ? 8 : 4);
  const int mts[2] = {4, 1};
  for (int mi = 0; mi < 2 && !c.ok; ++mi) {
    for (int CC = cc_first; CC >= 4 && !c.ok; CC /= 2) {
      const int MT = mts[mi];
      const int TM = (4 / c.WN) * MT;
      const int TPR = (TM >= 16) ? 2 : 1;
      const int BH = TM / TPR, BW = 16 * TPR;
      const int IH = (BH - 1) * g.IS + g.taps, IW = (BW - 1) * c.ISx + c.tapsx;
      const int IWq = bp_ceil_div(IW, c.ISx);
      const size_t lds = ((size_t)IH * c.ISx * IWq * CC + (size_t)c.tapsx * c.COB * CC) * sizeof(float);
      if (lds <= 64 * 1024) {
        c.CC = CC; c.MT = MT; c.TPR = TPR; c.BH = BH; c.IH = IH; c.IW = IW; c.IWq = IWq;
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

int bp_igemm_kernel_id(const ConvGeom& g) {
  const IgemmConfig c = igemm_config(g);
  return c.ok ? c.CC * 1000 + c.NT * 100 + c.WN * 10 + c.MT : -1;
}

int64_t bp_igemm_packed_floats(const ConvGeom& g) {
  const IgemmConfig c = igemm_config(g);
  if (!c.ok) return -1;
  return (int64_t)g.nphase * g.nphase * g.taps * c.tapsx * c.nchunk * c.cout_padP * c.CC;
}

int bp_igemm_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed,
                  hipStream_t st) {
  const IgemmConfig c = igemm_config(g);
  if (!c.ok) return BP_EUNSUPPORTED;
  PackArgs a{};
  a.w = w_torch; a.dst = packed; a.sa = wm.sa; a.sb = wm.sb;
  a.k = g.k; a.stride = g.stride; a.pad = g.pad; a.tapsy = g.taps; a.tapsx = c.tapsx; a.nphase = g.nphase;
  a.transposed = g.gather_transposed; a.cin_g = g.cin_g; a.cout_g = g.cout_g;
  a.CC = c.CC; a.nchunk = c.nchunk; a.cout_padP = c.cout_padP; a.PP = c.PP; a.COP = c.COP;
  a.total = bp_igemm_packed_floats(g);
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

int bp_igemm_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed,
                 const float* bias, const bp_view* out, hipStream_t st) {
  const IgemmConfig c = igemm_config(g);
  if (!c.ok) return BP_EUNSUPPORTED;
  IgemmArgs a{};
  a.in = in->ptr; a.in_h = in->h; a.in_w = in->w; a.in_cs = in->cstride; a.in_co = in->coff; a.cin = g.cin_g;
  a.out = out->ptr; a.out_h = out->h; a.out_w = out->w; a.out_cs = out->cstride; a.out_co = out->coff;
  a.cout = g.cout_g; a.wp = packed; a.bias = bias; a.pw = pw;
  a.tapsy = g.taps; a.tapsx = c.tapsx; a.ISy = g.IS; a.ISx = c.ISx; a.OS = g.OS; a.nphase = g.nphase;
  a.transposed = g.gather_transposed; a.PP = c.PP; a.COP = c.COP;
  a.stride = g.stride; a.pad = g.pad;
  a.TPR = c.TPR; a.BH = c.BH; a.nchunk = c.nchunk; a.cout_padP = c.cout_padP;
  a.IH = c.IH; a.IW = c.IW; a.IWq = c.IWq;
  a.vec_ok = (in->cstride % 4 == 0 && in->coff % 4 == 0 &&
              (reinterpret_cast<uintptr_t>(in->ptr) % 16 == 0)) ? 1 : 0;
  const int qh = bp_ceil_div(out->h, g.OS), qw = c.PP > 1 ? bp_ceil_div(out->w, c.PP) : bp_ceil_div(out->w, g.OS);
  a.tiles_x = bp_ceil_div(qw, 16 * c.TPR);
  a.tiles_y = bp_ceil_div(qh, c.BH);
  const int64_t gz = (int64_t)in->n * g.nphase * g.nphase;
  if (gz > 65535 || c.cout_padP / c.COB > 65535) return BP_EUNSUPPORTED;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)(c.cout_padP / c.COB), (unsigned)gz);
  if (c.MT == 4) return launch_mt<4>(c, a, grid, st);
  return launch_mt<1>(c, a, grid, st);
}
