// Data gradient of the heads' 8 -> 1 k5 layer (p_mu_out.2 / p_var_out.2, architecture_built.txt:104) in throughput mode:
// one fp32 gradient channel in, the 8-channel slot of the head (stored as bf16) out, with the producer's PReLU backward in
// the epilogue.  On the vector ALUs (small_conv_kernel<5, 1, 8, 16, false, true>) this launch was 0.34 ms of a 12 ms
// step -- 200 FMAs per pixel behind LDS reads -- for 0.6 GB of traffic.
//
// Matrix form: the layer gathers ONE channel, so the K index of an MFMA carries the tap window instead of channels:
//     K = (tap-row slot kg = 0..3, window column j = 0..7)      B[(kg, j), pixel x] = dy[r0 + kg, x - 2 + j]
//     M = (output row select rs = 0, 1; produced channel c)     A[(rs, c), (kg, j)] = w[c][4 - tap row][4 - j]
// (taps reversed: the data gradient of a correlation; j = 5..7 and tap rows outside 0..4 carry zeros).  Two K-steps cover
// the six gradient rows under a PAIR of output rows: 2 MFMAs per 16 pixels x 2 rows x 8 channels, 7 us of matrix time
// for the whole tensor: the kernel is one read of dy, one read of the slot's raw values and one bf16 store.
// A lane ends up with 4 consecutive channels of one pixel: 8-byte loads of the raw values, 8-byte stores.  (A variant with
// four output rows per pass -- two M tiles = the two channel halves, no dead K rows, 16-byte loads / stores -- measured
// 0.23 ms against this one's 0.18.)
// The window of a pixel starts at ANY 2-byte offset of the staged gradient row, so a fragment is five 2-byte LDS reads
// (the tile is 3 KB; the reads are a few per cent of the kernel).
#include "conv_bf16.hpp"

using namespace bpbf16;

// conv_igemm.hip: rows of n epilogue sums per workgroup -> sr->sums
int bp_stats_row_stride(int n);
size_t bp_stats_rows_bytes_n(int64_t rows, int n);
int bp_stats_rows_finish_n(double* ws, int64_t rows, int n, const IgemmStatsReq* sr, hipStream_t st);

namespace {

constexpr int HD_K = 5, HD_C = 8, HD_TW = 64, HD_TH = 16, HD_LW = HD_TW + 8, HD_LH = HD_TH + HD_K - 1;
constexpr int HD_PACKED = 2 * 64 * 8;

struct HdArgs {
  const float* in; int h, w, in_cs, in_co;        // dy: one channel
  u16* out; int out_cs, out_co;                   // 8 channels, bf16
  const u16* wp;
  int tiles_x, tiles_y, n;
  // ACT: out = g = d * act'(spw(raw)), rows {sum g, sum g*raw, sum_{t<=0} d*t} per workgroup (as small_conv_kernel)
  const u16* raw; int raw_cs, raw_co;
  PW spw;
  double* stat; int stat_stride;
};

struct HdPackArgs { const float* w; u16* dst; int64_t sa, sb; };

// [K-step t][lane (row = lane & 15, kg = lane >> 4)][j]: row = (rs, c); gradient row r0 + 4 t + kg is tap row
// 4 t + kg - rs of output row r0 + 2 + rs, i.e. weight row 4 - (4 t + kg - rs)
__global__ __launch_bounds__(256) void head_dgrad_pack_kernel(HdPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HD_PACKED) return;
  const int j = i & 7, lane = (i >> 3) & 63, t = i >> 9;
  const int row = lane & 15, kg = lane >> 4;
  const int rs = row >> 3, c = row & 7;
  const int ky = 4 - (4 * t + kg - rs), kx = 4 - j;
  float v = 0.f;
  if (ky >= 0 && ky < HD_K && kx >= 0) v = a.w[c * a.sb + ky * HD_K + kx];      // (gathered channel 0)
  a.dst[i] = f2bf(v);
}

template <bool ACT>
__global__ __launch_bounds__(256) void head_dgrad_kernel(HdArgs a) {
  __shared__ __attribute__((aligned(16))) u16 tile[HD_LH * HD_LW];
  __shared__ float ered[4][3 * HD_C];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int n = blockIdx.x / per_img, tr = blockIdx.x - n * per_img;
  const int ty0 = (tr / a.tiles_x) * HD_TH, tx0 = (tr % a.tiles_x) * HD_TW;

  // ---- stage the gradient tile: rows ty0 - 2 .., columns tx0 - 2 .. (zero outside the image), fp32 -> bf16
  {
    const float* img = a.in + (int64_t)n * a.h * a.w * a.in_cs + a.in_co;
    constexpr int NE = HD_LH * HD_LW, SL = (NE + 255) / 256;
    float v[SL];
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = min(tid + i * 256, NE - 1);
      const int row = e / HD_LW, col = e - row * HD_LW;
      const int gy = ty0 - 2 + row, gx = tx0 - 2 + col;
      const bool ok = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
      const float x = img[((int64_t)cy * a.w + cx) * a.in_cs];
      v[i] = ok ? x : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + i * 256;
      if (e < NE) tile[e] = f2bf(v[i]);
    }
  }
  bf8 wf[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) wf[t] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(a.wp + (t * 64 + lane) * 8));
  __syncthreads();

  const int x0 = wave * 16;
  const int ox = tx0 + x0 + lj, oxc = min(ox, a.w - 1);
  const int rs = kg >> 1, c0 = 4 * (kg & 1);
  float psc[4] = {1.f, 1.f, 1.f, 1.f}, psf[4] = {0.f, 0.f, 0.f, 0.f}, psl[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (ACT) {
    if (a.spw.scale) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { psc[r] = a.spw.scale[c0 + r]; psf[r] = a.spw.shift[c0 + r]; psl[r] = a.spw.slope[c0 + r]; }
    }
  }
  float es[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) es[q][r] = 0.f;

#pragma unroll 2
  for (int pr = 0; pr < HD_TH / 2; ++pr) {          // pairs of output rows ty0 + 2 pr, + 1
    const int oy = ty0 + 2 * pr + rs, oyc = min(oy, a.h - 1);
    uint2 rw = make_uint2(0u, 0u);
    if constexpr (ACT)
      rw = *reinterpret_cast<const uint2*>(a.raw + ((int64_t)(n * a.h + oyc) * a.w + oxc) * a.raw_cs + a.raw_co + c0);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // (slots 6, 7 of the eight rows lie outside the pair's window: zero DATA as well as zero weights -- a NaN of
      //  a gradient row must not reach pixels whose window does not hold it -- and no read past the tile)
      const bool dead = t == 1 && kg >= 2;
      const u16* p = tile + (dead ? 0 : 2 * pr + 4 * t + kg) * HD_LW + x0 + lj;
      const unsigned keep = dead ? 0u : 0xffffffffu;
      const unsigned e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3], e4 = p[4];
      const bf8 xf = __builtin_bit_cast(bf8, make_uint4((e0 | (e1 << 16)) & keep, (e2 | (e3 << 16)) & keep, e4 & keep, 0u));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf, acc, 0, 0, 0);
    }
    float g[4] = {acc[0], acc[1], acc[2], acc[3]};
    if constexpr (ACT) {
      const float rv[4] = {bf2f((u16)(rw.x & 0xffffu)), bf2f((u16)(rw.x >> 16)), bf2f((u16)(rw.y & 0xffffu)),
                           bf2f((u16)(rw.y >> 16))};
      const bool live = oy < a.h && ox < a.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = live ? acc[r] : 0.f;
        const float t = fmaf(rv[r], psc[r], psf[r]);
        const bool pos = t > 0.f;
        g[r] = pos ? d : d * psl[r];
        es[0][r] += g[r];
        es[1][r] = fmaf(g[r], rv[r], es[1][r]);
        if (!pos) es[2][r] = fmaf(d, t, es[2][r]);
      }
    }
    if (oy < a.h && ox < a.w)
      *reinterpret_cast<uint2*>(a.out + ((int64_t)(n * a.h + oy) * a.w + ox) * a.out_cs + a.out_co + c0) =
          make_uint2(pack2(g[0], g[1]), pack2(g[2], g[3]));
  }
  if constexpr (ACT) {
    // a lane's 8 pixels and the 32 lanes that share its channel quad in fp32 (256 terms), waves and workgroups in
    // double, fixed order (the scheme of small_conv_kernel)
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = es[q][r];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        v += __shfl_xor(v, 32, 64);
        es[q][r] = v;
      }
    if (lj == 0 && kg < 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) ered[wave][q * HD_C + c0 + r] = es[q][r];
    }
    __syncthreads();
    if (tid < a.stat_stride)
      a.stat[(int64_t)blockIdx.x * a.stat_stride + tid] =
          tid < 3 * HD_C ? ((double)ered[0][tid] + (double)ered[1][tid]) + ((double)ered[2][tid] + (double)ered[3][tid]) : 0.0;
  }
}

bool hd_geom(const ConvGeom& g) {
  // (no switch: the tiled kernel cannot stand in -- it wants 16-byte units of the gathered tensor, this one has 4-byte pixels)
  return g.gather_transposed && g.k == HD_K && g.stride == 1 && g.pad == 2 && g.nphase == 1 && g.IS == 1 && g.OS == 1 &&
         g.taps == HD_K && g.cin_g == 1 && g.cout_g == HD_C;
}
int64_t hd_rows(const bp_view* out) { return (int64_t)bp_ceil_div(out->w, HD_TW) * bp_ceil_div(out->h, HD_TH) * out->n; }

}  // namespace

// elements of this kernel's weight image (0: the kernel does not apply)
int64_t bp_bf16_head_packed_elems(const ConvGeom& g) { return hd_geom(g) ? HD_PACKED : 0; }

int bp_bf16_head_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st) {
  if (!hd_geom(g)) return BP_EUNSUPPORTED;
  HdPackArgs a{w_torch, dst, wm.sa, wm.sb};
  hipLaunchKernelGGL(head_dgrad_pack_kernel, dim3(HD_PACKED / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// mode: 0 plain data gradient, 3 activation backward of the produced slot in the epilogue (IgemmStatsReq)
bool bp_bf16_head_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int mode) {
  if (!hd_geom(g) || bias || !in || !out || (mode != 0 && mode != 3)) return false;
  if (in->dtype != BP_F32 || out->dtype != BP_BF16 || in->c != 1 || out->c != HD_C) return false;
  if (in->n != out->n || in->h != out->h || in->w != out->w) return false;
  if (out->cstride % 4 || out->coff % 4 || reinterpret_cast<uintptr_t>(out->ptr) % 8) return false;
  return hd_rows(out) < (1ll << 31);
}

size_t bp_bf16_head_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode) {
  if (mode != 3 || !bp_bf16_head_ok(g, in, out, nullptr, mode)) return 0;
  return bp_stats_rows_bytes_n(hd_rows(out), 3 * HD_C);
}

int bp_bf16_head_run(const ConvGeom& g, const bp_view* in, const u16* packed_head, const bp_view* out, hipStream_t st,
                     const IgemmStatsReq* sr) {
  HdArgs a{};
  a.in = reinterpret_cast<const float*>(in->ptr); a.h = in->h; a.w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = reinterpret_cast<u16*>(out->ptr); a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed_head; a.n = in->n;
  a.tiles_x = bp_ceil_div(out->w, HD_TW); a.tiles_y = bp_ceil_div(out->h, HD_TH);
  const int64_t rows = hd_rows(out);
  const dim3 grid((unsigned)rows), block(256);
  if (!sr) {
    hipLaunchKernelGGL(head_dgrad_kernel<false>, grid, block, 0, st, a);
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  const bp_view* r = sr->raw;
  if (sr->mode != 3) return BP_EUNSUPPORTED;
  if (!r || r->n != out->n || r->h != out->h || r->w != out->w || r->c != out->c || !sr->sums) return BP_EINVAL;
  if (r->dtype != BP_BF16 || r->cstride % 4 || r->coff % 4 || reinterpret_cast<uintptr_t>(r->ptr) % 8) return BP_EUNSUPPORTED;
  const int ns = 3 * HD_C;
  if (!sr->ws || sr->ws_bytes < bp_stats_rows_bytes_n(rows, ns)) return BP_EWORKSPACE;
  a.raw = reinterpret_cast<const u16*>(r->ptr); a.raw_cs = r->cstride; a.raw_co = r->coff;
  a.spw = sr->spw; a.stat = reinterpret_cast<double*>(sr->ws); a.stat_stride = bp_stats_row_stride(ns);
  hipLaunchKernelGGL(head_dgrad_kernel<true>, grid, block, 0, st, a);
  BP_CHECK_LAUNCH();
  return bp_stats_rows_finish_n(a.stat, rows, ns, sr, st);
}
