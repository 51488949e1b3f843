// bf16 matrix-core convolutions: configuration, weight packing and the entry points behind capi.hip
// (kernels: conv_bf16.hpp, instantiated per channel-chunk width in conv_bf16_cc*.hip).
#include "conv_bf16.hpp"

using namespace bpbf16;

int bp_bf16_launch_cc4(const BConfig& c, const BArgs& a, bool in_bf16, bool out_bf16, dim3 grid, hipStream_t st);
int bp_bf16_launch_cc8(const BConfig& c, const BArgs& a, bool in_bf16, bool out_bf16, dim3 grid, hipStream_t st);
int bp_bf16_launch_cc16(const BConfig& c, const BArgs& a, bool in_bf16, bool out_bf16, dim3 grid, hipStream_t st);
int bp_bf16_launch_cc32(const BConfig& c, const BArgs& a, bool in_bf16, bool out_bf16, dim3 grid, hipStream_t st);

// conv_bf16_flat.hip: flattened-K kernel for the unit-stride k7 head layers; its weight image follows the generic one
int64_t bp_bf16_flat_packed_elems(const ConvGeom& g);
int bp_bf16_flat_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st);
bool bp_bf16_flat_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int stats);
size_t bp_bf16_flat_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode);
int bp_bf16_flat_run(const ConvGeom& g, const bp_view* in, const PW& pw, const u16* packed_flat, const bp_view* out,
                     hipStream_t st, const IgemmStatsReq* sr);

// conv_bf16_ws.hip: weights-stationary kernel of the 128 -> 128 k3 trunk; its weight image follows the other two
int64_t bp_bf16_ws_packed_elems(const ConvGeom& g);
int bp_bf16_ws_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st);
bool bp_bf16_ws_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int mode);
size_t bp_bf16_ws_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out);
int bp_bf16_ws_run(const ConvGeom& g, const bp_view* in, const PW& pw, const u16* packed_ws, const bp_view* out,
                   hipStream_t st, const IgemmStatsReq* sr);

// conv_bf16_head.hip: data gradient (+ activation backward) of the heads' 8 -> 1 k5 layer; its weight image comes last
int64_t bp_bf16_head_packed_elems(const ConvGeom& g);
int bp_bf16_head_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st);
bool bp_bf16_head_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int mode);
size_t bp_bf16_head_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode);
int bp_bf16_head_run(const ConvGeom& g, const bp_view* in, const u16* packed_head, const bp_view* out, hipStream_t st,
                     const IgemmStatsReq* sr);

namespace {

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

static int64_t generic_packed_elems(const ConvGeom& g, const BConfig& c) {
  return (int64_t)g.nphase * g.nphase * g.taps * c.nrun * c.nchunk * c.cout_padP * 32;
}

// [generic image | flattened-K image (conv_bf16_flat.hip) | weights-stationary image (conv_bf16_ws.hip) | head image
// (conv_bf16_head.hip)], the last three where those kernels apply
int64_t bp_bf16_packed_elems(const ConvGeom& g) {
  const BConfig c = b_config(g);
  if (!c.ok) return -1;
  return generic_packed_elems(g, c) + bp_bf16_flat_packed_elems(g) + bp_bf16_ws_packed_elems(g) + bp_bf16_head_packed_elems(g);
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
  a.total = generic_packed_elems(g, c);
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  if (bp_bf16_flat_packed_elems(g) > 0) {
    const int rc = bp_bf16_flat_pack(g, wm, w_torch, a.dst + a.total, st);
    if (rc != BP_OK) return rc;
  }
  if (bp_bf16_ws_packed_elems(g) > 0) {
    const int rc = bp_bf16_ws_pack(g, wm, w_torch, a.dst + a.total + bp_bf16_flat_packed_elems(g), st);
    if (rc != BP_OK) return rc;
  }
  if (bp_bf16_head_packed_elems(g) > 0)
    return bp_bf16_head_pack(g, wm, w_torch, a.dst + a.total + bp_bf16_flat_packed_elems(g) + bp_bf16_ws_packed_elems(g), st);
  return BP_OK;
}

// conv_igemm.hip: partial rows of epilogue statistics -> sums
size_t bp_stats_rows_bytes(int64_t rows, int C);
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st);

static int64_t bf16_stat_rows(const ConvGeom& g, const BConfig& c, const bp_view* in, const bp_view* out) {
  const int qh = bp_ceil_div(out->h, g.OS), qw = bp_ceil_div(out->w, g.OS);
  return (int64_t)bp_ceil_div(qw, 16 * c.TPR) * bp_ceil_div(qh, c.BH) * in->n * g.nphase * g.nphase;
}

// workspace for bp_bf16_igemm_run with statistics (0: not available for this layer / these views)
size_t bp_bf16_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode) {
  const BConfig c = b_config(g);
  if (!c.ok) return 0;
  if (mode == 3) return bp_bf16_head_stats_workspace(g, in, out, mode);      // (activation backward: the head kernel only)
  if (!bp_bf16_igemm_ok(g, in, out)) return 0;
  if (bp_bf16_flat_ok(g, in, out, nullptr, mode)) return bp_bf16_flat_stats_workspace(g, in, out, mode);
  if (mode != 1) return 0;                                    // (mode 2: the flattened-K kernels only)
  const size_t generic = bp_stats_rows_bytes(bf16_stat_rows(g, c, in, out), g.cout_g);
  if (bp_bf16_ws_ok(g, in, out, nullptr, mode)) {        // (the larger of the two: bp_set_option may switch kernels later)
    const size_t ws = bp_bf16_ws_stats_workspace(g, in, out);
    return ws > generic ? ws : generic;
  }
  return generic;
}

int bp_bf16_igemm_run(const ConvGeom& g, const bp_view* in, const PW& pw, const void* packed, const float* bias,
                      const bp_view* out, hipStream_t st, const IgemmStatsReq* sr) {
  const BConfig c = b_config(g);
  if (!c.ok) return BP_EUNSUPPORTED;
  // (the head kernel reads its one gathered channel with scalar loads: any channel stride)
  if (!pw.scale && bp_bf16_head_ok(g, in, out, bias, sr ? sr->mode : 0))
    return bp_bf16_head_run(g, in, reinterpret_cast<const u16*>(packed) + generic_packed_elems(g, c) +
                            bp_bf16_flat_packed_elems(g) + bp_bf16_ws_packed_elems(g), out, st, sr);
  if (!bp_bf16_igemm_ok(g, in, out)) return BP_EUNSUPPORTED;
  if (bp_bf16_flat_ok(g, in, out, bias, sr ? sr->mode : 0))
    return bp_bf16_flat_run(g, in, pw, reinterpret_cast<const u16*>(packed) + generic_packed_elems(g, c), out, st, sr);
  if (bp_bf16_ws_ok(g, in, out, bias, sr ? sr->mode : 0))
    return bp_bf16_ws_run(g, in, pw, reinterpret_cast<const u16*>(packed) + generic_packed_elems(g, c) +
                          bp_bf16_flat_packed_elems(g), out, st, sr);
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
  const int64_t rows = (int64_t)grid.x * grid.y;
  if (sr) {
    const size_t need = bp_stats_rows_bytes(rows, g.cout_g);
    if (bias || sr->mode != 1 || !need) return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < need || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws); a.stat_c = g.cout_g;
  }
  int rc = BP_EUNSUPPORTED;
  switch (c.CC) {
    case 32: rc = bp_bf16_launch_cc32(c, a, ib, ob, grid, st); break;
    case 16: rc = bp_bf16_launch_cc16(c, a, ib, ob, grid, st); break;
    case 8: rc = bp_bf16_launch_cc8(c, a, ib, ob, grid, st); break;
    case 4: rc = bp_bf16_launch_cc4(c, a, ib, ob, grid, st); break;
  }
  if (rc != BP_OK || !sr) return rc;
  return bp_stats_rows_finish(a.stat, rows, g.cout_g, sr, st);
}

