// Weight gradient on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC.
//
//   dst[cy][cx][ky][kx] = sum_{n,q} act(X)[n, q*s + k - p, cx] * act(Y)[n, q, cy]
// X is the tensor on the fine grid (Conv2d: the layer input; ConvTranspose2d: the output
// gradient), Y the one on the coarse grid (Conv2d: dy; ConvTranspose2d: the layer input); the
// result is exactly the torch weight layout in both cases.
//   GEMM view: M = 16 X-channels, N = 16 Y-channels, K = pixels (4 per MFMA).
// A workgroup owns one kernel row ky, one 16-channel X tile and 16*NTY Y channels, and walks a
// strided subset ("split") of the BH x 16 pixel tiles; its 4 waves take different pixel rows and
// are summed through LDS at the end.  Partial results per split go to a workspace and a second
// kernel adds them in a fixed order, so the gradient is bitwise reproducible (no float atomics).
#include "common.hpp"
#include <cstdlib>
#include <vector>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int WG_BH = 8;    // pixel rows per tile (2 per wave)
constexpr int WG_BW = 16;   // pixel columns per tile
constexpr int KW_MAX = 9;

struct WgradArgs {
  const float* X; int xh, xw, xcs, xco, cx;
  const float* Y; int yh, yw, ycs, yco, cy;
  int n, k, stride, pad;
  PW pwx, pwy;
  float* ws;
  int ncxt, ncyg, nsplit, tiles_x, tiles_y, IW, IWq, CXP, CYP;
};

template <int NTY>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  constexpr int CYB = 16 * NTY;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int s = a.stride;
  float* lds_x = smem;                                   // [BH][s][IWq][16]
  float* lds_y = smem + (size_t)WG_BH * s * a.IWq * 16;  // [NTY][BH][BW][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;

  const int cxt = blockIdx.x % a.ncxt;
  const int cyg = blockIdx.x / a.ncxt;
  const int ky = blockIdx.y;
  const int split = blockIdx.z;
  const int cx0 = cxt * 16, cy0 = cyg * CYB;

  v4f acc[KW_MAX][NTY];
#pragma unroll
  for (int t = 0; t < KW_MAX; ++t)
#pragma unroll
    for (int nt = 0; nt < NTY; ++nt) acc[t][nt] = v4f{0.f, 0.f, 0.f, 0.f};

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.n * tiles_per_img;
  for (int tile = split; tile < ntiles; tile += a.nsplit) {
    const int n = tile / tiles_per_img;
    const int ty_ = (tile % tiles_per_img) / a.tiles_x, tx_ = tile % a.tiles_x;
    const int qy0 = ty_ * WG_BH, qx0 = tx_ * WG_BW;
    __syncthreads();
    // ---- stage X rows: row r <-> iy = (qy0+r)*s + ky - p, col c <-> ix = qx0*s - p + c
    const float* Xn = a.X + (int64_t)n * a.xh * a.xw * a.xcs + a.xco;
    for (int e = tid; e < WG_BH * a.IW * 16; e += 256) {
      const int ch = e & 15;
      const int c = (e >> 4) % a.IW;
      const int r = (e >> 4) / a.IW;
      const int iy = (qy0 + r) * s + ky - a.pad, ix = qx0 * s - a.pad + c;
      float v = 0.f;
      if (iy >= 0 && iy < a.xh && ix >= 0 && ix < a.xw && (qy0 + r) < a.yh && cx0 + ch < a.cx)
        v = pw_apply(a.pwx, cx0 + ch, Xn[((int64_t)iy * a.xw + ix) * a.xcs + cx0 + ch]);
      lds_x[((r * s + c % s) * a.IWq + c / s) * 16 + ch] = v;
    }
    // ---- stage Y tile
    const float* Yn = a.Y + (int64_t)n * a.yh * a.yw * a.ycs + a.yco;
    for (int e = tid; e < WG_BH * WG_BW * CYB; e += 256) {
      const int ch = e % CYB;
      const int c = (e / CYB) % WG_BW;
      const int r = e / (CYB * WG_BW);
      const int qy = qy0 + r, qx = qx0 + c;
      float v = 0.f;
      if (qy < a.yh && qx < a.yw && cy0 + ch < a.cy)
        v = pw_apply(a.pwy, cy0 + ch, Yn[((int64_t)qy * a.yw + qx) * a.ycs + cy0 + ch]);
      lds_y[(((ch >> 4) * WG_BH + r) * WG_BW + c) * 16 + (ch & 15)] = v;
    }
    __syncthreads();
    // ---- this wave's rows: 2 rows x 4 pixel groups
#pragma unroll
    for (int rr = 0; rr < WG_BH / 4; ++rr) {
      const int r = wave * (WG_BH / 4) + rr;
#pragma unroll
      for (int g = 0; g < WG_BW / 4; ++g) {
        const int qxl = 4 * g + kq;
        float bf[NTY];
#pragma unroll
        for (int nt = 0; nt < NTY; ++nt) bf[nt] = lds_y[((nt * WG_BH + r) * WG_BW + qxl) * 16 + li];
#pragma unroll
        for (int t = 0; t < KW_MAX; ++t) {
          if (t < a.k) {
            const float af = lds_x[((r * s + t % s) * a.IWq + qxl + t / s) * 16 + li];
#pragma unroll
            for (int nt = 0; nt < NTY; ++nt)
              acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[nt], acc[t][nt], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- sum the 4 waves through LDS and write this split's partial tile
  //      D[row = 4*(lane>>4)+r : X channel][col = lane&15 : Y channel]
  float* red = smem;  // [4 waves][NTY][64 lanes][4]
#pragma unroll
  for (int t = 0; t < KW_MAX; ++t) {
    if (t < a.k) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < NTY; ++nt)
        *reinterpret_cast<float4*>(red + ((wave * NTY + nt) * 64 + lane) * 4) =
            make_float4(acc[t][nt][0], acc[t][nt][1], acc[t][nt][2], acc[t][nt][3]);
      __syncthreads();
      for (int e = tid; e < NTY * 256; e += 256) {
        const int nt = e / 256, l = (e % 256) / 4, r = e % 4;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[((w * NTY + nt) * 64 + l) * 4 + r];
        const int cx = cx0 + 4 * (l >> 4) + r, cy = cy0 + nt * 16 + (l & 15);
        a.ws[((((int64_t)split * a.k + ky) * a.k + t) * a.CYP + cy) * a.CXP + cx] = sum;
      }
    }
  }
}

struct WreduceArgs {
  const float* ws; float* dst;
  int k, cx, cy, CXP, CYP, nsplit;
  int cx_total, cx_off, cy_off;     // position of this (cx, cy) block inside the full weight tensor
};

// WR_GRP: split groups per block (threads = 64 outputs x WR_GRP): 4 for the few-way splits of the fp32 kernels, 16 for
// the 128 ... 512-way splits of the bf16 and thin-layer kernels
template <int WR_GRP>
__device__ __forceinline__ void wgrad_reduce_block(const WreduceArgs& a, int64_t block, double (*sh)[64]) {
  // 64 consecutive outputs (workspace order [ky][kx][cy][cx], cx fastest: coalesced) x WR_GRP groups of splits per
  // block: a thread adds every WR_GRP-th split (two chains, loads independent of each other), the groups are then
  // added in a fixed order -> bitwise reproducible.  (With four groups a 512-way split is 128 trips per thread:
  // latency-bound, 30 us for 25 MB; sixteen cut the chain to 32.)
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = block * 64 + lane;
  const int64_t total = (int64_t)a.k * a.k * a.cy * a.cx;
  double s = 0.0;
  int cx = 0, cy = 0, t = 0;
  if (i < total) {
    cx = i % a.cx;
    cy = (i / a.cx) % a.cy;
    t = i / ((int64_t)a.cx * a.cy);
    const int64_t stride = (int64_t)a.k * a.k * a.CYP * a.CXP;
    const float* p = a.ws + ((int64_t)t * a.CYP + cy) * a.CXP + cx;
    double s0 = 0.0, s1 = 0.0;
    int sp = grp;
    for (; sp + WR_GRP < a.nsplit; sp += 2 * WR_GRP) {
      s0 += (double)p[sp * stride];
      s1 += (double)p[(sp + WR_GRP) * stride];
    }
    if (sp < a.nsplit) s0 += (double)p[sp * stride];
    s = s0 + s1;
  }
  sh[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && i < total) {
    double r = sh[0][lane];
#pragma unroll
    for (int g = 1; g < WR_GRP; ++g) r += sh[g][lane];
    a.dst[((int64_t)(a.cy_off + cy) * a.cx_total + a.cx_off + cx) * a.k * a.k + t] = (float)r;
  }
}

template <int WR_GRP>
__global__ __launch_bounds__(64 * WR_GRP) void wgrad_reduce_kernel(WreduceArgs a) {
  __shared__ double sh[WR_GRP][64];
  wgrad_reduce_block<WR_GRP>(a, blockIdx.x, sh);
}

// Deferred reductions (bp_wgrad_defer_begin / _flush): the reductions of many layers in one launch.  A block finds its
// job by scanning the table of first blocks (a few dozen entries, scalar loads); each job is reduced exactly as its own
// launch would reduce it, so deferring changes no bit of dW.
constexpr int WR_BATCH = 40;
struct WreduceBatch {
  int njobs;
  int first[WR_BATCH + 1];
  WreduceArgs job[WR_BATCH];
};
static_assert(sizeof(WreduceBatch) <= 4096, "kernel argument block");

template <int WR_GRP>
__global__ __launch_bounds__(64 * WR_GRP) void wgrad_reduce_batch_kernel(WreduceBatch b) {
  __shared__ double sh[WR_GRP][64];
  const int blk = blockIdx.x;
  int j = 0;
  while (j + 1 < b.njobs && blk >= b.first[j + 1]) ++j;
  wgrad_reduce_block<WR_GRP>(b.job[j], blk - b.first[j], sh);
}

struct WgradPlan {
  int NTY, ncxt, ncyg, nsplit, tiles_x, tiles_y, IW, IWq, CXP, CYP;
  size_t lds_bytes, ws_bytes;
  bool ok;
};

WgradPlan wgrad_plan(const bp_conv* cv, const bp_view* X, const bp_view* Y) {
  WgradPlan p{};
  if (cv->k > KW_MAX) return p;
  p.NTY = (Y->c > 16) ? 2 : 1;
  const int CYB = 16 * p.NTY;
  p.ncxt = bp_ceil_div(X->c, 16);
  p.ncyg = bp_ceil_div(Y->c, CYB);
  p.CXP = p.ncxt * 16;
  p.CYP = p.ncyg * CYB;
  p.tiles_x = bp_ceil_div(Y->w, WG_BW);
  p.tiles_y = bp_ceil_div(Y->h, WG_BH);
  const int64_t ntiles = (int64_t)Y->n * p.tiles_x * p.tiles_y;
  const int64_t base = (int64_t)p.ncxt * p.ncyg * cv->k;
  int64_t ns = (2048 + base - 1) / base;
  if (ns > ntiles) ns = ntiles;
  if (ns < 1) ns = 1;
  if (ns > 65535) ns = 65535;
  p.nsplit = (int)ns;
  p.IW = (WG_BW - 1) * cv->stride + cv->k;
  p.IWq = bp_ceil_div(p.IW, cv->stride);
  const size_t lds_main = ((size_t)WG_BH * cv->stride * p.IWq * 16 + (size_t)p.NTY * WG_BH * WG_BW * 16) * 4;
  const size_t lds_red = (size_t)4 * p.NTY * 64 * 4 * 4;
  p.lds_bytes = lds_main > lds_red ? lds_main : lds_red;
  p.ws_bytes = (size_t)p.nsplit * cv->k * cv->k * p.CYP * p.CXP * sizeof(float);
  p.ok = p.lds_bytes <= 64 * 1024;
  return p;
}

}  // namespace

int bp_wgrad_tiles(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                   size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);
int bp_wgrad_small(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                   size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);

int bp_wgrad_enc(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                 size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);
int bp_wgrad_thin(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                  size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);
int bp_wgrad_ws_f32(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                    size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);

// the k8 stride-4 encoder layer (conv_enc.hip), one-channel tails (conv_wgrad_thin.hip), the tap-packed few-channel
// kernel, then the tap-blocked one;
// BP_EUNSUPPORTED -> generic kernel
static int wgrad_fast(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                      size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  // the output-stationary kernel of the 128 <-> 128 k3 trunk layers (conv_wgrad_ws_f32.hip)
  int rc = bp_wgrad_ws_f32(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  if (rc == BP_OK && dry) {        // (size the workspace for either kernel: bp_set_option may switch later)
    size_t need2 = 0;
    int ns2, cx2, cy2;
    if (bp_wgrad_tiles(cv, X, pwx, Y, pwy, ws, ws_bytes, &need2, &ns2, &cx2, &cy2, st, true) == BP_OK && need2 > *need) *need = need2;
  }
  if (rc != BP_EUNSUPPORTED) return rc;
  rc = bp_wgrad_enc(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  if (rc == BP_EUNSUPPORTED) rc = bp_wgrad_thin(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  if (rc == BP_EUNSUPPORTED) rc = bp_wgrad_small(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  if (rc == BP_EUNSUPPORTED) rc = bp_wgrad_tiles(cv, X, pwx, Y, pwy, ws, ws_bytes, need, nsplit, cxp, cyp, st, dry);
  return rc;
}

// Per host thread: between bp_wgrad_defer_begin() and bp_wgrad_defer_flush() the reductions of calls flagged
// BP_IMPL_DEFER are collected, not launched: the flag is the caller's promise that the call's workspace is its own
// until the flush (cvae._Plan: one per fp32 layer).
struct DeferState { bool on = false, private_ws = false; std::vector<WreduceArgs> jobs; };
static thread_local DeferState t_defer;

static int wgrad_reduce(const float* ws, float* dst, int k, int cx, int cy, int CXP, int CYP, int nsplit,
                        hipStream_t st, int cx_total = -1, int cx_off = 0, int cy_off = 0, bool may_defer = true) {
  WreduceArgs r{};
  r.ws = ws; r.dst = dst; r.k = k; r.cx = cx; r.cy = cy; r.CXP = CXP; r.CYP = CYP; r.nsplit = nsplit;
  r.cx_total = cx_total < 0 ? cx : cx_total; r.cx_off = cx_off; r.cy_off = cy_off;
  if (t_defer.on && t_defer.private_ws && may_defer) {
    t_defer.jobs.push_back(r);
    return BP_OK;
  }
  const int64_t total = (int64_t)cy * cx * k * k;
  if (r.nsplit > 64) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, st, r);
  else hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, r);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

template <int WR_GRP>
static int wgrad_reduce_flush_class(const std::vector<WreduceArgs>& jobs, bool wide, hipStream_t st) {
  WreduceBatch b{};
  auto launch = [&]() {
    if (b.njobs == 0) return;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel<WR_GRP>, dim3((unsigned)b.first[b.njobs]), dim3(64 * WR_GRP), 0, st, b);
    b = WreduceBatch{};
  };
  for (const WreduceArgs& r : jobs) {
    if ((r.nsplit > 64) != wide) continue;
    const int64_t nb = ((int64_t)r.cy * r.cx * r.k * r.k + 63) / 64;
    if (b.njobs == WR_BATCH || b.first[b.njobs] + nb > 0x3fffffff) launch();
    b.job[b.njobs] = r;
    b.first[b.njobs + 1] = b.first[b.njobs] + (int)nb;
    ++b.njobs;
  }
  launch();
  BP_CHECK_LAUNCH();
  return BP_OK;
}

void bp_wgrad_private_ws(bool on) { t_defer.private_ws = on; }

int bp_wgrad_defer_begin_impl() {
  t_defer.on = true;
  t_defer.jobs.clear();
  return BP_OK;
}

int bp_wgrad_defer_flush_impl(hipStream_t st, int end) {
  int rc = BP_OK;
  // (the state is per host thread: a flush from another thread than the one that called bp_wgrad_defer_begin would find
  //  an empty list, report success and leave every deferred dW unreduced -- refuse it instead)
  if (end >= 0 && !t_defer.on) return BP_EINVAL;
  if (end < 0) t_defer.jobs.clear();          // abandon
  if (!t_defer.jobs.empty()) {
    rc = wgrad_reduce_flush_class<16>(t_defer.jobs, true, st);
    if (rc == BP_OK) rc = wgrad_reduce_flush_class<4>(t_defer.jobs, false, st);
    t_defer.jobs.clear();
  }
  if (end) t_defer.on = false;
  return rc;
}

// One or two channels on one side and many on the other (the k9 stem / head of the CGAN generator): the tap-packed
// few-channel kernel runs once per 16-channel chunk of the wide side, each chunk reduced into its block of dW.
static bool wide_side_chunks(const bp_conv* cv, const bp_view* X, const bp_view* Y, bool* on_x) {
  if (X->c > 16 && Y->c <= 2) *on_x = true;
  else if (Y->c > 16 && X->c <= 2) *on_x = false;
  else return false;
  bp_view sub = *on_x ? *X : *Y;
  sub.c = 16;
  size_t need = 0;
  int ns, cxp, cyp;
  const PW none{nullptr, nullptr, nullptr};
  return bp_wgrad_small(cv, *on_x ? &sub : X, none, *on_x ? Y : &sub, none, nullptr, 0, &need, &ns, &cxp, &cyp, nullptr,
                        true) == BP_OK;
}

static PW pw_offset(const PW& p, int c0) {
  return p.scale ? PW{p.scale + c0, p.shift + c0, p.slope + c0} : p;
}

static int wgrad_chunked(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst,
                         float* ws, size_t ws_bytes, size_t* need_out, hipStream_t st, bool dry, bool on_x) {
  const int wide = on_x ? X->c : Y->c;
  size_t need_max = 0;
  for (int c0 = 0; c0 < wide; c0 += 16) {
    bp_view sub = on_x ? *X : *Y;
    sub.c = wide - c0 < 16 ? wide - c0 : 16;
    sub.coff += c0;
    const bp_view* Xc = on_x ? &sub : X;
    const bp_view* Yc = on_x ? Y : &sub;
    const PW px = on_x ? pw_offset(pwx, c0) : pwx, py = on_x ? pwy : pw_offset(pwy, c0);
    size_t need = 0;
    int ns, cxp, cyp;
    const int rc = bp_wgrad_small(cv, Xc, px, Yc, py, ws, ws_bytes, &need, &ns, &cxp, &cyp, st, dry);
    if (rc != BP_OK) return rc;
    if (need > need_max) need_max = need;
    if (!dry) {
      const int rr = wgrad_reduce(ws, dst, cv->k, Xc->c, Yc->c, cxp, cyp, ns, st, X->c, on_x ? c0 : 0, on_x ? 0 : c0,
                                  /*may_defer=*/false);      // (the chunks share one workspace)
      if (rr != BP_OK) return rr;
    }
  }
  *need_out = need_max;
  return BP_OK;
}

// bf16 matrix-core weight gradient (conv_bf16.hip) + the same fixed-order reduction; operands may be fp32 or bf16
int bp_wgrad_bf16(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                  size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry);

size_t bp_wgrad_bf16_workspace(const bp_conv* cv, const bp_view* X, const bp_view* Y) {
  size_t need = 0;
  int ns, cxp, cyp;
  const PW none{nullptr, nullptr, nullptr};
  return bp_wgrad_bf16(cv, X, none, Y, none, nullptr, 0, &need, &ns, &cxp, &cyp, nullptr, true) == BP_OK ? need : 0;
}

int bp_wgrad_bf16_run(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst,
                      void* workspace, size_t workspace_bytes, hipStream_t st) {
  size_t need = 0;
  int ns, cxp, cyp;
  const int rc = bp_wgrad_bf16(cv, X, pwx, Y, pwy, reinterpret_cast<float*>(workspace), workspace_bytes, &need, &ns,
                               &cxp, &cyp, st, false);
  if (rc != BP_OK) return rc;
  return wgrad_reduce(reinterpret_cast<const float*>(workspace), dst, cv->k, X->c, Y->c, cxp, cyp, ns, st);
}

// conv_stem.hip: weight gradient of the 3 -> 16 k5 stem
bool bp_stem_wgrad_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy, const float* dbias);
size_t bp_stem_wgrad_workspace(const bp_view* X);
int bp_stem_wgrad(const bp_view* X, const PW& pwx, const bp_view* Y, float* dst, void* workspace, size_t workspace_bytes,
                  hipStream_t st);
// conv_wgrad_flat.hip: weight gradient of the 16 -> 8 k7 head layer
bool bp_wgrad_flat_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwy);
size_t bp_wgrad_flat_workspace(const bp_view* X);
int bp_wgrad_flat(const bp_view* X, const PW& pwx, const bp_view* Y, float* dst, void* workspace, size_t workspace_bytes,
                  hipStream_t st, bool shared);
// ... and of the thin stride-2 k4 layers (16 channels at full resolution, 32 at half)
bool bp_wgrad_flat_s2_ok(const bp_conv* cv, const bp_view* X, const bp_view* Y, const PW& pwx, const PW& pwy);
size_t bp_wgrad_flat_s2_workspace(const bp_view* Y);
int bp_wgrad_flat_s2(const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst, void* workspace,
                     size_t workspace_bytes, hipStream_t st, bool shared);
static size_t wgrad_general_workspace(const bp_conv* cv, const bp_view* X, const bp_view* Y);

size_t bp_wgrad_mfma_workspace(const bp_conv* cv, const bp_view* X, const bp_view* Y) {
  const size_t general = wgrad_general_workspace(cv, X, Y);
  if (bp_stem_wgrad_ok(cv, X, Y, PW{nullptr, nullptr, nullptr}, nullptr)) {
    const size_t stem = bp_stem_wgrad_workspace(X);
    return stem > general ? stem : general;
  }
  if (bp_wgrad_flat_ok(cv, X, Y, PW{nullptr, nullptr, nullptr})) {
    const size_t flat = bp_wgrad_flat_workspace(X);
    return flat > general ? flat : general;
  }
  if (bp_wgrad_flat_s2_ok(cv, X, Y, PW{nullptr, nullptr, nullptr}, PW{nullptr, nullptr, nullptr})) {
    const size_t flat = bp_wgrad_flat_s2_workspace(Y);
    return flat > general ? flat : general;
  }
  return general;
}

static size_t wgrad_general_workspace(const bp_conv* cv, const bp_view* X, const bp_view* Y) {
  size_t need = 0;
  int ns, cxp, cyp;
  const PW none{nullptr, nullptr, nullptr};
  bool on_x = false;
  if (wide_side_chunks(cv, X, Y, &on_x) &&
      wgrad_chunked(cv, X, none, Y, none, nullptr, nullptr, 0, &need, nullptr, true, on_x) == BP_OK)
    return need;
  if (wgrad_fast(cv, X, none, Y, none, nullptr, 0, &need, &ns, &cxp, &cyp, nullptr, true) == BP_OK) return need;
  const WgradPlan p = wgrad_plan(cv, X, Y);
  return p.ok ? p.ws_bytes : 0;
}

void bp_wgrad_tiles_target(int target);      // conv_wgrad_tiles.hip
namespace {
struct SharedTarget {                        // for the duration of one bp_wgrad_mfma call on this thread
  explicit SharedTarget(bool shared) {
    static const int t = getenv("BP_WT_SHARED_TARGET") ? atoi(getenv("BP_WT_SHARED_TARGET")) : 448;
    bp_wgrad_tiles_target(shared ? t : 0);
  }
  ~SharedTarget() { bp_wgrad_tiles_target(0); }
};
}  // namespace

int bp_wgrad_mfma(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy,
                  float* dst, void* workspace, size_t workspace_bytes, hipStream_t st, bool shared) {
  const SharedTarget guard(shared);
  if (bp_stem_wgrad_ok(cv, X, Y, pwy, nullptr)) return bp_stem_wgrad(X, pwx, Y, dst, workspace, workspace_bytes, st);
  if (bp_wgrad_flat_ok(cv, X, Y, pwy)) return bp_wgrad_flat(X, pwx, Y, dst, workspace, workspace_bytes, st, shared);
  if (bp_wgrad_flat_s2_ok(cv, X, Y, pwx, pwy))
    return bp_wgrad_flat_s2(X, pwx, Y, pwy, dst, workspace, workspace_bytes, st, shared);
  {
    bool on_x = false;
    if (wide_side_chunks(cv, X, Y, &on_x)) {
      size_t need = 0;
      return wgrad_chunked(cv, X, pwx, Y, pwy, dst, reinterpret_cast<float*>(workspace), workspace_bytes, &need, st,
                           false, on_x);
    }
  }
  {
    size_t need = 0;
    int ns, cxp, cyp;
    const int rc = wgrad_fast(cv, X, pwx, Y, pwy, reinterpret_cast<float*>(workspace), workspace_bytes, &need,
                              &ns, &cxp, &cyp, st, false);
    if (rc == BP_OK)
      return wgrad_reduce(reinterpret_cast<const float*>(workspace), dst, cv->k, X->c, Y->c, cxp, cyp, ns, st);
    if (rc != BP_EUNSUPPORTED) return rc;
  }
  const WgradPlan p = wgrad_plan(cv, X, Y);
  if (!p.ok) return BP_EUNSUPPORTED;
  if (!workspace || workspace_bytes < p.ws_bytes) return BP_EWORKSPACE;
  WgradArgs a{};
  a.X = X->ptr; a.xh = X->h; a.xw = X->w; a.xcs = X->cstride; a.xco = X->coff; a.cx = X->c;
  a.Y = Y->ptr; a.yh = Y->h; a.yw = Y->w; a.ycs = Y->cstride; a.yco = Y->coff; a.cy = Y->c;
  a.n = X->n; a.k = cv->k; a.stride = cv->stride; a.pad = cv->pad; a.pwx = pwx; a.pwy = pwy;
  a.ws = reinterpret_cast<float*>(workspace);
  a.ncxt = p.ncxt; a.ncyg = p.ncyg; a.nsplit = p.nsplit; a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y;
  a.IW = p.IW; a.IWq = p.IWq; a.CXP = p.CXP; a.CYP = p.CYP;
  dim3 grid((unsigned)(p.ncxt * p.ncyg), (unsigned)cv->k, (unsigned)p.nsplit);
  if (p.NTY == 2)
    hipLaunchKernelGGL((wgrad_kernel<2>), grid, dim3(256), p.lds_bytes, st, a);
  else
    hipLaunchKernelGGL((wgrad_kernel<1>), grid, dim3(256), p.lds_bytes, st, a);
  BP_CHECK_LAUNCH();
  WreduceArgs r{};
  r.ws = a.ws; r.dst = dst; r.k = cv->k; r.cx = X->c; r.cy = Y->c; r.CXP = p.CXP; r.CYP = p.CYP;
  r.nsplit = p.nsplit;
  const int64_t total = (int64_t)Y->c * X->c * cv->k * cv->k;
  if (r.nsplit > 64) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, st, r);
  else hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, r);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
