// Weights-stationary fp32 convolution of the generator's residual trunk (parity mode, BASELINE.json configs[1]):
// Conv2d k3 s1 p1, 128 -> 128 channels, forward and data gradient -- 16 launches = 24 % of the fp32 step
// (/root/reference/baryon_painter/models/utils.py:22-38 ResidualBlock; architecture_built.txt:54-93).
//
// The tiled igemm_dma_kernel runs these layers at 0.81 of the fp32 matrix peak (v_mfma_f32_16x16x4_f32 at the vector FMA
// rate: profiles/r03_mfma_util_f32.txt): weight slabs by LDS-DMA per tile, both operands read out of LDS, barriers per tap
// row.  The weights-stationary scheme of conv_bf16_ws.hip fits fp32 too if a workgroup owns HALF the produced channels:
// 9 taps x 128 gathered x 64 produced channels x 4 B = 295 KB = 288 registers per lane of four waves (wave w: channels
// [64 h + 16 w, + 16), one A operand register per k-step of 4 channels).  Per K-step group (tap, 16 channels) a wave reads
// ONE 16-byte B fragment per 16 pixels and issues FOUR MFMAs on it: 32 B/clk of LDS reads per CU, no weight traffic, one
// barrier per row of 4608 MFMAs.  The packed image is the tiled kernel's own ([tap][chunk of 16][co][16]: a lane's four
// k-steps of a chunk are one float4), so nothing else changes in the plan.  Numerics: the same exact-fp32 FMA chains
// (another summation order over K than the tiled kernel: taps outermost), batch-norm sums in double per lane.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

size_t bp_stats_rows_bytes(int64_t rows, int C);
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st);

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WF_C = 128, WF_R = 4, WF_NK = 288, WF_FV = 64;       // k-steps per wave; ... of which in VGPRs (rest AGPRs)

struct WfArgs {
  const float* in; int in_cs, in_co;
  float* out; int out_cs, out_co;
  int n, h;
  int iw, strips;                    // image width and column strips of W pixels per row (W = 16 G; iw = W * strips)
  int items;                         // images x bands x strips
  const float* wp;                   // tiled image [tap 9][chunk 8][co 128][16]
  PW pw;
  int BR, bands;
  double* stat;                      // rows [image x band][2][128]
};

template <int G> struct WfGeom {
  static constexpr int W = 16 * G, RP = W + 2;
  static constexpr int PS = (WF_R * RP + 15) / 16 * 16;          // 16-byte slots per plane (one plane per channel quad)
  static constexpr size_t img_bytes = (size_t)32 * PS * 16;
  static constexpr size_t lds_bytes = img_bytes + 3 * WF_C * sizeof(float);
};

template <int G, bool ACT, bool STATS, bool STRIPS = false>
__global__ __launch_bounds__(256) void ws3_f32_kernel(WfArgs a) {
  using GM = WfGeom<G>;
  constexpr int W = GM::W, RP = GM::RP, PS = GM::PS;
  constexpr int NU = 2 * G;                             // staging units (16 B) per thread and row
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float4* img = reinterpret_cast<float4*>(smem_f);      // [quad 32][ring row 4][RP]
  float* lpw = smem_f + GM::img_bytes / 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  // block id = item_lo + 8 * (half + 2 * item_hi): the two channel halves of a band read the same input rows -- on the same
  // XCD (workgroups go to the XCDs round-robin by id) the second read is an L2 hit (419 -> ~290 MB of HBM per launch)
  const int half = (blockIdx.x >> 3) & 1, item = (int)(blockIdx.x & 7) + 8 * (int)(blockIdx.x >> 4);
  if (item >= a.items) return;
  const int strip = item % a.strips, nb = item / a.strips;
  const int n = nb / a.bands, band = nb % a.bands;
  const int y0 = band * a.BR;
  const int y1 = min(y0 + a.BR, a.h);
  const int x0 = strip * W;          // images wider than a strip (the CGAN generator's 128-pixel trunk): the pad columns of
                                     // the ring rows then hold the neighbouring strips' pixels instead of zeros

  // staging: unit i = 8 pixels x 8 quads per wave instruction: pixel (lane & 7) + 8 (ub >> 2), quad (lane >> 3) + 8 (ub & 3),
  // ub = wave + 4 i -- eight consecutive pixels of one plane per eight lanes (conflict-free ds_write_b128)
  // (ub & 3 = wave, ub >> 2 = i: the quad is the same for every unit of a thread, the pixel advances by 8 -- one base and a
  //  uniform stride instead of NU registers each)
  const int s_q = (lane >> 3) + 8 * wave;
  const int s_slot0 = s_q * PS + 1 + (lane & 7);
  const unsigned s_off0 = (unsigned)((lane & 7) * a.in_cs + s_q * 4) * 4u;
  const unsigned s_st = (unsigned)(8 * a.in_cs) * 4u;
  const char* in_img = reinterpret_cast<const char*>(a.in + ((int64_t)n * a.h * a.iw + x0) * a.in_cs + a.in_co);
  const unsigned in_row = (unsigned)(a.iw * a.in_cs) * 4u;
  // halo columns x0 - 1 and x0 + W: wave 0, lane = (side, quad); outside the image they stay zero
  const int h_q = lane & 31, h_side = lane >> 5;
  const bool h_in = h_side ? x0 + W < a.iw : x0 > 0;
  const int h_slot = h_q * PS + (h_side ? RP - 1 : 0);
  const unsigned h_keep = h_in ? 0xffffffffu : 0u;
  const int h_off = ((h_in ? (h_side ? W : -1) : 0) * a.in_cs + h_q * 4) * 4;
  auto load_row = [&](int r, float4 (&raw)[NU], float4& hraw) {
    const char* rowp = in_img + (size_t)((unsigned)r * in_row);
#pragma unroll
    for (int i = 0; i < NU; ++i) raw[i] = *reinterpret_cast<const float4*>(rowp + (size_t)((unsigned)i * s_st) + s_off0);
    if (STRIPS && wave == 0) hraw = *reinterpret_cast<const float4*>(rowp + h_off);
  };
  auto act4 = [&](int q, float4 v) {
    if constexpr (ACT) {
      const float4 sc = *reinterpret_cast<const float4*>(lpw + q * 4);
      const float4 sf = *reinterpret_cast<const float4*>(lpw + WF_C + q * 4);
      const float4 sl = *reinterpret_cast<const float4*>(lpw + 2 * WF_C + q * 4);
      float t;
      t = fmaf(v.x, sc.x, sf.x); v.x = t > 0.f ? t : t * sl.x;          // (a NaN stays a NaN, as torch.relu)
      t = fmaf(v.y, sc.y, sf.y); v.y = t > 0.f ? t : t * sl.y;
      t = fmaf(v.z, sc.z, sf.z); v.z = t > 0.f ? t : t * sl.z;
      t = fmaf(v.w, sc.w, sf.w); v.w = t > 0.f ? t : t * sl.w;
    }
    return v;
  };
  auto commit_halo = [&](int rr, unsigned keep, const float4& hraw) {
    if (STRIPS && wave == 0) {
      const float4 v = act4(h_q, hraw);
      const unsigned k = keep & h_keep;
      auto m = [&](float f) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) & k); };
      img[h_slot + rr * RP] = make_float4(m(v.x), m(v.y), m(v.z), m(v.w));
    }
  };
  auto commit_row = [&](int r, bool inside, const float4 (&raw)[NU], const float4& hraw) {
    const int rr = (r + 1) & (WF_R - 1);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      float4 v = act4(s_q, raw[i]);
      if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
      img[s_slot0 + 8 * i + rr * RP] = v;
    }
    commit_halo(rr, inside ? 0xffffffffu : 0u, hraw);
  };

  // ---- prologue
  float4 raw0[NU], raw1[NU];
  float4 hraw0 = make_float4(0.f, 0.f, 0.f, 0.f), hraw1 = hraw0;
  const bool in0 = y0 - 1 >= 0, in2 = y0 + 1 < a.h;
  load_row(in0 ? y0 - 1 : y0, raw0, hraw0);
  load_row(y0, raw1, hraw1);
  float wf[WF_NK];                                      // k-step (tap, chunk q, j): channels 16 q + 4 kq' + j of lane group kq'
  {
    const float4* wsrc = reinterpret_cast<const float4*>(a.wp) + (size_t)(64 * half + 16 * wave + lm) * 4 + kq;
#pragma unroll
    for (int s = 0; s < 72; ++s) {                      // s = tap * 8 + q: [s][co 128][16 floats]
      const float4 t = wsrc[(size_t)s * WF_C * 4];
      wf[4 * s] = t.x; wf[4 * s + 1] = t.y; wf[4 * s + 2] = t.z; wf[4 * s + 3] = t.w;
    }
    // register classes by hand (see conv_bf16_ws.hip): an MFMA takes its A operand from either file
#pragma unroll
    for (int s = 0; s < WF_NK; ++s) {
      if (s < WF_FV) asm volatile("" : "+v"(wf[s]));
      else asm volatile("" : "+a"(wf[s]));
    }
  }
  // (with strips the pad columns are the halo, written by wave 0 with every row -- zeros where the image ends: no fill here,
  //  it would race with those writes)
  if (!STRIPS && tid < 256) {
    const int plane = tid >> 3, rr = (tid >> 1) & 3, side = tid & 1;
    img[plane * PS + rr * RP + (side ? RP - 1 : 0)] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (ACT) {
    for (int i = tid; i < WF_C; i += 256) { lpw[i] = a.pw.scale[i]; lpw[WF_C + i] = a.pw.shift[i]; lpw[2 * WF_C + i] = a.pw.slope[i]; }
    __syncthreads();
  }
  commit_row(y0 - 1, in0, raw0, hraw0);
  commit_row(y0, true, raw1, hraw1);
  load_row(in2 ? y0 + 1 : y0, raw0, hraw0);
  commit_row(y0 + 1, in2, raw0, hraw0);
  __syncthreads();

  double s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
  const int lbase = kq * PS + lm;
  char* out_img = reinterpret_cast<char*>(a.out + ((int64_t)n * a.h * a.iw + x0) * a.out_cs + a.out_co);
  const unsigned out_row = (unsigned)(a.iw * a.out_cs) * 4u;
  const unsigned o_off = (unsigned)(lm * a.out_cs + 64 * half + 16 * wave + 4 * kq) * 4u, o_g = (unsigned)(16 * a.out_cs) * 4u;

  v4f acc[2][G];
  float4 xf[2][G];
  auto epi = [&](auto P_, int g, int yp) {              // lane (lm, kq): channels 64 half + 16 wave + 4 kq + [0, 4) of pixel 16 g + lm
    constexpr int P = decltype(P_)::value;
    const v4f v = acc[P][g];
    char* rowp = out_img + (size_t)((unsigned)yp * out_row);
    *reinterpret_cast<float4*>(rowp + (o_off + (unsigned)g * o_g)) = make_float4(v[0], v[1], v[2], v[3]);
    if constexpr (STATS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const double d = (double)v[j]; s1[j] += d; s2[j] = fma(d, d, s2[j]); }
    }
  };

  // row pipeline as in conv_bf16_ws.hip: while row y multiplies into acc[P], row y - 1 (acc[P ^ 1]) is stored and summed and
  // input row y + 2 is activated into the free ring slot, in slices spread over the 72 K-step groups of the row
  auto row = [&](auto P_, auto PREV_, int y) {
    constexpr int P = decltype(P_)::value;
    constexpr bool PREV = decltype(PREV_)::value;
    const bool in_next = y + 2 < a.h;
    load_row(in_next ? y + 2 : a.h - 1, raw0, hraw0);
    const int rr_next = (y + 3) & (WF_R - 1);
    const unsigned keep = in_next ? 0xffffffffu : 0u;   // rows below the image are zero (a mask: no branch, and no 0 * NaN)
    int rbase[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) rbase[ty] = lbase + ((y + ty) & (WF_R - 1)) * RP;
#pragma unroll
    for (int s = 0; s < 72; ++s) {                      // s = (ty * 3 + tx) * 8 + q
      {
        const int sn = (s + 1) % 72;
        const int tn = sn >> 3, qn = sn & 7;
        const int base = (s == 71 ? rbase[1] : rbase[tn / 3]) + 4 * qn * PS + tn % 3;
#pragma unroll
        for (int g = 0; g < G; ++g) xf[(s + 1) & 1][g] = img[base + 16 * g];
      }
      if constexpr (PREV) {
        if (s >= 2 && s < 2 + G) epi(std::integral_constant<int, P ^ 1>{}, s - 2, y - 1);
      }
      if (s >= 16 && s < 16 + NU) {
        const int i = s - 16;
        float4 v = act4(s_q, raw0[i]);
        auto m = [&](float f) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, f) & keep); };
        img[s_slot0 + 8 * i + rr_next * RP] = make_float4(m(v.x), m(v.y), m(v.z), m(v.w));
      }
      if (s == 16 + NU) commit_halo(rr_next, keep, hraw0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 x = xf[s & 1][g];
          const float b = j == 0 ? x.x : j == 1 ? x.y : j == 2 ? x.z : x.w;
          const v4f z = {0.f, 0.f, 0.f, 0.f};
          acc[P][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[4 * s + j], b, (s == 0 && j == 0) ? z : acc[P][g], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  {
    const int rb0 = lbase + (y0 & (WF_R - 1)) * RP;
#pragma unroll
    for (int g = 0; g < G; ++g) xf[0][g] = img[rb0 + 16 * g];
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  row(I0{}, std::false_type{}, y0);
  int y = y0 + 1;
  for (; y + 1 < y1; y += 2) {
    row(I1{}, std::true_type{}, y);
    row(I0{}, std::true_type{}, y + 1);
  }
  if (y < y1) {
    row(I1{}, std::true_type{}, y);
#pragma unroll
    for (int g = 0; g < G; ++g) epi(I1{}, g, y);
  } else {
#pragma unroll
    for (int g = 0; g < G; ++g) epi(I0{}, g, y - 1);
  }

  if constexpr (STATS) {
    // a lane's doubles -> LDS [stat][channel 64][lm] -> this workgroup's 64 columns of the band's row (fixed order)
    double* red = reinterpret_cast<double*>(smem_f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = 16 * wave + 4 * kq + j;
      red[ch * 16 + lm] = s1[j];
      red[(64 + ch) * 16 + lm] = s2[j];
    }
    __syncthreads();
    if (tid < 128) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[tid * 16 + i];
      const int st = tid >> 6, ch = tid & 63;
      a.stat[((int64_t)item * 2 + st) * WF_C + 64 * half + ch] = t;
    }
  }
}

bool wf_enabled() {
  static const bool off = getenv("BP_F32_WS") && atoi(getenv("BP_F32_WS")) == 0;
  return !off;
}
int g_wf_override = -1;

// strip width / 16: the image itself up to 64 pixels, 64-pixel strips of wider images (multiples of 64)
int wf_G(int w) { return w == 64 ? 4 : w == 32 ? 2 : w == 16 ? 1 : (w > 64 && w % 64 == 0) ? 4 : 0; }
int wf_strips(int w) { return w > 64 ? w / 64 : 1; }

// rows per band: ~256 workgroups (two per band and strip: the channel halves), bands of >= 4 rows
void wf_bands(int n, int h, int strips, int* BR, int* bands) {
  int br = h;
  while (br > 4 && (int64_t)n * strips * bp_ceil_div(h, br) * 2 < 256) br = bp_ceil_div(br, 2);
  *BR = br;
  *bands = bp_ceil_div(h, br);
}

template <int G, bool ACT, bool STATS, bool STRIPS>
int wf_launch(const WfArgs& a, unsigned grid, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&ws3_f32_kernel<G, ACT, STATS, STRIPS>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)WfGeom<G>::lds_bytes);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((ws3_f32_kernel<G, ACT, STATS, STRIPS>), dim3(grid), dim3(256), WfGeom<G>::lds_bytes, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
template <int G, bool STRIPS = false>
int wf_launch_g(const WfArgs& a, bool act, bool stats, unsigned grid, hipStream_t st) {
  if (act) return stats ? wf_launch<G, true, true, STRIPS>(a, grid, st) : wf_launch<G, true, false, STRIPS>(a, grid, st);
  return stats ? wf_launch<G, false, true, STRIPS>(a, grid, st) : wf_launch<G, false, false, STRIPS>(a, grid, st);
}

}  // namespace

void bp_f32_ws_set(int v) { g_wf_override = v; }

bool bp_f32_ws_geom_ok(const ConvGeom& g) {
  return g.k == 3 && g.stride == 1 && g.pad == 1 && g.cin_g == WF_C && g.cout_g == WF_C && g.nphase == 1 && g.taps == 3 &&
         g.IS == 1 && g.OS == 1;
}

// `stats_mode`: 0 none, 1 batch-norm sums of the produced tensor (anything else: the tiled kernel)
bool bp_f32_ws_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int stats_mode) {
  if (!(g_wf_override < 0 ? wf_enabled() : g_wf_override != 0)) return false;
  if (!bp_f32_ws_geom_ok(g) || !in || !out || bias || (stats_mode != 0 && stats_mode != 1)) return false;
  if (in->dtype != BP_F32 || out->dtype != BP_F32 || in->c != WF_C || out->c != WF_C) return false;
  if (in->n != out->n || in->h != out->h || in->w != out->w || !wf_G(out->w)) return false;
  if (!bp_view_vec4(in) || !bp_view_vec4(out)) return false;
  if ((int64_t)in->h * in->w * in->cstride * 4 >= (int64_t)1 << 31 || (int64_t)out->h * out->w * out->cstride * 4 >= (int64_t)1 << 31)
    return false;
  int BR, bands;
  wf_bands(out->n, out->h, wf_strips(out->w), &BR, &bands);
  return (int64_t)out->n * bands * wf_strips(out->w) * 2 <= 0x7fffffff;
}

size_t bp_f32_ws_stats_workspace(const ConvGeom& g, const bp_view* out) {
  int BR, bands;
  wf_bands(out->n, out->h, wf_strips(out->w), &BR, &bands);
  return bp_stats_rows_bytes((int64_t)out->n * bands * wf_strips(out->w), g.cout_g);
}

int bp_f32_ws_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed_tiled, const bp_view* out,
                  hipStream_t st, const IgemmStatsReq* sr) {
  WfArgs a{};
  a.in = in->ptr; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_cs = out->cstride; a.out_co = out->coff;
  a.n = out->n; a.h = out->h; a.wp = packed_tiled; a.pw = pw;
  a.iw = out->w; a.strips = wf_strips(out->w);
  wf_bands(out->n, out->h, a.strips, &a.BR, &a.bands);
  const int64_t rows = (int64_t)out->n * a.bands * a.strips;
  a.items = (int)rows;
  if (sr) {
    const size_t need = bp_stats_rows_bytes(rows, g.cout_g);
    if (sr->mode != 1 || !need) return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < need || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  const bool act = pw.scale != nullptr;
  int rc;
  switch (wf_G(out->w)) {
    case 4: rc = a.strips > 1 ? wf_launch_g<4, true>(a, act, sr != nullptr, (unsigned)(bp_ceil_div((int)rows, 8) * 16), st)
                              : wf_launch_g<4>(a, act, sr != nullptr, (unsigned)(bp_ceil_div((int)rows, 8) * 16), st);
            break;
    case 2: rc = wf_launch_g<2>(a, act, sr != nullptr, (unsigned)(bp_ceil_div((int)rows, 8) * 16), st); break;
    default: rc = wf_launch_g<1>(a, act, sr != nullptr, (unsigned)(bp_ceil_div((int)rows, 8) * 16), st); break;
  }
  if (rc != BP_OK || !sr) return rc;
  return bp_stats_rows_finish(a.stat, rows, g.cout_g, sr, st);
}
