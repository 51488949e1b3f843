// Weights-stationary bf16 convolution of the generator's residual trunk: Conv2d k3 s1 p1, 128 -> 128 channels
// (/root/reference/baryon_painter/models/utils.py:22-38 ResidualBlock, 8 layers = 47 % of the step's FLOPs), forward
// and data gradient (the same correlation with the taps mirrored and the channel roles swapped: a different weight
// image, the same kernel).
//
// Why a kernel of its own: the tiled igemm_bf16_kernel streams the layer's 295 KB of weights through LDS once per
// 256-pixel tile -- five times the bytes of the tile's input -- and reads every MFMA operand out of LDS; it runs at
// 0.29 of the bf16 matrix peak (profiles/r03_mfma_util_bf16.txt), bound by LDS fragment reads and slab barriers.
// Here the weights never move after the prologue:
//   * one workgroup of four waves (one per SIMD, up to 512 registers each) per CU; wave w owns produced channels
//     [32 w, 32 w + 32) and keeps ITS weights for all 9 taps x 128 gathered channels in registers: 72 MFMA A-fragments
//     = 288 VGPR/AGPRs per lane (a CU's register file is 512 KB; the layer's weights are 295 KB);
//   * the workgroup walks a band of image rows top to bottom with a ring of four input rows in LDS (one new row per
//     produced row: every input byte is fetched once per band, + 2 halo rows per band);
//   * per K-step (tap, 32 channels) a wave reads one 16-byte B-fragment per 16 pixels and uses it for two MFMAs
//     (v_mfma_f32_16x16x32_bf16, its two 16-channel blocks): 128 B/clk of LDS reads per CU = half the LDS peak, no
//     weight reads at all, one barrier per ROW (9216 MFMAs per CU between barriers at W = 64);
//   * LDS image: 16 planes (one per 8-channel octet) of [ring row][pixel + 2 zero columns] 16-byte slots, plane stride
//     a multiple of 16 slots: the 16 lanes of every ds_read_b128 lane group hit 16 distinct bank quads whatever the
//     tap offset; staging writes 8 consecutive pixels of one plane per 8 lanes (conflict-free ds_write_b128);
//   * the producer's pending batch-norm + (leaky) ReLU is applied on the way into LDS, once per element (the 4 waves
//     all read the same image), NaN-propagating like torch.relu;
//   * epilogue per row from the accumulators: 16-byte bf16 stores (a lane holds 8 consecutive channels of a pixel: the
//     A-fragment rows are ordered so), training-mode batch-norm sums {sum y, sum y^2} of the values AS STORED per lane
//     in fp32 over the band (<= 256 terms), folded in double through LDS once per workgroup: one row per workgroup.
#include "conv_bf16.hpp"
#include <type_traits>

size_t bp_stats_rows_bytes(int64_t rows, int C);
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st);

namespace {
using namespace bpbf16;

constexpr int WS_C = 128;            // channels on both sides
constexpr int WS_R = 4;              // ring rows
constexpr int WS_FV = 32; // ... of which in VGPRs (the rest in AGPRs)
constexpr int WS_NF = 72;            // A-fragments per wave: 9 taps x 4 chunks of 32 channels x 2 blocks of 16 channels

struct WsArgs {
  const u16* in; int in_cs, in_co;
  u16* out; int out_cs, out_co;
  int n, h;
  const u16* wp;                     // [wave 4][fragment 72][lane 64][8]
  PW pw;
  int BR, bands;
  double* stat;                      // rows [workgroup][2][128]
};

template <int G> struct WsGeom {
  static constexpr int W = 16 * G, RP = W + 2;
  static constexpr int PS = (WS_R * RP + 15) / 16 * 16;         // 16-byte slots per plane
  static constexpr size_t img_bytes = (size_t)16 * PS * 16;
  static constexpr size_t lds_bytes = img_bytes + 3 * WS_C * sizeof(float);
};

// weight image: fragment f = ((ty*3 + tx)*4 + chunk)*2 + block of wave w; lane (lm, kq) holds A[row lm][k = 8 kq + j]:
// produced channel 32 w + 8 (lm >> 2) + 4 block + (lm & 3), gathered channel 32 chunk + 8 kq + j
struct WsPackArgs {
  const float* w; u16* dst;
  int64_t sa, sb;
  int transposed;
};
__global__ __launch_bounds__(256) void ws_pack_kernel(WsPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * WS_NF * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, f = (i >> 9) % WS_NF, w = i / (512 * WS_NF);
  const int lm = lane & 15, kq = lane >> 4;
  const int blk = f & 1, chunk = (f >> 1) & 3, tap = f >> 3;
  const int ty = tap / 3, tx = tap % 3;
  const int co = 32 * w + 8 * (lm >> 2) + 4 * blk + (lm & 3);
  const int ci = 32 * chunk + 8 * kq + j;
  // the gather-transposed form (data gradient of a Conv2d) visits the taps mirrored: bp_t_ky(0, 1, 1, 3, t) = 2 - t
  const int ky = a.transposed ? 2 - ty : ty, kx = a.transposed ? 2 - tx : tx;
  a.dst[i] = f2bf(a.w[ci * a.sa + co * a.sb + ky * 3 + kx]);
}

// scheduling masks of __builtin_amdgcn_sched_group_barrier
#define WS_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_VALU = 0x2, SG_MFMA = 0x8, SG_DSR = 0x100;

template <int G, bool ACT, bool STATS>
__global__ __launch_bounds__(256) void ws3_bf16_kernel(WsArgs a) {
  using GM = WsGeom<G>;
  constexpr int W = GM::W, RP = GM::RP, PS = GM::PS;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  uint4* img = reinterpret_cast<uint4*>(smem);
  float* lpw = reinterpret_cast<float*>(smem + GM::img_bytes / 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
  const int y0 = band * a.BR;
  const int y1 = min(y0 + a.BR, a.h);

  // staging units of this thread: unit i = 8 pixels x 8 octets per wave instruction
  //   octet = (lane >> 3) + 8 * (ub & 1), pixel = (lane & 7) + 8 * (ub >> 1), ub = wave + 4 i
  // addresses: a wave-uniform image base + a 32-bit byte offset per lane (bp_bf16_ws_ok bounds the tensor)
  int s_slot[G], s_oc[G];
  unsigned s_off[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int ub = wave + 4 * i;
    s_oc[i] = (lane >> 3) + 8 * (ub & 1);
    const int px = (lane & 7) + 8 * (ub >> 1);
    s_slot[i] = s_oc[i] * PS + 1 + px;
    s_off[i] = (unsigned)(px * a.in_cs + s_oc[i] * 8) * 2u;
  }
  const char* in_img = reinterpret_cast<const char*>(a.in + (int64_t)n * a.h * W * a.in_cs + a.in_co);
  const unsigned in_row = (unsigned)(W * a.in_cs) * 2u;
  auto load_row = [&](int r, uint4 (&raw)[G]) {        // (r inside the image)
    const char* rowp = in_img + (size_t)((unsigned)r * in_row);           // wave-uniform: a scalar base + a lane offset
#pragma unroll
    for (int i = 0; i < G; ++i) raw[i] = *reinterpret_cast<const uint4*>(rowp + s_off[i]);
  };
  // one half (4 channels: words 2h, 2h + 1) of a unit through the pending activation, in two phases that the row loop
  // places in different K-steps: (A) parameters + unpack + affine, (B) leaky ReLU + rounding
  auto act_a = [&](int oc, int h, unsigned w0, unsigned w1, float (&t)[4], float (&sl)[4]) {
    const float4 sc = *reinterpret_cast<const float4*>(lpw + oc * 8 + 4 * h);
    const float4 sf = *reinterpret_cast<const float4*>(lpw + WS_C + oc * 8 + 4 * h);
    const float4 sv = *reinterpret_cast<const float4*>(lpw + 2 * WS_C + oc * 8 + 4 * h);
    t[0] = fmaf(bf2f((u16)(w0 & 0xffffu)), sc.x, sf.x); t[1] = fmaf(bf2f((u16)(w0 >> 16)), sc.y, sf.y);
    t[2] = fmaf(bf2f((u16)(w1 & 0xffffu)), sc.z, sf.z); t[3] = fmaf(bf2f((u16)(w1 >> 16)), sc.w, sf.w);
    sl[0] = sv.x; sl[1] = sv.y; sl[2] = sv.z; sl[3] = sv.w;
  };
  auto act_b = [&](const float (&t)[4], const float (&sl)[4], unsigned& o0, unsigned& o1) {
    const float u0 = t[0] > 0.f ? t[0] : t[0] * sl[0], u1 = t[1] > 0.f ? t[1] : t[1] * sl[1];      // (a NaN stays a NaN, as torch.relu)
    const float u2 = t[2] > 0.f ? t[2] : t[2] * sl[2], u3 = t[3] > 0.f ? t[3] : t[3] * sl[3];
    o0 = pack2(u0, u1); o1 = pack2(u2, u3);
  };
  auto act_half = [&](int oc, int h, unsigned w0, unsigned w1, unsigned& o0, unsigned& o1) {
    float t[4], sl[4];
    act_a(oc, h, w0, w1, t, sl);
    act_b(t, sl, o0, o1);
  };
  auto commit_unit = [&](int i, int rr, bool inside, uint4 v) {
    if constexpr (ACT) {
      act_half(s_oc[i], 0, v.x, v.y, v.x, v.y);
      act_half(s_oc[i], 1, v.z, v.w, v.z, v.w);
    }
    if (!inside) v = make_uint4(0u, 0u, 0u, 0u);
    img[s_slot[i] + rr * RP] = v;
  };
  auto commit_row = [&](int r, bool inside, const uint4 (&raw)[G]) {
    const int rr = (r + 1) & (WS_R - 1);
#pragma unroll
    for (int i = 0; i < G; ++i) commit_unit(i, rr, inside, raw[i]);
  };

  // ---- prologue: first three input rows requested, then the weights; zero columns; activation parameters
  uint4 raw0[G], raw1[G], raw2[G];
  const bool in0 = y0 - 1 >= 0, in2 = y0 + 1 < a.h;
  load_row(in0 ? y0 - 1 : y0, raw0);
  load_row(y0, raw1);
  load_row(in2 ? y0 + 1 : y0, raw2);
  bf8 wf[WS_NF];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wp) + (size_t)wave * WS_NF * 64 + lane;
#pragma unroll
    for (int f = 0; f < WS_NF; ++f) wf[f] = __builtin_bit_cast(bf8, wsrc[f * 64]);
    // Register classes by hand: the first WS_FV fragments live in VGPRs, the others in AGPRs (an MFMA takes either as its
    // A operand).  Left to itself the allocator fills the 256 VGPRs, spills the rest to AGPRs and copies them back in
    // front of every use (181 v_accvgpr moves per 288 MFMAs, clustered at the head of each K-step).
#pragma unroll
    for (int f = 0; f < WS_NF; ++f) {
      if (f < WS_FV) asm volatile("" : "+v"(wf[f]));
      else asm volatile("" : "+a"(wf[f]));
    }
  }
  if (tid < 128) {
    const int plane = tid >> 3, rr = (tid >> 1) & 3, side = tid & 1;
    img[plane * PS + rr * RP + (side ? RP - 1 : 0)] = make_uint4(0u, 0u, 0u, 0u);
  }
  if constexpr (ACT) {
    for (int i = tid; i < WS_C; i += 256) { lpw[i] = a.pw.scale[i]; lpw[WS_C + i] = a.pw.shift[i]; lpw[2 * WS_C + i] = a.pw.slope[i]; }
    __syncthreads();
  }
  commit_row(y0 - 1, in0, raw0);
  commit_row(y0, true, raw1);
  commit_row(y0 + 1, in2, raw2);
  __syncthreads();

  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int lbase = kq * PS + lm;                       // slot of (plane kq, pixel lm) in ring row 0
  char* out_img = reinterpret_cast<char*>(a.out + (int64_t)n * a.h * W * a.out_cs + a.out_co);
  const unsigned out_row = (unsigned)(W * a.out_cs) * 2u;
  const unsigned o_off = (unsigned)(lm * a.out_cs + 32 * wave + 8 * kq) * 2u, o_g = (unsigned)(16 * a.out_cs) * 2u;

  // Software pipeline over rows: while row y is multiplied into acc[P], the accumulators of row y - 1 (acc[P ^ 1]) are
  // rounded, stored and summed, and input row y + 2 -- requested at the top of the row -- goes through its activation
  // into the ring slot nobody reads this row; both in slices of a few vector instructions placed between the MFMAs of a
  // K-step by sched_group_barrier patterns (one wave per SIMD: nobody else would fill the gaps).  The B fragments of
  // step s + 1 (the next row's step 0 included: its input row is resident) are requested before the MFMAs of step s.
  v4f acc[2][G][2];
  bf8 xf[2][G];
  uint4 pk[G];
  auto epi_pack = [&](auto P_, int g, int yp) {         // row yp from acc[P]: a lane's 8 consecutive channels of a pixel
    constexpr int P = decltype(P_)::value;
    pk[g] = make_uint4(pack2(acc[P][g][0][0], acc[P][g][0][1]), pack2(acc[P][g][0][2], acc[P][g][0][3]),
                       pack2(acc[P][g][1][0], acc[P][g][1][1]), pack2(acc[P][g][1][2], acc[P][g][1][3]));
    char* rowp = out_img + (size_t)((unsigned)yp * out_row);
    *reinterpret_cast<uint4*>(rowp + (o_off + (unsigned)g * o_g)) = pk[g];
  };
  auto epi_stats = [&](int g, int h) {                  // channels 4h .. 4h + 3 of the values AS STORED
    const unsigned w0 = h ? pk[g].z : pk[g].x, w1 = h ? pk[g].w : pk[g].y;
    const float v0 = bf2f((u16)(w0 & 0xffffu)), v1 = bf2f((u16)(w0 >> 16)), v2 = bf2f((u16)(w1 & 0xffffu)), v3 = bf2f((u16)(w1 >> 16));
    s1[4 * h] += v0; s2[4 * h] = fmaf(v0, v0, s2[4 * h]);
    s1[4 * h + 1] += v1; s2[4 * h + 1] = fmaf(v1, v1, s2[4 * h + 1]);
    s1[4 * h + 2] += v2; s2[4 * h + 2] = fmaf(v2, v2, s2[4 * h + 2]);
    s1[4 * h + 3] += v3; s2[4 * h + 3] = fmaf(v3, v3, s2[4 * h + 3]);
  };
  auto epi_all = [&](auto P_, int yp) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      epi_pack(P_, g, yp);
      if constexpr (STATS) { epi_stats(g, 0); epi_stats(g, 1); }
    }
  };

  auto row = [&](auto P_, auto PREV_, int y) {
    constexpr int P = decltype(P_)::value;
    constexpr bool PREV = decltype(PREV_)::value;
    const bool in_next = y + 2 < a.h;
    load_row(in_next ? y + 2 : a.h - 1, raw0);          // (always: beyond the band's last row it lands in a free slot)
    const int rr_next = (y + 3) & (WS_R - 1);
    const unsigned keep = in_next ? 0xffffffffu : 0u;   // rows below the image are zero
    int rbase[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) rbase[ty] = lbase + ((y + ty) & (WS_R - 1)) * RP;      // input row y - 1 + ty
    uint4 cv = make_uint4(0u, 0u, 0u, 0u);
    float ct[4] = {0.f, 0.f, 0.f, 0.f}, csl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const int ty = s / 12, c = (s / 3) & 3, tx = s % 3;
      // -- fragment reads of the next step (s == 35: step 0 of row y + 1, whose tap row 0 is input row y)
      {
        const int s1_ = (s + 1) % 36;
        const int ty1 = s1_ / 12, c1 = (s1_ / 3) & 3, tx1 = s1_ % 3;
        const int base1 = s == 35 ? rbase[1] : rbase[ty1];
#pragma unroll
        for (int g = 0; g < G; ++g) xf[(s + 1) & 1][g] = __builtin_bit_cast(bf8, img[base1 + 4 * c1 * PS + 16 * g + tx1]);
      }
      // -- side work of this step: one slice of <= 16 vector instructions (two per MFMA are free: an MFMA holds the
      //    SIMD's issue port for 8 of its 16 cycles, a vector instruction for 4)
      //    steps 1 .. 3G: previous row, group g: round + store | sums of channels 0-3 | of channels 4-7
      //    steps 13 .. 13 + 4G: input row y + 2, unit i: half 0 affine | half 0 ReLU | half 1 affine | half 1 ReLU + write
      if constexpr (PREV) {
        if (s >= 1 && s < 1 + 3 * G) {
          const int g = (s - 1) / 3, part = (s - 1) % 3;
          if (part == 0) epi_pack(std::integral_constant<int, P ^ 1>{}, g, y - 1);
          else if constexpr (STATS) epi_stats(g, part - 1);
        }
      }
      if (s >= 13 && s < 13 + 4 * G) {
        const int i = (s - 13) / 4, ph = (s - 13) % 4;
        if constexpr (ACT) {
          if (ph == 0) { cv = raw0[i]; act_a(s_oc[i], 0, cv.x, cv.y, ct, csl); }
          else if (ph == 1) act_b(ct, csl, cv.x, cv.y);
          else if (ph == 2) act_a(s_oc[i], 1, cv.z, cv.w, ct, csl);
          else act_b(ct, csl, cv.z, cv.w);
        } else {
          if (ph == 0) cv = raw0[i];
        }
        if (ph == 3) {          // (a mask, not a branch: a branch would cut the K-step's scheduling region in two)
          img[s_slot[i] + rr_next * RP] = make_uint4(cv.x & keep, cv.y & keep, cv.z & keep, cv.w & keep);
        }
      }
      // -- the step's MFMAs
      const int f = ((ty * 3 + tx) * 4 + c) * 2;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        acc[P][g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf[s & 1][g], s == 0 ? z : acc[P][g][0], 0, 0, 0);
        acc[P][g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f + 1], xf[s & 1][g], s == 0 ? z : acc[P][g][1], 0, 0, 0);
      }
      // -- issue order: a fragment read, then its share of the MFMAs with the side work's vector instructions between
#pragma unroll
      for (int g = 0; g < G; ++g) {
        WS_SGB(SG_DSR, 1);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  {
    // step 0 of the first row
    const int rb0 = lbase + (y0 & (WS_R - 1)) * RP;
#pragma unroll
    for (int g = 0; g < G; ++g) xf[0][g] = __builtin_bit_cast(bf8, img[rb0 + 16 * g]);
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
    epi_all(I1{}, y);
  } else {
    epi_all(I0{}, y - 1);
  }

  if constexpr (STATS) {
    // per-lane fp32 partials -> LDS [stat][channel][lm] -> one double row per workgroup (fixed order)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = 32 * wave + 8 * kq + j;
      red[ch * 16 + lm] = s1[j];
      red[(WS_C + ch) * 16 + lm] = s2[j];
    }
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += (double)red[tid * 16 + i];
    a.stat[(int64_t)blockIdx.x * 2 * WS_C + tid] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same scheme for the strided gather form  k4 s2 p1, 64 -> 128 channels  (forward of p_y_z_in.9, data gradient of the
// transposed layer p_y_z_in.16: /root/reference/trained_models/CVAE/fiducial/architecture_built.txt:51,94): 262 KB of
// weights = 64 A-fragments per wave; one produced row gathers four input rows (two new ones per produced row: a ring of
// six), the fine-grid pixels split by column parity into two sets of eight octet planes so that the 16 pixels of a
// B-fragment (stride 2 on the fine grid) are consecutive slots of one plane.
constexpr int W4_CI = 64, W4_R = 6, W4_NF = 64, W4_FV = 16;
template <int G> struct W4Geom {
  static constexpr int W = 16 * G, RP = W + 2;
  static constexpr int PS = (W4_R * RP + 15) / 16 * 16;
  static constexpr size_t img_bytes = (size_t)16 * PS * 16;
  static constexpr size_t lds_bytes = img_bytes + 3 * W4_CI * sizeof(float);
};

// fragment f = ((ky*4 + kx)*2 + chunk)*2 + block; lane (lm, kq): produced channel 32 w + 8 (lm >> 2) + 4 block + (lm & 3),
// gathered channel 32 chunk + 8 kq + j
__global__ __launch_bounds__(256) void ws4_pack_kernel(WsPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * W4_NF * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, f = (i >> 9) % W4_NF, w = i / (512 * W4_NF);
  const int lm = lane & 15, kq = lane >> 4;
  const int blk = f & 1, chunk = (f >> 1) & 1, tap = f >> 2;
  const int ky = tap >> 2, kx = tap & 3;
  const int co = 32 * w + 8 * (lm >> 2) + 4 * blk + (lm & 3);
  const int ci = 32 * chunk + 8 * kq + j;
  a.dst[i] = f2bf(a.w[ci * a.sa + co * a.sb + ky * 4 + kx]);
}

__device__ __forceinline__ void ws_act_a(const float* lpw, int C, int oc, int h, unsigned w0, unsigned w1, float (&t)[4], float (&sl)[4]) {
  const float4 sc = *reinterpret_cast<const float4*>(lpw + oc * 8 + 4 * h);
  const float4 sf = *reinterpret_cast<const float4*>(lpw + C + oc * 8 + 4 * h);
  const float4 sv = *reinterpret_cast<const float4*>(lpw + 2 * C + oc * 8 + 4 * h);
  t[0] = fmaf(bf2f((u16)(w0 & 0xffffu)), sc.x, sf.x); t[1] = fmaf(bf2f((u16)(w0 >> 16)), sc.y, sf.y);
  t[2] = fmaf(bf2f((u16)(w1 & 0xffffu)), sc.z, sf.z); t[3] = fmaf(bf2f((u16)(w1 >> 16)), sc.w, sf.w);
  sl[0] = sv.x; sl[1] = sv.y; sl[2] = sv.z; sl[3] = sv.w;
}
__device__ __forceinline__ void ws_act_b(const float (&t)[4], const float (&sl)[4], unsigned& o0, unsigned& o1) {
  const float u0 = t[0] > 0.f ? t[0] : t[0] * sl[0], u1 = t[1] > 0.f ? t[1] : t[1] * sl[1];      // (a NaN stays a NaN)
  const float u2 = t[2] > 0.f ? t[2] : t[2] * sl[2], u3 = t[3] > 0.f ? t[3] : t[3] * sl[3];
  o0 = pack2(u0, u1); o1 = pack2(u2, u3);
}

template <int G, bool ACT, bool STATS>
__global__ __launch_bounds__(256) void ws4_bf16_kernel(WsArgs a) {
  using GM = W4Geom<G>;
  constexpr int W = GM::W, RP = GM::RP, PS = GM::PS;
  constexpr int NS = 32;                                // K-steps per produced row: 16 taps x 2 chunks of 32 channels
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  uint4* img = reinterpret_cast<uint4*>(smem);          // [parity 2][octet 8][ring row 6][RP]
  float* lpw = reinterpret_cast<float*>(smem + GM::img_bytes / 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, kq = lane >> 4;
  const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
  const int y0 = band * a.BR;
  const int y1 = min(y0 + a.BR, a.h);                   // produced rows [y0, y1); a.h produced rows, 2 a.h gathered rows
  const int hin = 2 * a.h;

  // staging units of this thread, per gathered row: unit j = 8 fine pixels x 8 octets per wave instruction,
  // fine pixel (lane & 7) + 8 (wave + 4 j), octet lane >> 3; the pixel's parity (lane & 1) picks the plane set
  const int s_oc = lane >> 3;
  const int s_slot0 = ((lane & 1) * 8 + s_oc) * PS + 1 + ((lane & 7) >> 1) + 4 * wave;          // + 16 j + ring * RP
  const unsigned s_off0 = (unsigned)(((lane & 7) + 8 * wave) * a.in_cs + s_oc * 8) * 2u;       // + j * s_offj
  const unsigned s_offj = (unsigned)(32 * a.in_cs) * 2u;
  const char* in_img = reinterpret_cast<const char*>(a.in + (int64_t)n * hin * (2 * W) * a.in_cs + a.in_co);
  const unsigned in_row = (unsigned)(2 * W * a.in_cs) * 2u;
  auto load_row = [&](int r, uint4 (&raw)[G]) {         // (r inside the image)
    const char* rowp = in_img + (size_t)((unsigned)r * in_row);
#pragma unroll
    for (int j = 0; j < G; ++j) raw[j] = *reinterpret_cast<const uint4*>(rowp + (s_off0 + (unsigned)j * s_offj));
  };
  auto commit_row = [&](int ring, bool inside, const uint4 (&raw)[G]) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      uint4 v = raw[j];
      if constexpr (ACT) {
        float t[4], sl[4];
        ws_act_a(lpw, W4_CI, s_oc, 0, v.x, v.y, t, sl); ws_act_b(t, sl, v.x, v.y);
        ws_act_a(lpw, W4_CI, s_oc, 1, v.z, v.w, t, sl); ws_act_b(t, sl, v.z, v.w);
      }
      if (!inside) v = make_uint4(0u, 0u, 0u, 0u);
      img[s_slot0 + 16 * j + ring * RP] = v;
    }
  };
  // ring slot of gathered row r (>= -1): (r + 1) % 6
  auto ring_of = [](int r) { return (r + 1) % W4_R; };

  // ---- prologue: gathered rows 2 y0 - 1 .. 2 y0 + 2, the weights, zero columns, activation parameters
  uint4 rawa[G], rawb[G];
  const int r0 = 2 * y0 - 1;
  load_row(r0 >= 0 ? r0 : 0, rawa);
  load_row(r0 + 1, rawb);
  bf8 wf[W4_NF];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wp) + (size_t)wave * W4_NF * 64 + lane;
#pragma unroll
    for (int f = 0; f < W4_NF; ++f) wf[f] = __builtin_bit_cast(bf8, wsrc[f * 64]);
#pragma unroll
    for (int f = 0; f < W4_NF; ++f) {
      if (f < W4_FV) asm volatile("" : "+v"(wf[f]));
      else asm volatile("" : "+a"(wf[f]));
    }
  }
  if (tid < 192) {
    const int plane = tid / 12, rr = (tid / 2) % 6, side = tid & 1;
    img[plane * PS + rr * RP + (side ? RP - 1 : 0)] = make_uint4(0u, 0u, 0u, 0u);
  }
  if constexpr (ACT) {
    for (int i = tid; i < W4_CI; i += 256) { lpw[i] = a.pw.scale[i]; lpw[W4_CI + i] = a.pw.shift[i]; lpw[2 * W4_CI + i] = a.pw.slope[i]; }
    __syncthreads();
  }
  commit_row(ring_of(r0), r0 >= 0, rawa);
  commit_row(ring_of(r0 + 1), true, rawb);
  load_row(r0 + 2, rawa);
  load_row(r0 + 3 < hin ? r0 + 3 : hin - 1, rawb);
  commit_row(ring_of(r0 + 2), true, rawa);
  commit_row(ring_of(r0 + 3), r0 + 3 < hin, rawb);
  __syncthreads();

  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int lbase = kq * PS + lm;
  char* out_img = reinterpret_cast<char*>(a.out + (int64_t)n * a.h * W * a.out_cs + a.out_co);
  const unsigned out_row = (unsigned)(W * a.out_cs) * 2u;
  const unsigned o_off = (unsigned)(lm * a.out_cs + 32 * wave + 8 * kq) * 2u, o_g = (unsigned)(16 * a.out_cs) * 2u;

  v4f acc[2][G][2];
  bf8 xf[2][G];
  uint4 pk[G];
  auto epi_pack = [&](auto P_, int g, int yp) {
    constexpr int P = decltype(P_)::value;
    pk[g] = make_uint4(pack2(acc[P][g][0][0], acc[P][g][0][1]), pack2(acc[P][g][0][2], acc[P][g][0][3]),
                       pack2(acc[P][g][1][0], acc[P][g][1][1]), pack2(acc[P][g][1][2], acc[P][g][1][3]));
    char* rowp = out_img + (size_t)((unsigned)yp * out_row);
    *reinterpret_cast<uint4*>(rowp + (o_off + (unsigned)g * o_g)) = pk[g];
  };
  auto epi_stats = [&](int g, int h) {
    const unsigned w0 = h ? pk[g].z : pk[g].x, w1 = h ? pk[g].w : pk[g].y;
    const float v0 = bf2f((u16)(w0 & 0xffffu)), v1 = bf2f((u16)(w0 >> 16)), v2 = bf2f((u16)(w1 & 0xffffu)), v3 = bf2f((u16)(w1 >> 16));
    s1[4 * h] += v0; s2[4 * h] = fmaf(v0, v0, s2[4 * h]);
    s1[4 * h + 1] += v1; s2[4 * h + 1] = fmaf(v1, v1, s2[4 * h + 1]);
    s1[4 * h + 2] += v2; s2[4 * h + 2] = fmaf(v2, v2, s2[4 * h + 2]);
    s1[4 * h + 3] += v3; s2[4 * h + 3] = fmaf(v3, v3, s2[4 * h + 3]);
  };
  auto epi_all = [&](auto P_, int yp) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      epi_pack(P_, g, yp);
      if constexpr (STATS) { epi_stats(g, 0); epi_stats(g, 1); }
    }
  };

  // row y: gathered rows 2y - 1 + t, t = 0..3, live in ring slots (rb + t) % 6, rb = (2 y) % 6; rows t = 4, 5 are requested at
  // the top and committed in the second half of the K-steps
  auto row = [&](auto P_, auto PREV_, int y, int rb) {
    constexpr int P = decltype(P_)::value;
    constexpr bool PREV = decltype(PREV_)::value;
    const int ra = 2 * y + 3, rbw = 2 * y + 4;
    const bool in_a = ra < hin, in_b = rbw < hin;
    load_row(in_a ? ra : hin - 1, rawa);
    load_row(in_b ? rbw : hin - 1, rawb);
    const unsigned keep_a = in_a ? 0xffffffffu : 0u, keep_b = in_b ? 0xffffffffu : 0u;
    int ring[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) ring[t] = rb + t < W4_R ? rb + t : rb + t - W4_R;
    int vb[2][4];                                       // per-lane slot of (parity, tap row)
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) { vb[0][ky] = lbase + ring[ky] * RP; vb[1][ky] = vb[0][ky] + 8 * PS; }
    uint4 cv = make_uint4(0u, 0u, 0u, 0u);
    float ct[4] = {0.f, 0.f, 0.f, 0.f}, csl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int ky = s >> 3, c = (s >> 2) & 1, kx = s & 3;
      {   // fragments of the next step (s == NS - 1: step 0 of row y + 1, whose tap row 0 is this row's tap row 2)
        const int sn = (s + 1) % NS;
        const int kyn = sn >> 3, cn = (sn >> 2) & 1, kxn = sn & 3;
        const int base = vb[(kxn & 1) ^ 1][s == NS - 1 ? 2 : kyn] + 4 * cn * PS + ((kxn + 1) >> 1);
#pragma unroll
        for (int g = 0; g < G; ++g) xf[(s + 1) & 1][g] = __builtin_bit_cast(bf8, img[base + 16 * g]);
      }
      if constexpr (PREV) {
        if (s >= 1 && s < 1 + 3 * G) {
          const int g = (s - 1) / 3, part = (s - 1) % 3;
          if (part == 0) epi_pack(std::integral_constant<int, P ^ 1>{}, g, y - 1);
          else if constexpr (STATS) epi_stats(g, part - 1);
        }
      }
      if (s >= 13 && s < 13 + 4 * G) {                  // 2 G units, two steps each
        const int u = (s - 13) / 2, hh = (s - 13) % 2;
        const int rsel = u / G, j = u % G;
        if (hh == 0) {
          cv = rsel ? rawb[j] : rawa[j];
          if constexpr (ACT) { ws_act_a(lpw, W4_CI, s_oc, 0, cv.x, cv.y, ct, csl); ws_act_b(ct, csl, cv.x, cv.y); }
        } else {
          if constexpr (ACT) { ws_act_a(lpw, W4_CI, s_oc, 1, cv.z, cv.w, ct, csl); ws_act_b(ct, csl, cv.z, cv.w); }
          const unsigned keep = rsel ? keep_b : keep_a;
          img[s_slot0 + 16 * j + ring[4 + rsel] * RP] = make_uint4(cv.x & keep, cv.y & keep, cv.z & keep, cv.w & keep);
        }
      }
      const int f = ((ky * 4 + kx) * 2 + c) * 2;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const v4f z = {0.f, 0.f, 0.f, 0.f};
        acc[P][g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf[s & 1][g], s == 0 ? z : acc[P][g][0], 0, 0, 0);
        acc[P][g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f + 1], xf[s & 1][g], s == 0 ? z : acc[P][g][1], 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        WS_SGB(SG_DSR, 1);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  int rb = (2 * y0) % W4_R;
  {   // step 0 of the first row: tap (0, 0), chunk 0: odd plane set, column offset 0
    const int b0 = lbase + 8 * PS + rb * RP;
#pragma unroll
    for (int g = 0; g < G; ++g) xf[0][g] = __builtin_bit_cast(bf8, img[b0 + 16 * g]);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto next_rb = [](int v) { return v + 2 < W4_R ? v + 2 : v + 2 - W4_R; };
  row(I0{}, std::false_type{}, y0, rb);
  rb = next_rb(rb);
  int y = y0 + 1;
  for (; y + 1 < y1; y += 2) {
    row(I1{}, std::true_type{}, y, rb);
    rb = next_rb(rb);
    row(I0{}, std::true_type{}, y + 1, rb);
    rb = next_rb(rb);
  }
  if (y < y1) {
    row(I1{}, std::true_type{}, y, rb);
    epi_all(I1{}, y);
  } else {
    epi_all(I0{}, y - 1);
  }

  if constexpr (STATS) {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = 32 * wave + 8 * kq + j;
      red[ch * 16 + lm] = s1[j];
      red[(WS_C + ch) * 16 + lm] = s2[j];
    }
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += (double)red[tid * 16 + i];
    a.stat[(int64_t)blockIdx.x * 2 * WS_C + tid] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// ... and for the transposed form  k4 s2 p1, 128 -> 64 channels  (forward of p_y_z_in.16, data gradient of p_y_z_in.9:
// architecture_built.txt:51,94): four output phases (py, px), each a 2 x 2-tap correlation over the coarse grid.  Wave =
// phase: it keeps the phase's 4 taps x 128 gathered x 64 produced channels (64 A-fragments) and produces ALL 64 channels of
// its phase's pixels, so a B-fragment (16 coarse pixels x 32 channels) feeds four MFMAs: 64 B/clk of LDS reads per CU.
// The coarse rows y - 1, y, y + 1 serve both row phases: the same ring of four rows as the k3 kernel.  The pipeline unit
// is HALF a coarse row (32 pixels x 4 channel blocks = 32 accumulator registers, double-buffered).
constexpr int WT_CI = 128, WT_CO = 64, WT_NF = 64, WT_FV = 16;

// fragment f = ((ty*2 + tx)*4 + chunk)*4 + block of wave (phase) p; lane (lm, kq): produced channel
// 32 (block >> 1) + 8 (lm >> 2) + 4 (block & 1) + (lm & 3), gathered channel 32 chunk + 8 kq + j
__global__ __launch_bounds__(256) void wst_pack_kernel(WsPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * WT_NF * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, f = (i >> 9) % WT_NF, p = i / (512 * WT_NF);
  const int lm = lane & 15, kq = lane >> 4;
  const int blk = f & 3, chunk = (f >> 2) & 3, tap = f >> 4;
  const int ty = tap >> 1, tx = tap & 1, py = p >> 1, px = p & 1;
  const int co = 32 * (blk >> 1) + 8 * (lm >> 2) + 4 * (blk & 1) + (lm & 3);
  const int ci = 32 * chunk + 8 * kq + j;
  const int ky = bp_t_ky(py, 1, 2, 2, ty), kx = bp_t_ky(px, 1, 2, 2, tx);
  a.dst[i] = f2bf(a.w[ci * a.sa + co * a.sb + ky * 4 + kx]);
}

template <int G, bool ACT, bool STATS>
__global__ __launch_bounds__(256) void wst_bf16_kernel(WsArgs a) {
  using GM = WsGeom<G>;                                 // the k3 kernel's image: 16 octet planes x 4 ring rows x (W + 2) slots
  constexpr int W = GM::W, RP = GM::RP, PS = GM::PS;
  constexpr int HG = G >= 2 ? G / 2 : 1;                // pixel groups per pipeline unit
  constexpr int NUNIT = G / HG;                         // units per coarse row
  constexpr int NS = 16;                                // K-steps per unit: 4 taps x 4 chunks
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  uint4* img = reinterpret_cast<uint4*>(smem);
  float* lpw = reinterpret_cast<float*>(smem + GM::img_bytes / 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int lm = lane & 15, kq = lane >> 4;
  const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
  const int y0 = band * a.BR;                           // coarse rows [y0, y1); a.h coarse rows, 2 a.h produced rows
  const int y1 = min(y0 + a.BR, a.h);

  int s_slot[G], s_oc[G];
  unsigned s_off[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int ub = wave + 4 * i;
    s_oc[i] = (lane >> 3) + 8 * (ub & 1);
    const int pxl = (lane & 7) + 8 * (ub >> 1);
    s_slot[i] = s_oc[i] * PS + 1 + pxl;
    s_off[i] = (unsigned)(pxl * a.in_cs + s_oc[i] * 8) * 2u;
  }
  const char* in_img = reinterpret_cast<const char*>(a.in + (int64_t)n * a.h * W * a.in_cs + a.in_co);
  const unsigned in_row = (unsigned)(W * a.in_cs) * 2u;
  auto load_row = [&](int r, uint4 (&raw)[G]) {
    const char* rowp = in_img + (size_t)((unsigned)r * in_row);
#pragma unroll
    for (int i = 0; i < G; ++i) raw[i] = *reinterpret_cast<const uint4*>(rowp + s_off[i]);
  };
  auto commit_row = [&](int r, bool inside, const uint4 (&raw)[G]) {
    const int rr = (r + 1) & (WS_R - 1);
#pragma unroll
    for (int i = 0; i < G; ++i) {
      uint4 v = raw[i];
      if constexpr (ACT) {
        float t[4], sl[4];
        ws_act_a(lpw, WT_CI, s_oc[i], 0, v.x, v.y, t, sl); ws_act_b(t, sl, v.x, v.y);
        ws_act_a(lpw, WT_CI, s_oc[i], 1, v.z, v.w, t, sl); ws_act_b(t, sl, v.z, v.w);
      }
      if (!inside) v = make_uint4(0u, 0u, 0u, 0u);
      img[s_slot[i] + rr * RP] = v;
    }
  };

  // ---- prologue
  uint4 raw0[G], raw1[G];
  const bool in0 = y0 - 1 >= 0, in2 = y0 + 1 < a.h;
  load_row(in0 ? y0 - 1 : y0, raw0);
  load_row(y0, raw1);
  bf8 wf[WT_NF];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wp) + (size_t)wave * WT_NF * 64 + lane;
#pragma unroll
    for (int f = 0; f < WT_NF; ++f) wf[f] = __builtin_bit_cast(bf8, wsrc[f * 64]);
#pragma unroll
    for (int f = 0; f < WT_NF; ++f) {
      if (f < WT_FV) asm volatile("" : "+v"(wf[f]));
      else asm volatile("" : "+a"(wf[f]));
    }
  }
  if (tid < 128) {
    const int plane = tid >> 3, rr = (tid >> 1) & 3, side = tid & 1;
    img[plane * PS + rr * RP + (side ? RP - 1 : 0)] = make_uint4(0u, 0u, 0u, 0u);
  }
  if constexpr (ACT) {
    for (int i = tid; i < WT_CI; i += 256) { lpw[i] = a.pw.scale[i]; lpw[WT_CI + i] = a.pw.shift[i]; lpw[2 * WT_CI + i] = a.pw.slope[i]; }
    __syncthreads();
  }
  commit_row(y0 - 1, in0, raw0);
  commit_row(y0, true, raw1);
  load_row(in2 ? y0 + 1 : y0, raw0);
  commit_row(y0 + 1, in2, raw0);
  __syncthreads();

  float s1[16], s2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // fragment base of this lane and phase: plane kq, pixel lm + px (+ tx), tap row ty lives in ring slot (y + py + ty) & 3
  const int lbase = kq * PS + lm + px;
  char* out_img = reinterpret_cast<char*>(a.out + (int64_t)n * (2 * a.h) * (2 * W) * a.out_cs + a.out_co);
  const unsigned out_row = (unsigned)(2 * W * a.out_cs) * 2u;
  // produced pixel 2 (16 g + lm) + px of row 2 y + py; a lane's channels: 32 hb + 8 kq + [0, 8), hb = 0, 1
  const unsigned o_off = (unsigned)((2 * lm + px) * a.out_cs + 8 * kq) * 2u, o_g = (unsigned)(32 * a.out_cs) * 2u;

  v4f acc[2][HG][4];
  bf8 xf[2][HG];
  uint4 pk[HG][2];
  auto epi_pack = [&](auto P_, int g, int hb, int gg, int yp) {         // unit-local group g, channel half hb, row-global group gg
    constexpr int P = decltype(P_)::value;
    pk[g][hb] = make_uint4(pack2(acc[P][g][2 * hb][0], acc[P][g][2 * hb][1]), pack2(acc[P][g][2 * hb][2], acc[P][g][2 * hb][3]),
                           pack2(acc[P][g][2 * hb + 1][0], acc[P][g][2 * hb + 1][1]), pack2(acc[P][g][2 * hb + 1][2], acc[P][g][2 * hb + 1][3]));
    char* rowp = out_img + (size_t)((unsigned)(2 * yp + py) * out_row);
    *reinterpret_cast<uint4*>(rowp + (o_off + (unsigned)gg * o_g + 64u * (unsigned)hb)) = pk[g][hb];
  };
  auto epi_stats = [&](int g, int hb, int hh) {         // channels 32 hb + 8 kq + 4 hh + [0, 4) of the values AS STORED
    const unsigned w0 = hh ? pk[g][hb].z : pk[g][hb].x, w1 = hh ? pk[g][hb].w : pk[g][hb].y;
    const float v0 = bf2f((u16)(w0 & 0xffffu)), v1 = bf2f((u16)(w0 >> 16)), v2 = bf2f((u16)(w1 & 0xffffu)), v3 = bf2f((u16)(w1 >> 16));
    const int o = 8 * hb + 4 * hh;
    s1[o] += v0; s2[o] = fmaf(v0, v0, s2[o]);
    s1[o + 1] += v1; s2[o + 1] = fmaf(v1, v1, s2[o + 1]);
    s1[o + 2] += v2; s2[o + 2] = fmaf(v2, v2, s2[o + 2]);
    s1[o + 3] += v3; s2[o + 3] = fmaf(v3, v3, s2[o + 3]);
  };
  auto epi_all = [&](auto P_, int u, int yp) {
#pragma unroll
    for (int g = 0; g < HG; ++g)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        epi_pack(P_, g, hb, u * HG + g, yp);
        if constexpr (STATS) { epi_stats(g, hb, 0); epi_stats(g, hb, 1); }
      }
  };

  // unit (y, u): pixel groups u HG .. u HG + HG - 1 of coarse row y.  The first unit of a row requests coarse row y + 2; its
  // G staging units are committed over the row's units; the last unit of a row ends with the barrier.
  auto unit = [&](auto P_, auto PREV_, auto U_, int y, int yprev, int uprev) {
    constexpr int P = decltype(P_)::value;
    constexpr bool PREV = decltype(PREV_)::value;
    constexpr int U = decltype(U_)::value;
    const bool in_next = y + 2 < a.h;
    if constexpr (U == 0) load_row(in_next ? y + 2 : a.h - 1, raw0);
    const int rr_next = (y + 3) & (WS_R - 1);
    const unsigned keep = in_next ? 0xffffffffu : 0u;
    int rb[2];                                          // tap rows ty = 0, 1 of this phase
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) rb[ty] = lbase + ((y + py + ty) & (WS_R - 1)) * RP;
    // next unit's first fragments: same row (U + 1 < NUNIT) or row y + 1, whose tap row 0 is this row's tap row 1
    uint4 cv = make_uint4(0u, 0u, 0u, 0u);
    float ct[4] = {0.f, 0.f, 0.f, 0.f}, csl[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int CPU_ = (G + NUNIT - 1) / NUNIT;       // staging units committed per pipeline unit
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int t = s >> 2, c = s & 3;                  // tap (ty, tx) = (t >> 1, t & 1), channel chunk c
      {
        const int sn = (s + 1) % NS;
        const int tn = sn >> 2, cn = sn & 3;
        int base, gofs;
        if (s == NS - 1) {                              // first step of the next unit: tap (0, 0), chunk 0
          base = U + 1 < NUNIT ? rb[0] : rb[1];
          gofs = U + 1 < NUNIT ? 16 * HG * (U + 1) : 0;
        } else {
          base = rb[tn >> 1] + 4 * cn * PS + (tn & 1);
          gofs = 16 * HG * U;
        }
#pragma unroll
        for (int g = 0; g < HG; ++g) xf[(s + 1) & 1][g] = __builtin_bit_cast(bf8, img[base + gofs + 16 * g]);
      }
      if constexpr (PREV) {
        // previous unit: per group and channel half: round + store | sums of 4 channels | sums of 4 channels
        if (s < 6 * HG) {
          const int g = s / 6, r6 = s % 6, hb = r6 / 3, part = r6 % 3;
          if (part == 0) epi_pack(std::integral_constant<int, P ^ 1>{}, g, hb, uprev * HG + g, yprev);
          else if constexpr (STATS) epi_stats(g, hb, part - 1);
        }
      }
      if (s >= 8 && s < 8 + 2 * CPU_ && U * CPU_ + (s - 8) / 2 < G) {
        const int i = U * CPU_ + (s - 8) / 2, hh = (s - 8) % 2;
        if (hh == 0) {
          cv = raw0[i];
          if constexpr (ACT) { ws_act_a(lpw, WT_CI, s_oc[i], 0, cv.x, cv.y, ct, csl); ws_act_b(ct, csl, cv.x, cv.y); }
        } else {
          if constexpr (ACT) { ws_act_a(lpw, WT_CI, s_oc[i], 1, cv.z, cv.w, ct, csl); ws_act_b(ct, csl, cv.z, cv.w); }
          img[s_slot[i] + rr_next * RP] = make_uint4(cv.x & keep, cv.y & keep, cv.z & keep, cv.w & keep);
        }
      }
      const int f = (t * 4 + c) * 4;
#pragma unroll
      for (int g = 0; g < HG; ++g) {
        const v4f z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[P][g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f + nb], xf[s & 1][g], s == 0 ? z : acc[P][g][nb], 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < HG; ++g) {
        WS_SGB(SG_DSR, 1);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
        WS_SGB(SG_MFMA, 1); WS_SGB(SG_VALU, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (U == NUNIT - 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };

  {   // first fragments of the first unit: tap (0, 0), chunk 0
    const int b0 = lbase + ((y0 + py) & (WS_R - 1)) * RP;
#pragma unroll
    for (int g = 0; g < HG; ++g) xf[0][g] = __builtin_bit_cast(bf8, img[b0 + 16 * g]);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // NS is even, so the fragment buffer parity is the same for every unit; the accumulator parity alternates per unit
  if constexpr (NUNIT == 2) {
    unit(I0{}, std::false_type{}, I0{}, y0, y0, 0);
    unit(I1{}, std::true_type{}, I1{}, y0, y0, 0);
    for (int y = y0 + 1; y < y1; ++y) {
      unit(I0{}, std::true_type{}, I0{}, y, y - 1, 1);
      unit(I1{}, std::true_type{}, I1{}, y, y, 0);
    }
    epi_all(I1{}, 1, y1 - 1);
  } else {
    unit(I0{}, std::false_type{}, I0{}, y0, y0, 0);
    int y = y0 + 1;
    for (; y + 1 < y1; y += 2) {
      unit(I1{}, std::true_type{}, I0{}, y, y - 1, 0);
      unit(I0{}, std::true_type{}, I0{}, y + 1, y, 0);
    }
    if (y < y1) {
      unit(I1{}, std::true_type{}, I0{}, y, y - 1, 0);
      epi_all(I1{}, 0, y);
    } else {
      epi_all(I0{}, 0, y - 1);
    }
  }

  if constexpr (STATS) {
    // per-lane fp32 partials -> LDS [stat][channel 64][phase 4][lm 16] -> one double row per workgroup (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ch = 32 * (j >> 3) + 8 * kq + (j & 7);
      red[(ch * 4 + wave) * 16 + lm] = s1[j];
      red[((WT_CO + ch) * 4 + wave) * 16 + lm] = s2[j];
    }
    __syncthreads();
    if (tid < 2 * WT_CO) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < 64; ++i) t += (double)red[tid * 64 + i];
      a.stat[(int64_t)blockIdx.x * 2 * WT_CO + tid] = t;
    }
  }
}

bool ws_enabled() {
  static const bool off = getenv("BP_BF16_WS") && atoi(getenv("BP_BF16_WS")) == 0;
  return !off;
}
int g_ws_override = -1;       // bp_set_option("bf16_ws", v)

int ws_G(int w) { return w == 64 ? 4 : w == 32 ? 2 : w == 16 ? 1 : 0; }

// rows per band: whole image per workgroup when there are many images, else bands of >= 4 rows so that ~256 workgroups exist
void ws_bands(int n, int h, int* BR, int* bands) {
  int br = h;
  while (br > 4 && (int64_t)n * bp_ceil_div(h, br) < 256) br = bp_ceil_div(br, 2);
  *BR = br;
  *bands = bp_ceil_div(h, br);
}

template <int KIND, int G, bool ACT, bool STATS>
int ws_launch(const WsArgs& a, unsigned grid, hipStream_t st) {
  // (the transposed kernel folds its statistics through 32 KB of LDS: more than the image at W = 16)
  constexpr size_t lds = KIND == 3 ? WsGeom<G>::lds_bytes : KIND == 4 ? W4Geom<G>::lds_bytes
                         : (WsGeom<G>::lds_bytes > 32768 ? WsGeom<G>::lds_bytes : 32768);
  auto kern = [] {
    if constexpr (KIND == 3) return &ws3_bf16_kernel<G, ACT, STATS>;
    else if constexpr (KIND == 4) return &ws4_bf16_kernel<G, ACT, STATS>;
    else return &wst_bf16_kernel<G, ACT, STATS>;
  }();
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}
template <int KIND, int G>
int ws_launch_g(const WsArgs& a, bool act, bool stats, unsigned grid, hipStream_t st) {
  if (act) return stats ? ws_launch<KIND, G, true, true>(a, grid, st) : ws_launch<KIND, G, true, false>(a, grid, st);
  return stats ? ws_launch<KIND, G, false, true>(a, grid, st) : ws_launch<KIND, G, false, false>(a, grid, st);
}
template <int KIND>
int ws_launch_k(int G, const WsArgs& a, bool act, bool stats, unsigned grid, hipStream_t st) {
  switch (G) {
    case 4: return ws_launch_g<KIND, 4>(a, act, stats, grid, st);
    case 2: return ws_launch_g<KIND, 2>(a, act, stats, grid, st);
    default: return ws_launch_g<KIND, 1>(a, act, stats, grid, st);
  }
}

}  // namespace

void bp_bf16_ws_set(int v) { g_ws_override = v; }

// 3: the k3 s1 p1 128 -> 128 trunk layer (either direction); 4: the strided gather k4 s2 p1 64 -> 128; 5: the transposed
// form k4 s2 p1 128 -> 64; 0: none
int bp_bf16_ws_kind(const ConvGeom& g) {
  if (g.gather_transposed && g.k == 4 && g.stride == 2 && g.pad == 1 && g.nphase == 2 && g.taps == 2 && g.IS == 1 && g.OS == 2 &&
      g.cin_g == WT_CI && g.cout_g == WT_CO)
    return 5;
  if (g.nphase != 1 || g.OS != 1) return 0;
  if (g.k == 3 && g.stride == 1 && g.pad == 1 && g.cin_g == WS_C && g.cout_g == WS_C && g.taps == 3 && g.IS == 1) return 3;
  if (g.k == 4 && g.stride == 2 && g.pad == 1 && g.cin_g == W4_CI && g.cout_g == WS_C && g.taps == 4 && g.IS == 2 &&
      !g.gather_transposed)
    return 4;
  return 0;
}

int64_t bp_bf16_ws_packed_elems(const ConvGeom& g) {
  const int k = bp_bf16_ws_kind(g);
  return k == 3 ? (int64_t)4 * WS_NF * 64 * 8 : k == 4 ? (int64_t)4 * W4_NF * 64 * 8 : k == 5 ? (int64_t)4 * WT_NF * 64 * 8 : 0;
}

int bp_bf16_ws_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st) {
  const int k = bp_bf16_ws_kind(g);
  if (!k) return BP_EUNSUPPORTED;
  const WsPackArgs a{w_torch, dst, wm.sa, wm.sb, g.gather_transposed};
  if (k == 3) hipLaunchKernelGGL(ws_pack_kernel, dim3(4 * WS_NF * 64 * 8 / 256), dim3(256), 0, st, a);
  else if (k == 4) hipLaunchKernelGGL(ws4_pack_kernel, dim3(4 * W4_NF * 64 * 8 / 256), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wst_pack_kernel, dim3(4 * WT_NF * 64 * 8 / 256), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

// mode: 0 none, 1 batch-norm sums of the produced tensor
bool bp_bf16_ws_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int mode) {
  if (!(g_ws_override < 0 ? ws_enabled() : g_ws_override != 0)) return false;
  const int k = bp_bf16_ws_kind(g);
  if (!k || !in || !out || bias || (mode != 0 && mode != 1)) return false;
  if (in->dtype != BP_BF16 || out->dtype != BP_BF16 || in->c != g.cin_g || out->c != g.cout_g) return false;
  const bp_view* grid_v = k == 5 ? in : out;            // the view whose rows the kernel walks (the coarse one)
  if (in->n != out->n || !ws_G(grid_v->w)) return false;
  if (k == 3 && (in->w != out->w || in->h != out->h)) return false;
  if (k == 4 && (in->w != 2 * out->w || in->h != 2 * out->h)) return false;
  if (k == 5 && (out->w != 2 * in->w || out->h != 2 * in->h)) return false;
  if (in->cstride % 8 || in->coff % 8 || reinterpret_cast<uintptr_t>(in->ptr) % 16) return false;
  if (out->cstride % 8 || out->coff % 8 || reinterpret_cast<uintptr_t>(out->ptr) % 16) return false;
  // (row addresses are a 64-bit image base + 32-bit byte offsets inside the image)
  if ((int64_t)in->h * in->w * in->cstride * 2 >= (int64_t)1 << 31 || (int64_t)out->h * out->w * out->cstride * 2 >= (int64_t)1 << 31)
    return false;
  int BR, bands;
  ws_bands(grid_v->n, grid_v->h, &BR, &bands);
  if (mode == 1 && BR * ws_G(grid_v->w) > 256) return false;          // a lane's fp32 partial sums: <= 256 terms
  return (int64_t)out->n * bands <= 0x7fffffff;
}

size_t bp_bf16_ws_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out) {
  const bp_view* grid_v = bp_bf16_ws_kind(g) == 5 ? in : out;
  int BR, bands;
  ws_bands(grid_v->n, grid_v->h, &BR, &bands);
  return bp_stats_rows_bytes((int64_t)out->n * bands, g.cout_g);
}

int bp_bf16_ws_run(const ConvGeom& g, const bp_view* in, const PW& pw, const u16* packed_ws, const bp_view* out,
                   hipStream_t st, const IgemmStatsReq* sr) {
  WsArgs a{};
  a.in = reinterpret_cast<const u16*>(in->ptr); a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = reinterpret_cast<u16*>(out->ptr); a.out_cs = out->cstride; a.out_co = out->coff;
  const int kind = bp_bf16_ws_kind(g);
  const bp_view* grid_v = kind == 5 ? in : out;
  a.n = out->n; a.h = grid_v->h; a.wp = packed_ws; a.pw = pw;
  ws_bands(grid_v->n, grid_v->h, &a.BR, &a.bands);
  const int64_t rows = (int64_t)out->n * a.bands;
  if (sr) {
    const size_t need = bp_stats_rows_bytes(rows, g.cout_g);
    if (sr->mode != 1 || !need) return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < need || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws);
  }
  const bool act = pw.scale != nullptr;
  const int G = ws_G(grid_v->w);
  const int rc = kind == 3 ? ws_launch_k<3>(G, a, act, sr != nullptr, (unsigned)rows, st)
                 : kind == 4 ? ws_launch_k<4>(G, a, act, sr != nullptr, (unsigned)rows, st)
                             : ws_launch_k<5>(G, a, act, sr != nullptr, (unsigned)rows, st);
  if (rc != BP_OK || !sr) return rc;
  return bp_stats_rows_finish(a.stat, rows, g.cout_g, sr, st);
}
