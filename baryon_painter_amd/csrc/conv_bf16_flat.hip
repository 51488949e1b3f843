// Flattened-K bf16 matrix-core kernel for the UNIT-STRIDE few-channel layers at full resolution in throughput mode
// (BASELINE.json configs[3] / [4]):
//     p_mu_out.0 / p_var_out.0   k7  16 -> 8    forward (bf16 trunk in, fp32 out) and data gradient (8 -> 16, fp32 in,
//                                               bf16 out)
//     p_y_z_in.0  (the stem)     k5  3(+1) -> 16  forward (fp32 in, bf16 out, batch-norm sums from the epilogue) and the
//                                               data gradient restricted to the latent channel (16 -> 1, bf16 in, fp32 out)
// These launches were the largest single-layer items of the bf16 step in igemm_bf16_kernel (a 16-pixel tile per
// tap-row slab, weights re-streamed through LDS per tile, two barriers per tap row: 0.2-0.3 PFLOP/s, 1-2 TB/s); at bf16
// matrix speed each of them is 0.03-0.11 ms of MFMA and 0.15-0.25 ms of HBM.
//
// Scheme (the bf16 counterpart of conv_flat.hip):
//   * GEMM K = the FLATTENED (tap column, channel) index of one tap row -- KS x CIN consecutive bf16 of an NHWC row --
//     cut into blocks of 32 (v_mfma_f32_16x16x32_bf16); the packed weights carry zeros past KS x CIN, so a lane's
//     operand is always 8 consecutive bf16 of the staged row (one 16-byte LDS read; two 8-byte reads for CIN = 4).
//   * MFMA rows = produced channels.  With COUTP < 16 produced channels the 16 rows hold RS = 16 / COUTP OUTPUT ROWS:
//     row block rs uses tap row ky - rs, so one input-row fragment feeds all of them (weight fragment "p" =
//     [W[p] | W[p-1] | ...]).
//   * A wave owns 16 pixels x 4 output rows per pass: an input-row fragment is read once and multiplied with up to
//     4 / RS weight fragments; ALL weight fragments live in registers, loaded once per workgroup -- no weight traffic
//     through LDS, no barrier inside a tile.
//   * D = W x X: a lane ends up with 4 consecutive channels of one pixel -> 16-byte fp32 / 8-byte bf16 stores, 512
//     contiguous bytes per output row and instruction.
//   * Staging: every unit of the halo tile is loaded unconditionally from clamped coordinates, all loads of the tile
//     in flight at once, kept as raw words; the producer's pending batch-norm + (leaky) ReLU is applied and the value
//     rounded to bf16 on the way into LDS, zero padding after the activation (as torch pads the activated tensor).
//   * STATS: training-mode batch-norm sums {sum y, sum y^2} of the tensor AS STORED (bf16-rounded) from the
//     accumulators: a lane's 16 values in fp32, the 16 lanes of a channel quad by shuffles, the four waves through
//     LDS; one fixed-order row of doubles per tile (conv_igemm.hip folds the rows).
#include "conv_bf16.hpp"

using namespace bpbf16;

// conv_igemm.hip: partial rows of epilogue statistics -> sums (+ the fused batch-norm finalize)
size_t bp_stats_rows_bytes(int64_t rows, int C);
int bp_stats_rows_finish(double* ws, int64_t rows, int C, const IgemmStatsReq* sr, hipStream_t st);

namespace {

struct FbArgs {
  const void* in; int h, w, in_cs, in_co, cin;
  void* out; int out_cs, out_co, cout;
  const u16* wp;
  PW pw;
  int tiles_x, tiles_y, n;
  double* stat; int stat_c;
  // STATS == 2 (data gradient into a bf16 slot with batch-norm): the produced tensor is d(loss)/d(activated slot);
  // raw / spw = the slot's raw values and pending activation -> rows of {sum g, sum g*raw}, g = d * act'(spw(raw))
  const u16* raw; int raw_cs, raw_co;
  PW spw;
};

constexpr int FB_TW = 64, FB_TH = 16;          // (FB_R, output rows per pass: template parameter of the kernel, 4 by default)

template <int KS, int CIN> struct FbShape {
  static constexpr int KB = (KS * CIN + 31) / 32;
  static constexpr int LW = FB_TW + KS - 1, LH = FB_TH + KS - 1;
  static constexpr int ROWE = LW * CIN;                       // bf16 per staged row
  static_assert(ROWE % 8 == 0, "a staged row is a whole number of 8-element units");
  static constexpr int SLACK = 32;                            // the last K block reads up to one unit past a row
  static constexpr size_t LDS = ((size_t)LH * ROWE + SLACK) * 2 + 3 * 16 * sizeof(float) + 4 * 32 * sizeof(double);
};

// XCD-aware tile order: workgroups go to the 8 XCDs round-robin by linear id; give each XCD a contiguous run of
// tiles so that the halos shared by neighbouring tiles are fetched into ONE L2.
__device__ __forceinline__ int fb_tile_of_block(int n) {
  const int L = blockIdx.x;
  const int q = n >> 3, r = n & 7;
  const int xcd = L & 7, idx = L >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int KS, int CIN, int COUTP, bool IN_BF16, bool OUT_BF16, int STATS, int FB_R = 4>
__global__ __launch_bounds__(256, FB_R > 4 ? 2 : 3) void flatb_kernel(FbArgs a) {
  using S = FbShape<KS, CIN>;
  constexpr int KB = S::KB, ROWE = S::ROWE, LH = S::LH, PAD = KS / 2;
  constexpr int RS = 16 / COUTP;                // output rows per MFMA
  constexpr int NP = KS + RS - 1;               // weight fragments per K block
  constexpr int NS = FB_R / RS;                 // accumulator sets per pass
  constexpr int UPR = ROWE / 8;                 // 8-element staging units per row
  constexpr int NPX = CIN >= 8 ? 1 : 8 / CIN;   // pixels per unit
  constexpr int UPP = CIN >= 8 ? CIN / 8 : 1;   // units per pixel
  constexpr int NU = LH * UPR;
  constexpr int SLOTS = (NU + 255) / 256;
  static_assert(256 % UPP == 0 && (IN_BF16 || CIN <= 8), "staging layout");
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + LH * ROWE + S::SLACK);
  double* red = reinterpret_cast<double*>(lpw + 3 * 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;

  const int per_img = a.tiles_x * a.tiles_y;
  const int t = fb_tile_of_block(per_img * a.n);
  const int n = t / per_img, tr = t - n * per_img;
  const int ty0 = (tr / a.tiles_x) * FB_TH, tx0 = (tr % a.tiles_x) * FB_TW;

  // ---- stage the halo tile: all loads first (raw words), then activation + bf16 + LDS
  const int64_t img = (int64_t)n * a.h * a.w * a.in_cs + a.in_co;
  uint4 stage[SLOTS][IN_BF16 ? 1 : 2];
  int inside[SLOTS];                            // -1: no such unit; else bit p: pixel p of the unit lies in the image
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256;
    const int row = e / UPR, u = e - row * UPR;
    const int gy = ty0 - PAD + row;
    const int cy = min(max(gy, 0), a.h - 1);
    int flags = 0;
#pragma unroll
    for (int p = 0; p < NPX; ++p) {
      const int px = CIN >= 8 ? u / UPP : u * NPX + p;
      const int cu = CIN >= 8 ? u % UPP : 0;
      const int gx = tx0 - PAD + px;
      if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) flags |= 1 << p;
      const int cx = min(max(gx, 0), a.w - 1);
      const int64_t off = img + ((int64_t)cy * a.w + cx) * a.in_cs + cu * 8;
      if constexpr (IN_BF16) {
        stage[i][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(a.in) + off);
      } else if constexpr (CIN >= 8) {
        stage[i][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(a.in) + off);
        stage[i][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(a.in) + off + 4);
      } else {
        stage[i][p] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(a.in) + off);
      }
    }
    inside[i] = e < NU ? flags : -1;
  }
  const bool on = a.pw.scale != nullptr;
  if (tid < 16) {
    const bool ok = on && tid < a.cin;
    lpw[tid] = ok ? a.pw.scale[tid] : 1.f; lpw[16 + tid] = ok ? a.pw.shift[tid] : 0.f; lpw[32 + tid] = ok ? a.pw.slope[tid] : 1.f;
  }
  if (tid < S::SLACK / 8) *reinterpret_cast<uint4*>(lds + LH * ROWE + tid * 8) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  {
    // (256 % UPP == 0: a thread always handles the same channels -- its activation parameters once, in registers)
    const int c0 = CIN >= 8 ? (tid % UPP) * 8 : 0;
    float sc[8], sf[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = CIN >= 8 ? c0 + j : j % CIN;
      sc[j] = lpw[ch]; sf[j] = lpw[16 + ch]; sl[j] = lpw[32 + ch];
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      if (inside[i] < 0) continue;
      float raw[8], v[8];
      if constexpr (IN_BF16) {
        const unsigned w[4] = {stage[i][0].x, stage[i][0].y, stage[i][0].z, stage[i][0].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { raw[2 * j] = bf2f((u16)(w[j] & 0xffffu)); raw[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          raw[4 * q] = __builtin_bit_cast(float, stage[i][q].x); raw[4 * q + 1] = __builtin_bit_cast(float, stage[i][q].y);
          raw[4 * q + 2] = __builtin_bit_cast(float, stage[i][q].z); raw[4 * q + 3] = __builtin_bit_cast(float, stage[i][q].w);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = raw[j];
        if (on) { x = fmaf(x, sc[j], sf[j]); x = x > 0.f ? x : x * sl[j]; }
        const int p = NPX == 1 ? 0 : j / CIN;
        const int ch = CIN >= 8 ? c0 + j : j % CIN;
        v[j] = (((inside[i] >> p) & 1) && ch < a.cin) ? x : 0.f;
      }
      lds_store_unit<8>(lds + (tid + i * 256) * 8, v);
    }
  }
  // weights: registers, one 16-byte load per fragment and lane ([fragment p][K block][k octet][row][8] packed image).
  // Loaded AFTER the staging registers are dead: the kernel then fits three waves per SIMD (three workgroups per CU),
  // and the other workgroups' matrix work covers this L2 round trip.
  bf8 wf[NP][KB];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
      wf[p][kb] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(a.wp + ((p * KB + kb) * 64 + lane) * 8));
  __syncthreads();

  // ---- multiply: this wave's strip of 16 pixels, FB_R output rows per pass
  const int x0 = wave * 16;
  const int fbase = (x0 + lj) * CIN + kg * 8;
  const int ox = tx0 + x0 + lj;
  const int rs_l = (kg * 4) / COUTP;            // which output row of an MFMA this lane's 4 accumulator rows hold
  const int co_l = (kg * 4) % COUTP;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};       // STATS: this lane's 4 channels
  float psc[4] = {1.f, 1.f, 1.f, 1.f}, psf[4] = {0.f, 0.f, 0.f, 0.f}, psl[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (STATS == 2) {
    if (a.spw.scale) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { psc[r] = a.spw.scale[co_l + r]; psf[r] = a.spw.shift[co_l + r]; psl[r] = a.spw.slope[co_l + r]; }
    }
  }
#pragma unroll 1
  for (int pass = 0; pass < FB_TH / FB_R; ++pass) {
    v4f acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = v4f{0.f, 0.f, 0.f, 0.f};
    // STATS == 2: the slot's raw values under this pass's outputs, requested ahead of the MFMA chain
    uint2 rw[STATS == 2 ? NS : 1];
    if constexpr (STATS == 2) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int oy = min(ty0 + pass * FB_R + s * RS + rs_l, a.h - 1), oxc = min(ox, a.w - 1);
        rw[s] = *reinterpret_cast<const uint2*>(a.raw + ((int64_t)(n * a.h + oy) * a.w + oxc) * a.raw_cs + a.raw_co + co_l);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const u16* base = lds + (pass * FB_R) * ROWE + fbase;
#pragma unroll
    for (int jr = 0; jr < FB_R + KS - 1; ++jr) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const bf8 xf = lds_frag<(CIN < 8 ? 4 : 32)>(base + jr * ROWE + kb * 32);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int p = jr - s * RS;
          if (p >= 0 && p < NP) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[p][kb], xf, acc[s], 0, 0, 0);
        }
      }
    }
    // ---- store: 4 consecutive channels of pixel ox per lane and accumulator set
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int oy = ty0 + pass * FB_R + s * RS + rs_l;
      if (oy >= a.h || ox >= a.w) continue;
      const int64_t o = ((int64_t)(n * a.h + oy) * a.w + ox) * a.out_cs + a.out_co + co_l;
      if constexpr (OUT_BF16) {
        u16* q = reinterpret_cast<u16*>(a.out) + o;
        if (co_l + 3 < a.cout) *reinterpret_cast<uint2*>(q) = make_uint2(pack2(acc[s][0], acc[s][1]), pack2(acc[s][2], acc[s][3]));
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co_l + r < a.cout) q[r] = f2bf(acc[s][r]);
      } else {
        float* q = reinterpret_cast<float*>(a.out) + o;
        if (co_l + 3 < a.cout) *reinterpret_cast<float4*>(q) = make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co_l + r < a.cout) q[r] = acc[s][r];
      }
      if constexpr (STATS == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[s][r];
          if constexpr (OUT_BF16) v = bf2f(f2bf(v));
          s1[r] += v; s2[r] = fmaf(v, v, s2[r]);
        }
      }
      if constexpr (STATS == 2) {
        const float rv[4] = {bf2f((u16)(rw[s].x & 0xffffu)), bf2f((u16)(rw[s].x >> 16)), bf2f((u16)(rw[s].y & 0xffffu)),
                             bf2f((u16)(rw[s].y >> 16))};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float d = acc[s][r];
          if constexpr (OUT_BF16) d = bf2f(f2bf(d));                 // (what the separate pass would read back)
          const float t = fmaf(rv[r], psc[r], psf[r]);
          const float g = t > 0.f ? d : d * psl[r];
          s1[r] += g; s2[r] = fmaf(g, rv[r], s2[r]);
        }
      }
    }
  }
  if constexpr (STATS) {
    // (COUTP == 16 for every layer with a batch-norm here: lane kg holds channels 4 kg .. 4 kg + 3, the 16 lanes lj of
    //  a quarter-wave share them.)  16 values per lane in fp32, then doubles: lanes by shuffles, waves through LDS.
    static_assert(!STATS || COUTP == 16, "statistics: 16 produced channels");
    double d1[4], d2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { d1[r] = (double)s1[r]; d2[r] = (double)s2[r]; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) { d1[r] += __shfl_xor(d1[r], m, 16); d2[r] += __shfl_xor(d2[r], m, 16); }
    if (lj == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[wave * 32 + kg * 4 + r] = d1[r]; red[wave * 32 + 16 + kg * 4 + r] = d2[r]; }
    }
    __syncthreads();
    if (tid < 32) {
      const double v = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
      const int64_t row = t;                   // one row per tile, whatever workgroup computed it
      const int ch = tid & 15;
      if (ch < a.stat_c) a.stat[(row * 2 + (tid >> 4)) * a.stat_c + ch] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------- k7 16 -> 8, row ring
// The heads' first layer again (forward, bf16 in, bf16 / fp32 out), as a ROW WALKER: the tile kernel above stages a halo
// tile, loads 32 weight fragments, multiplies and stores, phase after phase -- 0.36 of the matrix pipe busy, 22 / 16 of
// the rows and the weights re-fetched per tile.  Here a workgroup owns a strip of 64 columns of one image and walks it top
// to bottom in steps of eight output rows: a ring of 24 input rows in LDS ([row][pixel + 6][16] bf16), the weights once
// per workgroup, the next step's eight rows requested from HBM before the step's 128 MFMAs per wave and committed
// (activation, bf16) after them into ring slots nobody reads, one barrier per step.  An input-row fragment feeds up to four
// row pairs (0.44 LDS reads per MFMA; the 4-row passes of the tile kernel: 0.63).  Same packed weights, same MFMA row /
// column meaning as flatb_kernel<7, 16, 8>.  Measured 0.29 ms against the tile kernel's 0.34-0.35 (not the 0.17 the HBM
// bytes allow: at 256 registers -- 128 of them weights -- two waves per SIMD is all the latency hiding there is; a second
// staging register set, i.e. a whole step of lead for the row loads, spilled and was slower: 0.36 ms).
struct FrArgs {
  const u16* in; int h, w, in_cs, in_co;
  void* out; int out_cs, out_co;
  const u16* wp;
  PW pw;
  int strips, n;
};

constexpr int FR_KS = 7, FR_CIN = 16, FR_TW = 64, FR_LW = FR_TW + FR_KS - 1, FR_ROWE = FR_LW * FR_CIN;
constexpr int FR_STEP = 8, FR_WIN = FR_STEP + FR_KS - 1, FR_NR = 24;          // rows per step, window rows, ring rows
constexpr int FR_KB = 4, FR_NP = 8;
constexpr size_t FR_LDS = ((size_t)FR_NR * FR_ROWE + 32) * 2 + 3 * 16 * sizeof(float);

template <bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void flatr_k7_kernel(FrArgs a) {
  constexpr int UPR = FR_LW * 2;                        // 8-channel units per row
  constexpr int NU = FR_STEP * UPR, SLOTS = (NU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + FR_NR * FR_ROWE + 32);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;
  const int n = blockIdx.x / a.strips, strip = blockIdx.x - n * a.strips;
  const int tx0 = strip * FR_TW;

  const bool on = a.pw.scale != nullptr;
  if (tid < 16) {
    lpw[tid] = on ? a.pw.scale[tid] : 1.f; lpw[16 + tid] = on ? a.pw.shift[tid] : 0.f; lpw[32 + tid] = on ? a.pw.slope[tid] : 1.f;
  }
  if (tid < 4) *reinterpret_cast<uint4*>(lds + FR_NR * FR_ROWE + tid * 8) = make_uint4(0u, 0u, 0u, 0u);     // slack behind the ring
  __syncthreads();
  const int c0 = (tid & 1) * 8;                         // (256 is even: a thread always stages the same channel octet)
  float sc[8], sf[8], sl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = lpw[c0 + j]; sf[j] = lpw[16 + c0 + j]; sl[j] = lpw[32 + c0 + j]; }

  const u16* img = a.in + (int64_t)n * a.h * a.w * a.in_cs + a.in_co + c0;
  // rows r0 .. r0 + 7 of the image (zero outside) -> registers; then -> ring slots (row mod 24)
  uint4 stage[SLOTS];
  unsigned inside = 0;
  auto fetch = [&](int r0) {
    inside = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = tid + i * 256;
      const int row = e / UPR, u = e - row * UPR;
      const int gy = r0 + row, gx = tx0 - 3 + (u >> 1);
      if (e < NU && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) inside |= 1u << i;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
      stage[i] = *reinterpret_cast<const uint4*>(img + ((int64_t)cy * a.w + cx) * a.in_cs);
    }
  };
  auto commit = [&](int r0) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = tid + i * 256;
      if (e >= NU) continue;
      const int row = e / UPR, u = e - row * UPR;
      const unsigned w[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = v[j];
        if (on) { x = fmaf(x, sc[j], sf[j]); x = x > 0.f ? x : x * sl[j]; }
        v[j] = ((inside >> i) & 1u) ? x : 0.f;
      }
      int rr = (r0 + row) % FR_NR;
      if (rr < 0) rr += FR_NR;
      lds_store_unit<8>(lds + rr * FR_ROWE + u * 8, v);
    }
  };

  // prologue: the first window, rows -3 .. 10 (two fetches of eight rows: -3 .. 4 and 5 .. 12)
  fetch(-3); commit(-3);
  fetch(5); commit(5);
  bf8 wf[FR_NP][FR_KB];
#pragma unroll
  for (int p = 0; p < FR_NP; ++p)
#pragma unroll
    for (int kb = 0; kb < FR_KB; ++kb)
      wf[p][kb] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(a.wp + ((p * FR_KB + kb) * 64 + lane) * 8));
  __syncthreads();

  const int x0 = wave * 16;
  const int fbase = (x0 + lj) * FR_CIN + kg * 8;
  const int ox = tx0 + x0 + lj;
  const int rs_l = kg >> 1, co_l = 4 * (kg & 1);        // this lane's row of a pair and its 4 channels
  for (int y0 = 0; y0 < a.h; y0 += FR_STEP) {
    // rows y0 + 13 .. y0 + 20: the part of the NEXT window that is not in LDS yet (rows up to y0 + 12 are)
    const bool more = y0 + FR_STEP < a.h;
    if (more) fetch(y0 + 13);
    __builtin_amdgcn_sched_barrier(0);                  // (the loads stay in front of the MFMA chain: left alone they sink to commit())
    v4f acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = v4f{0.f, 0.f, 0.f, 0.f};
    int rr = (y0 - 3) % FR_NR;
    if (rr < 0) rr += FR_NR;
    // the fragments of input row jr + 1 are read while row jr multiplies (two register sets, the stages fenced: left alone
    // the compiler reads a fragment right before its first MFMA -- 56 exposed LDS latencies per step at two waves per SIMD)
    bf8 xf[2][FR_KB];
    {
      const u16* rowp = lds + rr * FR_ROWE + fbase;
#pragma unroll
      for (int kb = 0; kb < FR_KB; ++kb) xf[0][kb] = lds_frag<32>(rowp + kb * 32);
    }
#pragma unroll
    for (int jr = 0; jr < FR_WIN; ++jr) {               // input row y0 - 3 + jr
      rr = rr + 1 == FR_NR ? 0 : rr + 1;
      if (jr + 1 < FR_WIN) {
        const u16* rowp = lds + rr * FR_ROWE + fbase;
#pragma unroll
        for (int kb = 0; kb < FR_KB; ++kb) xf[(jr + 1) & 1][kb] = lds_frag<32>(rowp + kb * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < FR_KB; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {                   // row pair s: output rows y0 + 2 s, + 1; tap-row slot p = jr - 2 s
          const int p = jr - 2 * s;
          if (p >= 0 && p < FR_NP) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[p][kb], xf[jr & 1][kb], acc[s], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int oy = y0 + 2 * s + rs_l;
      if (oy >= a.h || ox >= a.w) continue;
      const int64_t o = ((int64_t)(n * a.h + oy) * a.w + ox) * a.out_cs + a.out_co + co_l;
      if constexpr (OUT_BF16)
        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out) + o) = make_uint2(pack2(acc[s][0], acc[s][1]), pack2(acc[s][2], acc[s][3]));
      else
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) = make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
    }
    if (more) commit(y0 + 13);                          // slots of rows y0 - 11 .. y0 - 4: outside this step's window
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- stride 2, k4, p1
// The two thin stride-2 layers at the full / half resolution boundary (p_y_z_in.3: Conv 16 -> 32, p_y_z_in.22:
// ConvTranspose 32 -> 16) and each other's data gradients, in the same scheme.
//
// T form (ConvTranspose2d forward / data gradient of the strided Conv2d): 32 gathered channels on the COARSE grid, 16
// produced on the fine one.  Output phase (py, px) of coarse pixel (y, x) reads coarse rows y + iy0(py) + {0, 1} and
// pixels x + ix0(px) + {0, 1}: flattened K of a tap row = 2 pixels x 32 channels = two K blocks = two whole pixels, and
// the three pixels x-1, x, x+1 serve both px phases -- 3 fragment reads per input row feed 8-16 MFMAs.  All 16 weight
// fragments (4 phases x 2 tap rows x 2 K blocks) in registers; a wave = 16 coarse pixels x 4 coarse rows x 4 phases.
struct FtArgs {
  const u16* in; int h, w, in_cs, in_co;          // coarse grid
  void* out; int oh, ow, out_cs, out_co;          // fine grid (2h x 2w)
  const u16* wp;
  PW pw;
  int tiles_x, tiles_y, n;
  double* stat; int stat_c;
  const u16* raw; int raw_cs, raw_co;             // STATS == 2 (see FbArgs)
  PW spw;
};

constexpr int FT_TW = 32, FT_TH = 16, FT_LW = FT_TW + 2, FT_LH = FT_TH + 2, FT_C = 32;
constexpr int FT_ROWE = FT_LW * FT_C;
constexpr size_t FT_LDS = ((size_t)FT_LH * FT_ROWE) * 2 + 3 * FT_C * sizeof(float) + 4 * 32 * sizeof(double) + 3 * 16 * sizeof(float);

template <int STATS>
__global__ __launch_bounds__(256, 3) void flatb_t2_kernel(FtArgs a) {
  constexpr int NU = FT_LH * FT_LW * 4, SLOTS = (NU + 255) / 256;       // 8-channel units
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + FT_LH * FT_ROWE);
  double* red = reinterpret_cast<double*>(lpw + 3 * FT_C);
  float* lspw = reinterpret_cast<float*>(red + 4 * 32);      // STATS == 2: the produced slot's activation (registers are full)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int t = fb_tile_of_block(per_img * a.n);
  const int n = t / per_img, tr = t - n * per_img;
  const int ty0 = (tr / a.tiles_x) * FT_TH, tx0 = (tr % a.tiles_x) * FT_TW;

  const int64_t img = (int64_t)n * a.h * a.w * a.in_cs + a.in_co + (tid & 3) * 8;
  uint4 stage[SLOTS];
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256, pi = e >> 2;
    const int row = pi / FT_LW, px = pi - row * FT_LW;
    const int gy = ty0 - 1 + row, gx = tx0 - 1 + px;
    if (e < NU && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) inside |= 1u << i;
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    stage[i] = *reinterpret_cast<const uint4*>(a.in + img + ((int64_t)cy * a.w + cx) * a.in_cs);
  }
  const bool on = a.pw.scale != nullptr;
  if (on && tid < FT_C) { lpw[tid] = a.pw.scale[tid]; lpw[FT_C + tid] = a.pw.shift[tid]; lpw[2 * FT_C + tid] = a.pw.slope[tid]; }
  if constexpr (STATS == 2) {
    if (tid < 16) {
      const bool son = a.spw.scale != nullptr;
      lspw[tid] = son ? a.spw.scale[tid] : 1.f; lspw[16 + tid] = son ? a.spw.shift[tid] : 0.f; lspw[32 + tid] = son ? a.spw.slope[tid] : 1.f;
    }
  }
  __syncthreads();
  {
    const int c0 = (tid & 3) * 8;
    float sc[8], sf[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = on ? lpw[c0 + j] : 1.f; sf[j] = on ? lpw[FT_C + c0 + j] : 0.f; sl[j] = on ? lpw[2 * FT_C + c0 + j] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = tid + i * 256;
      if (e >= NU) continue;
      if (!on) {                         // (a data gradient: no pending activation, the raw words are the image)
        *reinterpret_cast<uint4*>(lds + e * 8) = ((inside >> i) & 1u) ? stage[i] : make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      const unsigned w[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = fmaf(v[j], sc[j], sf[j]);
        x = x > 0.f ? x : x * sl[j];
        v[j] = ((inside >> i) & 1u) ? x : 0.f;
      }
      lds_store_unit<8>(lds + e * 8, v);
    }
  }
  // weights [phase (py, px)][tap row ty][K block = tap column tx]: 16 fragments
  bf8 wf[4][2][2];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx)
        wf[ph][ty][tx] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(a.wp + (((ph * 2 + ty) * 2 + tx) * 64 + lane) * 8));
  __syncthreads();

  const int x0 = (wave & 1) * 16;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int oxb = 2 * (tx0 + x0 + lj);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = (wave >> 1) * 8 + pass * 4;              // first coarse row (tile-relative) of this pass
    v4f acc[4][4];                                          // [coarse row][phase]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) acc[r][ph] = v4f{0.f, 0.f, 0.f, 0.f};
    const u16* base = lds + (r0 * FT_LW + x0 + lj) * FT_C + kg * 8;
#pragma unroll
    for (int jr = 0; jr < 6; ++jr) {                        // LDS rows r0 + jr = coarse rows r0 - 1 + jr
      bf8 xf[3];
#pragma unroll
      for (int o = 0; o < 3; ++o) xf[o] = lds_frag<32>(base + (jr * FT_LW + o) * FT_C);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const int ty = jr - r - py;                       // coarse row r reads LDS rows r + py + ty
          if (ty < 0 || ty > 1) continue;
#pragma unroll
          for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
              acc[r][py * 2 + px] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[py * 2 + px][ty][tx], xf[px + tx], acc[r][py * 2 + px], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint2 rw[STATS == 2 ? 4 : 1];
      if constexpr (STATS == 2) {          // the slot's raw values under this coarse row's four phases, loaded together
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          const int oy = min(2 * (ty0 + r0 + r) + (ph >> 1), a.oh - 1), ox = min(oxb + (ph & 1), a.ow - 1);
          rw[ph] = *reinterpret_cast<const uint2*>(a.raw + ((int64_t)(n * a.oh + oy) * a.ow + ox) * a.raw_cs + a.raw_co + kg * 4);
        }
      }
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int oy = 2 * (ty0 + r0 + r) + (ph >> 1), ox = oxb + (ph & 1);
        if (oy >= a.oh || ox >= a.ow) continue;
        const int64_t o = ((int64_t)(n * a.oh + oy) * a.ow + ox) * a.out_cs + a.out_co + kg * 4;
        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out) + o) =
            make_uint2(pack2(acc[r][ph][0], acc[r][ph][1]), pack2(acc[r][ph][2], acc[r][ph][3]));
        if constexpr (STATS == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float v = bf2f(f2bf(acc[r][ph][q])); s1[q] += v; s2[q] = fmaf(v, v, s2[q]); }
        }
        if constexpr (STATS == 2) {
          const float rv[4] = {bf2f((u16)(rw[ph].x & 0xffffu)), bf2f((u16)(rw[ph].x >> 16)), bf2f((u16)(rw[ph].y & 0xffffu)),
                               bf2f((u16)(rw[ph].y >> 16))};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = bf2f(f2bf(acc[r][ph][q]));
            const float t = fmaf(rv[q], lspw[kg * 4 + q], lspw[16 + kg * 4 + q]);
            const float g = t > 0.f ? d : d * lspw[32 + kg * 4 + q];
            s1[q] += g; s2[q] = fmaf(g, rv[q], s2[q]);
          }
        }
      }
    }
  }
  if constexpr (STATS) {
    double d1[4], d2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { d1[r] = (double)s1[r]; d2[r] = (double)s2[r]; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) { d1[r] += __shfl_xor(d1[r], m, 16); d2[r] += __shfl_xor(d2[r], m, 16); }
    if (lj == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[wave * 32 + kg * 4 + r] = d1[r]; red[wave * 32 + 16 + kg * 4 + r] = d2[r]; }
    }
    __syncthreads();
    if (tid < 32) {
      const double v = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
      a.stat[((int64_t)t * 2 + (tid >> 4)) * a.stat_c + (tid & 15)] = v;
    }
  }
}

struct FtPackArgs { const float* w; u16* dst; int64_t sa, sb; };
// [phase][ty][tx][k octet][row = produced channel][8]; tap (t) of phase p reads weight element bp_t_ky(p, 1, 2, 2, t)
__global__ __launch_bounds__(256) void flatb_t2_pack_kernel(FtPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 16 * 512) return;
  int r = i;
  const int e = r % 8; r /= 8;
  const int row = r % 16; r /= 16;
  const int kg = r % 4; r /= 4;
  const int tx = r % 2; r /= 2;
  const int ty = r % 2; r /= 2;
  const int px = r % 2, py = r / 2;
  const int c = kg * 8 + e;
  const int ky = bp_t_ky(py, 1, 2, 2, ty), kx = bp_t_ky(px, 1, 2, 2, tx);
  a.dst[i] = f2bf(a.w[c * a.sa + row * a.sb + ky * 4 + kx]);
}

// S form (strided Conv2d forward / data gradient of the ConvTranspose2d): 16 gathered channels on the FINE grid, 32
// produced on the coarse one.  Flattened K of a tap row = 4 pixels x 16 channels = two K blocks, starting at fine pixel
// 2x - 1: consecutive in the staged row.  32 produced channels = two MFMA row tiles whose rows interleave channel quads
// (tile 0: channels 8q .. 8q+3, tile 1: 8q+4 .. 8q+7), so a lane's 8 values are 8 consecutive channels = one 16-byte
// store.  An input-row fragment feeds up to 2 output rows x 2 row tiles; 16 weight fragments in registers.
struct FsArgs {
  const u16* in; int h, w, in_cs, in_co;          // fine grid
  void* out; int oh, ow, out_cs, out_co;          // coarse grid
  const u16* wp;
  PW pw;
  int tiles_x, tiles_y, n;
  double* stat; int stat_c;
  const u16* raw; int raw_cs, raw_co;             // STATS == 2 (see FbArgs)
  PW spw;
};

constexpr int FS_TW = 32, FS_TH = 8, FS_LW = 2 * (FS_TW - 1) + 4, FS_LH = 2 * (FS_TH - 1) + 4, FS_C = 16;
constexpr int FS_ROWE = FS_LW * FS_C;
constexpr size_t FS_LDS = ((size_t)FS_LH * FS_ROWE + 32) * 2 + 3 * FS_C * sizeof(float) + 4 * 64 * sizeof(double) + 3 * 32 * sizeof(float);

template <int STATS>
__global__ __launch_bounds__(256, 3) void flatb_s2_kernel(FsArgs a) {
  constexpr int NU = FS_LH * FS_LW * 2, SLOTS = (NU + 255) / 256;       // 8-channel units
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + FS_LH * FS_ROWE + 32);
  double* red = reinterpret_cast<double*>(lpw + 3 * FS_C);
  float* lspw = reinterpret_cast<float*>(red + 4 * 64);      // STATS == 2: the produced slot's activation [3][32]
  if constexpr (STATS == 2) {
    if (threadIdx.x < 32) {
      const bool son = a.spw.scale != nullptr;
      lspw[threadIdx.x] = son ? a.spw.scale[threadIdx.x] : 1.f; lspw[32 + threadIdx.x] = son ? a.spw.shift[threadIdx.x] : 0.f;
      lspw[64 + threadIdx.x] = son ? a.spw.slope[threadIdx.x] : 1.f;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int t = fb_tile_of_block(per_img * a.n);
  const int n = t / per_img, tr = t - n * per_img;
  const int ty0 = (tr / a.tiles_x) * FS_TH, tx0 = (tr % a.tiles_x) * FS_TW;      // coarse (output) tile origin

  const int64_t img = (int64_t)n * a.h * a.w * a.in_cs + a.in_co + (tid & 1) * 8;
  uint4 stage[SLOTS];
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256, pi = e >> 1;
    const int row = pi / FS_LW, px = pi - row * FS_LW;
    const int gy = 2 * ty0 - 1 + row, gx = 2 * tx0 - 1 + px;
    if (e < NU && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) inside |= 1u << i;
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    stage[i] = *reinterpret_cast<const uint4*>(a.in + img + ((int64_t)cy * a.w + cx) * a.in_cs);
  }
  const bool on = a.pw.scale != nullptr;
  if (on && tid < FS_C) { lpw[tid] = a.pw.scale[tid]; lpw[FS_C + tid] = a.pw.shift[tid]; lpw[2 * FS_C + tid] = a.pw.slope[tid]; }
  if (tid < 4) *reinterpret_cast<uint4*>(lds + FS_LH * FS_ROWE + tid * 8) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  {
    const int c0 = (tid & 1) * 8;
    float sc[8], sf[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = on ? lpw[c0 + j] : 1.f; sf[j] = on ? lpw[FS_C + c0 + j] : 0.f; sl[j] = on ? lpw[2 * FS_C + c0 + j] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = tid + i * 256;
      if (e >= NU) continue;
      if (!on) {
        *reinterpret_cast<uint4*>(lds + e * 8) = ((inside >> i) & 1u) ? stage[i] : make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      const unsigned w[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = fmaf(v[j], sc[j], sf[j]);
        x = x > 0.f ? x : x * sl[j];
        v[j] = ((inside >> i) & 1u) ? x : 0.f;
      }
      lds_store_unit<8>(lds + e * 8, v);
    }
  }
  bf8 wf[4][2][2];                       // [tap row][K block][row tile]
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        wf[ky][kb][mt] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(a.wp + (((ky * 2 + kb) * 2 + mt) * 64 + lane) * 8));
  __syncthreads();

  const int x0 = (wave & 1) * 16, r0 = (wave >> 1) * 4;
  v4f acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[r][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
  const u16* base = lds + (2 * r0 * FS_LW + 2 * (x0 + lj)) * FS_C + kg * 8;
#pragma unroll
  for (int jr = 0; jr < 10; ++jr)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const bf8 xf = lds_frag<32>(base + jr * FS_ROWE + kb * 32);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ky = jr - 2 * r;
        if (ky < 0 || ky > 3) continue;
        acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][kb][0], xf, acc[r][0], 0, 0, 0);
        acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][kb][1], xf, acc[r][1], 0, 0, 0);
      }
    }
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  const int ox = tx0 + x0 + lj;
  uint4 rw[STATS == 2 ? 4 : 1];
  if constexpr (STATS == 2) {              // the slot's raw values under this lane's four outputs, loaded together
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oy = min(ty0 + r0 + r, a.oh - 1), oxc = min(ox, a.ow - 1);
      rw[r] = *reinterpret_cast<const uint4*>(a.raw + ((int64_t)(n * a.oh + oy) * a.ow + oxc) * a.raw_cs + a.raw_co + kg * 8);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = ty0 + r0 + r;
    if (oy >= a.oh || ox >= a.ow) continue;
    const int64_t o = ((int64_t)(n * a.oh + oy) * a.ow + ox) * a.out_cs + a.out_co + kg * 8;
    *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(a.out) + o) =
        make_uint4(pack2(acc[r][0][0], acc[r][0][1]), pack2(acc[r][0][2], acc[r][0][3]),
                   pack2(acc[r][1][0], acc[r][1][1]), pack2(acc[r][1][2], acc[r][1][3]));
    if constexpr (STATS == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float v = bf2f(f2bf(acc[r][q >> 2][q & 3])); s1[q] += v; s2[q] = fmaf(v, v, s2[q]); }
    }
    if constexpr (STATS == 2) {
      const unsigned w4[4] = {rw[r].x, rw[r].y, rw[r].z, rw[r].w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float rv = bf2f((u16)((q & 1) ? (w4[q >> 1] >> 16) : (w4[q >> 1] & 0xffffu)));
        const float d = bf2f(f2bf(acc[r][q >> 2][q & 3]));
        const float tt = fmaf(rv, lspw[kg * 8 + q], lspw[32 + kg * 8 + q]);
        const float g = tt > 0.f ? d : d * lspw[64 + kg * 8 + q];
        s1[q] += g; s2[q] = fmaf(g, rv, s2[q]);
      }
    }
  }
  if constexpr (STATS) {
    double d1[8], d2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { d1[q] = (double)s1[q]; d2[q] = (double)s2[q]; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
      for (int q = 0; q < 8; ++q) { d1[q] += __shfl_xor(d1[q], m, 16); d2[q] += __shfl_xor(d2[q], m, 16); }
    if (lj == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { red[wave * 64 + kg * 8 + q] = d1[q]; red[wave * 64 + 32 + kg * 8 + q] = d2[q]; }
    }
    __syncthreads();
    if (tid < 64) {
      const double v = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
      a.stat[((int64_t)t * 2 + (tid >> 5)) * a.stat_c + (tid & 31)] = v;
    }
  }
}

// [tap row ky][K block][row tile mt][k octet][row][8]: produced channel = 8 (row >> 2) + 4 mt + (row & 3),
// K index k = 32 kb + 8 kg + e -> (tap column kx = k / 16, gathered channel c = k % 16)
__global__ __launch_bounds__(256) void flatb_s2_pack_kernel(FtPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 16 * 512) return;
  int r = i;
  const int e = r % 8; r /= 8;
  const int row = r % 16; r /= 16;
  const int kg = r % 4; r /= 4;
  const int mt = r % 2; r /= 2;
  const int kb = r % 2, ky = r / 2;
  const int co = 8 * (row >> 2) + 4 * mt + (row & 3);
  const int k = kb * 32 + kg * 8 + e, kx = k / 16, c = k % 16;
  a.dst[i] = f2bf(a.w[c * a.sa + co * a.sb + ky * 4 + kx]);
}

// ---------------------------------------------------------------------------------------------- 32 <-> 64, stride 2
// The same two forms one level down (p_y_z_in.6: Conv 32 -> 64, p_y_z_in.19: ConvTranspose 64 -> 32, and each other's
// data gradients).  A layer's weights (64 fragments) no longer fit one wave: the four waves of a workgroup split them
// -- S form by halves of the 64 produced channels (two strips of 16 pixels x two halves), T form by output row phase
// (two strips x two phases) -- 32 fragments per wave, every wave reads the staged tile's fragments itself.
constexpr int GS_TW = 32, GS_TH = 4, GS_LW = 2 * (GS_TW - 1) + 4, GS_LH = 2 * (GS_TH - 1) + 4, GS_C = 32;
constexpr int GS_ROWE = GS_LW * GS_C;
constexpr size_t GS_LDS = ((size_t)GS_LH * GS_ROWE) * 2 + 3 * GS_C * sizeof(float) + 4 * 64 * sizeof(double);

template <bool STATS>
__global__ __launch_bounds__(256, 2) void flatb_s2w_kernel(FsArgs a) {
  constexpr int NU = GS_LH * GS_LW * 4, SLOTS = (NU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + GS_LH * GS_ROWE);
  double* red = reinterpret_cast<double*>(lpw + 3 * GS_C);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int t = fb_tile_of_block(per_img * a.n);
  const int n = t / per_img, tr = t - n * per_img;
  const int ty0 = (tr / a.tiles_x) * GS_TH, tx0 = (tr % a.tiles_x) * GS_TW;

  const int64_t img = (int64_t)n * a.h * a.w * a.in_cs + a.in_co + (tid & 3) * 8;
  uint4 stage[SLOTS];
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256, pi = e >> 2;
    const int row = pi / GS_LW, px = pi - row * GS_LW;
    const int gy = 2 * ty0 - 1 + row, gx = 2 * tx0 - 1 + px;
    if (e < NU && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) inside |= 1u << i;
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    stage[i] = *reinterpret_cast<const uint4*>(a.in + img + ((int64_t)cy * a.w + cx) * a.in_cs);
  }
  const bool on = a.pw.scale != nullptr;
  if (on && tid < GS_C) { lpw[tid] = a.pw.scale[tid]; lpw[GS_C + tid] = a.pw.shift[tid]; lpw[2 * GS_C + tid] = a.pw.slope[tid]; }
  __syncthreads();
  {
    const int c0 = (tid & 3) * 8;
    float sc[8], sf[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = on ? lpw[c0 + j] : 1.f; sf[j] = on ? lpw[GS_C + c0 + j] : 0.f; sl[j] = on ? lpw[2 * GS_C + c0 + j] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = tid + i * 256;
      if (e >= NU) continue;
      if (!on) {
        *reinterpret_cast<uint4*>(lds + e * 8) = ((inside >> i) & 1u) ? stage[i] : make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      const unsigned w[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = fmaf(v[j], sc[j], sf[j]);
        x = x > 0.f ? x : x * sl[j];
        v[j] = ((inside >> i) & 1u) ? x : 0.f;
      }
      lds_store_unit<8>(lds + e * 8, v);
    }
  }
  const int x0 = (wave & 1) * 16, half = wave >> 1;
  bf8 wf[4][4][2];                       // [tap row][K block = fine pixel of the tap row][row tile] of this channel half
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        wf[ky][kb][mt] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(
            a.wp + (((((half * 4 + ky) * 4 + kb) * 2 + mt) * 64) + lane) * 8));
  __syncthreads();

  v4f acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[r][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
  const u16* base = lds + (2 * (x0 + lj)) * GS_C + kg * 8;
#pragma unroll
  for (int jr = 0; jr < GS_LH; ++jr)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const bf8 xf = lds_frag<32>(base + jr * GS_ROWE + kb * GS_C);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ky = jr - 2 * r;
        if (ky < 0 || ky > 3) continue;
        acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][kb][0], xf, acc[r][0], 0, 0, 0);
        acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][kb][1], xf, acc[r][1], 0, 0, 0);
      }
    }
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  const int ox = tx0 + x0 + lj;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = ty0 + r;
    if (oy >= a.oh || ox >= a.ow) continue;
    const int64_t o = ((int64_t)(n * a.oh + oy) * a.ow + ox) * a.out_cs + a.out_co + half * 32 + kg * 8;
    *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(a.out) + o) =
        make_uint4(pack2(acc[r][0][0], acc[r][0][1]), pack2(acc[r][0][2], acc[r][0][3]),
                   pack2(acc[r][1][0], acc[r][1][1]), pack2(acc[r][1][2], acc[r][1][3]));
    if constexpr (STATS) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float v = bf2f(f2bf(acc[r][q >> 2][q & 3])); s1[q] += v; s2[q] = fmaf(v, v, s2[q]); }
    }
  }
  if constexpr (STATS) {
    double d1[8], d2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { d1[q] = (double)s1[q]; d2[q] = (double)s2[q]; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
      for (int q = 0; q < 8; ++q) { d1[q] += __shfl_xor(d1[q], m, 16); d2[q] += __shfl_xor(d2[q], m, 16); }
    if (lj == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { red[wave * 64 + kg * 8 + q] = d1[q]; red[wave * 64 + 32 + kg * 8 + q] = d2[q]; }
    }
    __syncthreads();
    if (tid < 128) {                     // (channel half, statistic, channel of the half): the two strips' waves
      const int h = tid >> 6, st = (tid >> 5) & 1, c = tid & 31;
      const double v = red[(2 * h) * 64 + st * 32 + c] + red[(2 * h + 1) * 64 + st * 32 + c];
      a.stat[((int64_t)t * 2 + st) * a.stat_c + 32 * h + c] = v;
    }
  }
}

// [channel half][ky][K block = tap column][row tile][k octet][row][8]; produced channel = 32 half + 8 (row >> 2) + 4 mt + (row & 3)
__global__ __launch_bounds__(256) void flatb_s2w_pack_kernel(FtPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 512) return;
  int r = i;
  const int e = r % 8; r /= 8;
  const int row = r % 16; r /= 16;
  const int kg = r % 4; r /= 4;
  const int mt = r % 2; r /= 2;
  const int kb = r % 4; r /= 4;
  const int ky = r % 4, half = r / 4;
  const int co = 32 * half + 8 * (row >> 2) + 4 * mt + (row & 3);
  a.dst[i] = f2bf(a.w[(kg * 8 + e) * a.sa + co * a.sb + ky * 4 + kb]);
}

constexpr int GT_TW = 32, GT_TH = 8, GT_LW = GT_TW + 2, GT_LH = GT_TH + 2, GT_C = 64;
constexpr int GT_ROWE = GT_LW * GT_C;
constexpr size_t GT_LDS = ((size_t)GT_LH * GT_ROWE) * 2 + 3 * GT_C * sizeof(float) + 4 * 64 * sizeof(double) + 3 * 32 * sizeof(float);

template <int STATS>
__global__ __launch_bounds__(256, 2) void flatb_t2w_kernel(FtArgs a) {
  constexpr int NU = GT_LH * GT_LW * 8, SLOTS = (NU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + GT_LH * GT_ROWE);
  double* red = reinterpret_cast<double*>(lpw + 3 * GT_C);
  float* lspw = reinterpret_cast<float*>(red + 4 * 64);      // STATS == 2: the produced slot's activation [3][32]
  if constexpr (STATS == 2) {
    if (threadIdx.x < 32) {
      const bool son = a.spw.scale != nullptr;
      lspw[threadIdx.x] = son ? a.spw.scale[threadIdx.x] : 1.f; lspw[32 + threadIdx.x] = son ? a.spw.shift[threadIdx.x] : 0.f;
      lspw[64 + threadIdx.x] = son ? a.spw.slope[threadIdx.x] : 1.f;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;
  const int per_img = a.tiles_x * a.tiles_y;
  const int t = fb_tile_of_block(per_img * a.n);
  const int n = t / per_img, tr = t - n * per_img;
  const int ty0 = (tr / a.tiles_x) * GT_TH, tx0 = (tr % a.tiles_x) * GT_TW;

  const int64_t img = (int64_t)n * a.h * a.w * a.in_cs + a.in_co + (tid & 7) * 8;
  uint4 stage[SLOTS];
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256, pi = e >> 3;
    const int row = pi / GT_LW, px = pi - row * GT_LW;
    const int gy = ty0 - 1 + row, gx = tx0 - 1 + px;
    if (e < NU && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) inside |= 1u << i;
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    stage[i] = *reinterpret_cast<const uint4*>(a.in + img + ((int64_t)cy * a.w + cx) * a.in_cs);
  }
  const bool on = a.pw.scale != nullptr;
  if (on && tid < GT_C) { lpw[tid] = a.pw.scale[tid]; lpw[GT_C + tid] = a.pw.shift[tid]; lpw[2 * GT_C + tid] = a.pw.slope[tid]; }
  __syncthreads();
  {
    const int c0 = (tid & 7) * 8;
    float sc[8], sf[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = on ? lpw[c0 + j] : 1.f; sf[j] = on ? lpw[GT_C + c0 + j] : 0.f; sl[j] = on ? lpw[2 * GT_C + c0 + j] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int e = tid + i * 256;
      if (e >= NU) continue;
      if (!on) {
        *reinterpret_cast<uint4*>(lds + e * 8) = ((inside >> i) & 1u) ? stage[i] : make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      const unsigned w[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((u16)(w[j] & 0xffffu)); v[2 * j + 1] = bf2f((u16)(w[j] >> 16)); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = fmaf(v[j], sc[j], sf[j]);
        x = x > 0.f ? x : x * sl[j];
        v[j] = ((inside >> i) & 1u) ? x : 0.f;
      }
      lds_store_unit<8>(lds + e * 8, v);
    }
  }
  const int x0 = (wave & 1) * 16, py = wave >> 1;
  bf8 wf[2][2][4][2];                    // [px][ty][K block = (tx, channel half)][row tile] of this wave's row phase
#pragma unroll
  for (int px = 0; px < 2; ++px)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          wf[px][ty][kb][mt] = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(
              a.wp + ((((((py * 2 + px) * 2 + ty) * 4 + kb) * 2 + mt) * 64) + lane) * 8));
  __syncthreads();

  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  const int oxb = 2 * (tx0 + x0 + lj);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = pass * 4;
    v4f acc[4][2][2];                    // [coarse row][px][row tile]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int px = 0; px < 2; ++px) { acc[r][px][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[r][px][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
    const u16* base = lds + ((r0 + py) * GT_LW + x0 + lj) * GT_C + kg * 8;      // LDS row of coarse row r, tap ty: r + py + ty
#pragma unroll
    for (int jr = 0; jr < 5; ++jr) {
      bf8 xf[3][2];
#pragma unroll
      for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) xf[o][hc] = lds_frag<32>(base + (jr * GT_LW + o) * GT_C + hc * 32);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ty = jr - r;
        if (ty < 0 || ty > 1) continue;
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx)
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) {
              acc[r][px][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[px][ty][tx * 2 + hc][0], xf[px + tx][hc], acc[r][px][0], 0, 0, 0);
              acc[r][px][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[px][ty][tx * 2 + hc][1], xf[px + tx][hc], acc[r][px][1], 0, 0, 0);
            }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint4 rw[STATS == 2 ? 2 : 1];
      if constexpr (STATS == 2) {
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int oy = min(2 * (ty0 + r0 + r) + py, a.oh - 1), ox = min(oxb + px, a.ow - 1);
          rw[px] = *reinterpret_cast<const uint4*>(a.raw + ((int64_t)(n * a.oh + oy) * a.ow + ox) * a.raw_cs + a.raw_co + kg * 8);
        }
      }
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int oy = 2 * (ty0 + r0 + r) + py, ox = oxb + px;
        if (oy >= a.oh || ox >= a.ow) continue;
        const int64_t o = ((int64_t)(n * a.oh + oy) * a.ow + ox) * a.out_cs + a.out_co + kg * 8;
        *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(a.out) + o) =
            make_uint4(pack2(acc[r][px][0][0], acc[r][px][0][1]), pack2(acc[r][px][0][2], acc[r][px][0][3]),
                       pack2(acc[r][px][1][0], acc[r][px][1][1]), pack2(acc[r][px][1][2], acc[r][px][1][3]));
        if constexpr (STATS == 1) {
#pragma unroll
          for (int q = 0; q < 8; ++q) { const float v = bf2f(f2bf(acc[r][px][q >> 2][q & 3])); s1[q] += v; s2[q] = fmaf(v, v, s2[q]); }
        }
        if constexpr (STATS == 2) {
          const unsigned w4[4] = {rw[px].x, rw[px].y, rw[px].z, rw[px].w};
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float rv = bf2f((u16)((q & 1) ? (w4[q >> 1] >> 16) : (w4[q >> 1] & 0xffffu)));
            const float d = bf2f(f2bf(acc[r][px][q >> 2][q & 3]));
            const float tt = fmaf(rv, lspw[kg * 8 + q], lspw[32 + kg * 8 + q]);
            const float g = tt > 0.f ? d : d * lspw[64 + kg * 8 + q];
            s1[q] += g; s2[q] = fmaf(g, rv, s2[q]);
          }
        }
      }
    }
  }
  if constexpr (STATS) {
    double d1[8], d2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { d1[q] = (double)s1[q]; d2[q] = (double)s2[q]; }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
      for (int q = 0; q < 8; ++q) { d1[q] += __shfl_xor(d1[q], m, 16); d2[q] += __shfl_xor(d2[q], m, 16); }
    if (lj == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { red[wave * 64 + kg * 8 + q] = d1[q]; red[wave * 64 + 32 + kg * 8 + q] = d2[q]; }
    }
    __syncthreads();
    if (tid < 64) {
      const double v = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
      a.stat[((int64_t)t * 2 + (tid >> 5)) * a.stat_c + (tid & 31)] = v;
    }
  }
}

// [py][px][ty][K block = (tx, channel half)][row tile][k octet][row][8]; produced channel = 8 (row >> 2) + 4 mt + (row & 3)
__global__ __launch_bounds__(256) void flatb_t2w_pack_kernel(FtPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 512) return;
  int r = i;
  const int e = r % 8; r /= 8;
  const int row = r % 16; r /= 16;
  const int kg = r % 4; r /= 4;
  const int mt = r % 2; r /= 2;
  const int kb = r % 4; r /= 4;
  const int ty = r % 2; r /= 2;
  const int px = r % 2, py = r / 2;
  const int co = 8 * (row >> 2) + 4 * mt + (row & 3);
  const int tx = kb >> 1, c = (kb & 1) * 32 + kg * 8 + e;
  const int ky = bp_t_ky(py, 1, 2, 2, ty), kx = bp_t_ky(px, 1, 2, 2, tx);
  a.dst[i] = f2bf(a.w[c * a.sa + co * a.sb + ky * 4 + kx]);
}

struct FbPackArgs {
  const float* w; u16* dst;
  int64_t sa, sb;
  int ks, cin, cinp, cout, coutp, KB, flip;
  int64_t total;
};

// [fragment p][K block][k octet kg][row i][8]: row i = (output row select rs, produced channel co), tap row ky = p - rs,
// K index k = 32 kb + 8 kg + e -> (tap column kx = k / cinp, gathered channel c = k % cinp); zero where no such tap /
// channel.  flip: the gather is the data gradient of a convolution (transposed form): tap t reads weight k - 1 - t.
__global__ __launch_bounds__(256) void flatb_pack_kernel(FbPackArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.total) return;
  int64_t r = i;
  const int e = r % 8; r /= 8;
  const int row = r % 16; r /= 16;
  const int kg = r % 4; r /= 4;
  const int kb = r % a.KB; r /= a.KB;
  const int p = (int)r;
  const int rs = row / a.coutp, co = row % a.coutp;
  const int ky = p - rs;
  const int k = kb * 32 + kg * 8 + e;
  const int kx = k / a.cinp, c = k % a.cinp;
  float v = 0.f;
  if (ky >= 0 && ky < a.ks && kx < a.ks && c < a.cin && co < a.cout) {
    const int wy = a.flip ? a.ks - 1 - ky : ky, wx = a.flip ? a.ks - 1 - kx : kx;
    v = a.w[c * a.sa + co * a.sb + wy * a.ks + wx];
  }
  a.dst[i] = f2bf(v);
}

// which instance serves this geometry: 0 none, 1 k7 16->8, 2 k7 8->16, 3 k5 3(4)->16 (stem forward), 4 k5 16->(1..4),
// 5 stride-2 k4 transposed form 32->16, 6 stride-2 k4 conv form 16->32, 7 / 8 the same for 64->32 / 32->64,
// 9 k5 8->(1..4) (second layer of the heads, bf16 in)
struct FbKind { int kind, ks, cinp, coutp; };
static inline bool fb_s2(int kind) { return kind >= 5 && kind <= 8; }
static FbKind fb_kind(const ConvGeom& g) {
  static const bool off = getenv("BP_BF16_NOFLAT") != nullptr;
  static const bool off5 = getenv("BP_BF16_NOFLAT5") != nullptr;
  static const bool off2 = getenv("BP_BF16_NOFLAT2") != nullptr;
  FbKind none{0, 0, 0, 0};
  if (!off && !off2 && g.k == 4 && g.stride == 2 && g.pad == 1) {
    if (g.gather_transposed && g.nphase == 2 && g.taps == 2 && g.IS == 1 && g.OS == 2 && g.cin_g == 32 && g.cout_g == 16)
      return FbKind{5, 4, 32, 16};
    if (!g.gather_transposed && g.nphase == 1 && g.taps == 4 && g.IS == 2 && g.OS == 1 && g.cin_g == 16 && g.cout_g == 32)
      return FbKind{6, 4, 16, 32};
    static const bool offw = getenv("BP_BF16_NOFLAT2W") != nullptr;
    if (!offw && g.gather_transposed && g.nphase == 2 && g.taps == 2 && g.IS == 1 && g.OS == 2 && g.cin_g == 64 && g.cout_g == 32)
      return FbKind{7, 4, 64, 32};
    if (!offw && !g.gather_transposed && g.nphase == 1 && g.taps == 4 && g.IS == 2 && g.OS == 1 && g.cin_g == 32 && g.cout_g == 64)
      return FbKind{8, 4, 32, 64};
  }
  if (off || g.stride != 1 || g.nphase != 1 || g.IS != 1 || g.OS != 1 || g.taps != g.k || g.pad != g.k / 2) return none;
  if (g.k == 7 && g.cin_g == 16 && g.cout_g == 8) return FbKind{1, 7, 16, 8};
  if (g.k == 7 && g.cin_g == 8 && g.cout_g == 16) return FbKind{2, 7, 8, 16};
  if (g.k == 5 && !off5 && g.cin_g <= 4 && g.cin_g >= 1 && g.cout_g == 16) return FbKind{3, 5, 4, 16};
  if (g.k == 5 && !off5 && g.cin_g == 16 && g.cout_g >= 1 && g.cout_g <= 4) return FbKind{4, 5, 16, 4};
  static const bool off9 = getenv("BP_BF16_NOFLAT9") != nullptr;
  if (g.k == 5 && !off9 && g.cin_g == 8 && g.cout_g >= 1 && g.cout_g <= 4 && !g.gather_transposed) return FbKind{9, 5, 8, 4};
  return none;
}

template <int KS, int CIN, int COUTP, bool IB, bool OB, int ST, int R = 4>
static void fb_launch(const FbArgs& a, dim3 grid, hipStream_t st) {
  auto k = flatb_kernel<KS, CIN, COUTP, IB, OB, ST, R>;
  constexpr size_t lds = FbShape<KS, CIN>::LDS;
  static const int once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds), 0);
  (void)once;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
}

}  // namespace

// elements of the flattened-K weight image of this layer (0: the kernel does not apply)
int64_t bp_bf16_flat_packed_elems(const ConvGeom& g) {
  const FbKind f = fb_kind(g);
  if (!f.kind) return 0;
  if (f.kind == 7 || f.kind == 8) return 64 * 512;
  if (fb_s2(f.kind)) return 16 * 512;
  const int KB = (f.ks * f.cinp + 31) / 32, NP = f.ks + 16 / f.coutp - 1;
  return (int64_t)NP * KB * 64 * 8;
}

int bp_bf16_flat_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st) {
  const FbKind f = fb_kind(g);
  if (!f.kind) return BP_EUNSUPPORTED;
  if (fb_s2(f.kind)) {
    FtPackArgs t{w_torch, dst, wm.sa, wm.sb};
    if (f.kind == 5) hipLaunchKernelGGL(flatb_t2_pack_kernel, dim3(32), dim3(256), 0, st, t);
    else if (f.kind == 6) hipLaunchKernelGGL(flatb_s2_pack_kernel, dim3(32), dim3(256), 0, st, t);
    else if (f.kind == 7) hipLaunchKernelGGL(flatb_t2w_pack_kernel, dim3(128), dim3(256), 0, st, t);
    else hipLaunchKernelGGL(flatb_s2w_pack_kernel, dim3(128), dim3(256), 0, st, t);
    BP_CHECK_LAUNCH();
    return BP_OK;
  }
  FbPackArgs a{};
  a.w = w_torch; a.dst = dst; a.sa = wm.sa; a.sb = wm.sb; a.ks = f.ks; a.cin = g.cin_g; a.cinp = f.cinp;
  a.cout = g.cout_g; a.coutp = f.coutp; a.KB = (f.ks * f.cinp + 31) / 32;
  a.flip = g.gather_transposed;
  a.total = bp_bf16_flat_packed_elems(g);
  hipLaunchKernelGGL(flatb_pack_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

static int64_t fb_tiles(const bp_view* out) {
  return (int64_t)bp_ceil_div(out->w, FB_TW) * bp_ceil_div(out->h, FB_TH) * out->n;
}
// tiles of the stride-2 instances: coarse-grid tiles of the gathered (T form) / produced (S form) tensor
static int64_t f2_tiles(int kind, const bp_view* in, const bp_view* out, int* tx, int* ty) {
  if (kind == 5) { *tx = bp_ceil_div(in->w, FT_TW); *ty = bp_ceil_div(in->h, FT_TH); }
  else if (kind == 6) { *tx = bp_ceil_div(out->w, FS_TW); *ty = bp_ceil_div(out->h, FS_TH); }
  else if (kind == 7) { *tx = bp_ceil_div(in->w, GT_TW); *ty = bp_ceil_div(in->h, GT_TH); }
  else { *tx = bp_ceil_div(out->w, GS_TW); *ty = bp_ceil_div(out->h, GS_TH); }
  return (int64_t)*tx * *ty * out->n;
}
static bool f2_ok(const FbKind& f, const bp_view* in, const bp_view* out) {
  if (in->dtype != BP_BF16 || out->dtype != BP_BF16 || in->n != out->n || in->c != f.cinp || out->c != f.coutp) return false;
  const bool tform = f.kind == 5 || f.kind == 7;
  if (tform ? (out->h != 2 * in->h || out->w != 2 * in->w) : (in->h != 2 * out->h || in->w != 2 * out->w)) return false;
  if (reinterpret_cast<uintptr_t>(in->ptr) % 16 || reinterpret_cast<uintptr_t>(out->ptr) % 16) return false;
  if (in->cstride % 8 || in->coff % 8) return false;
  const int ov = f.kind == 5 ? 4 : 8;                      // channels per vector store
  if (out->cstride % ov || out->coff % ov) return false;
  int tx, ty;
  return f2_tiles(f.kind, in, out, &tx, &ty) < (1ll << 31);
}

// stats: 0 none, 1 {sum y, sum y^2} of the produced tensor, 2 {sum g, sum g*raw} of a data gradient (IgemmStatsReq)
bool bp_bf16_flat_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int stats) {
  const FbKind f = fb_kind(g);
  if (!f.kind || bias || !in || !out) return false;
  static const bool m2wide = !(getenv("BP_BF16_M2_WIDE") && atoi(getenv("BP_BF16_M2_WIDE")) == 0);
  if (fb_s2(f.kind)) return f2_ok(f, in, out) && (stats != 2 || f.kind == 5 || (m2wide && f.kind != 8));
  if (stats == 1 && f.kind != 3) return false;
  if (stats == 2 && f.kind != 2) return false;
  if (stats > 2) return false;
  if (in->c != g.cin_g || out->c != g.cout_g || in->h != out->h || in->w != out->w || in->n != out->n) return false;
  const bool ib = in->dtype == BP_BF16, ob = out->dtype == BP_BF16;
  // element types of the instances: the bf16 trunk on one side; the few-channel edge on the other is fp32, or bf16 where
  // the 8-channel slot of a head is stored as bf16 (kinds 1, 2)
  if (f.kind == 1 ? !ib : f.kind == 2 ? !ob : (f.kind == 4 || f.kind == 9) ? !(ib && !ob) : !(!ib && ob)) return false;
  if (reinterpret_cast<uintptr_t>(in->ptr) % 16 || reinterpret_cast<uintptr_t>(out->ptr) % 16) return false;
  // staged units: 8 channels (16 bytes of bf16 / two float4) or, for the 4-channel stem, one float4 per pixel
  if (f.kind == 3) {
    if (in->cstride % 4 || in->coff % 4 || in->coff + 4 > in->cstride) return false;      // (reads a whole channel quad)
  } else if ((in->cstride * (ib ? 2 : 4)) % 16 || (in->coff * (ib ? 2 : 4)) % 16) return false;
  if (g.cout_g >= 4 && (out->cstride % 4 || out->coff % 4)) return false;                 // 4-channel vector stores
  return fb_tiles(out) < (1ll << 31);
}

size_t bp_bf16_flat_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode) {
  if (!bp_bf16_flat_ok(g, in, out, nullptr, mode)) return 0;
  const FbKind f = fb_kind(g);
  if (fb_s2(f.kind)) {
    int tx, ty;
    return bp_stats_rows_bytes(f2_tiles(f.kind, in, out, &tx, &ty), g.cout_g);
  }
  return bp_stats_rows_bytes(fb_tiles(out), g.cout_g);
}

int bp_bf16_flat_run(const ConvGeom& g, const bp_view* in, const PW& pw, const u16* packed_flat, const bp_view* out,
                     hipStream_t st, const IgemmStatsReq* sr) {
  const FbKind f = fb_kind(g);
  if (fb_s2(f.kind)) {
    int tx, ty;
    const int64_t rows = f2_tiles(f.kind, in, out, &tx, &ty);
    double* stat = nullptr;
    if (sr) {
      const size_t need = bp_stats_rows_bytes(rows, g.cout_g);
      if ((sr->mode != 1 && !(sr->mode == 2 && f.kind != 8)) || !need) return BP_EUNSUPPORTED;
      if (!sr->ws || sr->ws_bytes < need || !sr->sums) return BP_EWORKSPACE;
      stat = reinterpret_cast<double*>(sr->ws);
    }
    const dim3 grid((unsigned)rows), block(256);
    if (sr && sr->mode == 2) {          // as a data gradient: the producer's batch-norm backward sums (kinds 5, 6, 7)
      const bp_view* r = sr->raw;
      const int rv = f.kind == 5 ? 4 : 8;                       // raw channels per vector load
      if (!r || r->dtype != BP_BF16 || r->n != out->n || r->h != out->h || r->w != out->w || r->c != out->c ||
          r->cstride % rv || r->coff % rv || reinterpret_cast<uintptr_t>(r->ptr) % (2 * rv))
        return BP_EUNSUPPORTED;
      const u16* rp = reinterpret_cast<const u16*>(r->ptr);
#define BP_F2M2(KERNEL, ARGS, LDS_)                                                                                   \
      do {                                                                                                            \
        static const int once2 = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<2>),                 \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)), 0); \
        (void)once2;                                                                                                  \
        hipLaunchKernelGGL(KERNEL<2>, grid, block, LDS_, st, ARGS);                                                   \
      } while (0)
      if (f.kind == 6) {
        FsArgs t{reinterpret_cast<const u16*>(in->ptr), in->h, in->w, in->cstride, in->coff, out->ptr, out->h, out->w,
                 out->cstride, out->coff, packed_flat, pw, tx, ty, in->n, stat, g.cout_g, rp, r->cstride, r->coff, sr->spw};
        BP_F2M2(flatb_s2_kernel, t, FS_LDS);
      } else {
        FtArgs t{reinterpret_cast<const u16*>(in->ptr), in->h, in->w, in->cstride, in->coff, out->ptr, out->h, out->w,
                 out->cstride, out->coff, packed_flat, pw, tx, ty, in->n, stat, g.cout_g, rp, r->cstride, r->coff, sr->spw};
        if (f.kind == 5) BP_F2M2(flatb_t2_kernel, t, FT_LDS);
        else BP_F2M2(flatb_t2w_kernel, t, GT_LDS);
      }
#undef BP_F2M2
      BP_CHECK_LAUNCH();
      return bp_stats_rows_finish(stat, rows, g.cout_g, sr, st);
    }
#define BP_F2(KERNEL, ARGS, LDS_)                                                                                        \
    do {                                                                                                               \
      static const int once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<true>),                  \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)),      \
                               (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<false>),                 \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)), 0); \
      (void)once;                                                                                                      \
      if (sr) hipLaunchKernelGGL(KERNEL<true>, grid, block, LDS_, st, ARGS);                                           \
      else hipLaunchKernelGGL(KERNEL<false>, grid, block, LDS_, st, ARGS);                                             \
    } while (0)
    if (f.kind == 5 || f.kind == 7) {
      FtArgs t{reinterpret_cast<const u16*>(in->ptr), in->h, in->w, in->cstride, in->coff, out->ptr, out->h, out->w,
               out->cstride, out->coff, packed_flat, pw, tx, ty, in->n, stat, g.cout_g, nullptr, 0, 0, PW{nullptr, nullptr, nullptr}};
      if (f.kind == 5) BP_F2(flatb_t2_kernel, t, FT_LDS);
      else BP_F2(flatb_t2w_kernel, t, GT_LDS);
    } else {
      FsArgs t{reinterpret_cast<const u16*>(in->ptr), in->h, in->w, in->cstride, in->coff, out->ptr, out->h, out->w,
               out->cstride, out->coff, packed_flat, pw, tx, ty, in->n, stat, g.cout_g, nullptr, 0, 0, PW{nullptr, nullptr, nullptr}};
      if (f.kind == 6) BP_F2(flatb_s2_kernel, t, FS_LDS);
      else BP_F2(flatb_s2w_kernel, t, GS_LDS);
    }
#undef BP_F2
    BP_CHECK_LAUNCH();
    if (!sr) return BP_OK;
    return bp_stats_rows_finish(stat, rows, g.cout_g, sr, st);
  }
  FbArgs a{};
  a.in = in->ptr; a.h = in->h; a.w = in->w; a.in_cs = in->cstride; a.in_co = in->coff; a.cin = g.cin_g;
  a.out = out->ptr; a.out_cs = out->cstride; a.out_co = out->coff; a.cout = g.cout_g;
  a.wp = packed_flat; a.pw = pw; a.n = in->n;
  a.tiles_x = bp_ceil_div(out->w, FB_TW); a.tiles_y = bp_ceil_div(out->h, FB_TH);
  const int64_t rows = fb_tiles(out);
  if (sr) {
    const size_t need = bp_stats_rows_bytes(rows, g.cout_g);
    if (!((sr->mode == 1 && f.kind == 3) || (sr->mode == 2 && f.kind == 2)) || !need) return BP_EUNSUPPORTED;
    if (!sr->ws || sr->ws_bytes < need || !sr->sums) return BP_EWORKSPACE;
    a.stat = reinterpret_cast<double*>(sr->ws); a.stat_c = g.cout_g;
    if (sr->mode == 2) {
      const bp_view* r = sr->raw;
      if (!r || r->dtype != BP_BF16 || r->n != out->n || r->h != out->h || r->w != out->w || r->c != out->c ||
          r->cstride % 4 || r->coff % 4 || reinterpret_cast<uintptr_t>(r->ptr) % 8)
        return BP_EUNSUPPORTED;
      a.raw = reinterpret_cast<const u16*>(r->ptr); a.raw_cs = r->cstride; a.raw_co = r->coff; a.spw = sr->spw;
    }
  }
  const dim3 grid((unsigned)rows);
  const bool ib = in->dtype == BP_BF16, ob = out->dtype == BP_BF16;
  switch (f.kind) {
    case 1: {
      // the row walker where there are enough strips to fill the GPU (the training batch, the paint sub-batches)
      static const bool no_ring = getenv("BP_FLATR") && atoi(getenv("BP_FLATR")) == 0;
      const int strips = bp_ceil_div(out->w, FR_TW);
      if (!no_ring && (int64_t)in->n * strips >= 128 && (int64_t)in->n * strips < (1ll << 31) && (ob ? out->cstride % 4 == 0 : true)) {
        FrArgs r{};
        r.in = reinterpret_cast<const u16*>(in->ptr); r.h = in->h; r.w = in->w; r.in_cs = in->cstride; r.in_co = in->coff;
        r.out = out->ptr; r.out_cs = out->cstride; r.out_co = out->coff; r.wp = packed_flat; r.pw = pw;
        r.strips = strips; r.n = in->n;
        static const int once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(flatr_k7_kernel<true>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FR_LDS),
                                 (void)hipFuncSetAttribute(reinterpret_cast<const void*>(flatr_k7_kernel<false>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FR_LDS), 0);
        (void)once;
        const dim3 rgrid((unsigned)(in->n * strips));
        if (ob) hipLaunchKernelGGL(flatr_k7_kernel<true>, rgrid, dim3(256), FR_LDS, st, r);
        else hipLaunchKernelGGL(flatr_k7_kernel<false>, rgrid, dim3(256), FR_LDS, st, r);
        break;
      }
      static const int rr = getenv("BP_FLAT_R") ? atoi(getenv("BP_FLAT_R")) : 4;
      if (ob && rr == 8) fb_launch<7, 16, 8, true, true, 0, 8>(a, grid, st);
      else if (ob && rr == 16) fb_launch<7, 16, 8, true, true, 0, 16>(a, grid, st);
      else if (ob) fb_launch<7, 16, 8, true, true, 0>(a, grid, st);
      else fb_launch<7, 16, 8, true, false, 0>(a, grid, st);
      break;
    }
    case 2: if (ib) {
              static const int rr = getenv("BP_FLAT_R") ? atoi(getenv("BP_FLAT_R")) : 4;
              if (sr && rr == 8) fb_launch<7, 8, 16, true, true, 2, 8>(a, grid, st);
              else if (sr && rr == 16) fb_launch<7, 8, 16, true, true, 2, 16>(a, grid, st);
              else if (sr) fb_launch<7, 8, 16, true, true, 2>(a, grid, st);
              else fb_launch<7, 8, 16, true, true, 0>(a, grid, st);
            } else {
              if (sr) fb_launch<7, 8, 16, false, true, 2>(a, grid, st);
              else fb_launch<7, 8, 16, false, true, 0>(a, grid, st);
            }
            break;
    case 9: fb_launch<5, 8, 4, true, false, 0>(a, grid, st); break;
    case 3: if (sr) fb_launch<5, 4, 16, false, true, 1>(a, grid, st);
            else fb_launch<5, 4, 16, false, true, 0>(a, grid, st);
            break;
    case 4: fb_launch<5, 16, 4, true, false, 0>(a, grid, st); break;
    default: return BP_EUNSUPPORTED;
  }
  BP_CHECK_LAUNCH();
  if (!sr) return BP_OK;
  return bp_stats_rows_finish(a.stat, rows, g.cout_g, sr, st);
}
