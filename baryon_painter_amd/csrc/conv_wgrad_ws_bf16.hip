// Output-stationary bf16 weight gradient of the generator's residual trunk: Conv2d k3 s1 p1, 128 <-> 128 channels, bf16
// activations and gradients, fp32 accumulation (BASELINE.json configs[3]; /root/reference/baryon_painter/models/utils.py:22-38).
//
// wgrad_bf16_kernel<3,3,1,2,2,2,2,4> runs these eight layers at 0.25 of the bf16 matrix peak (profiles/r03_mfma_util_bf16.txt):
// a 4 x 32 pixel tile per staging round -- 144 MFMAs of work per wave between two barriers, both tensors re-staged per tile.
// Here, as in conv_wgrad_ws_f32.hip, the OUTPUT stays put: a workgroup owns one (64 ci x 64 co) block of dW for all nine
// taps (wave (wi, wo): 32 x 32 channels = nine accumulator tiles of v_mfma_f32_32x32x16_bf16) and walks an image top to
// bottom with a ring of eight X rows and four dY rows in LDS, staged TWO rows ahead: the fragments of a row's first k-step
// are requested before the barrier that ends the previous row, so the matrix pipe never waits for a row boundary.
// Both operands are "8 pixels of one channel" while the tensors are channel-major: LDS images [16-channel tile][row][pixel][16]
// read through ds_read_b64_tr_b16 (two reads per fragment; the images of neighbouring tiles are 128 bytes apart modulo the
// 256-byte bank row, so the two 16-lane groups of a 32-lane half -- same pixels, neighbouring tiles -- never share a bank).
// Per k-step of 16 pixels a wave reads nine X fragments + one dY fragment for nine 32 x 32 x 16 MFMAs: with the 16 x 16 x 32
// shape the same work is 36 MFMAs and 40 fragment reads per 32 pixels, and the wave's in-order issue stream (an MFMA
// holds the issue port for 8 cycles, every LDS read for 4) is what bounded it: 72 MFMAs = 1152 matrix cycles per row
// against ~1100 issue cycles.
// The two workgroups that gather the same X half (and the two that gather the same dY half) get block ids that differ by a
// multiple of 8: the dispatcher is observed to place such blocks on one XCD, where the second reader hits the L2 (speed
// only: nothing depends on it).
#include "conv_bf16.hpp"
#include <type_traits>

namespace {
using namespace bpbf16;

constexpr int WB_C = 128, WB_B = 64, WB_RX = 8, WB_RY = 4;

struct WoArgs {
  const u16* X; int x_cs, x_co;
  const u16* Y; int y_cs, y_co;
  int n, h;
  PW pwx;
  float* ws;                         // [split][tap 9][co 128][ci 128]
  int BR, bands, nsplit;
};

template <int G> struct WoGeom {
  static constexpr int W = 16 * G, RPX = W + 2;
  // bf16 elements of one channel tile's image, [ring row][pixel][16], + 64: neighbouring tiles 128 B apart mod 256 B
  static constexpr int XT = WB_RX * RPX * 16 + 64;
  static constexpr int YT = WB_RY * W * 16 + 64;
  static constexpr size_t lds_bytes = (size_t)(4 * XT + 4 * YT) * 2 + 3 * WB_B * sizeof(float);
};

__device__ __forceinline__ s4 wo_tr(const u16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf8 wo_frag(s4 a, s4 b) {
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf8, v);
}

template <int G, bool ACT>
__global__ __launch_bounds__(256) void wgrad_ws_bf16_kernel(WoArgs a) {
  using GM = WoGeom<G>;
  constexpr int W = GM::W, RPX = GM::RPX, XT = GM::XT, YT = GM::YT;
  constexpr int KS = W / 32;                            // k-steps per row
  constexpr int NU = W / 32;                            // 16-byte staging units per thread, row and tensor: W * 8 / 256
  extern __shared__ __attribute__((aligned(16))) u16 smem_o[];
  u16* xs = smem_o;                                     // [tile 4][ring 8][RPX][16]
  u16* ys = smem_o + 4 * XT;                            // [tile 4][ring 4][W][16]
  float* lpw = reinterpret_cast<float*>(smem_o + 4 * XT + 4 * YT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wo = wave & 1;
  const int li = lane & 15, kq = lane >> 4;
  // block id = split_lo + 8 * (pair + 4 * split_hi): the four channel-block pairs of a split share an XCD
  const int split = (int)(blockIdx.x & 7) + 8 * (int)(blockIdx.x >> 5), pair = (blockIdx.x >> 3) & 3;
  if (split >= a.nsplit) return;                        // (uniform per block; the grid is padded to a multiple of 32)
  const int cib = pair & 1, cob = pair >> 1;
  const int n = split / a.bands, band = split % a.bands;
  const int y0 = band * a.BR;
  const int y1 = min(y0 + a.BR, a.h);

  // staging: unit i of a row: lane bits [0] half of a 16-channel tile, [1:2] pixel & 3, [3:4] tile, [5] pixel bit 2; wave and
  // i the pixel's higher bits -- eight lanes write 4 pixels x 32 B = 128 contiguous bytes of one tile image
  const int s_half = lane & 1, s_tile = (lane >> 3) & 3;
  const int s_px = ((lane >> 1) & 3) + 4 * (lane >> 5) + 8 * wave;              // + 32 i
  const int s_oct = 2 * s_tile + s_half;
  const char* x_img = reinterpret_cast<const char*>(a.X + (int64_t)n * a.h * W * a.x_cs + a.x_co + WB_B * cib);
  const char* y_img = reinterpret_cast<const char*>(a.Y + (int64_t)n * a.h * W * a.y_cs + a.y_co + WB_B * cob);
  const unsigned x_row = (unsigned)(W * a.x_cs) * 2u, y_row = (unsigned)(W * a.y_cs) * 2u;
  const unsigned x_off = (unsigned)(s_px * a.x_cs + 8 * s_oct) * 2u, x_st = (unsigned)(32 * a.x_cs) * 2u;
  const unsigned y_off = (unsigned)(s_px * a.y_cs + 8 * s_oct) * 2u, y_st = (unsigned)(32 * a.y_cs) * 2u;
  // (a row's units as NAMED members: arrays carried from one k-step's code to the next ended up in scratch memory)
  struct Raw { uint4 u0, u1; };
  auto load_x = [&](int r) {
    const char* rowp = x_img + (size_t)((unsigned)r * x_row);
    Raw raw;
    raw.u0 = *reinterpret_cast<const uint4*>(rowp + x_off);
    raw.u1 = NU > 1 ? *reinterpret_cast<const uint4*>(rowp + (x_off + x_st)) : make_uint4(0u, 0u, 0u, 0u);
    return raw;
  };
  auto load_y = [&](int r) {
    const char* rowp = y_img + (size_t)((unsigned)r * y_row);
    Raw raw;
    raw.u0 = *reinterpret_cast<const uint4*>(rowp + y_off);
    raw.u1 = NU > 1 ? *reinterpret_cast<const uint4*>(rowp + (y_off + y_st)) : make_uint4(0u, 0u, 0u, 0u);
    return raw;
  };
  auto act8 = [&](uint4 v) {
    if constexpr (ACT) {
      const unsigned wd[4] = {v.x, v.y, v.z, v.w};
      unsigned o[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 sc = *reinterpret_cast<const float4*>(lpw + 8 * s_oct + 4 * h);
        const float4 sf = *reinterpret_cast<const float4*>(lpw + WB_B + 8 * s_oct + 4 * h);
        const float4 sl = *reinterpret_cast<const float4*>(lpw + 2 * WB_B + 8 * s_oct + 4 * h);
        float t0 = fmaf(bf2f((u16)(wd[2 * h] & 0xffffu)), sc.x, sf.x), t1 = fmaf(bf2f((u16)(wd[2 * h] >> 16)), sc.y, sf.y);
        float t2 = fmaf(bf2f((u16)(wd[2 * h + 1] & 0xffffu)), sc.z, sf.z), t3 = fmaf(bf2f((u16)(wd[2 * h + 1] >> 16)), sc.w, sf.w);
        t0 = t0 > 0.f ? t0 : t0 * sl.x; t1 = t1 > 0.f ? t1 : t1 * sl.y;           // (a NaN stays a NaN, as torch.relu)
        t2 = t2 > 0.f ? t2 : t2 * sl.z; t3 = t3 > 0.f ? t3 : t3 * sl.w;
        o[2 * h] = pack2(t0, t1); o[2 * h + 1] = pack2(t2, t3);
      }
      v = make_uint4(o[0], o[1], o[2], o[3]);
    }
    return v;
  };
  auto put_x = [&](int r, unsigned keep, const Raw raw) {              // X row r -> ring slot (r + 1) & 7, columns 1 .. W
    const int rr = (r + 1) & (WB_RX - 1);
    u16* base = xs + s_tile * XT + ((rr * RPX + 1 + s_px) * 16 + 8 * s_half);
    const uint4 v = act8(raw.u0);
    *reinterpret_cast<uint4*>(base) = make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
    if constexpr (NU > 1) {
      const uint4 w = act8(raw.u1);
      *reinterpret_cast<uint4*>(base + 32 * 16) = make_uint4(w.x & keep, w.y & keep, w.z & keep, w.w & keep);
    }
  };
  auto put_y = [&](int r, const Raw raw) {                             // dY row r -> ring slot r & 3
    const int rr = r & (WB_RY - 1);
    u16* base = ys + s_tile * YT + ((rr * W + s_px) * 16 + 8 * s_half);
    *reinterpret_cast<uint4*>(base) = raw.u0;
    if constexpr (NU > 1) *reinterpret_cast<uint4*>(base + 32 * 16) = raw.u1;
  };

  // ---- prologue: X rows y0 - 1 .. y0 + 2, dY rows y0, y0 + 1
  Raw rx, ry;
  if (tid < 128) {                                      // zero columns 0 and W + 1: [tile 4][ring 8][side 2] x two 16-byte halves
    const int t = tid >> 5, rr = (tid >> 2) & 7, side = (tid >> 1) & 1, hf = tid & 1;
    *reinterpret_cast<uint4*>(xs + t * XT + ((rr * RPX + (side ? RPX - 1 : 0)) * 16 + 8 * hf)) = make_uint4(0u, 0u, 0u, 0u);
  }
  if constexpr (ACT) {
    for (int i = tid; i < WB_B; i += 256) {
      lpw[i] = a.pwx.scale[WB_B * cib + i]; lpw[WB_B + i] = a.pwx.shift[WB_B * cib + i]; lpw[2 * WB_B + i] = a.pwx.slope[WB_B * cib + i];
    }
    __syncthreads();
  }
  {
    const bool i0 = y0 - 1 >= 0, i2 = y0 + 1 < a.h, i3 = y0 + 2 < a.h;
    rx = load_x(i0 ? y0 - 1 : y0);
    ry = load_x(y0);
    put_x(y0 - 1, i0 ? 0xffffffffu : 0u, rx);
    put_x(y0, 0xffffffffu, ry);
    rx = load_x(i2 ? y0 + 1 : y0);
    ry = load_x(i3 ? y0 + 2 : y0);
    put_x(y0 + 1, i2 ? 0xffffffffu : 0u, rx);
    put_x(y0 + 2, i3 ? 0xffffffffu : 0u, ry);
    rx = load_y(y0);
    ry = load_y(i2 ? y0 + 1 : y0);
    put_y(y0, rx);
    put_y(y0 + 1, ry);
  }
  __syncthreads();

  typedef float v16f __attribute__((ext_vector_type(16)));
  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // v_mfma_f32_32x32x16_bf16 operands: lane l = (c = l & 31, h = l >> 5) holds A[row c][k = 8 h + j] / B[k = 8 h + j][col c].
  // The transposing read works per 16-lane group: group q = l >> 4 reads channel tile q & 1 (rows 16 (q & 1) + i), pixels
  // 8 h + 0..3 (first read) and 8 h + 4..7 (second); lane i of a group addresses pixel (i >> 2), channels 4 (i & 3).
  const int q = lane >> 4, h = lane >> 5;
  const int trl = (8 * h + (li >> 2)) * 16 + 4 * (li & 3);
  const u16* xw = xs + (2 * wi + (q & 1)) * XT + trl;
  const u16* yw = ys + (2 * wo + (q & 1)) * YT + trl;
  constexpr int KS16 = W / 16;                          // k-steps of 16 pixels per row
  bf8 xf[9], yf;
  auto frag_at = [&](const u16* p) { return wo_frag(wo_tr(p), wo_tr(p + 4 * 16)); };
  // Global loads run THREE rows ahead of their commit (three register sets, rotated by unrolling the row loop three
  // times: moving a set into another would wait for its loads on the spot).  Row y: at its top it requests X row y + 5 and
  // dY row y + 4 into the youngest set; at its end it commits the oldest (X row y + 3, dY row y + 2: two rows ahead of their
  // first reader, into slots nobody reads any more) and passes the row's barrier.
  // Inside a row everything is static: the k-steps are unrolled, so a fragment address is one of six per-row base registers
  // (X rows y - 1 .. y + 2, dY rows y, y + 1) plus an immediate -- measured with the k-step as a run-time loop: 151 scalar and
  // 88 vector address instructions per row beside 36 MFMAs and 84 LDS instructions, the wave's in-order issue stream 58 %
  // busy and the matrix pipe 35 %.  Fragment pipeline at TAP granularity: a tap's X fragment is re-requested for the NEXT
  // k-step right behind the MFMA that consumed it (eight MFMAs of cover); the next k-step is the same row's, or k-step 0 of
  // row y + 1 (resident since the barrier that ended row y - 1).
  Raw sx[3], sy[3];
  auto ld_x = [&](int r) { return load_x(r < a.h ? r : a.h - 1); };
  auto ld_y = [&](int r) { return load_y(r < a.h ? r : a.h - 1); };
  sx[0] = ld_x(y0 + 3); sy[0] = ld_y(y0 + 2);
  sx[1] = ld_x(y0 + 4); sy[1] = ld_y(y0 + 3);
  sx[2] = sx[0]; sy[2] = sy[0];
  {
    const u16* xb0 = xw + ((y0 & (WB_RX - 1)) * RPX) * 16;
    yf = frag_at(yw + ((y0 & (WB_RY - 1)) * W) * 16);
#pragma unroll
    for (int t = 0; t < 9; ++t) xf[t] = frag_at(xw + ((((y0 + t / 3) & (WB_RX - 1)) * RPX) + t % 3) * 16);
    (void)xb0;
  }
  auto row = [&](auto S_, int y) {
    constexpr int S = decltype(S_)::value;
    sx[(S + 2) % 3] = ld_x(y + 5);
    sy[(S + 2) % 3] = ld_y(y + 4);
    const u16* xb[4];                                   // X rows y - 1 + ty, ty = 0 .. 3 (3: row y + 1's third tap row)
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) xb[ty] = xw + (((y + ty) & (WB_RX - 1)) * RPX) * 16;
    const u16* yb[2] = {yw + ((y & (WB_RY - 1)) * W) * 16, yw + (((y + 1) & (WB_RY - 1)) * W) * 16};
#pragma unroll
    for (int ks = 0; ks < KS16; ++ks) {
      const bool last = ks == KS16 - 1;
      const int kn = last ? 0 : ks + 1;
      const bf8 yfn = frag_at(yb[last ? 1 : 0] + 16 * kn * 16);
      // the oldest set goes to LDS in the middle of the row (its loads were issued three rows ago): one unit's activation
      // per k-step, its ~50 vector instructions dealt between the MFMAs by the group pattern below (an MFMA holds the issue
      // port for 8 of its 32 cycles: six others fit beside it)
      if (ks == 1 % KS16) put_x(y + 3, y + 3 < a.h ? 0xffffffffu : 0u, sx[S]);
      if (ks == 2 % KS16) put_y(y + 2, sy[S]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[t], yf, acc[t], 0, 0, 0);
        xf[t] = frag_at(xb[t / 3 + (last ? 1 : 0)] + (16 * kn + t % 3) * 16);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      yf = yfn;
    }
    // the barrier publishes this row's LDS writes; they are older than the last k-step's 20 fragment reads (LDS operations
    // retire in order), so the wave need not drain those: wait until at most 8 are outstanding
    // (only where a whole k-step -- fenced by sched_barrier -- lies behind the writes: W = 64)
    if constexpr (KS16 >= 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (int y = y0; y < y1; y += 3) {
    row(std::integral_constant<int, 0>{}, y);
    if (y + 1 < y1) row(std::integral_constant<int, 1>{}, y + 1);
    if (y + 2 < y1) row(std::integral_constant<int, 2>{}, y + 2);
  }

  // ---- partial sums: D[row 8 g + 4 h + r][col c] of tap t (register 4 g + r) = dW[t][co 64 cob + 32 wo + c][ci 64 cib + 32 wi + 8 g + 4 h + r]
  const int c = lane & 31;
  float* dst = a.ws + ((int64_t)split * 9 * WB_C + (WB_B * cob + 32 * wo + c)) * WB_C + WB_B * cib + 32 * wi + 4 * h;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + (int64_t)t * WB_C * WB_C + 8 * g) =
          make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
}

bool wo_enabled() {
  static const bool off = getenv("BP_BF16_WGRAD_WS") && atoi(getenv("BP_BF16_WGRAD_WS")) == 0;
  return !off;
}
int g_wo_override = -1;

int wo_G(int w) { return w == 64 ? 4 : w == 32 ? 2 : 0; }

void wo_bands(int n, int h, int* BR, int* bands) {      // ~256 workgroups = 64 splits x 4 channel-block pairs
  int br = h;
  while (br > 8 && (int64_t)n * bp_ceil_div(h, br) < 64) br = bp_ceil_div(br, 2);
  *BR = br;
  *bands = bp_ceil_div(h, br);
}

template <int G, bool ACT>
int wo_launch(const WoArgs& a, unsigned grid, hipStream_t st) {
  static const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ws_bf16_kernel<G, ACT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)WoGeom<G>::lds_bytes);
  if (optin != hipSuccess) return BP_ELAUNCH;
  hipLaunchKernelGGL((wgrad_ws_bf16_kernel<G, ACT>), dim3(grid), dim3(256), WoGeom<G>::lds_bytes, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

}  // namespace

void bp_bf16_wgrad_ws_set(int v) { g_wo_override = v; }

// Same contract as wb_launch (conv_wgrad_bf16.hip): BP_EUNSUPPORTED -> the tiled kernel.
int bp_wgrad_ws_bf16(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* ws,
                     size_t ws_bytes, size_t* need, int* nsplit, int* cxp, int* cyp, hipStream_t st, bool dry) {
  if (!(g_wo_override < 0 ? wo_enabled() : g_wo_override != 0)) return BP_EUNSUPPORTED;
  if (cv->transposed || cv->k != 3 || cv->stride != 1 || cv->pad != 1 || cv->cin != WB_C || cv->cout != WB_C) return BP_EUNSUPPORTED;
  if (X->dtype != BP_BF16 || Y->dtype != BP_BF16 || X->c != WB_C || Y->c != WB_C || pwy.scale) return BP_EUNSUPPORTED;
  if (X->n != Y->n || X->h != Y->h || X->w != Y->w || !wo_G(X->w)) return BP_EUNSUPPORTED;
  if (X->cstride % 8 || X->coff % 8 || reinterpret_cast<uintptr_t>(X->ptr) % 16) return BP_EUNSUPPORTED;
  if (Y->cstride % 8 || Y->coff % 8 || reinterpret_cast<uintptr_t>(Y->ptr) % 16) return BP_EUNSUPPORTED;
  if ((int64_t)X->h * X->w * X->cstride * 2 >= (int64_t)1 << 31 || (int64_t)Y->h * Y->w * Y->cstride * 2 >= (int64_t)1 << 31)
    return BP_EUNSUPPORTED;
  WoArgs a{};
  wo_bands(X->n, X->h, &a.BR, &a.bands);
  const int64_t splits = (int64_t)X->n * a.bands;
  if (splits * 4 + 32 > 0x7fffffff) return BP_EUNSUPPORTED;
  *nsplit = (int)splits; *cxp = WB_C; *cyp = WB_C;
  *need = (size_t)splits * 9 * WB_C * WB_C * sizeof(float);
  if (dry) return BP_OK;
  if (!ws || ws_bytes < *need) return BP_EWORKSPACE;
  a.X = reinterpret_cast<const u16*>(X->ptr); a.x_cs = X->cstride; a.x_co = X->coff;
  a.Y = reinterpret_cast<const u16*>(Y->ptr); a.y_cs = Y->cstride; a.y_co = Y->coff;
  a.n = X->n; a.h = X->h; a.pwx = pwx; a.ws = ws; a.nsplit = (int)splits;
  const bool act = pwx.scale != nullptr;
  const unsigned grid = (unsigned)(bp_ceil_div((int)splits, 8) * 32);          // (split_lo 8) x (pair 4) x split_hi
  if (wo_G(X->w) == 4) return act ? wo_launch<4, true>(a, grid, st) : wo_launch<4, false>(a, grid, st);
  return act ? wo_launch<2, true>(a, grid, st) : wo_launch<2, false>(a, grid, st);
}
