// Flattened-K bf16 matrix-core kernel for the UNIT-STRIDE k = 7 layers of the generator heads in throughput mode
// (BASELINE.json configs[3] / [4]): p_mu_out.0 / p_var_out.0 forward (16 -> 8 channels at full resolution) and their
// data gradient (8 -> 16).  These two launches were the largest single-layer item of the bf16 step (2 x 0.87 ms,
// 0.2-0.3 PFLOP/s in igemm_bf16_kernel<16,1,1,4,6,...>: a 16-pixel x 8-channel tile per tap-row slab, weights
// re-streamed through LDS per tile, two barriers per tap row); at bf16 matrix speed the layer is 0.11 ms of MFMA and
// 0.17 ms of HBM.
//
// Scheme (the bf16 counterpart of conv_flat.hip):
//   * GEMM K = the FLATTENED (tap column, channel) index of one tap row -- 7 x CIN consecutive bf16 of an NHWC row --
//     cut into blocks of 32 (v_mfma_f32_16x16x32_bf16); the packed weights carry zeros past 7 x CIN, so a lane's
//     operand is always ONE 16-byte LDS read of 8 consecutive bf16 of the staged row, whatever the tap.
//   * MFMA rows = produced channels.  With 8 produced channels the 16 rows hold TWO output rows: row block rs uses
//     tap row ky - rs, so one input-row fragment feeds both (weight fragment "pair p" = [W[p] | W[p-1]]).
//   * A wave owns 16 pixels x 4 output rows per pass: an input-row fragment is read once and multiplied with up to
//     two (8 channels) / four (16 channels) weight fragments; ALL weight fragments (32 / 14) live in registers, loaded
//     once per workgroup -- no weight traffic, no barrier inside a tile.
//   * D = W x X: a lane ends up with 4 consecutive channels of one pixel -> 16-byte fp32 / 8-byte bf16 stores, 512
//     contiguous bytes per output row and instruction.
//   * Staging: every unit of the halo tile is loaded unconditionally from clamped coordinates, all loads of the tile
//     in flight at once, kept as raw words; the producer's pending batch-norm + (leaky) ReLU is applied and the value
//     rounded to bf16 on the way into LDS, zero padding after the activation (as torch pads the activated tensor).
#include "conv_bf16.hpp"

using namespace bpbf16;

namespace {

struct FbArgs {
  const void* in; int h, w, in_cs, in_co;
  void* out; int out_cs, out_co;
  const u16* wp;
  PW pw;
  int tiles_x, tiles_y, n;
};

constexpr int FB_K = 7, FB_PAD = 3;
constexpr int FB_TW = 64, FB_TH = 16, FB_R = 4;
constexpr int FB_LW = FB_TW + FB_K - 1, FB_LH = FB_TH + FB_K - 1;

template <int CIN> struct FbShape {
  static constexpr int KB = (FB_K * CIN + 31) / 32;
  static constexpr int ROWE = FB_LW * CIN;                    // bf16 per staged row
  static constexpr int SLACK = 32;                            // the last K block reads up to one pixel past a row
  static constexpr size_t LDS = ((size_t)FB_LH * ROWE + SLACK) * 2 + 3 * CIN * sizeof(float);
};

// XCD-aware tile order: workgroups go to the 8 XCDs round-robin by linear id; give each XCD a contiguous run of
// tiles so that the halos shared by neighbouring tiles are fetched into ONE L2.
__device__ __forceinline__ int fb_tile_of_block(int n) {
  const int L = blockIdx.x;
  const int q = n >> 3, r = n & 7;
  const int xcd = L & 7, idx = L >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int CIN, int COUT, bool IN_BF16, bool OUT_BF16>
__global__ __launch_bounds__(256, 3) void flatb_k7_kernel(FbArgs a) {
  using S = FbShape<CIN>;
  constexpr int KB = S::KB, ROWE = S::ROWE;
  constexpr int RS = 16 / COUT;                 // output rows per MFMA
  constexpr int NP = FB_K + RS - 1;             // weight fragments per K block ("pairs")
  constexpr int NS = FB_R / RS;                 // accumulator sets per pass
  constexpr int UPP = CIN / 8;                  // 8-channel staging units per pixel
  constexpr int NU = FB_LH * FB_LW * UPP;
  constexpr int SLOTS = (NU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) u16 smem_fb[];
  u16* lds = smem_fb;
  float* lpw = reinterpret_cast<float*>(lds + FB_LH * ROWE + S::SLACK);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kg = lane >> 4;

  const int per_img = a.tiles_x * a.tiles_y;
  const int t = fb_tile_of_block(per_img * a.n);
  const int n = t / per_img, tr = t - n * per_img;
  const int ty0 = (tr / a.tiles_x) * FB_TH, tx0 = (tr % a.tiles_x) * FB_TW;

  // ---- stage the halo tile: all loads first (raw words), then activation + bf16 + LDS
  const int64_t img = (int64_t)n * a.h * a.w * a.in_cs + a.in_co;
  RawUnit<8, IN_BF16> stage[SLOTS];
  int inside[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int e = tid + i * 256;
    const int pi = e / UPP, cu = e - pi * UPP;
    const int row = pi / FB_LW, px = pi - row * FB_LW;
    const int gy = ty0 - FB_PAD + row, gx = tx0 - FB_PAD + px;
    const bool ok = e < NU && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    inside[i] = e >= NU ? -1 : (ok ? 1 : 0);
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    load_unit_raw<8, IN_BF16>(a.in, img + ((int64_t)cy * a.w + cx) * a.in_cs + cu * 8, stage[i]);
  }
  const bool on = a.pw.scale != nullptr;
  if (on && tid < CIN) {
    lpw[tid] = a.pw.scale[tid]; lpw[CIN + tid] = a.pw.shift[tid]; lpw[2 * CIN + tid] = a.pw.slope[tid];
  }
  if (tid < S::SLACK / 8) *reinterpret_cast<uint4*>(lds + FB_LH * ROWE + tid * 8) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  {
    // (256 % UPP == 0: a thread always handles the same channel octet -- its activation parameters once, in registers)
    const int cu = tid % UPP;
    float sc[8], sf[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = on ? lpw[cu * 8 + j] : 1.f; sf[j] = on ? lpw[CIN + cu * 8 + j] : 0.f; sl[j] = on ? lpw[2 * CIN + cu * 8 + j] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      if (inside[i] < 0) continue;
      float raw[8], v[8];
      unpack_unit<8, IN_BF16>(stage[i], raw);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = raw[j];
        if (on) { x = fmaf(x, sc[j], sf[j]); x = x > 0.f ? x : x * sl[j]; }
        v[j] = inside[i] ? x : 0.f;
      }
      lds_store_unit<8>(lds + (tid + i * 256) * 8, v);
    }
  }
  // weights: registers, one 16-byte load per fragment and lane ([pair][K block][k octet][row][8] packed image).
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
  const int rs_l = (kg * 4) / COUT;             // which output row of an MFMA this lane's 4 accumulator rows hold
  const int co_l = (kg * 4) % COUT;
#pragma unroll 1
  for (int pass = 0; pass < FB_TH / FB_R; ++pass) {
    v4f acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = v4f{0.f, 0.f, 0.f, 0.f};
    const u16* base = lds + (pass * FB_R) * ROWE + fbase;
#pragma unroll
    for (int jr = 0; jr < FB_R + FB_K - 1; ++jr) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const bf8 xf = lds_frag<32>(base + jr * ROWE + kb * 32);
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
      if constexpr (OUT_BF16)
        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.out) + o) =
            make_uint2(pack2(acc[s][0], acc[s][1]), pack2(acc[s][2], acc[s][3]));
      else
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) =
            make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
    }
  }
}

struct FbPackArgs {
  const float* w; u16* dst;
  int64_t sa, sb;
  int cin, cout, KB, RS, NP, flip;
  int64_t total;
};

// [pair p][K block][k octet kg][row i][8]: row i = (output row select rs, produced channel co), tap row ky = p - rs,
// K index k = 32 kb + 8 kg + e -> (tap column kx = k / cin, gathered channel c = k % cin); zero where no such tap.
// flip: the gather is the data gradient of a convolution (transposed form): tap t reads weight element k - 1 - t.
__global__ __launch_bounds__(256) void flatb_pack_kernel(FbPackArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.total) return;
  int64_t r = i;
  const int e = r % 8; r /= 8;
  const int row = r % 16; r /= 16;
  const int kg = r % 4; r /= 4;
  const int kb = r % a.KB; r /= a.KB;
  const int p = (int)r;
  const int rs = row / a.cout, co = row % a.cout;
  const int ky = p - rs;
  const int k = kb * 32 + kg * 8 + e;
  const int kx = k / a.cin, c = k % a.cin;
  float v = 0.f;
  if (rs < a.RS && ky >= 0 && ky < FB_K && kx < FB_K) {
    const int wy = a.flip ? FB_K - 1 - ky : ky, wx = a.flip ? FB_K - 1 - kx : kx;
    v = a.w[c * a.sa + co * a.sb + wy * FB_K + wx];
  }
  a.dst[i] = f2bf(v);
}

static bool fb_shape(const ConvGeom& g, int* KB, int* RS, int* NP) {
  static const bool off = getenv("BP_BF16_NOFLAT") != nullptr;
  if (off) return false;
  if (g.k != FB_K || g.stride != 1 || g.pad != FB_PAD || g.nphase != 1 || g.taps != FB_K || g.IS != 1 || g.OS != 1)
    return false;
  if (!((g.cin_g == 16 && g.cout_g == 8) || (g.cin_g == 8 && g.cout_g == 16))) return false;
  *KB = (FB_K * g.cin_g + 31) / 32;
  *RS = 16 / g.cout_g;
  *NP = FB_K + *RS - 1;
  return true;
}

}  // namespace

// elements of the flattened-K weight image of this layer (0: the kernel does not apply)
int64_t bp_bf16_flat_packed_elems(const ConvGeom& g) {
  int KB, RS, NP;
  if (!fb_shape(g, &KB, &RS, &NP)) return 0;
  return (int64_t)NP * KB * 64 * 8;
}

int bp_bf16_flat_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, u16* dst, hipStream_t st) {
  FbPackArgs a{};
  if (!fb_shape(g, &a.KB, &a.RS, &a.NP)) return BP_EUNSUPPORTED;
  a.w = w_torch; a.dst = dst; a.sa = wm.sa; a.sb = wm.sb; a.cin = g.cin_g; a.cout = g.cout_g;
  a.flip = g.gather_transposed;
  a.total = bp_bf16_flat_packed_elems(g);
  hipLaunchKernelGGL(flatb_pack_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  BP_CHECK_LAUNCH();
  return BP_OK;
}

bool bp_bf16_flat_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, bool stats) {
  int KB, RS, NP;
  if (!fb_shape(g, &KB, &RS, &NP) || bias || stats || !in || !out) return false;
  if (in->c != g.cin_g || out->c != g.cout_g || in->h != out->h || in->w != out->w || in->n != out->n) return false;
  // the forward reads the bf16 trunk and writes the fp32 head; its data gradient reads fp32 and writes bf16
  const bool ib = in->dtype == BP_BF16, ob = out->dtype == BP_BF16;
  if (!((g.cin_g == 16 && ib && !ob) || (g.cin_g == 8 && !ib && ob))) return false;
  const int ie = ib ? 2 : 4, oe = ob ? 2 : 4;
  if ((in->cstride * ie) % 16 || (in->coff * ie) % 16 || reinterpret_cast<uintptr_t>(in->ptr) % 16) return false;
  if ((out->cstride * oe) % (4 * oe) || (out->coff * oe) % (4 * oe) || reinterpret_cast<uintptr_t>(out->ptr) % 16) return false;
  const int64_t tiles = (int64_t)bp_ceil_div(out->w, FB_TW) * bp_ceil_div(out->h, FB_TH) * out->n;
  return tiles < (1ll << 31);
}

int bp_bf16_flat_run(const ConvGeom& g, const bp_view* in, const PW& pw, const u16* packed_flat, const bp_view* out,
                     hipStream_t st) {
  FbArgs a{};
  a.in = in->ptr; a.h = in->h; a.w = in->w; a.in_cs = in->cstride; a.in_co = in->coff;
  a.out = out->ptr; a.out_cs = out->cstride; a.out_co = out->coff;
  a.wp = packed_flat; a.pw = pw; a.n = in->n;
  a.tiles_x = bp_ceil_div(out->w, FB_TW); a.tiles_y = bp_ceil_div(out->h, FB_TH);
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.n)), block(256);
  if (g.cin_g == 16) {
    auto k = flatb_k7_kernel<16, 8, true, false>;
    static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)FbShape<16>::LDS), 0);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, FbShape<16>::LDS, st, a);
  } else {
    auto k = flatb_k7_kernel<8, 16, false, true>;
    static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)FbShape<8>::LDS), 0);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, FbShape<8>::LDS, st, a);
  }
  BP_CHECK_LAUNCH();
  return BP_OK;
}
