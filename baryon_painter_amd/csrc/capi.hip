// C ABI of the convolution entry points (include/bp_hip.h): shape validation on the host, then
// dispatch to the MFMA kernels (conv_igemm.hip / conv_wgrad.hip) or the direct ones.
#include "common.hpp"
#include <cstring>

int64_t bp_igemm_packed_floats(const ConvGeom& g);
int bp_igemm_kernel_id(const ConvGeom& g);
int bp_igemm_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, hipStream_t st);
size_t bp_igemm_pack_job_bytes();
int bp_igemm_pack_job(const ConvGeom& g, const WeightMap& wm, const float* w_torch, float* packed, void* job,
                      int64_t* nblocks);
int bp_igemm_pack_jobs(const void* jobs_dev, const int64_t* first_block_dev, int njobs, int64_t total_blocks,
                       hipStream_t st);
int bp_igemm_run(const ConvGeom& g, const bp_view* in, const PW& pw, const float* packed, const float* bias,
                 const bp_view* out, hipStream_t st, const IgemmStatsReq* stats = nullptr);
size_t bp_igemm_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode);
int bp_direct_gather(const ConvGeom& g, const WeightMap& wm, const bp_view* in, const PW& pw, const float* w_torch,
                     const float* bias, const bp_view* out, hipStream_t st);
int bp_direct_wgrad(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst,
                    hipStream_t st);
size_t bp_wgrad_mfma_workspace(const bp_conv* cv, const bp_view* X, const bp_view* Y);
int bp_wgrad_mfma(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst,
                  void* workspace, size_t workspace_bytes, hipStream_t st, bool shared);
extern "C" int bp_sums_to_float(const double* sums, int32_t c, float* dst, void* stream);
// conv_bf16.hip
bool bp_bf16_igemm_ok(const ConvGeom& g, const bp_view* in, const bp_view* out);
int64_t bp_bf16_packed_elems(const ConvGeom& g);
int bp_bf16_pack(const ConvGeom& g, const WeightMap& wm, const float* w_torch, void* packed, hipStream_t st);
int bp_bf16_igemm_run(const ConvGeom& g, const bp_view* in, const PW& pw, const void* packed, const float* bias,
                      const bp_view* out, hipStream_t st, const IgemmStatsReq* stats = nullptr);
size_t bp_bf16_stats_workspace(const ConvGeom& g, const bp_view* in, const bp_view* out, int mode);
size_t bp_wgrad_bf16_workspace(const bp_conv* cv, const bp_view* X, const bp_view* Y);
int bp_wgrad_bf16_run(const bp_conv* cv, const bp_view* X, const PW& pwx, const bp_view* Y, const PW& pwy, float* dst,
                      void* workspace, size_t workspace_bytes, hipStream_t st);

void bp_f32_ws_set(int v);        // conv_ws_f32.hip
void bp_f32_wgrad_ws_set(int v);  // conv_wgrad_ws_f32.hip
void bp_bf16_wgrad_ws_set(int v); // conv_wgrad_ws_bf16.hip
bool bp_f32_ws_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int stats_mode);
void bp_bf16_ws_set(int v);       // conv_bf16_ws.hip
int bp_bf16_ws_kind(const ConvGeom& g);
bool bp_bf16_ws_ok(const ConvGeom& g, const bp_view* in, const bp_view* out, const float* bias, int mode);

namespace {

bool conv_ok(const bp_conv* cv) {
  return cv && cv->cin > 0 && cv->cout > 0 && cv->k > 0 && cv->stride > 0 && cv->pad >= 0 && cv->out_pad >= 0 &&
         (cv->transposed == 0 || cv->transposed == 1) && (cv->transposed || cv->out_pad == 0) &&
         cv->out_pad < cv->stride + (cv->stride == 1);
}

// x: module input, y: module output (fp32 views unless `any`)
bool shapes_ok(const bp_conv* cv, const bp_view* x, const bp_view* y, bool any = false) {
  if (any ? (!bp_view_ok_any(x) || !bp_view_ok_any(y)) : (!bp_view_ok(x) || !bp_view_ok(y))) return false;
  if (x->n != y->n || x->c != cv->cin || y->c != cv->cout) return false;
  return y->h == bp_conv_out_extent(cv, x->h) && y->w == bp_conv_out_extent(cv, x->w) && y->h > 0 && y->w > 0;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

void bp_wgrad_private_ws(bool on);
struct WgradPrivateWs {
  explicit WgradPrivateWs(bool on) { bp_wgrad_private_ws(on); }
  ~WgradPrivateWs() { bp_wgrad_private_ws(false); }
};
int bp_wgrad_defer_begin_impl();
int bp_wgrad_defer_flush_impl(hipStream_t st, int end);

extern "C" {

int bp_version(void) { return 100; }

int bp_set_option(const char* name, int value) {
  if (!name) return BP_EINVAL;
  if (!strcmp(name, "bf16_ws")) { bp_bf16_ws_set(value); return BP_OK; }
  if (!strcmp(name, "f32_ws")) { bp_f32_ws_set(value); return BP_OK; }
  if (!strcmp(name, "f32_wgrad_ws")) { bp_f32_wgrad_ws_set(value); return BP_OK; }
  if (!strcmp(name, "bf16_wgrad_ws")) { bp_bf16_wgrad_ws_set(value); return BP_OK; }
  return BP_EUNSUPPORTED;
}

const char* bp_strerror(int code) {
  switch (code) {
    case BP_OK: return "ok";
    case BP_EINVAL: return "invalid argument (null pointer or inconsistent shapes)";
    case BP_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case BP_ELAUNCH: return "kernel launch failed";
    case BP_EWORKSPACE: return "workspace missing or too small";
  }
  return "unknown error";
}

int64_t bp_conv_packed_floats(const bp_conv* cv, int dir) {
  if (!conv_ok(cv) || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return -1;
  const ConvGeom g = dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv);
  return bp_igemm_packed_floats(g);
}

int bp_conv_kernel_id(const bp_conv* cv, int dir) {
  if (!conv_ok(cv) || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return -1;
  return bp_igemm_kernel_id(dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv));
}

int bp_conv_pack(const bp_conv* cv, int dir, const float* w_torch, float* packed, void* stream) {
  if (!conv_ok(cv) || !w_torch || !packed || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return BP_EINVAL;
  const ConvGeom g = dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv);
  return bp_igemm_pack(g, bp_wmap(cv, dir), w_torch, packed, bp_stream(stream));
}

int32_t bp_conv_pack_job_bytes(void) { return (int32_t)bp_igemm_pack_job_bytes(); }

int bp_conv_pack_job(const bp_conv* cv, int dir, const float* w_torch, float* packed, void* job, int64_t* nblocks) {
  if (!conv_ok(cv) || !w_torch || !packed || !job || !nblocks || (dir != BP_PACK_FWD && dir != BP_PACK_BWD))
    return BP_EINVAL;
  const ConvGeom g = dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv);
  return bp_igemm_pack_job(g, bp_wmap(cv, dir), w_torch, packed, job, nblocks);
}

int bp_conv_pack_jobs(const void* jobs_dev, const int64_t* first_block_dev, int32_t njobs, int64_t total_blocks,
                      void* stream) {
  if (!jobs_dev || !first_block_dev || njobs < 0 || total_blocks < 0) return BP_EINVAL;
  return bp_igemm_pack_jobs(jobs_dev, first_block_dev, njobs, total_blocks, bp_stream(stream));
}

int64_t bp_conv_bf16_packed_elems(const bp_conv* cv, int dir) {
  if (!conv_ok(cv) || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return -1;
  return bp_bf16_packed_elems(dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv));
}

int bp_conv_bf16_pack(const bp_conv* cv, int dir, const float* w_torch, void* packed, void* stream) {
  if (!conv_ok(cv) || !w_torch || !packed || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return BP_EINVAL;
  const ConvGeom g = dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv);
  return bp_bf16_pack(g, bp_wmap(cv, dir), w_torch, packed, bp_stream(stream));
}

int bp_conv_ws_kind(const bp_conv* cv, int dir, const bp_view* in, const bp_view* out) {
  if (!conv_ok(cv) || (dir != BP_PACK_FWD && dir != BP_PACK_BWD) || !bp_view_ok_any(in) || !bp_view_ok_any(out)) return 0;
  const ConvGeom g = dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv);
  if (in->dtype == BP_F32 && out->dtype == BP_F32) return bp_f32_ws_ok(g, in, out, nullptr, 0) ? 3 : 0;
  return bp_bf16_ws_ok(g, in, out, nullptr, 0) ? bp_bf16_ws_kind(g) : 0;
}

int bp_conv_bf16_supported(const bp_conv* cv, int dir, const bp_view* in, const bp_view* out) {
  if (!conv_ok(cv) || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return 0;
  if ((in && !bp_view_ok_any(in)) || (out && !bp_view_ok_any(out))) return 0;
  return bp_bf16_igemm_ok(dir == BP_PACK_FWD ? bp_geom_forward(cv) : bp_geom_backward_data(cv), in, out) ? 1 : 0;
}

int bp_conv_forward(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw, const float* packed_fwd,
                    const float* w_torch, const float* bias, const bp_view* y, int impl, void* stream) {
  if (!conv_ok(cv) || !shapes_ok(cv, x, y, impl == BP_IMPL_BF16)) return BP_EINVAL;
  const ConvGeom g = bp_geom_forward(cv);
  if (impl == BP_IMPL_BF16) {
    if (!packed_fwd) return BP_EINVAL;
    return bp_bf16_igemm_run(g, x, bp_pw(x_pw), packed_fwd, bias, y, bp_stream(stream));
  }
  if (impl == BP_IMPL_AUTO && packed_fwd) {
    // the register-resident kernels (conv_flat.hip) want 16-byte aligned produced views: AUTO takes the direct kernel
    // for the odd view instead of refusing it (BP_IMPL_MFMA, asked for by name, still reports BP_EUNSUPPORTED)
    const int rc = bp_igemm_run(g, x, bp_pw(x_pw), packed_fwd, bias, y, bp_stream(stream));
    if (rc != BP_EUNSUPPORTED || !w_torch) return rc;
    impl = BP_IMPL_DIRECT;
  }
  if (impl == BP_IMPL_AUTO) impl = BP_IMPL_DIRECT;
  if (impl == BP_IMPL_MFMA) {
    if (!packed_fwd) return BP_EINVAL;
    return bp_igemm_run(g, x, bp_pw(x_pw), packed_fwd, bias, y, bp_stream(stream));
  }
  if (!w_torch) return BP_EINVAL;
  return bp_direct_gather(g, bp_wmap(cv, BP_PACK_FWD), x, bp_pw(x_pw), w_torch, bias, y, bp_stream(stream));
}

size_t bp_conv_stats_workspace(const bp_conv* cv, int dir, const bp_view* x, const bp_view* y, int impl) {
  if (!conv_ok(cv) || !shapes_ok(cv, x, y, impl == BP_IMPL_BF16) || (dir != BP_PACK_FWD && dir != BP_PACK_BWD)) return 0;
  if (impl == BP_IMPL_BF16)
    return dir == BP_PACK_FWD ? bp_bf16_stats_workspace(bp_geom_forward(cv), x, y, 1)
                              : bp_bf16_stats_workspace(bp_geom_backward_data(cv), y, x, 2);
  return dir == BP_PACK_FWD ? bp_igemm_stats_workspace(bp_geom_forward(cv), x, y, 1)
                            : bp_igemm_stats_workspace(bp_geom_backward_data(cv), y, x, 2);
}

int bp_conv_forward_stats(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw, const float* packed_fwd,
                          const bp_view* y, double* sums, void* workspace, size_t workspace_bytes, int impl,
                          void* stream) {
  if (!conv_ok(cv) || !shapes_ok(cv, x, y, impl == BP_IMPL_BF16) || !packed_fwd || !sums) return BP_EINVAL;
  const IgemmStatsReq sr{1, nullptr, PW{nullptr, nullptr, nullptr}, sums, workspace, workspace_bytes};
  if (impl == BP_IMPL_BF16)
    return bp_bf16_igemm_run(bp_geom_forward(cv), x, bp_pw(x_pw), packed_fwd, nullptr, y, bp_stream(stream), &sr);
  return bp_igemm_run(bp_geom_forward(cv), x, bp_pw(x_pw), packed_fwd, nullptr, y, bp_stream(stream), &sr);
}

int bp_conv_forward_bn(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw, const float* packed_fwd,
                       const bp_view* y, double* sums, const bp_bn_train* bn, void* workspace, size_t workspace_bytes,
                       int impl, void* stream) {
  if (!conv_ok(cv) || !shapes_ok(cv, x, y, impl == BP_IMPL_BF16) || !packed_fwd || !sums || !bn || bn->count <= 0 ||
      !bn->scale || !bn->shift)
    return BP_EINVAL;
  const BnFin fin{bn->count, bn->gamma, bn->beta, bn->eps, bn->momentum, bn->running_mean, bn->running_var,
                  bn->num_batches_tracked, bn->scale, bn->shift, bn->save_mean, bn->save_invstd};
  const IgemmStatsReq sr{1, nullptr, PW{nullptr, nullptr, nullptr}, sums, workspace, workspace_bytes, &fin};
  if (impl == BP_IMPL_BF16)
    return bp_bf16_igemm_run(bp_geom_forward(cv), x, bp_pw(x_pw), packed_fwd, nullptr, y, bp_stream(stream), &sr);
  return bp_igemm_run(bp_geom_forward(cv), x, bp_pw(x_pw), packed_fwd, nullptr, y, bp_stream(stream), &sr);
}

int bp_conv_backward_data_stats(const bp_conv* cv, const bp_view* dy, const float* packed_bwd, const bp_view* dx,
                                const bp_view* x_raw, const bp_pointwise* x_pw, double* sums, void* workspace,
                                size_t workspace_bytes, void* stream) {
  const bool b16 = (dy && dy->dtype == BP_BF16) || (dx && dx->dtype == BP_BF16);        // the bf16 kernels' epilogue
  if (!conv_ok(cv) || !shapes_ok(cv, dx, dy, b16) || !packed_bwd || !sums || !bp_view_ok_any(x_raw)) return BP_EINVAL;
  if (x_raw->n != dx->n || x_raw->h != dx->h || x_raw->w != dx->w || x_raw->c != dx->c) return BP_EINVAL;
  const IgemmStatsReq sr{2, x_raw, bp_pw(x_pw), sums, workspace, workspace_bytes};
  if (b16)
    return bp_bf16_igemm_run(bp_geom_backward_data(cv), dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, dx,
                             bp_stream(stream), &sr);
  if (!bp_view_ok(x_raw)) return BP_EINVAL;
  return bp_igemm_run(bp_geom_backward_data(cv), dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, dx,
                      bp_stream(stream), &sr);
}

size_t bp_conv_backward_data_act_workspace(const bp_conv* cv, const bp_view* dy, const bp_view* g) {
  const bool b16 = g && g->dtype == BP_BF16;              // the bf16 kernel's epilogue (conv_bf16_head.hip)
  if (!conv_ok(cv) || !shapes_ok(cv, g, dy, b16)) return 0;
  if (b16) return bp_bf16_stats_workspace(bp_geom_backward_data(cv), dy, g, 3);
  return bp_igemm_stats_workspace(bp_geom_backward_data(cv), dy, g, 3);
}

int bp_conv_backward_data_act(const bp_conv* cv, const bp_view* dy, const float* packed_bwd, const bp_view* g,
                              const bp_view* x_raw, const bp_pointwise* x_pw, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream) {
  const bool b16 = g && g->dtype == BP_BF16;
  if (!conv_ok(cv) || !shapes_ok(cv, g, dy, b16) || !packed_bwd || !sums || !bp_view_ok_any(x_raw)) return BP_EINVAL;
  if (x_raw->n != g->n || x_raw->h != g->h || x_raw->w != g->w || x_raw->c != g->c) return BP_EINVAL;
  const IgemmStatsReq sr{3, x_raw, bp_pw(x_pw), sums, workspace, workspace_bytes};
  if (b16) {
    // (only the head kernel has this epilogue: the generic bf16 kernel would ignore the request)
    const ConvGeom gm = bp_geom_backward_data(cv);
    if (!bp_bf16_stats_workspace(gm, dy, g, 3)) return BP_EUNSUPPORTED;
    return bp_bf16_igemm_run(gm, dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, g, bp_stream(stream), &sr);
  }
  if (!bp_view_ok(x_raw)) return BP_EINVAL;
  return bp_igemm_run(bp_geom_backward_data(cv), dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, g,
                      bp_stream(stream), &sr);
}

int bp_conv_backward_data(const bp_conv* cv, const bp_view* dy, const float* packed_bwd, const float* w_torch,
                          const bp_view* dx, int impl, void* stream) {
  if (!conv_ok(cv) || !shapes_ok(cv, dx, dy, impl == BP_IMPL_BF16)) return BP_EINVAL;
  const ConvGeom g = bp_geom_backward_data(cv);
  if (impl == BP_IMPL_BF16) {
    if (!packed_bwd) return BP_EINVAL;
    return bp_bf16_igemm_run(g, dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, dx, bp_stream(stream));
  }
  if (impl == BP_IMPL_AUTO && packed_bwd) {
    const int rc = bp_igemm_run(g, dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, dx, bp_stream(stream));
    if (rc != BP_EUNSUPPORTED || !w_torch) return rc;
    impl = BP_IMPL_DIRECT;
  }
  if (impl == BP_IMPL_AUTO) impl = BP_IMPL_DIRECT;
  if (impl == BP_IMPL_MFMA) {
    if (!packed_bwd) return BP_EINVAL;
    return bp_igemm_run(g, dy, PW{nullptr, nullptr, nullptr}, packed_bwd, nullptr, dx, bp_stream(stream));
  }
  if (!w_torch) return BP_EINVAL;
  return bp_direct_gather(g, bp_wmap(cv, BP_PACK_BWD), dy, PW{nullptr, nullptr, nullptr}, w_torch, nullptr, dx,
                          bp_stream(stream));
}

size_t bp_conv_backward_weight_workspace(const bp_conv* cv, const bp_view* x, const bp_view* dy) {
  if (!conv_ok(cv) || !shapes_ok(cv, x, dy, true)) return 0;
  const bp_view* X = cv->transposed ? dy : x;
  const bp_view* Y = cv->transposed ? x : dy;
  // (the larger of the fp32 and the bf16 kernel's needs: the caller sizes one workspace per layer)
  size_t main = (x->dtype == BP_F32 && dy->dtype == BP_F32) ? bp_wgrad_mfma_workspace(cv, X, Y) : 0;
  const size_t mb = bp_wgrad_bf16_workspace(cv, X, Y);
  if (mb > main) main = mb;
  return align256(main) + align256(bp_channel_sums_workspace(dy)) + align256((size_t)2 * dy->c * sizeof(double));
}

int bp_conv_backward_weight(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw, const bp_view* dy,
                            float* dw_torch, float* dbias, void* workspace, size_t workspace_bytes, int impl,
                            void* stream) {
  const bool shared = (impl & BP_IMPL_SHARED) != 0;
  const WgradPrivateWs priv((impl & BP_IMPL_DEFER) != 0);      // (this call's reduction may wait for the flush)
  impl &= ~(BP_IMPL_SHARED | BP_IMPL_DEFER);
  if (!conv_ok(cv) || !shapes_ok(cv, x, dy, impl == BP_IMPL_BF16) || !dw_torch) return BP_EINVAL;
  const bp_view* X = cv->transposed ? dy : x;
  const bp_view* Y = cv->transposed ? x : dy;
  const PW none{nullptr, nullptr, nullptr};
  const PW pwx = cv->transposed ? none : bp_pw(x_pw);
  const PW pwy = cv->transposed ? bp_pw(x_pw) : none;
  hipStream_t st = bp_stream(stream);
  if (impl == BP_IMPL_BF16) {
    if (dbias) return BP_EUNSUPPORTED;       // (bias gradients: the fp32 path; the CVAE's convolutions have none)
    const size_t need = bp_wgrad_bf16_workspace(cv, X, Y);
    if (!need) return BP_EUNSUPPORTED;
    if (!workspace || workspace_bytes < need) return BP_EWORKSPACE;
    return bp_wgrad_bf16_run(cv, X, pwx, Y, pwy, dw_torch, workspace, workspace_bytes, st);
  }
  const size_t ws_main = align256(bp_wgrad_mfma_workspace(cv, X, Y));
  if (impl == BP_IMPL_AUTO) impl = ws_main ? BP_IMPL_MFMA : BP_IMPL_DIRECT;
  if (impl == BP_IMPL_MFMA || dbias) {
    if (!workspace || workspace_bytes < bp_conv_backward_weight_workspace(cv, x, dy)) return BP_EWORKSPACE;
  }
  int rc;
  if (impl == BP_IMPL_MFMA) {
    if (!ws_main) return BP_EUNSUPPORTED;
    rc = bp_wgrad_mfma(cv, X, pwx, Y, pwy, dw_torch, workspace, ws_main, st, shared);
  } else {
    rc = bp_direct_wgrad(cv, X, pwx, Y, pwy, dw_torch, st);
  }
  if (rc != BP_OK) return rc;
  if (dbias) {
    char* base = reinterpret_cast<char*>(workspace) + ws_main;
    const size_t ws_sums = align256(bp_channel_sums_workspace(dy));
    double* sums = reinterpret_cast<double*>(base + ws_sums);
    rc = bp_channel_sums(dy, sums, base, ws_sums, stream);
    if (rc != BP_OK) return rc;
    rc = bp_sums_to_float(sums, dy->c, dbias, stream);
  }
  return rc;
}

int bp_wgrad_defer_begin(void) { return bp_wgrad_defer_begin_impl(); }
int bp_wgrad_defer_flush(int end, void* stream) { return bp_wgrad_defer_flush_impl(bp_stream(stream), end); }

}  // extern "C"
