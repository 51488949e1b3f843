/*
 * bp_hip.h -- C ABI of the MI355X (gfx950) kernels behind the baryon_painter CVAE hot path.
 *
 * The reference has no FFI: its arithmetic boundary is the set of stock torch.nn modules that
 * `build_sequential` instantiates (/root/reference/baryon_painter/models/utils.py:114-157) plus
 * the tensor expressions in models/cvae.py:63-146.  Each entry point below replaces one of those
 * call sites (cited per function).  Signatures use plain pointers and sizes only; all buffers are
 * device memory owned by the caller, nothing is allocated or synchronised inside, every launch
 * goes to `stream` (a hipStream_t passed as void*), so a call sequence can be captured in a
 * hipGraph.  Return value: BP_OK or a negative error code; nothing is launched on error.
 *
 * Data layout: activations are NHWC fp32.  A `bp_view` addresses channels [coff, coff+c) of a
 * buffer whose pixels are `cstride` floats apart, so channel concatenation (cvae.py:72,109) is
 * "write into a slice".  Convolution outputs are stored RAW (pre batch-norm / pre activation);
 * the per-channel affine + leaky-ReLU that follows them in the module list (BatchNorm2d + ReLU /
 * PReLU / LeakyReLU) is described by a `bp_pointwise` and applied by the CONSUMER while it stages
 * its input ("lazy activation"), which keeps every activation at one HBM write + one read.
 */
#ifndef BP_HIP_H
#define BP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_OK 0
#define BP_EINVAL (-1)       /* inconsistent shapes / null pointers */
#define BP_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define BP_ELAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch */
#define BP_EWORKSPACE (-4)   /* workspace too small */

/* Element type of a view.  fp32 is the reference's arithmetic (parity mode, BASELINE.json configs[1]); bf16
 * storage with fp32 accumulation is the throughput mode of configs[3] and is accepted by the entry points that say
 * so (convolutions, batch-norm statistics, activation / batch-norm backward, residual tail); every other entry
 * point returns BP_EUNSUPPORTED for a bf16 view. */
enum { BP_F32 = 0, BP_BF16 = 1 };

/* NHWC view: element (n,y,x,ch) lives at ptr[((n*h + y)*w + x)*cstride + coff + ch] (in elements of `dtype`). */
typedef struct bp_view {
  float* ptr;      /* device pointer (to 2-byte elements when dtype == BP_BF16) */
  int32_t n, h, w;
  int32_t c;       /* channels in this view            */
  int32_t cstride; /* channels of the underlying buffer */
  int32_t coff;    /* first channel of the view         */
  int32_t dtype;   /* BP_F32 (0) or BP_BF16             */
} bp_view;

/* Per-channel t = x*scale[ch] + shift[ch];  y = t > 0 ? t : t*slope[ch].
 * scale == NULL means identity (no transform at all).  ReLU: slope 0; PReLU: slope a;
 * identity-with-affine: slope 1.  Arrays have `c` entries of the view they accompany. */
typedef struct bp_pointwise {
  const float* scale;
  const float* shift;
  const float* slope;
} bp_pointwise;

/* torch.nn.Conv2d / ConvTranspose2d hyper-parameters (utils.py:128-131, square kernels). */
typedef struct bp_conv {
  int32_t transposed; /* 0: Conv2d, 1: ConvTranspose2d */
  int32_t cin, cout;  /* module in_channels / out_channels */
  int32_t k, stride, pad, out_pad;
} bp_conv;

enum { BP_IMPL_AUTO = 0, BP_IMPL_DIRECT = 1, BP_IMPL_MFMA = 2, BP_IMPL_BF16 = 3 };
/* OR-ed into `impl` of bp_conv_backward_weight: the launch shares the GPU with kernels of another stream (the
 * training step runs weight gradients beside the data-gradient chain), so persistent kernels take one workgroup
 * per CU instead of two and leave room for the other stream's workgroups (and the tiled weight gradients take fewer
 * split-K workgroups).  Results are deterministic per schedule; between the two schedules the grouping of tiles into
 * fp32 partial sums differs, so weight gradients may differ in their last bits. */
enum { BP_IMPL_SHARED = 0x100 };
/* OR-ed into `impl` of bp_conv_backward_weight between bp_wgrad_defer_begin / _flush: this call's workspace is not
 * touched by anyone else before the flush, so its split-K reduction may be deferred (see bp_wgrad_defer_begin). */
enum { BP_IMPL_DEFER = 0x200 };
enum { BP_PACK_FWD = 0, BP_PACK_BWD = 1 };

/* ---- library ------------------------------------------------------------------------------ */
int bp_version(void);
/* Kernel-selection switches for A/B measurements and tests inside ONE process (the environment variables of the same
 * names are read once, at the first call).  "bf16_ws" / "f32_ws": 1 / 0 = use / do not use the weights-stationary kernels
 * (csrc/conv_bf16_ws.hip: bf16 128 -> 128 k3 and 64 -> 128 k4 s2; csrc/conv_ws_f32.hip: fp32 128 -> 128 k3), -1 = back to the
 * environment's choice (BP_BF16_WS / BP_F32_WS); "f32_wgrad_ws" / "bf16_wgrad_ws": the output-stationary weight gradients of
 * the same trunk layers (csrc/conv_wgrad_ws_f32.hip, csrc/conv_wgrad_ws_bf16.hip; BP_F32_WGRAD_WS / BP_BF16_WGRAD_WS).
 * Not thread-safe against concurrent launches; results of either kernel meet the same tolerances.
 * Returns BP_EUNSUPPORTED for an unknown name. */
int bp_set_option(const char* name, int value);
const char* bp_strerror(int code);

/* ---- convolution (replaces nn.Conv2d / nn.ConvTranspose2d forward + their autograd backward,
 *      utils.py:128-131, painter.py:227) ----------------------------------------------------- */

/* Number of floats of the packed weight image for direction `dir` (BP_PACK_FWD feeds
 * bp_conv_forward, BP_PACK_BWD feeds bp_conv_backward_data). */
int64_t bp_conv_packed_floats(const bp_conv* cv, int dir);

/* Which igemm_kernel<CC,NT,WN,MT> instantiation serves this layer/direction, encoded as
 * CC*1000 + NT*100 + WN*10 + MT, plus 100000 / 200000 for the LDS-DMA pipelined igemm_dma_kernel with 4 / 8 waves and 300000 for
 * igemm_dmaf_kernel (tap groups, fused output phases)
 * (so a profile's kernel names can be matched to layers). */
int bp_conv_kernel_id(const bp_conv* cv, int dir);

/* Re-layout torch-format weights (Conv2d: [cout][cin][k][k]; ConvTranspose2d: [cin][cout][k][k])
 * into the MFMA image [phase][tap][cin_chunk][cout_pad][chunk] (zero padded).  Must be re-run
 * whenever the weights change (once per optimiser step). */
int bp_conv_pack(const bp_conv* cv, int dir, const float* w_torch, float* packed, void* stream);

/* Batched packing: every layer's weights in ONE launch (a step re-packs ~60 small images).  bp_conv_pack_job
 * fills a host record of bp_conv_pack_job_bytes() bytes describing bp_conv_pack(cv, dir, w_torch, packed) and
 * returns its workgroup count (BP_EUNSUPPORTED for the few layers served by the vector-ALU kernel: pack those
 * with bp_conv_pack).  The caller uploads the records back to back, plus first_block[njobs + 1] = exclusive
 * prefix sums of the workgroup counts, and launches them with bp_conv_pack_jobs; the pointers inside the records
 * must stay valid (parameters are views of one flat buffer, packed images belong to the plan). */
int32_t bp_conv_pack_job_bytes(void);
int bp_conv_pack_job(const bp_conv* cv, int dir, const float* w_torch, float* packed, void* job,
                     int64_t* nblocks);
int bp_conv_pack_jobs(const void* jobs_dev, const int64_t* first_block_dev, int32_t njobs,
                      int64_t total_blocks, void* stream);

/* bf16 matrix-core path (configs[3]): impl == BP_IMPL_BF16 in the three calls below takes a bf16 packed image
 * (bp_conv_bf16_pack, bp_conv_bf16_packed_elems 2-byte elements) and views of either element type on both sides;
 * products are bf16 x bf16, accumulation and the weight gradient are fp32.  bp_conv_bf16_supported says whether
 * the kernels take a layer / direction with the given views (NULL: any).  The packed image is opaque: for the thin
 * full-resolution layers (unit-stride k7 / k5 heads and stem, stride-2 k4 16<->32 and 32<->64) it carries a second,
 * flattened-K weight image behind the generic one (csrc/conv_bf16_flat.hip), and the run picks the kernel by the
 * views it is given -- always size the buffer with bp_conv_bf16_packed_elems. */
int64_t bp_conv_bf16_packed_elems(const bp_conv* cv, int dir);
/* Which weights-stationary kernel serves this layer / direction with these views (`in` = the gathered tensor of the
 * direction: x for BP_PACK_FWD, dy for BP_PACK_BWD; `out` = the produced one): 3 = the k3 s1 128 -> 128 trunk kernel (bf16
 * views: csrc/conv_bf16_ws.hip; fp32 views: csrc/conv_ws_f32.hip), 4 = the bf16 strided gather k4 s2 64 -> 128, 5 = the bf16
 * transposed form k4 s2 128 -> 64, 0 = none
 * (the tiled / flattened-K kernels).  For profiles and bench.py's kernel labels; follows bp_set_option. */
int bp_conv_ws_kind(const bp_conv* cv, int dir, const bp_view* in, const bp_view* out);
int bp_conv_bf16_pack(const bp_conv* cv, int dir, const float* w_torch, void* packed, void* stream);
int bp_conv_bf16_supported(const bp_conv* cv, int dir, const bp_view* in, const bp_view* out);

/* y_raw = conv(act(x)) [+ bias].  `x_pw` may be NULL (identity). */
int bp_conv_forward(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw,
                    const float* packed_fwd, const float* w_torch, const float* bias,
                    const bp_view* y, int impl, void* stream);

/* dx = d(loss)/d(act(x)) given dy = d(loss)/d(y_raw).  (The activation's own derivative is
 * applied afterwards by bp_act_backward.) */
int bp_conv_backward_data(const bp_conv* cv, const bp_view* dy, const float* packed_bwd,
                          const float* w_torch, const bp_view* dx, int impl, void* stream);

/* Convolution + per-channel sums in one launch (fp32 matrix-core kernels only): the sums are taken from the
 * accumulators in the kernel's epilogue, so the streaming pass of bp_channel_sums / bp_act_backward over the
 * produced tensor is not needed (batch-norm statistics of utils.py:146-147 and the two reductions of its backward).
 *   impl: BP_IMPL_MFMA (fp32 views) or BP_IMPL_BF16 (bf16 packed image, views of either type; forward only: the sums
 *       are those of the tensor as stored, i.e. of the bf16-rounded values where y is bf16).
 *   bp_conv_stats_workspace(cv, dir, x, y, impl)  bytes of workspace for layer `cv` on module input x / output y
 *       (BP_PACK_FWD: bp_conv_forward_stats, BP_PACK_BWD: bp_conv_backward_data_stats); 0 = this layer's kernel
 *       has no statistics epilogue (bias, pixel-packed or vector-ALU kernels, channel count not a power of two):
 *       use the separate passes.
 *   bp_conv_forward_stats        y_raw = conv(act(x)); sums[2*cout] = {sum y, sum y^2} as bp_channel_sums(y).
 *   bp_conv_backward_data_stats  dx = d(loss)/d(act(x)) as bp_conv_backward_data; sums[2*cin] = {sum g, sum g*x_raw},
 *       g = dx * act'(x_pw(x_raw)): the first two sums of bp_act_backward(dx, NULL, x_raw, x_pw, ...).  With bf16
 *       views (dx and x_raw bf16, packed_bwd the bf16 image) the bf16 kernels run: their two full-resolution
 *       flattened-K data gradients have the epilogue (the layers behind the generator's 16-channel 512^2 slots), g
 *       from the bf16-rounded dx as stored; workspace query with impl = BP_IMPL_BF16. */
size_t bp_conv_stats_workspace(const bp_conv* cv, int dir, const bp_view* x, const bp_view* y, int impl);
int bp_conv_forward_stats(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw, const float* packed_fwd,
                          const bp_view* y, double* sums, void* workspace, size_t workspace_bytes, int impl,
                          void* stream);
int bp_conv_backward_data_stats(const bp_conv* cv, const bp_view* dy, const float* packed_bwd, const bp_view* dx,
                                const bp_view* x_raw, const bp_pointwise* x_pw, double* sums, void* workspace,
                                size_t workspace_bytes, void* stream);

/* Data gradient + the producer's WHOLE activation backward in one launch, for a producer without batch-norm (the
 * heads' PReLU layers, cvae.py:57-58 via utils.py:135): what is stored is g = dx * act'(x_pw(x_raw)) -- exactly what
 * bp_act_backward(dx, NULL, x_raw, x_pw, NULL, g = dx, ...) would leave in place of dx -- and sums[3*cin] are its
 * three sums {sum g, sum g*x_raw, sum_{t<=0} dx*t} (the last is the PReLU slope gradient, bp_prelu_slope_grad).
 * Saves the separate pass' read of dx and its write of g (x_raw is read either way).
 * bp_conv_backward_data_act_workspace: bytes of workspace, 0 = this layer's kernel has no such epilogue (only the
 * vector-ALU kernel of the 8 -> 1 k5 layer has one: 0.35 ms of a 15 ms bf16 step). */
size_t bp_conv_backward_data_act_workspace(const bp_conv* cv, const bp_view* dy, const bp_view* g);
int bp_conv_backward_data_act(const bp_conv* cv, const bp_view* dy, const float* packed_bwd, const bp_view* g,
                              const bp_view* x_raw, const bp_pointwise* x_pw, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream);

/* bp_conv_forward_stats followed by bp_bn_finalize(sums, ...) of the produced tensor, with the finalize folded into
 * the launch that sums the epilogue's partial rows (one small launch less per batch-norm layer and step; the same
 * numbers bit for bit).  Single-device training only: under data parallelism the sums are all-reduced between the two
 * calls (utils.py:146-147 on the global batch).  The fields are bp_bn_finalize's arguments. */
typedef struct bp_bn_train {
  double count;                       /* elements per channel: n * h * w of y */
  const float* gamma; const float* beta; /* NULL: 1 / 0 */
  float eps, momentum;
  float* running_mean; float* running_var; int64_t* num_batches_tracked; /* updated in place; NULL: skipped */
  float* scale; float* shift;         /* the pending batch-norm of y as a bp_pointwise */
  double* save_mean; double* save_invstd; /* for bp_bn_backward_finalize; NULL: skipped */
} bp_bn_train;
int bp_conv_forward_bn(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw, const float* packed_fwd,
                       const bp_view* y, double* sums, const bp_bn_train* bn, void* workspace, size_t workspace_bytes,
                       int impl, void* stream);

/* dw (torch layout, overwritten) and optionally dbias (NULL to skip) from act(x) and dy. */
size_t bp_conv_backward_weight_workspace(const bp_conv* cv, const bp_view* x, const bp_view* dy);
int bp_conv_backward_weight(const bp_conv* cv, const bp_view* x, const bp_pointwise* x_pw,
                            const bp_view* dy, float* dw_torch, float* dbias, void* workspace,
                            size_t workspace_bytes, int impl, void* stream);

/* Deferred split-K reductions.  Between bp_wgrad_defer_begin() and bp_wgrad_defer_flush(end = 1, ...) on one host
 * thread, a bp_conv_backward_weight call flagged BP_IMPL_DEFER launches the layer's partial-sum kernel but only
 * RECORDS its reduction into dw_torch;
 * bp_wgrad_defer_flush launches every recorded reduction on `stream` (two launches for a whole network instead of
 * one per layer; each dw is reduced in the same order as without deferral: bit-identical).  The caller must give each
 * flagged call its own workspace until the flush, and flush on the stream (or behind the streams) the calls ran on.
 * Measured on the fiducial step: -0.3 ms for the fp32 layers (28-way splits, partial sums of 0.6 ... 16 MB); the
 * 128 ... 512-way splits of the bf16 kernels are better reduced at once, while their partial sums are still in L2.
 * end = 0 flushes and keeps deferring; end < 0 drops what was recorded and stops (error paths).  The recorded state is
 * per host thread: a flush (end >= 0) on a thread without an active bp_wgrad_defer_begin returns BP_EINVAL instead of
 * silently leaving the deferred gradients unreduced.  (Replaces nothing in the reference: torch's autograd launches one cuDNN
 * weight-gradient kernel per layer, cvae.py:392 loss.backward().) */
int bp_wgrad_defer_begin(void);
int bp_wgrad_defer_flush(int end, void* stream);

/* ---- batch norm (replaces nn.BatchNorm2d, utils.py:146-147) -------------------------------- */

/* Per-channel double sums[2*c] = {sum x, sum x^2} over all pixels of `x` (deterministic two-stage
 * reduction).  workspace: bp_channel_sums_workspace(x) bytes.  Under data parallelism the caller
 * all-reduces `sums` across ranks before bp_bn_finalize (global-batch statistics). */
size_t bp_channel_sums_workspace(const bp_view* x);
int bp_channel_sums(const bp_view* x, double* sums, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Train mode: batch mean / biased variance from `sums` and `count` (= N*H*W over all ranks),
 * writes the consumer's pointwise (scale = gamma*invstd, shift = beta - mean*scale), saves
 * mean/invstd (in double: the backward's cancellation needs them exact) for backward and updates running stats (momentum, unbiased variance) and
 * num_batches_tracked (int64, may be NULL). */
int bp_bn_finalize(const double* sums, double count, int32_t c, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                   double* save_mean, double* save_invstd, void* stream);

/* Eval mode: scale/shift from the running statistics. */
int bp_bn_eval_pointwise(int32_t c, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float eps,
                         float* scale, float* shift, void* stream);

/* ---- activation / batch-norm backward ------------------------------------------------------ */

/* Backward through y = leaky(raw*scale+shift [+ skip], slope):
 *   g = (dout [+ dout2]) * act'(t)           written to `g` (may alias dout; NULL: sums only, for
 *                                            layers that continue with bp_act_bn_backward_apply)
 *   sums[3*c] (double) = { sum g, sum g*raw, sum (dout[+dout2]) * t * [t<=0] }  (third: d slope)
 * `act_out`, if given, is the saved activated output and supplies the sign of t (residual
 * blocks, where t includes the skip); otherwise t is recomputed from raw and `pw`. */
size_t bp_act_backward_workspace(const bp_view* raw);
int bp_act_backward(const bp_view* dout, const bp_view* dout2, const bp_view* raw,
                    const bp_pointwise* pw, const bp_view* act_out, const bp_view* g,
                    double* sums, void* workspace, size_t workspace_bytes, void* stream);

/* bp_act_backward followed by bp_bn_backward_finalize(sums, ...) of the same layer, the finalize folded into the
 * launch that adds the partial sums (one tiny launch less per batch-norm layer on the backward chain, where every
 * launch's latency is exposed; the same numbers bit for bit).  Single-device training only: under data parallelism
 * the sums are all-reduced between the two.  The fields are bp_bn_backward_finalize's arguments. */
typedef struct bp_bn_backward_fin {
  double count;
  const float* gamma;                    /* NULL: 1 */
  const double* save_mean; const double* save_invstd;
  float param_grad_scale;
  float* dgamma; float* dbeta;           /* NULL: skipped */
  double* coef_abc;                      /* [4*c] */
} bp_bn_backward_fin;
int bp_act_backward_bn(const bp_view* dout, const bp_view* dout2, const bp_view* raw,
                       const bp_pointwise* pw, const bp_view* act_out, const bp_view* g, double* sums,
                       const bp_bn_backward_fin* fin, void* workspace, size_t workspace_bytes, void* stream);

/* From the (all-reduced) sums: dgamma, dbeta and the per-channel coefficients {A, mg, B, mean} of
 *   d_raw = A*(g - mg) + B*(raw - mean)   (batch-norm backward as an affine map of (g, raw)),
 * kept and evaluated in double like the reference's CPU batch_norm_backward (its accumulate type).
 * dgamma/dbeta are multiplied by `param_grad_scale` (1/world_size under data parallelism, where
 * the sums are global but every rank's loss is normalised by its local batch; 1 otherwise). */
int bp_bn_backward_finalize(const double* sums, double count, int32_t c, const float* gamma,
                            const double* save_mean, const double* save_invstd,
                            float param_grad_scale, float* dgamma, float* dbeta,
                            double* coef_abc /* [4*c] */, void* stream);

/* out = A*(g - mg) + B*(raw - mean)  (out may alias g). */
int bp_bn_backward_apply(const bp_view* g, const bp_view* raw, const double* coef_abc,
                         const bp_view* out, void* stream);

/* The same map with g recomputed on the fly from (dout [+ dout2], raw, pw, act_out) exactly as
 * bp_act_backward defines it: pairs with bp_act_backward(..., g = NULL, ...) so that g never
 * travels through HBM (out may alias dout). */
int bp_act_bn_backward_apply(const bp_view* dout, const bp_view* dout2, const bp_view* raw,
                             const bp_pointwise* pw, const bp_view* act_out, const double* coef_abc,
                             const bp_view* out, void* stream);

/* dst[ch] = (float) sums[ch]  (e.g. a bias gradient from bp_channel_sums of dy). */
int bp_sums_to_float(const double* sums, int32_t c, float* dst, void* stream);

/* PReLU slope gradient (one shared slope, utils.py:137): dslope = sum over channels of sums[2]. */
int bp_prelu_slope_grad(const double* sums, int32_t c, float* dslope, void* stream);

/* ---- residual block tail (utils.py:35-37): out = leaky(raw*scale+shift + act(skip), slope) -- */
int bp_residual_forward(const bp_view* raw, const bp_pointwise* pw, const bp_view* skip,
                        const bp_pointwise* skip_pw, float slope, const bp_view* out, void* stream);

/* ---- layout glue ---------------------------------------------------------------------------- */

/* NCHW (N,c,H,W) -> channels [coff, coff+c) of `out`, and the aux label(s) (N,caux) broadcast to
 * constant planes in the next caux channels (merge_aux_label, utils.py:159-182). aux may be NULL. */
int bp_nchw_to_view(const float* src_nchw, int32_t c, const float* aux, int32_t caux,
                    const bp_view* out, void* stream);
/* dst = [softplus](act(src)) as NCHW; `softplus` = 1 applies torch's Softplus(beta=1,threshold=20). */
int bp_view_to_nchw(const bp_view* src, const bp_pointwise* pw, int32_t softplus, float* dst_nchw,
                    void* stream);
int bp_fill(float* dst, int64_t n, float value, void* stream);

/* ---- paint() pipeline (painter.py:371-392 around cvae.py:149-162; utils/data_transforms.py:72-97) ----------------
 * The raw tile goes in, the physical tile comes out: the reference's "shift-log" range compression and its inverse
 * are fused into the layout kernels on either side of the network, with the float32 / float64 promotion of the
 * reference's NumPy expressions, so device and host transforms agree bit for bit up to libm's exp (<= 1 ulp).
 *   bp_paint_load : out[n,:,:,ch<c] = (float) (log((double) raw / sigma_k[n][0] + 1) / sigma_k[n][1]), and the aux
 *                   label(s) as constant planes in the next caux channels (= transform + bp_nchw_to_view)
 *   bp_paint_store: dst = (float) ((double) (expf32(act(src) * (float) k_sigma[n][0]) - 1.f) * k_sigma[n][1]),
 *                   NCHW (= bp_view_to_nchw + inverse transform; softplus as in bp_view_to_nchw) */
int bp_paint_load(const float* raw_nchw, int32_t c, const double* sigma_k, const float* aux, int32_t caux,
                  const bp_view* out, void* stream);
/* bp_paint_load into TWO views of the same shape at once (cvae.py:87-88 and :104-105 merge the same y with the aux
 * label for the prior network and for the generator's concatenated input): one evaluation of the transform. */
int bp_paint_load2(const float* raw_nchw, int32_t c, const double* sigma_k, const float* aux, int32_t caux,
                   const bp_view* out, const bp_view* out2, void* stream);
int bp_paint_store(const bp_view* src, const bp_pointwise* pw, int32_t softplus, const double* k_sigma,
                   float* dst_nchw, void* stream);
/* eps (L, n, per_tile) standard normal for the sampler of cvae.py:64-65 from Philox4x32-10 keyed on `seed`, counter
 * (element group, l, tile id): a tile's noise depends on (seed, its GLOBAL id) only, not on batch, stream or rank
 * (torch.randn on the device in the reference: same distribution, no reproducible stream to match). */
int bp_philox_normal(uint64_t seed, const int64_t* tile_ids, int32_t n, int32_t L, int32_t per_tile, float* eps,
                     void* stream);
/* The same with the key read from device memory at run time: a launch captured in a hipGraph then serves every seed
 * (the reference draws fresh torch.randn per tile, cvae.py:64: planes of a light cone must not share their noise). */
int bp_philox_normal_dev(const uint64_t* seed_dev, const int64_t* tile_ids, int32_t n, int32_t L, int32_t per_tile,
                         float* eps, void* stream);

/* ---- latent heads: reparametrisation sampler + KL (cvae.py:63-66, 76-77, 126-130) ----------- */
typedef struct bp_latent {
  int32_t n;      /* batch M                         */
  int32_t L;      /* samples per datum (cvae.py:21)  */
  int32_t zc, zh, zw;
  float min_z_var;
} bp_latent;

/* q_raw / p_raw: raw head outputs (N,zh,zw,2*zc) with their pointwise (BN+ReLU); p_raw may be
 * NULL (no prior network: standard-normal prior).  eps: (L,N,zc,zh,zw) standard normal.
 * Outputs: stats4 (4,N,zc,zh,zw) = {z_mu, z_log_var, prior_mu, prior_log_var} (NCHW planes),
 * z (L*N, zh, zw, zc) view, kl_sum: one double = sum[...] of cvae.py:129-130 (un-normalised). */
int bp_latent_forward(const bp_latent* lt, const bp_view* q_raw, const bp_pointwise* q_pw,
                      const bp_view* p_raw, const bp_pointwise* p_pw, const float* eps,
                      float* stats4, const bp_view* z, double* kl_sum, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Gradients w.r.t. the ACTIVATED head outputs (N,zh,zw,2*zc): reparametrisation + KL terms. */
int bp_latent_backward(const bp_latent* lt, const bp_view* dz, const float* stats4,
                       const float* eps, const float* seed /* device scalar dL/dELBO */,
                       float beta_kl, const bp_view* dq_act, const bp_view* dp_act, void* stream);

/* ---- Gaussian log-likelihood head (cvae.py:132-146) ----------------------------------------- */
typedef struct bp_loglik {
  int32_t n;  /* M */
  int32_t L;
  int32_t c, h, w;
  int32_t mu_softplus;  /* last activation of the mean head: 1 softplus, 0 none */
  int32_t predict_var;
  float alpha_var, beta_kl, likelihood_scaling;
} bp_loglik;

/* x: NCHW (M,c,H,W); mu_raw / var_raw: (L*M,H,W,c) raw head outputs with optional pointwise.
 * Writes x_mu (NCHW, L*M) [and x_log_var], and stats (floats):
 *   [0]=ELBO [1]=KL_term [2..2+c)=log_likelihood [..+c)=fixed_var [..+c)=free_var. */
size_t bp_loglik_workspace(const bp_loglik* ll);
int bp_loglik_forward(const bp_loglik* ll, const float* x_nchw, const bp_view* mu_raw,
                      const bp_view* var_raw, const double* kl_sum, float* x_mu_nchw,
                      float* x_log_var_nchw, float* stats, void* workspace,
                      size_t workspace_bytes, void* stream);
/* d(seed*ELBO)/d(mu_raw), d/d(var_raw). */
int bp_loglik_backward(const bp_loglik* ll, const float* x_nchw, const bp_view* mu_raw,
                       const bp_view* var_raw, const float* seed, const bp_view* d_mu_raw,
                       const bp_view* d_var_raw, void* stream);

/* ---- GAN heads (the CGAN of trained_models/README.md:95-144; no code in the reference) ------ */
/* out = f(act(in)), kind 0 identity, 1 tanh (generator output, README.md:128), 2 sigmoid. */
int bp_unary_forward(const bp_view* in, const bp_pointwise* pw, int32_t kind, const bp_view* out,
                     void* stream);
/* sum over samples [n0,n1) of BCE(sigmoid(raw), target) evaluated on the logits; and its gradient
 * d_raw = scale * (sigmoid(raw) - target) for those samples. */
int bp_bce_logits(const bp_view* raw, int32_t n0, int32_t n1, float target, double* sum,
                  void* workspace, size_t workspace_bytes, void* stream);
int bp_bce_logits_grad(const bp_view* raw, int32_t n0, int32_t n1, float target, float scale,
                       const bp_view* d_raw, void* stream);
/* sum |fake - x| ; d_raw = (d_fake + l1_scale*sign(fake - x)) * (1 - fake^2) through the Tanh. */
int bp_l1_sum(const bp_view* fake, const float* x_nchw, double* sum, void* workspace,
              size_t workspace_bytes, void* stream);
int bp_tanh_l1_backward(const bp_view* fake, const float* x_nchw, const bp_view* d_fake,
                        float l1_scale, const bp_view* d_raw, void* stream);

/* ---- device-side batch assembly (replaces BAHAMASDataset.get_stack + transform on the host,
 *      utils/datasets.py:305-404): the stacks live in HBM, one launch builds a field of a batch.
 * desc100/desc150: n records {const float* base; int32 pitch; int32 r0,rr,rc,c0,cr,cc; int32 pad}
 *   (source row = r0 + rr*r + rc*c, col = c0 + cr*r + cc*c: the dihedral tile permutation);
 * xform: n records {double scale, inv_sigma, inv_k; int32 mode; int32 pad}, mode 1 = shift-log.
 * out: (n,1,tile,tile) float32 = transform(scale * (tile100 + tile150)). */
int bp_gather_tiles(const void* desc100, const void* desc150, const void* xform, int32_t n,
                    int32_t tile, float* out_nchw, void* stream);

/* ---- optimiser (replaces torch.optim.Adam.step, painter.py:93,228; same arithmetic) --------- */
int bp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, int32_t step, void* stream);
/* The same update with its scalars read from device memory - hyper[6] = {lr, beta1, beta2, eps,
 * 1 - beta1^step, sqrt(1 - beta2^step)} - so that the launch can sit in a captured hipGraph of the whole
 * training step while the learning rate and the step number keep changing. */
int bp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                     const float* hyper, void* stream);

/* ---- data parallel: small-message all-reduce over peer memory (csrc/peer_comm.hip) --------------------------
 * The reference is single-device; sharding its minibatch keeps its arithmetic only if every BatchNorm2d sees the
 * statistics of the GLOBAL batch (painter.py:224 feeds the whole minibatch through the module): 44 dependent all-reduces
 * of <= 2 KB per step.  One process per GPU; each rank calls bp_peer_create (allocates its fine-grained exchange buffer,
 * returns an opaque communicator and an IPC handle of bp_peer_handle_bytes() bytes), the host exchanges the handles
 * (any transport: torch.distributed all_gather), each rank passes the world_size handles, in rank order, to bp_peer_open.
 * bp_peer_all_reduce then sums `n` (<= bp_peer_max_doubles()) doubles in place over the ranks with ONE kernel on `stream`:
 * peer stores over xGMI, per-rank flags, a fixed rank-order sum (bitwise the same result on every rank).  Every rank must
 * issue the same sequence of calls.  Spins are bounded (`spin_limit` polls, <= 0: default): a timeout is counted in the
 * communicator's status word (bp_peer_status, synchronising) and the call's result is then undefined. */
int bp_peer_handle_bytes(void);
int bp_peer_max_doubles(void);
int bp_peer_slots(void);                 /* slots of the exchange ring: a step must issue fewer than half as many collectives */
int bp_peer_create(int rank, int world, void** comm_out, void* handle_out);
int bp_peer_open(void* comm, const void* handles);
int bp_peer_all_reduce(void* comm, double* data, int n, int64_t spin_limit, void* stream);
int64_t bp_peer_status(void* comm);
/* Bind `comm` (NULL: unbind) to the CALLING host thread.  While bound, every launch that finishes a layer's batch-norm
 * statistics together with the finalize arithmetic -- bp_conv_forward_bn (the `count` of bp_bn_train is then the GLOBAL
 * pixel count) and bp_act_backward_bn (`count` global, `pscale` = 1 / world size) -- exchanges each channel's two sums with
 * the other ranks inside that kernel, in place of a separate all-reduce: the collective number advances as in
 * bp_peer_all_reduce, so every rank must issue the same sequence of both kinds of call (cvae.py:224 semantics: the
 * statistics of the global batch, as one nn.BatchNorm2d over the whole minibatch computes them). */
int bp_peer_bind(void* comm);
/* The per-channel exchange of the bound form on its own, one collective: data[ch] and data[c + ch] (2 c <=
 * bp_peer_max_doubles()) summed over the ranks by workgroup ch -- the start-up self-test of that path (dist.py). */
int bp_peer_exchange_check(void* comm, double* data, int c, void* stream);
int bp_peer_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* BP_HIP_H */
