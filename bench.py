#!/usr/bin/env python3
"""Benchmark of the MI355X-native CVAE / CGAN hot path on synthetic 512x512 dark-matter -> pressure tiles.

    python bench.py --gpus N --steps K --warmup W

HEADLINE (metric / value / config / dtype / roofline of the JSON line): the CVAE fiducial train step (forward +
backward + Adam), fp32, 64 tiles per GPU -- BASELINE.json configs[1].  The same line carries, as bounded secondary
legs (their own objects, each with its own roofline; none of them changes the headline fields):
  "paint" -- configs[4] per GPU: CVAEPainter.paint_stream, raw host tile -> physical host tile, hipGraph forward;
  "bf16"  -- configs[3] per GPU: the same train step with bf16 activations / gradients in the generator trunk
             (+ its own paint leg);
  "cgan"  -- configs[2]: CGAN fiducial alternating D+G iteration, batch 64 (single GPU only);
  "cpu_baseline" -- oracle/torch_ref.py (the reference's torch.nn.functional graph) on the host cores, configs[0].
A compact copy of the secondary legs' numbers is kept in config["other_configs"].  --legs selects legs.

N > 1: the driver launches one rank per GPU with torch.distributed.run; started plainly (`python bench.py --gpus N`,
no WORLD_SIZE in the environment) this script launches that itself as a CHILD process before touching the GPU and
relays rank 0's JSON line.  The batch is sharded data-parallel (weak scaling: 64 tiles per GPU), gradients are
all-reduced as one flat buffer and batch-norm statistics are all-reduced per layer (global-batch arithmetic).
Timing: W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides, MAX over ranks.
  roofline -- the dominant kernel's achieved rate from HIP events recorded around every launch (on the stream the
              kernels are launched on) in extra steps of the serial schedule right after the timed region.
"""
import argparse
import json
import re
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (spec, no sparsity)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a streaming copy reaches
CGAN_FLOP_PER_TILE = 0.609e12 * 1.0   # alternating D+G iteration ~ 255 + 354 GFLOP per 512^2 tile
TILE = 512
BATCH_PER_GPU = 64


def source_hash():
    """Hash of the kernel sources: stamps profiles/pmc_traffic.json (tools/pmc_to_json.py) so that counter-derived
    traffic figures are only quoted for the kernels they were measured on (the GPU box has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "baryon_painter_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_name(kind, unit, lib):
    import ctypes as C
    from baryon_painter_amd import _lib as L
    if kind in ("bn_stats", "act_backward", "bn_apply", "residual"):
        return {"bn_stats": "channel_sums", "act_backward": "act_backward", "bn_apply": "bn_backward_apply",
                "residual": "residual_forward"}[kind] + ("_bf16_kernel" if unit.out.dt == L.BF16 else "_fast_kernel")
    if getattr(unit, "bf16", False):
        cv = unit.cv
        tag = "%s%d->%d k%ds%d" % ("T" if cv.transposed else "C", cv.cin, cv.cout, cv.k, cv.stride)
        if kind == "backward_weight":
            return "wgrad_bf16_kernel[%s]" % tag
        # the weights-stationary kernels (csrc/conv_bf16_ws.hip) serve a layer's forward AND data gradient with one
        # instance family: one label, so that the dominant kernel is picked by instance, not by direction
        fwd = kind == "forward"
        vin = unit.inp.view if fwd else unit.out.grad
        vout = unit.out.view if fwd else unit.dx
        ws = lib.bp_conv_ws_kind(C.byref(cv), L.PACK_FWD if fwd else L.PACK_BWD, C.byref(vin), C.byref(vout)) \
            if vin is not None and vout is not None else 0
        if ws:
            return "ws%d_bf16_kernel[%s]" % (ws, tag)
        return ("igemm_bf16_kernel[%s fwd]" if fwd else "igemm_bf16_kernel[%s dgrad]") % tag
    if kind == "backward_weight":
        cv = unit.cv
        cx, cy = (cv.cout, cv.cin) if cv.transposed else (cv.cin, cv.cout)
        if (cv.transposed, cv.cin, cv.cout, cv.k, cv.stride, cv.pad) == (0, 3, 16, 5, 1, 2):
            return "stem_wgrad_kernel"
        if (cv.transposed, cv.cin, cv.cout, cv.k, cv.stride, cv.pad) == (0, 16, 8, 7, 1, 3):
            return "wgrad_flat_kernel"
        if (cx, cy, cv.k, cv.stride, cv.pad) == (16, 32, 4, 2, 1):
            return "wgrad_flat_s2_kernel"
        if cy == 1 and cv.stride == 1 and ((cx == 1 and cv.k in (3, 5)) or (4 < cx <= 8 and cv.k == 5)):
            return "wgrad_c1_kernel<%d>" % cv.k if cx == 1 else "wgrad_cy1_kernel<%d,8,16>" % cv.k
        if cx <= 16 and cy <= 16 and not (cx == 16 and cy == 16) and (cv.k, cv.stride) in ((3, 1), (5, 1), (7, 1), (4, 2), (8, 4)):
            return "wgrad_small_kernel[k%ds%d %d,%d]" % (cv.k, cv.stride, cx, cy)
        return "wgrad_tiles_kernel[k%ds%d %s]" % (cv.k, cv.stride, "wide" if (cx > 16 and cy > 16) else "thin")
    cv = unit.cv
    if kind == "backward_data" and getattr(unit, "_sub", None) is not None:
        cv = unit._sub["cv"]           # data gradient restricted to a channel slice
    elif kind in ("forward", "backward_data"):
        # the fp32 weights-stationary trunk kernel (csrc/conv_ws_f32.hip) takes the layer by its views, not by its id
        fwd = kind == "forward"
        vin, vout = (unit.inp.view, unit.out.view) if fwd else (unit.out.grad, unit.dx)
        if vin is not None and vout is not None and \
                lib.bp_conv_ws_kind(C.byref(cv), L.PACK_FWD if fwd else L.PACK_BWD, C.byref(vin), C.byref(vout)) == 3:
            return "ws3_f32_kernel[C%d->%d k%ds%d]" % (cv.cin, cv.cout, cv.k, cv.stride)
    kid = lib.bp_conv_kernel_id(C.byref(cv), L.PACK_FWD if kind == "forward" else L.PACK_BWD)
    if kid in (780001, 780002):
        return "enc0_fwd_kernel<%d>" % (kid - 780000)
    if kid == 760000:
        return "enc_fwd_kernel"
    if kid == 770000:
        return "enc_dgrad_kernel"
    if kid == 700000:
        return "stem_forward_kernel"
    if kid == 750000:
        return "flat_h7_kernel"
    if kid == 740000:
        return "flat_t64_kernel"
    if kid == 730000:
        return "flat_g4_kernel"
    if kid == 710000:
        return "flat_k7_kernel"
    if kid == 720000:
        return "flat_t4_kernel"
    if 800000 <= kid < 900000:
        return "tiny_%s_kernel<%d,%d,%d,%d>" % ("transposed" if kid % 10 else "gather", kid // 10000 % 10, kid // 1000 % 10,
                                                kid // 100 % 10, kid // 10 % 10)
    if kid >= 900000:
        return "small_conv_kernel<%d,%d,%d>" % (kid // 1000 % 100, kid // 10 % 100, kid % 10)
    dma, kid = divmod(kid, 100000)
    return "%s<%d,%d,%d,%d>" % (("igemm_kernel", "igemm_dma_kernel", "igemm_dma8_kernel", "igemm_dmaf_kernel",
                                 "igemm_wres_kernel")[dma], kid // 1000,
                                kid // 100 % 10, kid // 10 % 10, kid % 10)


def host_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(seconds_budget=30.0):
    """Reference arithmetic on the host cores (oracle/torch_ref.py: the reference's torch.nn.functional graph +
    autograd + torch Adam).  SURVEY.md 8d: BASELINE.json configs[0] exactly -- batch 4 of 256x256 tiles, 10 warm-up
    + 50 timed train steps, median -- plus, as secondary figures on what is left of the time budget, batch 4 at
    512x512 and paint (sample_P) of one 512x512 tile."""
    from baryon_painter_amd.models import arch as A
    from baryon_painter_amd.utils import synthetic as syn
    from oracle.cvae_oracle import CVAEOracle
    from oracle.torch_ref import TorchRefCVAE
    torch.set_num_threads(host_cores())
    t_begin = time.time()

    def train_steps(tile, n, warm, timed, budget):
        arch = A.fiducial_architecture(tile)
        m = TorchRefCVAE(arch, syn.fill_params(CVAEOracle(arch).param_shapes(), 7))
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        x, y, aux = syn.synthetic_batch(n, tile, tile, seed=5)
        eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=6)
        times, t0 = [], time.time()
        for it in range(warm + timed):
            t1 = time.time()
            elbo = m.forward(x, y, aux, eps)
            opt.zero_grad()
            (-elbo).backward()
            opt.step()
            if it >= warm:
                times.append(time.time() - t1)
            if len(times) >= 3 and time.time() - t0 > budget:
                break
        return m, arch, float(np.median(times)), len(times)

    _, _, med256, k256 = train_steps(256, 4, 10, 50, seconds_budget * 0.6)
    left = max(4.0, seconds_budget - (time.time() - t_begin))
    m, arch, med512, k512 = train_steps(512, 4, 1, 6, left * 0.8)
    m.training = False
    x, y, aux = syn.synthetic_batch(1, 512, 512, seed=7)
    z = syn.synthetic_eps((1, *arch["dim_z"]), seed=8)
    tp = []
    for _ in range(6):
        t1 = time.time()
        m.sample_P(y, aux, z=z)
        tp.append(time.time() - t1)
    medp = float(np.median(tp[1:]))
    return {"value": 4 / med256, "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"BASELINE.json configs[0]: {k256} timed train steps (fwd+bwd+Adam, after 10 warm-up) of batch 4 x "
                      f"256x256, fp32, oracle/torch_ref.py (torch.nn.functional on CPU), median {med256:.3f} s/step",
            "train_512_tiles_per_sec": 4 / med512,
            "train_512_sample": f"{k512} timed steps of batch 4 x 512x512, median {med512:.3f} s/step",
            "paint_512_tiles_per_sec": 1 / medp,
            "paint_512_sample": f"5 timed sample_P of one 512x512 tile (eval mode), median {medp * 1e3:.1f} ms"}


def _reduce_max_time(dt, dev, world):
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _timed(step, steps, warmup, world):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides."""
    out = None
    for _ in range(warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    return time.perf_counter() - t0, out


def _free():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _pmc_leg(dtype):
    """This dtype's leg of profiles/pmc_traffic.json (tools/pmc_to_json.py: per kernel instance and launch shape, per
    step) -- only when the stamp says the counters were collected on these very kernel sources."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pj.get("source_hash") == source_hash():
            return pj.get("legs", {}).get(dtype)
    except Exception:
        pass
    return None


def _traffic(kernel, dtype):
    """HBM bytes per launch of the kernel bench.py labels `kernel`, from the PMC passes of the SAME dtype's step
    (rocprofv3 cannot run inside bench.py); launch-weighted over the launch shapes the label covers."""
    leg = _pmc_leg(dtype)
    if leg is None:
        return None
    tab = leg["by_label"]
    if kernel in tab:
        return tab[kernel]["hbm_bytes_per_launch"]
    fam = re.sub(r"[<\[].*", "", kernel)          # (streaming passes: the counter file names the family)
    return tab[fam]["hbm_bytes_per_launch"] if fam in tab else None


def _collective_report(sync, step):
    """Two further steps of the timed schedule with HIP events around every all-reduce."""
    n0, g0, b0, f0 = sync.n_small, sync.n_grad, sync.bytes_grad, getattr(sync, "n_fused", 0)
    sync.timing = []
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ev, sync.timing = sync.timing, None
    per = {}
    for e0, e1, kind in ev:
        per.setdefault(kind, []).append(e0.elapsed_time(e1) * 1e3)
    stats = {k: {"per_step": len(v) // 2, "ms_per_step": round(sum(v) / 2e3, 3), "min_us": round(min(v), 1),
                 "median_us": round(float(np.median(v)), 1), "max_us": round(max(v), 1)} for k, v in per.items()}
    sync.check()                                  # (a timed-out peer-memory collective would have corrupted the steps)
    n_grad = (sync.n_grad - g0) // 2
    return {"collectives_per_step": {"batch_norm_statistics": (sync.n_small - n0) // 2, "gradient_buffers": n_grad,
                                     # exchanged inside the kernels that finalize the statistics: no launch of their own
                                     "batch_norm_statistics_fused_into_finalize": (getattr(sync, "n_fused", 0) - f0) // 2},
            "gradient_bytes_per_step": (sync.bytes_grad - b0) // 2,
            "inside_collectives": stats,
            # (what RAN, from the count of gradient collectives: BP_GRAD_COMM=1 alone selects a second communicator but
            #  not the early schedule)
            "gradient_all_reduce": "generator trunk + heads (97 % of the bytes) early on the weight-gradient stream, own "
                                   "communicator (BP_EARLY_ALLREDUCE=1); the rest after the backward pass"
                                   if n_grad > 1 else "one flat buffer after the backward pass",
            "statistics_transport": "IPC-mapped peer memory (csrc/peer_comm.hip): inside the finalize kernels where a layer has "
                                    "one, else one kernel per collective"
                                    if getattr(sync, "peer", None) is not None else "process group",
            "backend": torch.distributed.get_backend()}


# ---------------------------------------------------------------------------------------------- CVAE train step + paint
def cvae_leg(args, dtype, dev, world, rank, sync, steps, warmup, paint=True):
    """One CVAE leg: timed train steps, collectives report (N > 1), per-kernel HIP-event profile of the serial
    schedule -> roofline, paint_stream.  Returns the leg's dict (every rank runs it; rank 0's is printed)."""
    import contextlib
    from baryon_painter_amd.models import arch as A
    from baryon_painter_amd.models.cvae import CVAE
    from baryon_painter_amd.optim import FlatAdam
    from baryon_painter_amd.utils import synthetic as syn

    arch = A.fiducial_architecture(args.tile)
    torch.manual_seed(1234)                      # same initial weights on every rank
    with contextlib.redirect_stdout(sys.stderr):     # the model announces itself like the reference does
        model = CVAE(arch, dev, sync=sync, dtype=dtype)
    opt = FlatAdam(model, lr=1e-3) if not args.torch_adam else torch.optim.Adam(model.parameters(), lr=1e-3)
    n = args.batch
    # synthetic tiles: a few distinct ones, tiled up to the batch (generation cost, not arithmetic)
    nb = min(n, 8)
    x, y, aux = syn.synthetic_batch(nb, args.tile, args.tile, seed=1234 + rank)
    reps = (n + nb - 1) // nb
    x = torch.from_numpy(np.tile(x, (reps, 1, 1, 1))[:n]).to(dev)
    y = torch.from_numpy(np.tile(y, (reps, 1, 1, 1))[:n]).to(dev)
    aux = torch.from_numpy(np.tile(aux, reps)[:n]).to(dev)

    def eager_step():
        elbo = model(x, y, aux)
        opt.zero_grad()
        (-elbo).backward()
        opt.step()
        return elbo

    step = eager_step
    use_graph = args.graph and world == 1 and not args.torch_adam
    if use_graph:
        eager_step()                                  # (sizes workspaces, claims gradient buffers)
        gstep = model.make_graphed_train_step(opt, n)
        step = lambda: gstep(x, y, aux)

    dt, elbo = _timed(step, steps, warmup, world)
    dt = _reduce_max_time(dt, dev, world)
    plan = model._last
    final_elbo = float(elbo.detach())
    coll = _collective_report(sync, step) if sync is not None else None

    # ---- kernel times for the roofline: PROF_STEPS further steps of the SERIAL schedule (not timed above).
    # In the timed region the weight gradients run on a second stream and share the CUs with the data gradients
    # and the batch-norm passes, so HIP events around a launch there measure "this kernel while sharing the GPU",
    # not the kernel.  Serially every launch has the GPU to itself; the events sit on the stream the kernels are
    # launched on (torch's current stream).
    PROF_STEPS = 2
    model.overlap_weight_gradients(False)
    plan.prof = []                               # HIP events around every convolution launch
    for _ in range(PROF_STEPS):
        eager_step()                             # (eager: the events are recorded by the launch hooks)
    torch.cuda.synchronize()
    prof_events, plan.prof = plan.prof, None
    model.overlap_weight_gradients(True)

    paint_leg = _paint_leg(args, model, dtype, dev, world, rank, n) if paint else None

    per = {}
    for e0, e1, unit, kind, nstreams in prof_events:
        name = kernel_name(kind, unit, model._lib)
        d = per.setdefault(name, {"ms": 0.0, "launches": 0, "flop": 0.0, "bytes": 0.0})
        d["ms"] += e0.elapsed_time(e1)
        d["launches"] += 1
        if kind in ("forward", "backward_data", "backward_weight"):
            d["flop"] += 2.0 * unit.macs(kind)
        d["bytes"] += unit.algorithmic_bytes(kind, nstreams)
    if args.layers and rank == 0:
        lay = {}
        for e0, e1, unit, kind, nstreams in prof_events:
            fl = 2.0 * unit.macs(kind) if kind in ("forward", "backward_data", "backward_weight") else 0.0
            d = lay.setdefault((unit.name, kind), [0.0, fl, unit.algorithmic_bytes(kind, nstreams),
                                                   kernel_name(kind, unit, model._lib)])
            d[0] += e0.elapsed_time(e1) / PROF_STEPS
        for (name, kind), (ms, fl, by, kn) in sorted(lay.items(), key=lambda kv: -kv[1][0]):
            print(f"{name:28s} {kind:16s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TF/s  {by / ms / 1e6:8.1f} GB/s  {kn}",
                  file=sys.stderr)
    conv = {k: v for k, v in per.items() if v["flop"] > 0}
    # fp32: the step is matrix-core bound (AI ~100 FLOP/B vs a ridge of 20): dominant = the convolution kernel with
    # the most time, priced in TFLOP/s.  bf16: HBM-bound by the survey's accounting (ridge 310 FLOP/B): dominant =
    # whichever kernel takes the most time, priced in algorithmic bytes per second (its matrix-core rate beside it).
    hbm = dtype == "bf16"
    pool = per if hbm else conv
    dom = max(pool, key=lambda k: pool[k]["ms"])
    d = per[dom]
    conv_ms = sum(v["ms"] for v in conv.values()) / PROF_STEPS
    serial_ms = sum(v["ms"] for v in per.values()) / PROF_STEPS
    tflops_dom = d["flop"] / (d["ms"] * 1e-3) / 1e12
    gbs_dom = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    mfma_peak = PEAK_BF16_MFMA_TFLOPS if "bf16" in dom else PEAK_FP32_MFMA_TFLOPS
    if hbm:
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(gbs_dom, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs_dom / PEAK_HBM_GBS, 4), "traffic": _traffic(dom, dtype),
                    "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                    "mfma_tflops": round(tflops_dom, 1), "mfma_frac": round(tflops_dom / mfma_peak, 4)}
    else:
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(tflops_dom, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tflops_dom / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": _traffic(dom, dtype),
                    "flop_per_launch": d["flop"] / d["launches"],
                    "algorithmic_bytes_per_launch": d["bytes"] / d["launches"]}
    all_bytes = sum(v["bytes"] for v in per.values()) / PROF_STEPS
    step_s = dt / steps
    step_flop = n * 61.4e9 * (args.tile / 512) ** 2                  # 3 x forward convention
    step_bytes = n * (310e6 if hbm else 615e6) * (args.tile / 512) ** 2   # fused-ideal activation traffic
    top = sorted(per.items(), key=lambda kv: -kv[1]["ms"])[:14]
    roofline.update({
        "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // PROF_STEPS,
        "share_of_step": round(d["ms"] / PROF_STEPS / (step_s * 1e3), 3),
        "measured_on": f"{PROF_STEPS} steps of the serial schedule run right after the timed region (there weight "
                       "gradients overlap the backward chain on a second stream: per-launch times are not kernel times)",
        "serial_kernels_ms_per_step": round(serial_ms, 2), "conv_kernels_ms_per_step": round(conv_ms, 2),
        # Sigma of the counter-derived HBM bytes of ONE step of this dtype (every dispatch) over the survey's fused-ideal bytes
        "whole_step": {"traffic_bytes": (_pmc_leg(dtype) or {}).get("step_bytes"),
                       "traffic_ratio": (round(_pmc_leg(dtype)["step_bytes"] / step_bytes, 3) if _pmc_leg(dtype) else None),
                       "tflops": round(step_flop / step_s / 1e12, 1),
                       "mfma_frac": round(step_flop / step_s / 1e12 / (PEAK_BF16_MFMA_TFLOPS if hbm else PEAK_FP32_MFMA_TFLOPS), 4),
                       "algorithmic_GBs": round(step_bytes / step_s / 1e9, 1),
                       "hbm_frac": round(step_bytes / step_s / 1e9 / PEAK_HBM_GBS, 4),
                       "algorithmic_GBs_timed_kernels": round(all_bytes / step_s / 1e9, 1)},
        "top_kernels": {k: {"ms_per_step": round(v["ms"] / PROF_STEPS, 3), "launches": v["launches"] // PROF_STEPS,
                            "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2),
                            "GBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in top}})
    leg = {
        "metric": "cvae_train_tiles_per_sec", "value": round(world * n * steps / dt, 2), "unit": "tiles/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(step_s * 1e3, 3),
        "dtype": dtype,
        "config": {"workload": f"CVAE fiducial train step (fwd+bwd+Adam), batch {n}/GPU of "
                               f"{args.tile}x{args.tile} tiles, " + (
                                   "fp32 (BASELINE.json configs[1])" if dtype == "f32" else
                                   "bf16 activations/gradients in the generator trunk, fp32 accumulation, master "
                                   "weights and statistics (BASELINE.json configs[3] per GPU)"),
                   "tile": args.tile, "batch_per_gpu": n, "global_batch": n * world,
                   "parallelism": f"dp{world}",
                   "batch_norm": "single device" if world == 1 else
                                 ("local statistics (--local-bn)" if args.local_bn else "global (all-reduced statistics)"),
                   "optimizer": "torch.optim.Adam(lr=1e-3)" if args.torch_adam
                                else "FlatAdam(lr=1e-3) = torch.optim.Adam arithmetic, fused",
                   "schedule": ("weight gradients on a second HIP stream beside the rest of the backward pass"
                                + ("; the whole step replayed from one hipGraph" if use_graph else ""))
                               if os.environ.get("BP_SIDE_WGRAD", "1") != "0" else "single stream",
                   "final_elbo": final_elbo},
        "roofline": roofline, "paint": paint_leg}
    if coll is not None:
        leg["config"].update(coll)
    del model, opt, plan, prof_events
    _free()
    return leg


def _paint_leg(args, model, dtype, dev, world, rank, n):
    """paint() (SURVEY.md 8d metric (B), BASELINE.json configs[4]): RAW host tiles in -> physical host tiles out through
    CVAEPainter.paint_stream -- device-side transforms, Philox per-tile prior noise, one captured hipGraph per batch
    (prior + sampler + generator on four streams), pinned double-buffered H2D / D2H on side streams.  This rank
    paints its contiguous share of the tiles; no collective."""
    from baryon_painter_amd.painter import CVAEPainter
    from baryon_painter_amd.utils.datasets import SyntheticTileDataset
    model.train(False)
    # tiles per captured graph: independent of the training batch -- eval-mode tiles do not couple, and 128 (fp32) / 256
    # (bf16) tiles per replay amortise the graph's ~70 launch gaps (bf16: 22.8k -> 25.0k tiles/s resident)
    pb = args.paint_batch if args.paint_batch > 0 else (256 if dtype == "bf16" else 128)
    ds = SyntheticTileDataset(n_sample=8, tile_size=args.tile, seed=3)
    pt = CVAEPainter.__new__(CVAEPainter)
    pt.model, pt.compute_device, pt.sync = model, dev, None
    pt.input_field, pt.label_fields = ds.input_field, ds.label_fields
    pt.transform, pt.inverse_transform = ds.transform, ds.inverse_transform
    # (per rank; under N > 1 every rank pins its own in / out buffers: 2 x 2 GB instead of 2 x 4 GB)
    n_paint = args.paint_tiles if world == 1 else min(args.paint_tiles, 2048)
    raw = np.stack([ds.raw_fields(i)[0] for i in range(8)])
    zs = np.array([ds.raw_fields(i)[2] for i in range(8)])
    reps_p = (n_paint + 7) // 8
    tin = torch.from_numpy(np.tile(raw, (reps_p, 1, 1))[:n_paint]).pin_memory()
    zin = np.tile(zs, reps_p)[:n_paint]
    tout = torch.empty((n_paint, args.tile, args.tile), dtype=torch.float32).pin_memory()
    ids = np.arange(n_paint, dtype=np.int64) + rank * n_paint
    with torch.no_grad():
        pt.paint_stream(tin[:2 * pb], zin[:2 * pb], batch_size=pb, tile_ids=ids[:2 * pb], out=tout[:2 * pb])   # capture
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        pt.paint_stream(tin, zin, batch_size=pb, tile_ids=ids, out=tout)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        tp = _reduce_max_time(time.perf_counter() - t0p, dev, world)
        # the captured forward alone, tiles resident in HBM (no transforms, no copies): the kernel-side ceiling
        g = model.paint_graph(pb)
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        for _ in range(10):
            g["graph"].replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0p) / 10
    # every painted tile is finite and positive-definite pressure; tiles 0 and 8 are the same raw tile with different
    # Philox streams (tile ids): they must differ, and differ little (the prior noise is a small modulation)
    o = tout.numpy()
    assert np.isfinite(o).all() and (o >= 0).all(), "paint_stream produced non-finite / negative pressure"
    if n_paint > 8:
        rel = np.abs(o[0] - o[8]).sum() / max(np.abs(o[0]).sum(), 1e-30)
        assert 0 < rel < 1.0, f"tiles 0 and 8 (same input, different noise stream) differ by {rel}"
    alg = 190e6 * (args.tile / 512) ** 2 * (0.5 if dtype == "bf16" else 1.0)   # fused-ideal
    #                                                      activation bytes of one painted tile (fp32; half in bf16)
    flop = 20.25e9 * (args.tile / 512) ** 2      # paint = prior + generator forward
    rate = n_paint / tp
    leg = {"metric": "paint_tiles_per_sec", "value": round(world * rate, 1), "unit": "tiles/s",
           "tiles": n_paint, "batch": pb, "ms_per_batch": round(tp / (n_paint / pb) * 1e3, 3),
           "resident_tiles_per_sec": round(world * pb / tg, 1), "ms_per_batch_graph_only": round(tg * 1e3, 3),
           # fp32 paint is matrix-core bound like the fp32 train step (AI ~107 FLOP/B), bf16 paint HBM-bound
           "roofline": ({"bound": "hbm", "achieved": round(alg * rate / 1e9, 1), "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": round(alg * rate / 1e9 / PEAK_HBM_GBS, 4),
                         "algorithmic_bytes_per_tile": alg, "mfma_tflops": round(flop * rate / 1e12, 1)}
                        if dtype == "bf16" else
                        {"bound": "mfma", "achieved": round(flop * rate / 1e12, 1),
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flop * rate / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                         "flop_per_tile": flop, "algorithmic_bytes_per_tile": alg}),
           "pcie_bytes_per_tile": 2 * 4 * args.tile ** 2,
           "note": "host float32 raw tile in -> host float32 physical tile out (pinned memory), per rank its "
                   "own tiles, no collective; 'resident' = the captured forward alone on tiles already in HBM; "
                   "roofline per GPU"}
    model.train(True)
    return leg


# ---------------------------------------------------------------------------------------------- CGAN iteration
def cgan_leg(args, dev, world, rank, sync, steps, warmup):
    """BASELINE.json configs[2]: CGAN fiducial, alternating discriminator / generator iteration."""
    import contextlib
    from baryon_painter_amd.models.cgan import CGAN
    from baryon_painter_amd.utils import synthetic as syn
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(sys.stderr):
        model = CGAN(tile_size=args.tile, device=dev, sync=sync)
    n = args.batch
    nb = min(n, 8)
    x, y, z = syn.synthetic_batch(nb, args.tile, args.tile, seed=1234 + rank)
    reps = (n + nb - 1) // nb
    x = torch.from_numpy(np.tanh(3 * np.tile(x, (reps, 1, 1, 1))[:n] - 0.5).astype(np.float32)).to(dev)
    y = torch.from_numpy(np.tile(y, (reps, 1, 1, 1))[:n]).to(dev)
    z = torch.from_numpy(np.tile(z, reps)[:n]).to(dev)
    opt_g = torch.optim.Adam(model.g_parameters(), lr=5e-5, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(model.d_parameters(), lr=5e-5, betas=(0.5, 0.999))
    step = lambda: model.train_step(x, y, z, opt_g, opt_d)
    dt, losses = _timed(step, steps, warmup, world)
    dt = _reduce_max_time(dt, dev, world)
    losses = {k: float(v) for k, v in losses.items()}
    # one further iteration of the serial schedule with HIP events around every launch (as the CVAE leg)
    plan = model._plan(n)
    side, plan.side = plan.side, None
    plan.prof = []
    step()
    torch.cuda.synchronize()
    ev, plan.prof = plan.prof, None
    plan.side = side
    per = {}
    for e0, e1, unit, kind, nstreams in ev:
        name = kernel_name(kind, unit, model._lib) + (
            f"[{unit.name} {kind}]" if kind in ("forward", "backward_data", "backward_weight") else "")
        d = per.setdefault(name, {"ms": 0.0, "launches": 0, "flop": 0.0, "bytes": 0.0})
        d["ms"] += e0.elapsed_time(e1)
        d["launches"] += 1
        if kind in ("forward", "backward_data", "backward_weight"):
            d["flop"] += 2.0 * unit.macs(kind)
        d["bytes"] += unit.algorithmic_bytes(kind, nstreams)
    conv = {k: v for k, v in per.items() if v["flop"] > 0}
    dom = max(conv, key=lambda k: conv[k]["ms"])
    d = per[dom]
    tf = d["flop"] / (d["ms"] * 1e-3) / 1e12
    step_s = dt / steps
    flop_step = sum(v["flop"] for v in conv.values())          # counted from the launches of one iteration
    top = sorted(per.items(), key=lambda kv: -kv[1]["ms"])[:10]
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                "flop_per_launch": d["flop"] / d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                "launches_per_step": d["launches"], "share_of_step": round(d["ms"] / (step_s * 1e3), 3),
                "measured_on": "one iteration of the serial schedule right after the timed region",
                "serial_kernels_ms_per_step": round(sum(v["ms"] for v in per.values()), 2),
                "whole_step": {"tflops": round(flop_step / step_s / 1e12, 1),
                               "mfma_frac": round(flop_step / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                               "flop_per_tile_counted": flop_step / n,
                               "flop_per_tile_survey": CGAN_FLOP_PER_TILE * (args.tile / 512) ** 2},
                "top_kernels": {k: {"ms_per_step": round(v["ms"], 3), "launches": v["launches"],
                                    "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in top}}
    leg = {"metric": "cgan_train_tiles_per_sec", "value": round(world * n * steps / dt, 2), "unit": "tiles/s",
           "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(step_s * 1e3, 3), "dtype": "f32",
           "config": {"workload": f"CGAN fiducial alternating D+G iteration, batch {n} of {args.tile}x{args.tile} tiles, "
                                  "fp32 (BASELINE.json configs[2]); parity vs own restatement only (no reference code)",
                      "parallelism": f"dp{world}", "global_batch": n * world, "losses": losses},
           "roofline": roofline}
    del model, opt_g, opt_d, plan, ev
    _free()
    return leg


# ---------------------------------------------------------------------------------------------- driver
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (nothing in this
    process has touched the GPU) and pass its output through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _compact(leg):
    """The few numbers of a secondary leg that config['other_configs'] repeats."""
    if not isinstance(leg, dict) or "value" not in leg:
        return leg
    r = leg.get("roofline") or {}
    out = {"value": leg["value"], "unit": leg["unit"]}
    if "ms_per_step" in leg:
        out["ms_per_step"] = leg["ms_per_step"]
    if "ms_per_batch" in leg:
        out["ms_per_batch"] = leg["ms_per_batch"]
        out["resident_tiles_per_sec"] = leg["resident_tiles_per_sec"]
    out["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic") if k in r}
    if "whole_step" in r:
        out["roofline"]["whole_step"] = r["whole_step"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="tiles per GPU")
    ap.add_argument("--tile", type=int, default=TILE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--local-bn", action="store_true", help="do not all-reduce batch-norm statistics")
    ap.add_argument("--layers", action="store_true", help="also print a per-layer table to stderr")
    ap.add_argument("--no-paint", action="store_true", help="skip the paint() throughput legs")
    ap.add_argument("--paint-tiles", type=int, default=4096,
                    help="tiles streamed through paint() per rank (configs[4] streams 100k: the un-overlapped first upload / "
                         "last download of the pipeline are 8 %% of a 1024-tile run and 2 %% of this one)")
    ap.add_argument("--paint-batch", type=int, default=0, help="tiles per captured paint graph (0: 128 for fp32, 256 for bf16)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused FlatAdam")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step (forward + backward + Adam, same launches) from one hipGraph "
                         "(CVAE.make_graphed_train_step; single GPU).  Measured SLOWER than the eager schedule at "
                         "batch 64 x 512^2; it pays at the reference's small minibatches")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="headline arithmetic. f32: the reference's (configs[1]); bf16: bf16 activations / gradients "
                         "in the generator trunk, fp32 accumulation, master weights and statistics (configs[3])")
    ap.add_argument("--workload", choices=["cvae", "cgan", "paint"], default="cvae",
                    help="headline workload. cvae: BASELINE.json configs[1]; cgan: configs[2] (alternating D/G step); "
                         "paint: configs[4] per GPU alone (paint_stream on a freshly built model: no training kernel runs -- "
                         "what tools/prof_paint.sh profiles)")
    ap.add_argument("--legs", default=None,
                    help="comma list of secondary legs beside the headline: bf16,cgan (default: bf16,cgan on one GPU, "
                         "bf16 on several; 'none' for the headline alone)")
    ap.add_argument("--secondary-steps", type=int, default=8, help="timed steps of the bf16 leg (CGAN: a quarter)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                 # (before anything touches the GPU)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("BP_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0                                             # rehearsal: all ranks share GPU 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    sync = None
    if world > 1 or os.environ.get("BP_SYNC_FORCE") == "1":     # (=1: one rank drives the data-parallel schedule)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BP_DIST_BACKEND", "nccl")       # "gloo" only to rehearse on a one-GPU box
        if backend == "nccl" and os.environ.get("BP_PG_EAGER") != "1":
            # LAZY communicator creation (no device_id=): with the eager form every all-reduce of the step costs ~80 us
            # more (measured with one rank driving the schedule through RCCL, tools/rccl_one_rank.sh: 47.5 vs 43.8 ms
            # per fp32 step, 44 statistics collectives; the kernels' own time inside the collectives is the same
            # 0.65-0.75 ms either way) -- the host no longer runs ahead of the GPU.  torch.cuda.set_device above
            # already pins the rank's device.
            dist.init_process_group("nccl", rank=rank, world_size=world)
        elif backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        from baryon_painter_amd.dist import Sync
        sync = Sync(sync_bn=not args.local_bn)

    if args.legs is None:
        legs = ["bf16", "cgan"] if world == 1 else ["bf16"]
    else:
        legs = [s for s in args.legs.split(",") if s and s != "none"]
    t_start = time.time()

    if args.workload == "paint":
        import contextlib
        from baryon_painter_amd.models import arch as A
        from baryon_painter_amd.models.cvae import CVAE
        torch.manual_seed(1234)
        with contextlib.redirect_stdout(sys.stderr):
            model = CVAE(A.fiducial_architecture(args.tile), dev, sync=None, dtype=args.dtype)
        leg = _paint_leg(args, model, args.dtype, dev, world, rank, args.batch)
        if rank == 0:
            leg.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                        "data": "synthetic (8 distinct raw tiles per rank, repeated; random-init weights)",
                        "config": {"workload": f"paint_stream, {args.paint_tiles} raw {args.tile}x{args.tile} tiles per GPU through "
                                               "the CVAE fiducial prior + generator (random-init weights), batch "
                                               f"{min(args.batch, 64)}, host tile in -> host tile out"}})
            print(json.dumps(leg), flush=True)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    if args.workload == "cgan":
        head = cgan_leg(args, dev, world, rank, sync, args.steps, args.warmup)
        legs = []
    else:
        head = cvae_leg(args, args.dtype, dev, world, rank, sync, args.steps, args.warmup, paint=not args.no_paint)
    extra = {}
    for name in legs:
        try:
            if name == "bf16" and not (args.workload == "cvae" and args.dtype == "bf16"):
                extra["bf16"] = cvae_leg(args, "bf16", dev, world, rank, sync, args.secondary_steps, 2,
                                         paint=not args.no_paint)
            elif name == "cgan" and args.workload != "cgan":
                extra["cgan"] = cgan_leg(args, dev, world, rank, sync, max(2, args.secondary_steps // 4), 1)
        except Exception as e:                      # a secondary leg must not cost the headline
            import traceback
            traceback.print_exc()
            if world > 1:
                # ... but under N > 1 the other ranks are inside this leg's collectives and would wait for this rank for
                # ever: leave loudly instead (the launcher then ends every rank; a hang would outlive the driver's clock)
                sys.stderr.write(f"rank {rank}: secondary leg '{name}' failed under world size {world}: aborting all ranks\n")
                sys.stderr.flush()
                os._exit(3)
            extra[name] = {"error": f"{type(e).__name__}: {e}"}
            _free()

    if rank == 0:
        out = {k: head[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}
        out.update({"higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"],
                    "data": "synthetic (8 distinct log-normal field tiles per rank, repeated to fill the batch; random-init weights)",
                    "config": head["config"]})
        other = {}
        if head.get("paint"):
            other["paint (configs[4] per GPU)"] = _compact(head["paint"])
        if "bf16" in extra:
            other["bf16 (configs[3] per GPU)"] = _compact(extra["bf16"])
            if isinstance(extra["bf16"], dict) and extra["bf16"].get("paint"):
                other["paint bf16"] = _compact(extra["bf16"]["paint"])
        if "cgan" in extra:
            other["cgan (configs[2])"] = _compact(extra["cgan"])
        if other:
            out["config"]["other_configs"] = other
        # (the per-kernel table first, the numbers a reader wants last: a log tail keeps the end of the line)
        out["roofline"] = head["roofline"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if head.get("paint") is not None:
            out["paint"] = head["paint"]
        out.update(extra)
        out["wall_s"] = round(time.time() - t_start, 1)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
