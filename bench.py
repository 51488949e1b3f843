#!/usr/bin/env python3
"""Headline benchmark: CVAE fiducial train step (forward + backward + Adam) on synthetic
512x512 dark-matter -> pressure tiles, fp32, 64 tiles per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

N>1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL); the batch is
sharded data-parallel (weak scaling: 64 tiles per GPU), gradients are all-reduced as one flat
buffer and batch-norm statistics are all-reduced per layer (global-batch arithmetic).
Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- the dominant kernel's achieved fp32 MFMA rate from HIP events recorded around
                  every convolution launch inside the timed region,
  cpu_baseline -- oracle/torch_ref.py (the reference's torch.nn.functional graph) timed on the
                  host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a streaming copy reaches
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
        return ("wgrad_bf16_kernel[%s]" if kind == "backward_weight" else
                "igemm_bf16_kernel[%s fwd]" if kind == "forward" else "igemm_bf16_kernel[%s dgrad]") % tag
    if kind == "backward_weight":
        cv = unit.cv
        cx, cy = (cv.cout, cv.cin) if cv.transposed else (cv.cin, cv.cout)
        if (cv.transposed, cv.cin, cv.cout, cv.k, cv.stride, cv.pad) == (0, 3, 16, 5, 1, 2):
            return "stem_wgrad_kernel"
        if (cv.transposed, cv.cin, cv.cout, cv.k, cv.stride, cv.pad) == (0, 16, 8, 7, 1, 3):
            return "wgrad_flat_kernel"
        if (cx, cy, cv.k, cv.stride, cv.pad) == (16, 32, 4, 2, 1):
            return "wgrad_flat_s2_kernel"
        if cx <= 16 and cy <= 16 and not (cx == 16 and cy == 16) and (cv.k, cv.stride) in ((3, 1), (5, 1), (7, 1), (4, 2), (8, 4)):
            return "wgrad_small_kernel[k%ds%d %d,%d]" % (cv.k, cv.stride, cx, cy)
        return "wgrad_tiles_kernel[k%ds%d %s]" % (cv.k, cv.stride, "wide" if (cx > 16 and cy > 16) else "thin")
    cv = unit.cv
    if kind == "backward_data" and getattr(unit, "_sub", None) is not None:
        cv = unit._sub["cv"]           # data gradient restricted to a channel slice
    kid = lib.bp_conv_kernel_id(C.byref(cv), L.PACK_FWD if kind == "forward" else L.PACK_BWD)
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


def bench_cgan(args, dev, world, rank, sync=None):
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
    for _ in range(args.warmup):
        model.train_step(x, y, z, opt_g, opt_d)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = model.train_step(x, y, z, opt_g, opt_d)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    print(json.dumps({
        "metric": "cgan_train_tiles_per_sec", "value": round(world * n * args.steps / dt, 2), "unit": "tiles/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"CGAN fiducial alternating D+G iteration, batch {n} of {args.tile}x{args.tile} tiles, fp32 "
                               "(BASELINE.json configs[2]); parity vs own restatement only (no reference code)",
                   "parallelism": f"dp{world}", "global_batch": n * world,
                   "losses": {k: float(v) for k, v in losses.items()}}}), flush=True)


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
    ap.add_argument("--no-paint", action="store_true", help="skip the paint() throughput leg")
    ap.add_argument("--paint-tiles", type=int, default=1024, help="tiles streamed through paint() per rank")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused FlatAdam")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step (forward + backward + Adam, same launches) from one hipGraph "
                         "(CVAE.make_graphed_train_step; single GPU).  Measured SLOWER than the eager schedule at "
                         "batch 64 x 512^2 (fp32 48.5 vs 46.2 ms, bf16 24.8 vs 23.6: the graph re-packs every layer's "
                         "weights and the host is never the bottleneck); it pays at the reference's small minibatches")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: the reference's arithmetic (configs[1], the headline); bf16: bf16 activations / gradients "
                         "in the generator trunk, fp32 accumulation, master weights and statistics (configs[3])")
    ap.add_argument("--workload", choices=["cvae", "cgan"], default="cvae",
                    help="cvae: BASELINE.json configs[1] (the headline); cgan: configs[2] (alternating D/G step)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
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
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        from baryon_painter_amd.dist import Sync
        sync = Sync(sync_bn=not args.local_bn)

    from baryon_painter_amd.models import arch as A
    from baryon_painter_amd.models.cvae import CVAE
    from baryon_painter_amd.utils import synthetic as syn

    if args.workload == "cgan":
        return bench_cgan(args, dev, world, rank, sync)

    arch = A.fiducial_architecture(args.tile)
    torch.manual_seed(1234)                      # same initial weights on every rank
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the model announces itself like the reference does
        model = CVAE(arch, dev, sync=sync, dtype=args.dtype)
    from baryon_painter_amd.optim import FlatAdam
    opt = FlatAdam(model, lr=1e-3) if not args.torch_adam else torch.optim.Adam(model.parameters(), lr=1e-3)
    n = args.batch
    # synthetic tiles: a few distinct ones, tiled up to the batch (generation cost, not arithmetic)
    nb = min(n, 8)
    x, y, aux = syn.synthetic_batch(nb, args.tile, args.tile, seed=1234 + rank)
    reps = (n + nb - 1) // nb
    x = torch.from_numpy(np.tile(x, (reps, 1, 1, 1))[:n]).to(dev)
    y = torch.from_numpy(np.tile(y, (reps, 1, 1, 1))[:n]).to(dev)
    aux = torch.from_numpy(np.tile(aux, reps)[:n]).to(dev)

    if os.environ.get("BP_MAIN_PRIORITY"):           # experiment: the main chain on a stream of another priority
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["BP_MAIN_PRIORITY"])))

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

    for _ in range(args.warmup):
        step()
    plan = model._last
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        elbo = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_elbo = float(elbo.detach())

    # ---- collectives (N > 1): two further steps of the timed schedule with HIP events around every all-reduce
    coll = None
    if sync is not None:
        n0, g0, b0 = sync.n_small, sync.n_grad, sync.bytes_grad
        sync.timing = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ev, sync.timing = sync.timing, None
        ms = {}
        for e0, e1, kind in ev:
            ms[kind] = ms.get(kind, 0.0) + e0.elapsed_time(e1) / 2
        coll = {"collectives_per_step": {"batch_norm_statistics": (sync.n_small - n0) // 2,
                                         "gradient_buffers": (sync.n_grad - g0) // 2},
                "gradient_bytes_per_step": (sync.bytes_grad - b0) // 2,
                "ms_per_step_inside_collectives": {k: round(v, 3) for k, v in ms.items()},
                "gradient_all_reduce": "generator trunk + heads (97 % of the bytes) on the weight-gradient stream right "
                                       "behind the trunk's last weight gradient, own communicator; the rest after the "
                                       "backward pass" if sync.overlap else "one flat buffer after the backward pass",
                "backend": torch.distributed.get_backend()}

    # ---- kernel times for the roofline: PROF_STEPS further steps of the SERIAL schedule (not timed above).
    # In the timed region the weight gradients run on a second stream and share the CUs with the data gradients
    # and the batch-norm passes (that is where 7 % of the throughput comes from), so HIP events around a launch
    # there measure "this kernel while sharing the GPU", not the kernel.  Serially every launch has the GPU to
    # itself; the events sit on the stream the kernels are launched on (torch's current stream).
    PROF_STEPS = 2
    model.overlap_weight_gradients(False)
    plan.prof = []                               # HIP events around every convolution launch
    for _ in range(PROF_STEPS):
        eager_step()                             # (eager: the events are recorded by the launch hooks)
    torch.cuda.synchronize()
    prof_events, plan.prof = plan.prof, None
    model.overlap_weight_gradients(True)

    # ---- paint() (SURVEY.md 8d metric (B), BASELINE.json configs[4]): RAW host tiles in -> physical host tiles out
    # through CVAEPainter.paint_stream -- device-side transforms, Philox per-tile prior noise, one captured hipGraph
    # per batch (prior + sampler + generator on four streams), pinned double-buffered H2D / D2H on side streams.
    # This rank paints its contiguous share of the tiles; no collective.
    paint_leg = None
    if not args.no_paint:
        from baryon_painter_amd.painter import CVAEPainter
        from baryon_painter_amd.utils.datasets import SyntheticTileDataset
        model.train(False)
        pb = min(n, 64)
        ds = SyntheticTileDataset(n_sample=8, tile_size=args.tile, seed=3)
        pt = CVAEPainter.__new__(CVAEPainter)
        pt.model, pt.compute_device, pt.sync = model, dev, None
        pt.input_field, pt.label_fields = ds.input_field, ds.label_fields
        pt.transform, pt.inverse_transform = ds.transform, ds.inverse_transform
        n_paint = args.paint_tiles
        raw = np.stack([ds.raw_fields(i)[0] for i in range(8)])
        zs = np.array([ds.raw_fields(i)[2] for i in range(8)])
        reps_p = (n_paint + 7) // 8
        tin = torch.from_numpy(np.tile(raw, (reps_p, 1, 1))[:n_paint]).pin_memory()
        zin = np.tile(zs, reps_p)[:n_paint]
        tout = torch.empty((n_paint, args.tile, args.tile), dtype=torch.float32).pin_memory()
        ids = np.arange(n_paint, dtype=np.int64) + rank * n_paint
        with torch.no_grad():
            pt.paint_stream(tin[:2 * pb], zin[:2 * pb], batch_size=pb, tile_ids=ids[:2 * pb], out=tout[:2 * pb])   # capture
            torch.cuda.synchronize()
            t0p = time.perf_counter()
            pt.paint_stream(tin, zin, batch_size=pb, tile_ids=ids, out=tout)
            tp = time.perf_counter() - t0p
            # the captured forward alone, tiles resident in HBM (no transforms, no copies): the kernel-side ceiling
            g = model.paint_graph(pb)
            torch.cuda.synchronize()
            t0p = time.perf_counter()
            for _ in range(10):
                g["graph"].replay()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t0p) / 10
        assert np.isfinite(tout[-1].numpy()).all()
        alg = 190e6 * (args.tile / 512) ** 2 * (0.5 if args.dtype == "bf16" else 1.0)   # fused-ideal
        #                                                      activation bytes of one painted tile (fp32; half in bf16)
        flop = 20.25e9 * (args.tile / 512) ** 2      # paint = prior + generator forward
        paint_leg = {"metric": "paint_tiles_per_sec", "value": round(world * n_paint / tp, 1), "unit": "tiles/s",
                     "tiles": n_paint, "batch": pb, "ms_per_batch": round(tp / (n_paint / pb) * 1e3, 3),
                     "resident_tiles_per_sec": round(world * pb / tg, 1), "ms_per_batch_graph_only": round(tg * 1e3, 3),
                     # fp32 paint is matrix-core bound like the fp32 train step (AI ~107 FLOP/B), bf16 paint HBM-bound
                     "roofline": ({"bound": "hbm", "achieved": round(alg * n_paint / tp / 1e9, 1), "peak": PEAK_HBM_GBS,
                                   "unit": "GB/s", "frac": round(alg * n_paint / tp / 1e9 / PEAK_HBM_GBS, 4),
                                   "algorithmic_bytes_per_tile": alg} if args.dtype == "bf16" else
                                  {"bound": "mfma", "achieved": round(flop * n_paint / tp / 1e12, 1),
                                   "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(flop * n_paint / tp / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "flop_per_tile": flop, "algorithmic_bytes_per_tile": alg}),
                     "pcie_bytes_per_tile": 2 * 4 * args.tile ** 2,
                     "note": "host float32 raw tile in -> host float32 physical tile out (pinned memory), per rank its "
                             "own tiles, no collective; 'resident' = the captured forward alone on tiles already in "
                             "HBM"}
        model.train(True)

    if rank == 0:
        # ---- roofline of the dominant kernel
        per = {}
        for e0, e1, unit, kind, nstreams in prof_events:
            name = kernel_name(kind, unit, model._lib)
            d = per.setdefault(name, {"ms": 0.0, "launches": 0, "flop": 0.0, "bytes": 0.0})
            d["ms"] += e0.elapsed_time(e1)
            d["launches"] += 1
            if kind in ("forward", "backward_data", "backward_weight"):
                d["flop"] += 2.0 * unit.macs(kind)
            d["bytes"] += unit.algorithmic_bytes(kind, nstreams)
        if args.layers:
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
        # the most time, priced in TFLOP/s.  bf16: HBM-bound (ridge 310 FLOP/B): dominant = whichever kernel takes the
        # most time, priced in algorithmic bytes per second.
        hbm = args.dtype == "bf16"
        pool = per if hbm else conv
        dom = max(pool, key=lambda k: pool[k]["ms"])
        d = per[dom]
        conv_ms = sum(v["ms"] for v in conv.values()) / PROF_STEPS
        traffic = None
        try:       # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside bench.py); only quoted when
            #        the stamp says the counters were collected on these very kernel sources
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pj.get("source_hash") == source_hash():
                traffic = pj["hbm_bytes_per_launch"].get(dom)
        except Exception:
            pass
        if hbm:
            achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": d["bytes"] / d["launches"]}
        else:
            achieved = d["flop"] / (d["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                        "flop_per_launch": d["flop"] / d["launches"]}
        all_bytes = sum(v["bytes"] for v in per.values()) / PROF_STEPS
        roofline.update({
            "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // PROF_STEPS,
            "share_of_step": round(d["ms"] / PROF_STEPS / (dt / args.steps * 1e3), 3),
            "measured_on": f"{PROF_STEPS} steps of the serial schedule (BP_SIDE_WGRAD=0 equivalent) run right after "
                           "the timed region; in the timed region weight gradients overlap the rest of the "
                           "backward pass on a second stream, so per-launch times there are not kernel times",
            "all_conv_kernels_ms_per_step": round(conv_ms, 2),
            "whole_step": {"tflops": round(n * 61.4e9 * (args.tile / 512) ** 2 / (dt / args.steps) / 1e12, 1),
                           "algorithmic_GBs_timed_kernels": round(all_bytes / (dt / args.steps) / 1e9, 1)},
            "per_kernel": {k: {"ms_per_step": round(v["ms"] / PROF_STEPS, 3),
                               "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2),
                               "GBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                           for k, v in sorted(per.items(), key=lambda kv: -kv[1]["ms"])}})
        out = {
            "metric": "cvae_train_tiles_per_sec", "value": round(world * n * args.steps / dt, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"CVAE fiducial train step (fwd+bwd+Adam), batch {n}/GPU of "
                                   f"{args.tile}x{args.tile} tiles, " + (
                                       "fp32 (BASELINE.json configs[1])" if args.dtype == "f32" else
                                       "bf16 activations/gradients in the generator trunk, fp32 accumulation, master "
                                       "weights and statistics (BASELINE.json configs[3] per GPU)"),
                       "tile": args.tile, "batch_per_gpu": n, "global_batch": n * world,
                       "parallelism": f"dp{world}", "batch_norm": "local" if args.local_bn or world == 1 and False
                       else ("global (all-reduced statistics)" if world > 1 else "single device"),
                       "optimizer": "torch.optim.Adam(lr=1e-3)" if args.torch_adam else "FlatAdam(lr=1e-3) = torch.optim.Adam arithmetic, fused",
                       "schedule": "weight gradients on a second HIP stream beside the rest of the backward pass"
                                   + ("; the whole step replayed from one hipGraph" if use_graph else "")
                                   if os.environ.get("BP_SIDE_WGRAD", "1") != "0" else "single stream",
                       "final_elbo": final_elbo},
            "roofline": roofline,
        }
        out["paint"] = paint_leg
        if coll is not None:
            out["config"].update(coll)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
