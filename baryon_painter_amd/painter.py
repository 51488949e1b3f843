"""Painter API: train / validate / paint / checkpoint, with the reference's signatures.

Mirrors ``baryon_painter.painter`` (/root/reference/baryon_painter/painter.py:16-545).  The loop
semantics that affect results are kept: pseudo-epoch bookkeeping (one ``scheduler.step()`` per
pseudo epoch, painter.py:179-190), adaptive batch size by rebuilding the DataLoader
(:210-215), validation losses computed under ``no_grad`` but in TRAIN mode (:85,306-314 -- they
update batch-norm running statistics, as in the reference), ``paint`` switching to eval mode
for good (:372), file names and the text format of the statistics logs (:127-131,476-484).
Out of scope and therefore absent: the matplotlib / cosmotools diagnostics
(``validation_plotting``); ``show_plots`` / ``save_plots`` / ``plot_*`` arguments are accepted
and ignored, and the two reference crashes they trigger do not occur.
"""
import collections
import os
import pickle

import numpy as np
import torch
import torch.utils.data

from .models import cvae as _cvae
from .utils import datasets

try:                                    # the reference pickles its metadata with dill
    import dill as _pickler
except ImportError:                     # pragma: no cover
    _pickler = pickle


class Painter:
    """Abstract base class for a baryon painter (painter.py:16-31)."""

    def __init__(self):
        raise NotImplementedError("This is an abstract base class.")

    def load_state_from_file(self, filename):
        raise NotImplementedError("This is an abstract base class.")

    def paint(self, input, **kwargs):
        raise NotImplementedError("This is an abstract base class.")


class CVAEPainter(Painter):
    def __init__(self, filename=None, training_data_set=None, test_data_set=None, architecture="test",
                 compute_device="cuda:0", sync=None, dtype="f32"):
        """The reference's arguments (painter.py:34-38) plus ``sync`` (data parallel, baryon_painter_amd.dist.Sync) and
        ``dtype``: "f32" = the reference's arithmetic, "bf16" = the bf16 throughput mode of ``models.cvae.CVAE``."""
        self.sync = sync
        self.dtype = dtype
        if filename is not None:
            self.load_state_from_file(filename, compute_device)
        else:
            self.architecture = architecture
            self.compute_device = compute_device
            self.model = _cvae.CVAE(architecture, torch.device(compute_device), sync=sync, dtype=dtype)
        self.training_data = training_data_set
        self.test_data = test_data_set

    def load_training_data(self, filename):
        self.data_path = os.path.dirname(filename)
        with open(filename, "rb") as f:
            self.training_data_file_info = pickle.load(f)

    def load_test_data(self, filename):
        self.test_data_path = os.path.dirname(filename)
        with open(filename, "rb") as f:
            self.test_data_file_info = pickle.load(f)

    # ------------------------------------------------------------------------------ training
    def _loader(self, batch_size):
        if self.sync is not None and self.sync.world_size > 1:
            return _ShardedLoader(self.training_data, batch_size, self.sync)
        if getattr(self, "device_assembler", None) is not None:
            return _DeviceLoader(self.device_assembler, len(self.training_data), batch_size)
        return torch.utils.data.DataLoader(self.training_data, batch_size=batch_size, shuffle=True)

    def use_device_assembly(self, k_values, mode="shift-log"):
        """Keep the training stacks in HBM and assemble batches with one gather launch per field
        (utils.datasets.DeviceTileAssembler) instead of the host DataLoader; same shuffle order."""
        self.device_assembler = datasets.DeviceTileAssembler(self.training_data, self.compute_device,
                                                             k_values=k_values, mode=mode)

    def train(self, n_epoch=5, n_pepoch=None, learning_rate=1e-4, batch_size=1,
              adaptive_learning_rate=None, adaptive_batch_size=None,
              validation_pepochs=[0, 1], validation_batch_size=4,
              validation_loss_frequency=100, validation_loss_batch_size=16,
              checkpoint_frequency=1000, statistics_report_frequency=50,
              loss_plot_frequency=1000, mavg_window_size=20,
              plot_sample_var=False, plot_power_spectra=["auto"], plot_histogram=["log"],
              show_plots=True, save_plots=False, output_path=None, verbose=True,
              pepoch_size=3136, var_anneal_fn=None, KL_anneal_fn=None, graph_step=False):
        """Train.  1 pseudo epoch = ``pepoch_size`` samples (3136 by default; the checked-in
        script uses 1568).  Returns ``(training_stats, validation_stats)``.

        ``graph_step=True`` (extension): forward + backward + Adam of full minibatches replay from one hipGraph
        (``CVAE.make_graphed_train_step``) - same arithmetic, ~2x the throughput at the reference's minibatch
        sizes of 4-24 tiles, where a step is bound by its ~450 kernel launches.  Single device only."""
        if self.training_data is None:
            raise RuntimeError("Trying to train but no training data specified.")
        if len(validation_pepochs) > 0 and self.test_data is None:
            raise RuntimeError("Trying to validate but no test data specified.")
        model = self.model
        model.train(True)
        if adaptive_batch_size is not None or batch_size <= 0:
            batch_size = adaptive_batch_size(0)
        dataloader = self._loader(batch_size)

        # torch.optim.Adam arithmetic, one fused launch over the flat parameter buffer
        from .optim import FlatAdam
        optimizer = FlatAdam(model, lr=learning_rate)
        scheduler = None
        if adaptive_learning_rate is not None:
            if callable(adaptive_learning_rate):
                scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, adaptive_learning_rate)
            elif isinstance(adaptive_learning_rate, dict):
                scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=adaptive_learning_rate["step_size"],
                                                            gamma=adaptive_learning_rate["gamma"])
            elif adaptive_learning_rate == "avoid_plateau":
                scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="max", factor=0.1,
                                                                       patience=10, threshold=0.0001,
                                                                       threshold_mode="rel", cooldown=0, min_lr=0,
                                                                       eps=1e-08)

        # statistics labels: "log_likelihood_0" -> "log_likelihood_pressure_0" (painter.py:114-121)
        nf = self.training_data.n_feature_per_field
        stats_labels = model.get_stats_labels()
        for j, f in enumerate(self.training_data.label_fields):
            for k in range(nf):
                stats_labels = [l.replace(f"{j * nf + k}", f"{f}_{k}") for l in stats_labels]
        stats_labels += ["lr", "batch_size"]

        ckpt_template = train_file = val_file = idx_file = None
        if output_path is not None:
            os.makedirs(output_path, exist_ok=True)
            ckpt_template = os.path.join(output_path, "checkpoint_sample{sample:0>10}_batch{batch}_epoch{epoch}{suffix}")
            train_file = os.path.join(output_path, "training_stats.txt")
            val_file = os.path.join(output_path, "validation_stats.txt")
            idx_file = os.path.join(output_path, "training_sample_indicies.txt")
        elif save_plots:
            raise ValueError("save_plots=True requires output_path to be set.")
        rank0 = self.sync is None or self.sync.rank == 0
        if not rank0:
            ckpt_template = train_file = val_file = idx_file = None

        training_stats = TrainingStats(stats_labels, mavg_window_size, stats_filename=train_file)
        validation_stats = TrainingStats(stats_labels, mavg_window_size, stats_filename=val_file,
                                         dump_to_file_frequency=1)
        if n_pepoch is None:
            n_pepoch = n_epoch * len(self.training_data) // pepoch_size

        sample_indices = []
        n_samples = n_batches = 0
        last_pepoch_at = last_val = last_report = last_ckpt = 0
        i_epoch = i_pepoch = i_batch = 0
        world = 1 if self.sync is None else self.sync.world_size
        ELBO = None
        graphed_steps = {}

        while i_epoch < n_epoch:
            i_epoch = n_samples // len(self.training_data)
            if verbose:
                model.check_gpu()
            if i_pepoch >= n_pepoch:
                break
            for i_batch, batch_data in enumerate(dataloader):
                if n_samples - pepoch_size >= last_pepoch_at or n_samples == 0:
                    if n_samples != 0:
                        i_pepoch += 1
                        last_pepoch_at = n_samples
                        if i_pepoch >= n_pepoch:
                            break
                        if scheduler is not None:
                            if adaptive_learning_rate == "avoid_plateau":
                                scheduler.step(float(ELBO.item()))
                            else:
                                scheduler.step()
                    if callable(var_anneal_fn):
                        model.alpha_var = var_anneal_fn(i_pepoch)
                    if callable(KL_anneal_fn):
                        model.beta_KL = KL_anneal_fn(i_pepoch)
                    if i_pepoch in validation_pepochs:
                        self.validate(validation_batch_size=validation_batch_size, plot_sample_var=plot_sample_var)
                    if adaptive_batch_size is not None:
                        new_bs = adaptive_batch_size(i_pepoch)
                        if new_bs != batch_size:
                            batch_size = new_bs
                            dataloader = self._loader(batch_size)
                            break

                x = torch.cat(batch_data[0][1:], dim=1).to(model.device)
                y = batch_data[0][0].to(model.device)
                aux = batch_data[2].to(device=model.device, dtype=y.dtype) if len(batch_data) > 2 else None

                if graph_step and self.sync is None:
                    n_b = int(y.shape[0])
                    if n_b not in graphed_steps and n_b == batch_size:
                        graphed_steps[n_b] = model.make_graphed_train_step(optimizer, n_b)
                    stepper = graphed_steps.get(n_b)
                else:
                    stepper = None
                if stepper is not None:
                    if callable(var_anneal_fn) or callable(KL_anneal_fn):
                        raise NotImplementedError("graph_step with annealed loss weights (they are captured constants)")
                    ELBO = stepper(x, y, aux)
                else:
                    ELBO = model(x, y, aux)
                    optimizer.zero_grad()
                    (-ELBO).backward()
                    optimizer.step()

                n_samples += x.size(0) * world
                n_batches += 1
                with torch.no_grad():
                    sample_indices += list(np.asarray(batch_data[1]))
                    lr = [g["lr"] for g in optimizer.param_groups]
                    training_stats.push_loss(n_samples, *model.get_stats(), lr[0], batch_size)
                    if n_samples - validation_loss_frequency >= last_val:
                        last_val = n_samples
                        stats = self.validate(validation_batch_size=validation_loss_batch_size, compute_loss=True)
                        validation_stats.push_loss(n_samples, *stats, lr[0], batch_size)
                    if n_samples - checkpoint_frequency >= last_ckpt and ckpt_template is not None:
                        last_ckpt = n_samples
                        base = ckpt_template.format(epoch=i_epoch, batch=i_batch, sample=n_samples, suffix="")
                        self.save_state_to_file((base + "_state", base + "_meta"))
                    if n_samples - statistics_report_frequency >= last_report and statistics_report_frequency > 0:
                        last_report = n_samples
                        if rank0:
                            print("Epoch: [{}/{}], P-Epoch: [{}/{}], Batch: [{}/{}], Loss: {:.3e}".format(
                                i_epoch, n_epoch, i_pepoch, n_pepoch, i_batch,
                                len(self.training_data) // (batch_size * world),
                                training_stats.loss_terms["ELBO"]["mavg"][-1]))
                            print("Processed batches: {}, processed samples: {}, batch size: {}, learning rate: {}"
                                  .format(n_batches, n_samples, batch_size, " ".join("{:.1e}".format(v) for v in lr)))
                            print(training_stats.get_pretty_str(n_col=1))
                        if idx_file is not None:
                            with open(idx_file, "wb") as f:
                                pickle.dump(sample_indices, f)

        self.validate(validation_batch_size=validation_batch_size, plot_sample_var=plot_sample_var)
        if ckpt_template is not None:
            base = ckpt_template.format(epoch=i_epoch, batch=i_batch, sample=n_samples, suffix="_final")
            self.save_state_to_file((base + "_state", base + "_meta"))
            self.save_state_to_file((os.path.join(output_path, "model_state"), os.path.join(output_path, "model_meta")))
        training_stats.flush_to_file()
        validation_stats.flush_to_file()
        return training_stats, validation_stats

    def validate(self, validation_batch_size=8, compute_loss=False, validation_redshift=None,
                 plot_samples=1, plot_sample_var=False, plot_power_spectra=["auto"], plot_histogram=["log"],
                 histogram_n_sample=1, show_plots=True, save_plots=False, filename_template="{plot_type}.png"):
        """A random test batch through the model (painter.py:295-367).  With ``compute_loss`` the
        loss terms (``model.get_stats()``) are returned; otherwise the reference draws diagnostic
        plots from a prior sample -- here the sample is drawn (same device work, same random
        stream) and returned as ``(x, y, x_pred[, x_pred_var])`` NumPy arrays instead."""
        model = self.model
        with torch.no_grad():
            fields, indicies, z = self.test_data.get_batch(size=validation_batch_size, z=validation_redshift)
            x = torch.tensor(np.concatenate(fields[1:], axis=1), device=model.device)
            y = torch.tensor(fields[0], device=model.device)
            aux = torch.tensor(z, device=model.device, dtype=y.dtype)
            if compute_loss:
                model(x, y, aux)
                return model.get_stats()
            if plot_sample_var and model.predict_var:
                x_pred, x_var = model.sample_P(y, return_var=True, aux_label=aux)
                return x.cpu().numpy(), y.cpu().numpy(), x_pred.cpu().numpy(), x_var.cpu().numpy()
            x_pred = model.sample_P(y, aux_label=aux)
            return x.cpu().numpy(), y.cpu().numpy(), x_pred.cpu().numpy()

    # ------------------------------------------------------------------------------ inference
    def paint(self, input, z=0.0, transform=True, inverse_transform=True):
        """Paint one tile (painter.py:371-392): dark-matter tile (H,W) at redshift z -> pressure."""
        self.model.train(False)
        with torch.no_grad():
            y = self.transform(input, field=self.input_field, z=z) if transform and self.transform is not None \
                else input
            y = np.asarray(y)
            y = y.reshape(1, *y.shape)
            if y.shape != (1, *self.model.dim_y):
                raise ValueError(f"Shape mismatch between input and model: {input.shape} vs {self.model.dim_y}")
            yt = torch.tensor(y, device=self.compute_device, dtype=torch.float32)
            aux = torch.tensor(z, device=self.compute_device, dtype=yt.dtype)
            prediction = self.model.sample_P(yt, aux_label=aux).cpu().numpy()
        if inverse_transform and self.inverse_transform is not None:
            if len(self.label_fields) > 1:
                raise NotImplementedError("Painting with more than one output field is not supported yet.")
            return self.inverse_transform(prediction, field=self.label_fields[0], z=z)
        return prediction

    def paint_batch(self, inputs, z, transform=True, inverse_transform=True, batch_size=64, use_graph=True):
        """Throughput form of ``paint``: many tiles (N,H,W) with redshifts (N,) in batches through the
        same eval-mode forward (BASELINE.json configs[4]); per-tile results equal ``paint``'s up to
        the prior noise draw."""
        self.model.train(False)
        inputs = np.asarray(inputs)
        zs = np.broadcast_to(np.asarray(z, dtype=np.float64), (inputs.shape[0],))
        out = []
        with torch.no_grad():
            for s in range(0, inputs.shape[0], batch_size):
                chunk = inputs[s:s + batch_size]
                zc = zs[s:s + batch_size]
                if transform and self.transform is not None:
                    y = np.stack([np.asarray(self.transform(t, field=self.input_field, z=float(zz)))
                                  for t, zz in zip(chunk, zc)])
                else:
                    y = chunk.reshape(chunk.shape[0], 1, *chunk.shape[-2:])
                y = y.reshape(y.shape[0], *self.model.dim_y)
                yt = torch.tensor(y, device=self.compute_device, dtype=torch.float32)
                aux = torch.tensor(zc, device=self.compute_device, dtype=torch.float32)
                graphed = use_graph and self.model._eps_override is None and yt.shape[0] == batch_size
                sample = self.model.sample_P_graphed if graphed else self.model.sample_P
                pred = sample(yt, aux_label=aux).cpu().numpy()
                if inverse_transform and self.inverse_transform is not None:
                    pred = np.stack([self.inverse_transform(p[None], field=self.label_fields[0], z=float(zz))
                                     for p, zz in zip(pred, zc)])
                out.append(pred)
        return np.concatenate(out, axis=0)

    # ---- throughput pipeline (BASELINE.json configs[4]) ---------------------------------------------------------
    def _shift_log_parameters(self, zs):
        """(sigma_in, k_in, k_out, sigma_out) per tile for the device-side transforms, read out of the compiled host
        transforms (the reference's "shift-log" range compression, data_transforms.py:72-97); anything else has no
        device form."""
        from .utils import data_transforms as T

        def shape_only(st):          # (by name: a transform chain restored from a checkpoint holds re-imported functions)
            return getattr(st, "__module__", None) == T.__name__ and \
                getattr(st, "__name__", None) in ("atleast_3d", "squeeze", "as_float32")

        def find(compiled, direction, field):
            if compiled is None:
                raise NotImplementedError("paint_stream needs the painter's transforms (transform=None has no device form)")
            func = getattr(compiled, "func", None)
            steps = getattr(func, "steps", None) or [func]
            found = None
            for st in steps:
                if isinstance(st, T._RangeCompress) and st.direction == direction and found is None:
                    if st.modes[field].lower() != "shift-log":
                        raise NotImplementedError("device-side transforms implement the 'shift-log' mode only")
                    found = (float(st.k_values[field]), compiled.stats[field])
                elif not shape_only(st):
                    # a custom scaling step in the chain would be silently dropped on the device path
                    raise NotImplementedError(f"transform step {st!r} has no device form")
            if found is None:
                raise NotImplementedError("paint_stream(transform=True) needs the painter's shift-log range compression")
            return found
        if len(self.label_fields) != 1:
            raise NotImplementedError("Painting with more than one output field is not supported yet.")
        k_in, st_in = find(self.transform, 0, self.input_field)
        k_out, st_out = find(self.inverse_transform, 1, self.label_fields[0])
        s_in = np.sqrt(T.interpolate_z_many(st_in, zs, "var"))          # (vectorised: no Python call per tile)
        s_out = np.sqrt(T.interpolate_z_many(st_out, zs, "var"))
        return s_in, k_in, k_out, s_out

    def can_paint_stream(self, z=0.0):
        """Whether ``paint_stream`` has a device form for this painter (single-channel tiles, one label field, the
        'shift-log' range compression on both sides, L = 1 and a prior network) -- WITHOUT side effects: nothing is
        captured, no random number is drawn.  ``lightcone.paint_plane`` asks this before it draws a plane's seed, so that
        a NotImplementedError raised later, from inside a capture, is an error and not a silent fall-back."""
        model = self.model
        if model.dim_y[0] != 1 or getattr(model, "L", 1) != 1 or getattr(model, "prior_network", None) is None:
            return False
        try:
            self._shift_log_parameters(np.atleast_1d(np.asarray(z, dtype=np.float64))[:1])
        except NotImplementedError:
            return False
        return True

    def release_paint_buffers(self):
        """Free the page-locked host staging buffers ``paint_stream`` keeps between calls: two parameter blocks plus up to
        four (batch, 1, H, W) fp32 buffers -- 256 MiB of pinned memory at batch 64 of 512^2 tiles -- which otherwise live
        as long as the painter (page-locking them costs tens of milliseconds per call, hence the cache)."""
        self.__dict__.pop("_paint_host_buffers", None)

    def paint_stream(self, inputs, z, batch_size=64, tile_ids=None, seed=0, rank=0, world_size=1, out=None):
        """Paint MANY raw tiles: ``inputs`` (N, H, W) float32 host array (NumPy, memory map, or a pinned torch tensor),
        redshifts ``z`` (scalar or (N,)) -> (N, H, W) float32 physical tiles.  The production form of ``paint``
        (process_SLICS.py:201-218 calls it tile by tile):

          * the transform and its inverse run on the device, fused into the layout kernels on either side of the
            network (``bp_paint_load`` / ``bp_paint_store``), bit-compatible with the host transforms;
          * batches of ``batch_size`` tiles replay ONE captured hipGraph (prior + sampler + generator on four streams);
          * host->device and device->host copies go through pinned double buffers on their own streams, so that
            batch b+1 is uploaded and batch b-1 downloaded while batch b is painted;
          * the prior noise of tile i comes from a counter-based generator keyed on (``seed``, ``tile_ids[i]``
            [default: i], int64), so the result does not depend on ``batch_size`` or on how tiles are dealt to ranks;
            the seed travels in the per-batch parameter block: one captured graph serves every seed;
          * ``rank`` / ``world_size``: this process paints the contiguous block of tiles that is its share (one
            process per GPU, no collective: tiles are independent) and returns (block, (lo, hi))."""
        model = self.model
        model.train(False)
        dev = model.device
        cy, H, W = model.dim_y
        if cy != 1:
            raise NotImplementedError("paint_stream paints single-channel input tiles")
        N = len(inputs)
        if tuple(inputs.shape[1:]) != (H, W):
            raise ValueError(f"Shape mismatch between input and model: {tuple(inputs.shape)} vs {model.dim_y}")
        zs = np.broadcast_to(np.asarray(z, dtype=np.float64), (N,))
        ids = np.arange(N, dtype=np.int64) if tile_ids is None else np.asarray(tile_ids, dtype=np.int64)
        per = (N + world_size - 1) // world_size
        lo, hi = min(rank * per, N), min((rank + 1) * per, N)
        B = int(batch_size)
        s_in, k_in, k_out, s_out = self._shift_log_parameters(zs[lo:hi])     # (NotImplementedError before any capture)
        g = model.paint_graph(B)
        torch_in = isinstance(inputs, torch.Tensor)
        result = out if out is not None else np.empty((hi - lo, H, W), np.float32)
        torch_out = isinstance(result, torch.Tensor)
        main = torch.cuda.current_stream(dev)
        up, down = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        # Two slots = the graph's own two buffer sets (CVAE.paint_graph): uploads land where bp_paint_load reads,
        # downloads leave from where bp_paint_store writes.  Per slot one pinned parameter block (one copy per batch).
        layout = g["block_layout"]
        # pinned host buffers are kept between calls (page-locking 4 x 64 MiB costs tens of milliseconds per call);
        # release_paint_buffers() frees them; the tile buffers are only allocated for NumPy inputs / outputs
        cache = self.__dict__.setdefault("_paint_host_buffers", {})
        key = (B, H, W, g["block_bytes"], str(dev))
        if key not in cache:
            cache.clear()
            cache[key] = [{"h_blk": torch.zeros(g["block_bytes"], dtype=torch.uint8).pin_memory(), "h_in": None,
                           "h_out": None} for _ in range(2)]
        slots = []
        for gs, hb in zip(g["slots"], cache[key]):
            h_blk = hb["h_blk"]
            hv = {}
            for name, (o, dt, shape) in layout.items():
                nb = torch.tensor([], dtype=dt).element_size() * int(np.prod(shape))
                hv[name] = h_blk[o:o + nb].view(dt).view(shape).numpy()
            hv["seed"][0] = np.array(int(seed) & 0xFFFFFFFFFFFFFFFF, dtype=np.uint64).astype(np.int64)
            if not torch_in and hb["h_in"] is None:
                hb["h_in"] = torch.empty((B, 1, H, W), dtype=torch.float32).pin_memory()
            if not torch_out and hb["h_out"] is None:
                hb["h_out"] = torch.empty((B, 1, H, W), dtype=torch.float32).pin_memory()
            slots.append({"g": gs, "h_blk": h_blk, "hv": hv, "h_in": hb["h_in"], "h_out": hb["h_out"],
                          "ev_up": torch.cuda.Event(), "ev_done": torch.cuda.Event(), "ev_down": torch.cuda.Event(),
                          "pending": None})

        def harvest(sl):
            if sl["pending"] is None:
                return
            a, b = sl["pending"]
            sl["ev_down"].synchronize()
            if not torch_out:
                result[a - lo:b - lo] = sl["h_out"][:b - a, 0].numpy()
            sl["pending"] = None

        starts = list(range(lo, hi, B))

        def upload(bi):
            """Fill slot bi % 2's pinned buffers with batch bi and start its host-to-device copies."""
            a = starts[bi]
            b = min(a + B, hi)
            m = b - a
            sl = slots[bi % 2]
            gs, hv = sl["g"], sl["hv"]
            sl["ev_up"].synchronize()                     # the slot's previous upload has left its pinned buffers
            hv["xf_in"][:m, 0], hv["xf_in"][:m, 1] = s_in[a - lo:b - lo], k_in
            hv["xf_out"][:m, 0], hv["xf_out"][:m, 1] = k_out, s_out[a - lo:b - lo]
            hv["aux"][:m, 0] = zs[a:b]
            hv["tile_ids"][:m] = ids[a:b]
            if m < B:                                     # a short last batch: pad with its last tile's parameters
                for k in ("xf_in", "xf_out", "aux", "tile_ids"):
                    hv[k][m:] = hv[k][m - 1]
            if torch_in:
                src = inputs[a:b].reshape(m, 1, H, W)
            else:
                sl["h_in"][:m, 0].numpy()[...] = np.asarray(inputs[a:b], dtype=np.float32)
                src = sl["h_in"][:m]
            up.wait_event(sl["ev_done"])                  # the slot's previous batch has been painted (inputs read)
            with torch.cuda.stream(up):
                gs["raw"][:m].copy_(src, non_blocking=True)
                gs["block"].copy_(sl["h_blk"], non_blocking=True)
                sl["ev_up"].record(up)

        with torch.no_grad():
            if starts:
                upload(0)
            for bi, a in enumerate(starts):
                b = min(a + B, hi)
                m = b - a
                sl = slots[bi % 2]
                gs = sl["g"]
                # The NEXT batch's upload is enqueued BEFORE this batch's graph: copies enqueued behind a graph launch
                # only start once that graph has drained (measured: tools/paint_probe.py -- the upload then sits on the
                # critical path, 1.2 ms per 64 tiles); enqueued ahead of it they run beside it.
                if bi + 1 < len(starts):
                    upload(bi + 1)
                harvest(sl)                                   # this slot's previous batch has left the device
                main.wait_event(sl["ev_up"])
                main.wait_event(sl["ev_down"])                # ... and its previous output has been downloaded
                gs["graph"].replay()
                sl["ev_done"].record(main)
                down.wait_event(sl["ev_done"])
                with torch.cuda.stream(down):
                    dst = result[a - lo:b - lo].reshape(m, 1, H, W) if torch_out else sl["h_out"][:m]
                    dst.copy_(gs["out"][:m], non_blocking=True)
                    sl["ev_down"].record(down)
                sl["pending"] = (a, b)
            for sl in slots:
                harvest(sl)
            torch.cuda.synchronize(dev)
        return (result, (lo, hi)) if world_size > 1 else result

    # ------------------------------------------------------------------------------ checkpoints
    def save_state_to_file(self, filename, mode="model_state_dict+metadata"):
        """(state_path, meta_path): ``torch.save(state_dict)`` + pickled metadata with the
        compiled transforms (painter.py:395-418).  The state dict has the reference's keys."""
        if not isinstance(filename, (tuple, list)):
            raise ValueError("filename needs to be a tuple of (state_filename, meta_filename).")
        td = self.training_data
        d = {"L": td.L, "n_grid": td.n_grid, "tile_L": td.tile_L, "n_tile": td.n_tile, "tile_size": td.tile_size,
             "input_field": td.input_field, "label_fields": td.label_fields, "scale_to_SLICS": td.scale_to_SLICS,
             "transform": datasets.compile_transform(transform=td.transform_func, stats=td.stats),
             "inverse_transform": datasets.compile_transform(transform=td.inverse_transform_func, stats=td.stats),
             "model_architecture": self.architecture}
        with open(filename[1], "wb") as f:
            _pickler.dump(d, f)
        torch.save({k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}, filename[0])

    def load_state_from_file(self, filename, compute_device="cuda:0"):
        if not isinstance(filename, (tuple, list)):
            raise ValueError("filename needs to be a tuple of (state_filename, meta_filename).")
        self.compute_device = compute_device
        state_dict = torch.load(filename[0], map_location=torch.device(self.compute_device))
        with open(filename[1], "rb") as f:
            d = _pickler.load(f)
        self.model = _cvae.CVAE(d["model_architecture"], torch.device(self.compute_device),
                                sync=getattr(self, "sync", None), dtype=getattr(self, "dtype", "f32"))
        self.model.load_state_dict(state_dict)
        self.architecture = d["model_architecture"]
        for k in ("L", "n_grid", "tile_L", "n_tile", "tile_size", "input_field", "label_fields", "scale_to_SLICS"):
            setattr(self, k, d[k])
        self.transform = d.get("transform")
        self.inverse_transform = d.get("inverse_transform")


def dataloader_shuffle_order(n):
    """The index order ``DataLoader(dataset, shuffle=True)`` would produce next, consuming the global
    torch RNG the same way (iterator base seed first, then the RandomSampler's generator seed)."""
    torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter._base_seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())    # RandomSampler.__iter__
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


class _DeviceLoader:
    """Yields the same ``(fields, indices, redshifts)`` batches as the reference's DataLoader, in the
    same shuffled order, with the tiles assembled on the GPU."""

    def __init__(self, assembler, n, batch_size):
        self.asm, self.n, self.batch_size = assembler, n, batch_size

    def __iter__(self):
        order = dataloader_shuffle_order(self.n)
        for s in range(0, self.n, self.batch_size):
            idx = order[s:s + self.batch_size]
            x, y, z = self.asm.get_batch(idx)
            yield [y, x], torch.tensor(idx), z

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size


class _ShardedLoader:
    """Data-parallel replacement for ``DataLoader(shuffle=True)``: every rank draws the SAME
    global permutation (seeded, advanced per pass) and takes its slice of each global batch
    (baryon_painter_amd.dist.shard_indices), so the union over ranks is what one device would
    have processed."""

    def __init__(self, dataset, batch_size, sync, seed=20190101):
        self.dataset, self.batch_size, self.sync = dataset, batch_size, sync
        self.seed, self.epoch = seed, 0

    def __iter__(self):
        from .dist import shard_indices
        rng = np.random.Generator(np.random.PCG64([self.seed, self.epoch]))
        self.epoch += 1
        perm = rng.permutation(len(self.dataset))
        collate = torch.utils.data.default_collate
        for idx in shard_indices(perm, self.sync.rank, self.sync.world_size, self.batch_size):
            yield collate([self.dataset[int(i)] for i in idx])

    def __len__(self):
        return len(self.dataset) // (self.batch_size * self.sync.world_size)


class TrainingStats:
    """Loss log with moving average and the reference's text format (painter.py:447-545):
    header ``# Batch nr, sample nr, <labels>``, one line ``batch sample v0 v1 ...`` per push."""

    def __init__(self, loss_terms=[], moving_average_window=100, dump_to_file_frequency=10, stats_filename=None):
        self.mavg_window = moving_average_window
        self.n_batches = 0
        self.n_processed_samples = []
        self.last_dump_to_file = 0
        self.dump_to_file_frequency = dump_to_file_frequency
        self.loss_terms = collections.OrderedDict((t, {"all": [], "mavg": []}) for t in loss_terms)
        self.stats_filename = stats_filename
        if stats_filename is not None:
            with open(stats_filename, "w") as f:
                f.write("# Batch nr, sample nr, {}\n".format(", ".join(loss_terms)))

    def __del__(self):
        try:
            self.flush_to_file()
        except Exception:
            pass

    def push_loss(self, n_sample, *args):
        self.n_batches += 1
        self.n_processed_samples.append(n_sample)
        for value, term in zip(args, self.loss_terms.values()):
            term["all"].append(value)
            term["mavg"].append(np.mean(term["all"][-min(self.n_batches, self.mavg_window):]))
        if self.stats_filename is not None and self.n_batches - self.dump_to_file_frequency >= self.last_dump_to_file:
            self.flush_to_file()

    def flush_to_file(self):
        if self.stats_filename is None:
            return
        with open(self.stats_filename, "a") as f:
            for s in range(self.last_dump_to_file, self.n_batches):
                f.write(self.get_str(s) + "\n")
        self.last_dump_to_file = self.n_batches

    def get_str(self, idx=-1):
        batch = idx if idx >= 0 else self.n_batches + idx + 1
        return f"{batch} {self.n_processed_samples[idx]} " + "".join(f"{t['all'][idx]} " for t in self.loss_terms.values())

    def get_pretty_str(self, n_col=1):
        width = max(len(k) for k in self.loss_terms)
        out, in_row = "", 0
        for key, term in self.loss_terms.items():
            out += "{key:<{width}s}: {value:8.3e}     ".format(key=key, width=width, value=term["mavg"][-1])
            in_row += 1
            if in_row >= n_col:
                out += "\n"
                in_row = 0
        return out


class CGANPainter(Painter):
    """Painter around the conditional GAN (models/cgan.py) with the ``paint`` keywords of the
    reference's external ``GAN_Painter`` (scripts/create_lightcone.py:47-54,
    process_SLICS.py:170-172).  Field transform: the reference's CGAN used a "shift-log-cam" map into
    the tanh range (trained_models/CGAN/fiducial/transform.pickle: log(x/sigma+1)/k0 - k1 with
    k = [4, 1]); it is applied here on top of the dataset's statistics."""

    K = (4.0, 1.0)

    def __init__(self, training_data_set=None, tile_size=512, compute_device="cuda:0", n_res=9, filename=None):
        from .models.cgan import CGAN
        self.compute_device = compute_device
        self.model = CGAN(tile_size=tile_size, device=compute_device, n_res=n_res)
        self.training_data = training_data_set
        self.stats = None if training_data_set is None else training_data_set.stats
        self.input_field, self.label_fields = "dm", ["pressure"]
        if filename is not None:
            self.load_state_from_file(filename, compute_device)

    # ---- the CGAN's own field transform
    def _sigma(self, field, z):
        from .utils.data_transforms import interpolate_z
        return float(np.sqrt(interpolate_z(self.stats[field], z)["var"]))

    def transform(self, x, field, z):
        return (np.log(np.asarray(x, np.float64) / self._sigma(field, z) + 1) / self.K[0] - self.K[1]).astype(np.float32)

    def inverse_transform(self, y, field, z):
        return (np.exp((np.asarray(y, np.float64) + self.K[1]) * self.K[0]) - 1) * self._sigma(field, z)

    def train(self, n_iter=1000, batch_size=6, learning_rate=5e-5, lr_decay=0.85, lr_decay_every=1568, verbose=False):
        """Alternating D / G iterations (README.md:97,130-139): Adam(betas=(0.5, 0.999)), lr x0.85 every
        1568 iterations, batch 6.  Returns the list of loss dicts."""
        if self.training_data is None:
            raise RuntimeError("Trying to train but no training data specified.")
        m = self.model
        m.train(True)
        opt_g = torch.optim.Adam(m.g_parameters(), lr=learning_rate, betas=(0.5, 0.999))
        opt_d = torch.optim.Adam(m.d_parameters(), lr=learning_rate, betas=(0.5, 0.999))
        sched = [torch.optim.lr_scheduler.StepLR(o, step_size=lr_decay_every, gamma=lr_decay) for o in (opt_g, opt_d)]
        ds, log = self.training_data, []
        rng = np.random.default_rng(0)
        for it in range(n_iter):
            idx = rng.integers(0, len(ds), batch_size)
            dm, pr, zs = [], [], []
            for i in idx:
                d, p, z = ds.raw_fields(int(i)) if hasattr(ds, "raw_fields") else self._raw(ds, int(i))
                dm.append(self.transform(d, "dm", z)[None])
                pr.append(self.transform(p, "pressure", z)[None])
                zs.append(z)
            losses = m.train_step(torch.from_numpy(np.stack(pr)), torch.from_numpy(np.stack(dm)),
                                  torch.tensor(zs, dtype=torch.float32), opt_g, opt_d)
            for s in sched:
                s.step()
            log.append({k: float(v) for k, v in losses.items()})
            if verbose and it % 50 == 0:
                print(it, log[-1])
        return log

    @staticmethod
    def _raw(ds, i):
        z = ds.sample_idx_to_redshift(i)
        return ds.get_input_sample(i, transform=False), ds.get_label_sample(i, transform=False)[0], z

    def paint(self, input, z=0.0, transform=True, inverse_transform=True):
        self.model.train(False)
        y = self.transform(input, "dm", z) if transform else np.asarray(input, np.float32)
        t = self.model.tile_size
        if y.shape != (t, t):
            raise ValueError(f"Shape mismatch between input and model: {np.shape(input)} vs {(1, t, t)}")
        pred = self.model.generate(torch.from_numpy(y.reshape(1, 1, t, t)), torch.tensor([z])).cpu().numpy()
        if inverse_transform:
            return self.inverse_transform(pred[0, 0], "pressure", z)
        return pred

    def save_state_to_file(self, filename):
        torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, filename)

    def load_state_from_file(self, filename, compute_device="cuda:0"):
        self.model.load_state_dict(torch.load(filename, map_location=torch.device(compute_device)))
