#!/usr/bin/env python3
"""Training entry point of the single-scale CVAE (drop-in for the reference's
scripts/CVAE_single_scale.py:18-198): same constants, same architecture helpers, same schedules
and the same ``painter.train(...)`` call -- on the MI355X-native painter.

The reference edits constants in the file; they are kept as constants here, with environment
overrides so the script can be exercised without the BAHAMAS stacks:
  BP_DATA_PATH    training stacks directory (train_files_info.pickle + .npy); if absent, a seeded
                  synthetic tile dataset of the same protocol is used
  BP_OUTPUT_PATH  run directory (default ../output/)
  BP_DEVICE       compute device (default cuda:0)
  BP_N_PEPOCH / BP_TILE  shorten the run / shrink the tiles for smoke tests
  BP_DTYPE         f32 (default: the reference's arithmetic) or bf16 (bf16 activations / gradients in the generator
                   trunk, fp32 accumulation, master weights and statistics)
  BP_DIST_BACKEND  collective backend under torch.distributed.run (default nccl = RCCL; gloo to rehearse the
                  multi-rank path with all ranks on one GPU)
Launch under ``python -m torch.distributed.run --nproc-per-node N`` for data-parallel training
(RCCL all-reduce of one flat gradient buffer + batch-norm statistics).
"""
import os
import pickle
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from baryon_painter_amd.utils import datasets, data_transforms      # noqa: E402
import baryon_painter_amd.painter                                    # noqa: E402
from baryon_painter_amd.models import cvae                           # noqa: E402

if __name__ == "__main__":
    data_path = os.environ.get("BP_DATA_PATH", "../../painting_baryons/training_data/BAHAMAS/stacks_new/")
    output_path = os.environ.get("BP_OUTPUT_PATH", "../output/")
    compute_device = os.environ.get("BP_DEVICE", "cuda:0")

    n_training_stack = 11
    n_validation_stack = 3
    n_scale = 1
    n_aux_label = 1
    label_fields = ["pressure"]
    redshifts = [0.0, 0.125, 0.25, 0.375, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0]

    sync = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        backend = os.environ.get("BP_DIST_BACKEND", "nccl")       # "gloo" only to rehearse on a one-GPU box
        if backend != "nccl":
            local_rank = 0
        compute_device = f"cuda:{local_rank}"
        torch.cuda.set_device(local_rank)
        # (lazy communicator creation -- no device_id=: the eager form costs ~80 us per all-reduce of a step, bench.py)
        dist.init_process_group(backend)
        from baryon_painter_amd.dist import Sync
        sync = Sync()

    range_compress_transform, range_compress_inv_transform = data_transforms.create_range_compress_transforms(
        k_values={"dm": 4.0, "pressure": 4}, modes={"dm": "shift-log", "pressure": "shift-log"}, eps=1e-4)
    transform = data_transforms.chain_transformations([range_compress_transform, data_transforms.atleast_3d,
                                                       data_transforms.as_float32])
    inv_transform = data_transforms.chain_transformations([data_transforms.squeeze, range_compress_inv_transform])

    info_file = os.path.join(data_path, "train_files_info.pickle")
    if os.path.exists(info_file):
        with open(info_file, "rb") as f:
            training_files_info = pickle.load(f)
        common = dict(redshifts=redshifts, label_fields=label_fields, transform=transform,
                      inverse_transform=inv_transform, n_feature_per_field=n_scale, tile_permutations=True,
                      mmap_mode="r", scale_to_SLICS=True, subtract_minimum=False)
        training_dataset = datasets.BAHAMASDataset(files=training_files_info, root_path=data_path,
                                                   n_stack=n_training_stack, stack_offset=n_validation_stack, **common)
        validation_dataset = datasets.BAHAMASDataset(data=training_dataset.data, n_stack=n_validation_stack,
                                                     stack_offset=0, **common)
    else:
        tile = int(os.environ.get("BP_TILE", "512"))
        print(f"{info_file} not found: using seeded synthetic {tile}x{tile} tiles")
        training_dataset = datasets.SyntheticTileDataset(n_sample=1 << 14, tile_size=tile, redshifts=redshifts, seed=1)
        validation_dataset = datasets.SyntheticTileDataset(n_sample=1 << 10, tile_size=tile, redshifts=redshifts, seed=2)

    n_x_feature = len(training_dataset.label_fields) * n_scale
    dim_x = (n_x_feature, training_dataset.tile_size, training_dataset.tile_size)
    dim_y = (n_scale, training_dataset.tile_size, training_dataset.tile_size)
    dim_z = (1, training_dataset.tile_size // 32, training_dataset.tile_size // 32)

    def head(last_activation):
        return (cvae.conv_block(16, 8, kernel=7, bias=False, batchnorm=False, activation="PReLU")
                + cvae.conv_block(8, n_x_feature, kernel=5, bias=False, batchnorm=False, activation="PReLU")
                + cvae.conv_block(n_x_feature, n_x_feature, kernel=3, bias=False, batchnorm=False,
                                  activation=last_activation))

    test_net = {
        "type": "Type-1", "dim_x": dim_x, "dim_y": dim_y, "dim_z": dim_z, "n_x_features": n_x_feature,
        "aux_label": True,
        "prior_z_y": cvae.conv_down(in_channel=1 + n_aux_label, channels=[8, 16, 32], scales=[2, 4, 4])
        + cvae.conv_block(32, 2 * dim_z[0], kernel=5) + [("unflatten", (2, *dim_z))],
        "q_x_in": cvae.conv_down(in_channel=n_x_feature, channels=[8, 16, 32], scales=[2, 4, 4]),
        "q_y_in": cvae.conv_down(in_channel=1 + n_aux_label, channels=[8, 16, 32], scales=[2, 4, 4]),
        "q_x_y_out": cvae.conv_block(64, 2 * dim_z[0], kernel=5) + [("unflatten", (2, *dim_z))],
        "p_y_in": None,
        "p_z_in": cvae.conv_up(1, channels=[1, 1, 1], scales=[2, 4, 4], bias=False, batchnorm=True),
        "p_y_z_in": cvae.conv_block(n_aux_label + n_scale + 1, 16, kernel=5)
        + cvae.conv_down(in_channel=16, channels=[32, 64, 128], scales=[2, 2, 2])
        + [("residual block", cvae.res_block(128)) for _ in range(4)]
        + cvae.conv_up(128, channels=[64, 32, 16], scales=[2, 2, 2], bias=False, batchnorm=True, activation="ReLU"),
        "p_y_z_out": (head("softplus"), head(None)),          # mean and variance heads
        "min_x_var": 1e-7, "min_z_var": 1e-7, "L": 1,
    }

    painter = baryon_painter_amd.painter.CVAEPainter(training_data_set=training_dataset,
                                                     test_data_set=validation_dataset,
                                                     architecture=test_net, compute_device=compute_device, sync=sync,
                                                     dtype=os.environ.get("BP_DTYPE", "f32"))
    print(painter.model)

    def adaptive_batch_size(pepoch, min_batch_size=1, max_batch_size=24):
        for start, size in [(32, 24), (16, 16), (8, 8), (0, 4)]:
            if pepoch >= start:
                return min(size, max_batch_size)
        return min_batch_size

    def adaptive_lr(pepoch):
        step, min_gamma = 32, 1e-6
        min_pepoch = 64 - step
        if pepoch < min_pepoch:
            return 1
        return max(min_gamma, 0.5 ** ((pepoch - min_pepoch) // step))

    run_name = ("single_scale_max_z2_res4_var_prior_late_prelu_log_shift_softmax_lr1e-3_tile_perm_slow_decay"
                "_switched_sets")
    output_path = os.path.join(output_path, run_name)
    os.makedirs(output_path, exist_ok=True)
    with open(os.path.join(output_path, "architecture.txt"), "w") as f:
        f.write(repr(painter.model.architecture))
    with open(os.path.join(output_path, "architecture_built.txt"), "w") as f:
        f.write(repr(painter.model))

    painter.train(n_epoch=1, n_pepoch=int(os.environ.get("BP_N_PEPOCH", "256")), learning_rate=1e-3, batch_size=4,
                  adaptive_learning_rate=adaptive_lr, adaptive_batch_size=adaptive_batch_size,
                  pepoch_size=1568, validation_loss_frequency=72, validation_loss_batch_size=24,
                  validation_pepochs=[0, 1, 2, 5, 10, 15, 20, 30, 40, 50, 60, 70, 80, 90, 100, 120, 140, 160, 180,
                                      200, 230, 260, 290, 350, 400],
                  validation_batch_size=8, checkpoint_frequency=20000, statistics_report_frequency=400,
                  loss_plot_frequency=0, mavg_window_size=50, show_plots=False, save_plots=True,
                  plot_sample_var=True, plot_power_spectra=["auto", "cross"], plot_histogram=["log"],
                  output_path=output_path, verbose=True)
