#!/usr/bin/env python3
"""DDP-style trainer — flag-compatible with the reference's train.py (train.py:306-337), one process
per GPU (`torchrun --nproc-per-node=G train.py ...`), running the native MI355X step
(osu_diffusion_amd.training.NativeTrainer) with ONE RCCL all-reduce per step on a flat fp32
gradient arena instead of DDP's bucketed reducer.

`--data-path <root>` streams windows from `<root>/TrackNNNNN/beatmaps/*.osu` exactly like the reference's loader
(tracks split per rank, train.py:165-170, then per DataLoader worker; `.osu` files read by osu_diffusion_amd.beatmap — the
third-party `slider` parser is not needed); `--synthetic` trains on synthetic windows with the same tensor contract.
"""
import argparse
import logging
import os
from glob import glob
from time import time

import torch
import torch.distributed as dist

from osu_diffusion_amd.beatmap import open_beatmap_sequence, track_catalogue
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.synthetic import synthetic_windows
from osu_diffusion_amd.windows import WindowIterableFactory, get_data_loader, synthetic_sequences
from osu_diffusion_amd.training import NativeTrainer, shard_range

feature_size = 19


def create_logger(logging_dir, rank):
    """train.py:73-91."""
    if rank == 0:
        logging.basicConfig(level=logging.INFO, format="[\033[34m%(asctime)s\033[0m] %(message)s", datefmt="%Y-%m-%d %H:%M:%S",
                            handlers=[logging.StreamHandler(), logging.FileHandler(f"{logging_dir}/log.txt")])
        return logging.getLogger(__name__)
    logger = logging.getLogger(__name__)
    logger.addHandler(logging.NullHandler())
    return logger


def main(args):
    assert torch.cuda.is_available(), "Training currently requires at least one GPU."
    distributed = "RANK" in os.environ
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.dist)  # "nccl" == RCCL on ROCm
    world_size = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    assert args.global_batch_size % world_size == 0, "Batch size must be divisible by world size."
    device = rank % torch.cuda.device_count()
    seed = args.global_seed * world_size + rank
    torch.manual_seed(seed)
    torch.cuda.set_device(device)
    print(f"Starting rank={rank}, seed={seed}, world_size={world_size}.")

    if rank == 0:
        os.makedirs(args.results_dir, exist_ok=True)
        experiment_index = len(glob(f"{args.results_dir}/*"))
        experiment_dir = f"{args.results_dir}/{experiment_index:03d}-{args.model.replace('/', '-')}"
        checkpoint_dir = f"{experiment_dir}/checkpoints"
        os.makedirs(checkpoint_dir, exist_ok=True)
        logger = create_logger(experiment_dir, rank)
        logger.info(f"Experiment directory created at {experiment_dir}")
    else:
        checkpoint_dir = ""
        logger = create_logger(None, rank)

    model = DiT_models[args.model](num_classes=args.num_classes, context_size=feature_size - 3 + 128,
                                   class_dropout_prob=0.2, precision=args.precision).to(device)
    diffusion = create_diffusion(timestep_respacing="", noise_schedule=args.noise_schedule, use_l1=args.l1_loss)
    logger.info(f"DiT Parameters: {sum(p.numel() for p in model.parameters()):,}")
    logger.info(f"arithmetic tier: {args.precision}" + (" (every Linear product of the blocks -- forward, data and weight gradients -- on e4m3 "
                                                         "operands, delayed per-tensor scaling; attention / adaLN / embedders bf16; fp32 masters)"
                                                         if args.precision == "fp8" else ""))
    # AdamW(lr, wd=0) + EMA 0.9999 + init broadcast; the exchange options are flags of this script (--zero1, --grad-wire, --native-comm)
    if args.select_exchange and world_size > 1 and not args.zero1 and not args.native_comm:
        # which exchange this node runs faster is MEASURED -- 3 steps each on synthetic windows of the run's shape, with lr = 0 on throw-away
        # trainers (the weights the run starts from are not touched; moments and EMA copies die with the probe) -- and agreed by all ranks
        from osu_diffusion_amd.synthetic import synthetic_windows as probe_windows  # (a local name: `synthetic_windows` is the module-level import the batch generator below closes over)
        from osu_diffusion_amd.training import select_exchange_schedule

        probe = []
        for i in range(2):
            (px, po, pc), py = probe_windows(args.global_batch_size // world_size, args.seq_len, args.num_classes, seed=977 * rank + i, train_offsets=True)
            probe.append(((px.to(device), po.to(device), pc.to(device)), py.to(device)))
        model.train()
        sel = select_exchange_schedule(lambda shard_optimizer: NativeTrainer(model, diffusion, lr=0.0, shard_optimizer=shard_optimizer, broadcast_init=False),
                                       probe, steps=3, warmup=1, device=device)
        args.zero1 = sel["name"] == "zero1"
        logger.info(f"exchange schedule selected by measurement: {sel}")
    trainer = NativeTrainer(model, diffusion, lr=args.lr, shard_optimizer=args.zero1,
                            wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None, native_comm=args.native_comm)
    model.train()
    if args.ckpt is not None:
        trainer.load_checkpoint(torch.load(args.ckpt, map_location="cpu", weights_only=False), lr=args.lr,
                                relearn_embeds=args.relearn_embeds)
        logger.info(f"Restored from checkpoint at {args.ckpt}")

    batch_size = args.global_batch_size // world_size
    assert args.synthetic or args.data_path, "pass --data-path <dataset root> or --synthetic"
    loader = None
    if not args.synthetic:
        # train.py:165-190: this rank's track range -> interleaved, worker-sharded stream of augmented windows
        dataset_start, dataset_end = shard_range(args.data_start, args.data_end, rank, world_size)
        loader = get_data_loader(track_catalogue(args.data_path), dataset_start, dataset_end,
                                 WindowIterableFactory(args.seq_len, args.stride, open_fn=open_beatmap_sequence),
                                 cycle_length=max(1, batch_size // 2), batch_size=batch_size, num_workers=args.num_workers,
                                 shuffle=True, pin_memory=True, drop_last=True)
    elif args.synthetic_maps > 0:
        # the reference's loader contract end to end (windows.py): sequences -> overlapping windows with random phase, flips
        # and time offsets -> batches; tracks split per rank (train.py:165-170), then per DataLoader worker
        catalogue = synthetic_sequences(args.synthetic_maps, min_len=args.seq_len, max_len=8 * args.seq_len,
                                        seed=args.global_seed, first_id=0)
        dataset_start, dataset_end = shard_range(0, len(catalogue), rank, world_size)
        loader = get_data_loader(catalogue, dataset_start, dataset_end, WindowIterableFactory(args.seq_len, args.stride),
                                 cycle_length=max(1, batch_size // 2), batch_size=batch_size, num_workers=args.num_workers, shuffle=True,
                                 pin_memory=True, drop_last=True)
    else:
        dataset_start, dataset_end = shard_range(args.data_start, args.data_end, rank, world_size)
    logger.info(f"Dataset contains {(dataset_end - dataset_start):,} beatmap sets ({'synthetic' if args.synthetic else args.data_path})")

    train_steps, log_steps, start_time = 0, 0, time()
    running_loss = torch.zeros((), device=device)  # accumulated on the device: no host sync per step
    logger.info(f"Training for {args.epochs} epochs...")
    if args.embed_only_epochs > 0:  # train.py:223-225
        logger.info(f"Freezing non-embedding layers for {args.embed_only_epochs} epochs")
        trainer.embed_only = True
    for epoch in range(args.epochs):
        logger.info(f"Beginning epoch {epoch}...")
        if 0 < args.embed_only_epochs == epoch:  # train.py:237-241
            logger.info("Un-freezing non-embedding layers")
            trainer.embed_only = False
            trainer.lr = 1e-4
        def batches():
            if loader is not None:
                yield from loader
                return
            for it in range(args.steps_per_epoch):
                track = dataset_start + (epoch * args.steps_per_epoch + it) % max(1, dataset_end - dataset_start)
                yield synthetic_windows(batch_size, args.seq_len, args.num_classes, seed=track)

        for (x, o, c), y in batches():
            # --refine: the refine-model recipe of train_nodist.py:222 (t = 0 for every sample; sample.py --refine-ckpt uses it)
            t0 = torch.zeros(x.shape[0], dtype=torch.long) if args.refine else None
            terms = trainer.step(x, o, c, y, t=t0)
            running_loss += terms[2].mean()
            log_steps += 1
            train_steps += 1
            if train_steps % args.log_every == 0:
                torch.cuda.synchronize()
                steps_per_sec = log_steps / (time() - start_time)
                avg_loss = running_loss / log_steps
                if distributed:
                    dist.all_reduce(avg_loss, op=dist.ReduceOp.SUM)
                logger.info(f"(step={train_steps:07d}) Train Loss: {avg_loss.item() / world_size:.4f}, "
                            f"Train Steps/Sec: {steps_per_sec:.2f}")
                running_loss.zero_()
                log_steps, start_time = 0, time()
            if train_steps % args.ckpt_every == 0 and train_steps > 0:
                # sharded optimizer (--zero1): the moments / EMA of the other ranks' shards are all-gathered first -- a
                # COLLECTIVE, so every rank takes part before rank 0 alone writes the file (train.py:285-297)
                trainer.sync_sharded_state()
                if rank == 0:
                    path = f"{checkpoint_dir}/{train_steps:07d}.pt"
                    torch.save(trainer.checkpoint(args), path)
                    logger.info(f"Saved checkpoint to {path}")
                if distributed:
                    dist.barrier()
    model.eval()
    logger.info("Done!")
    if distributed:
        dist.destroy_process_group()


# flag -> (type, default) for the reference's flag set (train.py:306-337); same names, same defaults
REFERENCE_FLAGS = {
    "data-path": (str, None), "num-classes": (int, 52670), "data-start": (int, 0), "data-end": (int, 13402),
    "results-dir": (str, "results"), "epochs": (int, 1400), "global-batch-size": (int, 256), "global-seed": (int, 0),
    "num-workers": (int, 4), "log-every": (int, 100), "ckpt-every": (int, 50000), "seq-len": (int, 128), "stride": (int, 16),
    "use-amp": (bool, True),            # accepted; bf16 MFMA tier with fp32 masters, no GradScaler needed
    "ckpt": (str, None), "dist": (str, "nccl"),
    "fine-tune-ids": (str, None),       # declared but unused by the reference too
    "noise-schedule": (str, "squaredcos_cap_v2"), "l1-loss": (bool, True), "lr": (float, 1e-4),
    "relearn-embeds": (bool, False), "embed-only-epochs": (int, 0),
}


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    for flag, (kind, default) in REFERENCE_FLAGS.items():
        p.add_argument("--" + flag, type=kind, default=default)
    p.add_argument("--model", type=str, choices=list(DiT_models.keys()), default="DiT-B")
    # additions of this build
    p.add_argument("--refine", action="store_true", help="train the t = 0 refine model (train_nodist.py's recipe)")
    p.add_argument("--synthetic", action="store_true", help="train on synthetic windows (no dataset needed)")
    p.add_argument("--synthetic-maps", type=int, default=0,
                   help="with --synthetic: stream windows from this many synthetic hit-object sequences through the "
                        "reference's loader contract (windows.py) instead of drawing windows directly")
    p.add_argument("--steps-per-epoch", type=int, default=1000)
    p.add_argument("--precision", choices=["bf16", "fp32", "fp8"], default="bf16",
                   help="bf16: the reference's --use-amp tier (bf16 MFMA, fp32 masters); fp32: the parity tier; fp8: the bf16 tier with every "
                        "Linear product of the blocks (forward, data and weight gradients) on e4m3 operands with delayed scaling "
                        "(BASELINE config 5: DiT-XL, seq-len 256)")
    p.add_argument("--zero1", action="store_true",
                   help="multi-GPU: reduce-scatter / sharded AdamW + EMA / all-gather instead of the gradient all-reduce")
    p.add_argument("--select-exchange", action="store_true",
                   help="multi-GPU: time 3 steps each of the all-reduce and the --zero1 exchange at start-up and keep the faster (every rank the same)")
    p.add_argument("--grad-wire", choices=["fp32", "bf16"], default="fp32", help="--zero1: dtype of the gradients on the wire")
    p.add_argument("--native-comm", action="store_true",
                   help="multi-GPU: the exchange through the library's own RCCL communicator (C ABI) instead of torch.distributed")
    return p.parse_args(argv)


if __name__ == "__main__":
    main(parse_args())
