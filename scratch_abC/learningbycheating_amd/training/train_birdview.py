"""Behaviour cloning of the privileged agent (reference training/train_birdview.py), MI355X-native:
BirdViewPolicyModelSS(resnet18) on 7x192x192 maps, L1 (choice='l1', train_birdview.py:161) between the selected
branch and ground-truth waypoints in pixels."""
import argparse
import os
import time
from pathlib import Path

import torch
import torch.distributed as dist

from ..bird_view.models.birdview import BirdViewPolicyModelSS
from ..bird_view.utils import bz_utils as bzu
from .data import make_loaders
from ..bird_view.utils.train_utils import one_hot
from ..parallel import broadcast_module
from .native import NativeTrainer

BACKBONE = "resnet18"
GAP = 5
N_STEP = 5
SAVE_EPOCHS = [1, 2, 4, 8, 16, 32, 64, 128, 256, 384, 512, 768, 1000]


def train_or_eval(trainer, data, is_train, config, is_first_epoch):
    """reference train_birdview.py:102-153"""
    tick = time.time()
    for i, (rgb_image, birdview, location, command, speed) in enumerate(data):
        command = one_hot(command).to(config["device"])
        loss = trainer.step(birdview, speed, command, target=location.float().contiguous(), update=is_train and not is_first_epoch, train_mode=is_train)
        if (i % int(config["log_iterations"]) == 0) or (not is_train) or is_first_epoch:
            bzu.log.scalar(is_train=is_train, loss_mean=loss.mean().item())
        now = time.time()
        bzu.log.scalar(is_train=is_train, fps=1.0 / max(now - tick, 1e-9))
        tick = now
        if is_first_epoch and i == 10:
            break


def train(config):
    rank, world, device = config["rank"], config["world_size"], config["device"]
    bzu.log.init(config["log_dir"], rank)
    bzu.log.save_config({k: v for k, v in config.items() if k not in ("rank", "world_size")})
    net = BirdViewPolicyModelSS(config["model_args"]["backbone"]).to(device)
    net.precision = config.get("precision", "fp32")
    if config["resume"]:
        # the reference takes glob('model-*.th')[-1] unsorted (train_birdview.py:164-169); sort numerically instead
        ckpts = sorted(Path(config["log_dir"]).glob("model-*.th"), key=lambda p: int(p.stem.split("-")[1]))
        if ckpts:
            net.load_state_dict(torch.load(str(ckpts[-1]), map_location=device))
    broadcast_module(net)
    bs = config["data_args"]["batch_size"]
    data_train, data_val = make_loaders(config, device, rank, world)
    trainer = NativeTrainer(net, None, bs, (7, 192, 192), device, phase="birdview", lr=config["optimizer_args"]["lr"], world_size=world)
    for epoch in range(int(config["max_epoch"]) + 1):
        net.train()
        train_or_eval(trainer, data_train, True, config, epoch == 0)
        net.eval()                              # reference train_birdview.py:175-176: validation pass after every epoch
        train_or_eval(trainer, data_val, False, config, epoch == 0)
        net.train()
        if epoch in SAVE_EPOCHS and rank == 0:
            torch.save(net.state_dict(), str(Path(config["log_dir"]) / ("model-%d.th" % epoch)))
        rec = bzu.log.end_epoch()
        if rank == 0:
            print(rec)
    return net


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--log_dir", required=True)
    parser.add_argument("--log_iterations", default=1000)
    parser.add_argument("--max_epoch", default=1000)
    parser.add_argument("--dataset_dir", default=None)
    parser.add_argument("--batch_size", type=int, default=256)
    parser.add_argument("--x_jitter", type=int, default=5)
    parser.add_argument("--y_jitter", type=int, default=0)
    parser.add_argument("--angle_jitter", type=int, default=5)
    parser.add_argument("--gap", type=int, default=5)
    parser.add_argument("--max_frames", type=int, default=None)
    parser.add_argument("--cmd-biased", action="store_true")
    parser.add_argument("--resume", action="store_true")
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--synthetic", type=int, default=2048)
    parser.add_argument("--iters_per_epoch", type=int, default=1000)
    parser.add_argument("--precision", choices=["fp32", "bf16", "bf16_mfma"], default="fp32",
                        help="fp32 = the reference arithmetic; bf16 = bf16 MFMA operands + bf16 activation storage, f32 master weights")
    parsed = parser.parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a ROCm GPU")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")
    config = {
        "log_dir": parsed.log_dir, "log_iterations": parsed.log_iterations, "max_epoch": parsed.max_epoch,
        "device": torch.device("cuda", local), "precision": parsed.precision, "optimizer_args": {"lr": parsed.lr}, "resume": parsed.resume,
        "data_args": {"dataset_dir": parsed.dataset_dir, "batch_size": parsed.batch_size, "n_step": N_STEP, "gap": parsed.gap,
                      "crop_x_jitter": parsed.x_jitter, "crop_y_jitter": parsed.y_jitter, "angle_jitter": parsed.angle_jitter,
                      "max_frames": parsed.max_frames, "cmd_biased": parsed.cmd_biased},
        "model_args": {"model": "birdview_dian", "input_channel": 7, "backbone": BACKBONE},
        "synthetic": parsed.synthetic, "iters_per_epoch": parsed.iters_per_epoch, "rank": rank, "world_size": world,
    }
    train(config)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
