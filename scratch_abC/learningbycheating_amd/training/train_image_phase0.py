"""Phase-0 warm-up of the sensorimotor agent (reference training/train_image_phase0.py), MI355X-native.

The reference moves the teacher's map-space waypoints to the CPU every step, projects them with cv2.projectPoints in
float64 and ships them back (train_image_phase0.py:67-79).  Here the same pinhole projection + clip + normalised L1
against the *selected* branch runs in one HIP kernel (csrc/loss.hip kind 0); no device->host round trip."""
import argparse
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from ..bird_view.models.birdview import BirdViewPolicyModelSS
from ..bird_view.models.image import ImagePolicyModelSS
from ..bird_view.utils import bz_utils as bzu
from .data import make_loaders
from ..bird_view.utils.train_utils import one_hot
from ..parallel import broadcast_module
from .native import NativeTrainer, camera_struct

BACKBONE = "resnet34"
GAP = 5
N_STEP = 5
PIXELS_PER_METER = 5
CROP_SIZE = 192
SAVE_EPOCHS = [1, 2, 4, 8, 16, 32, 64, 128, 256, 384, 512, 768, 1000]


def train_or_eval(trainer, data, is_train, config, is_first_epoch):
    """reference train_image_phase0.py:152-209"""
    tick = time.time()
    for i, (rgb_image, birdview, location, command, speed) in enumerate(data):
        command = one_hot(command).to(config["device"])
        loss = trainer.step(rgb_image, speed, command, birdview=birdview, update=is_train and not is_first_epoch, train_mode=is_train)
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
    teacher_backbone = "resnet18"
    if config["teacher_args"]["model_path"]:
        teacher_backbone = bzu.log.load_config(config["teacher_args"]["model_path"])["model_args"]["backbone"]
    net = ImagePolicyModelSS(config["model_args"]["backbone"], pretrained=config["model_args"]["imagenet_pretrained"]).to(device)
    teacher_net = BirdViewPolicyModelSS(teacher_backbone).to(device)
    net.precision = teacher_net.precision = config.get("precision", "fp32")
    if config["teacher_args"]["model_path"]:
        teacher_net.load_state_dict(torch.load(config["teacher_args"]["model_path"], map_location=device))
    teacher_net.eval()
    broadcast_module(net)
    broadcast_module(teacher_net)
    bs = config["data_args"]["batch_size"]
    data_train, data_val = make_loaders(config, device, rank, world)
    cam = camera_struct(**{k: float(v) for k, v in config["camera_args"].items()})
    trainer = NativeTrainer(net, teacher_net, bs, (3, 160, 384), device, phase=0, lr=config["optimizer_args"]["lr"], world_size=world, camera=cam)
    for epoch in range(int(config["max_epoch"]) + 1):
        net.train()
        train_or_eval(trainer, data_train, True, config, epoch == 0)
        net.eval()                              # reference train_image_phase0.py:236-237: validation pass after every epoch
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
    parser.add_argument("--max_epoch", default=2)
    parser.add_argument("--pretrained", action="store_true")
    parser.add_argument("--teacher_path", default=None)
    parser.add_argument("--fixed_offset", type=float, default=4.0)
    parser.add_argument("--dataset_dir", default=None)
    parser.add_argument("--batch_size", type=int, default=96)
    parser.add_argument("--augment", choices=["None", "medium", "medium_harder", "super_hard"], default=None)
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--synthetic", type=int, default=2048)
    parser.add_argument("--iters_per_epoch", type=int, default=1000)
    parser.add_argument("--precision", choices=["fp32", "bf16", "bf16_mfma"], default="fp32",
                        help="fp32 = the reference arithmetic; bf16 = bf16 MFMA operands + bf16 activation storage, f32 master weights")
    parsed = parser.parse_args(argv)
    if parsed.pretrained:
        raise SystemExit("--pretrained downloads ImageNet weights (reference resnet.py:175-178); no network here")
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a ROCm GPU")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")
    config = {
        "log_dir": parsed.log_dir, "log_iterations": parsed.log_iterations, "max_epoch": parsed.max_epoch,
        "device": torch.device("cuda", local), "precision": parsed.precision, "optimizer_args": {"lr": parsed.lr},
        "data_args": {"dataset_dir": parsed.dataset_dir, "batch_size": parsed.batch_size, "n_step": N_STEP, "gap": GAP,
                      "augment": parsed.augment, "num_workers": 8},
        "model_args": {"model": "image_ss", "imagenet_pretrained": parsed.pretrained, "backbone": BACKBONE},
        "camera_args": {"w": 384, "h": 160, "fov": 90, "world_y": 1.4, "fixed_offset": parsed.fixed_offset},
        "teacher_args": {"model_path": parsed.teacher_path},
        "synthetic": parsed.synthetic, "iters_per_epoch": parsed.iters_per_epoch, "rank": rank, "world_size": world,
    }
    train(config)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
