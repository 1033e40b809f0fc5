"""Phase-1 training of the sensorimotor agent (reference training/train_image_phase1.py), MI355X-native.

Same flags, config.json schema and model-%d.th naming as the reference; the loop body runs on the native executor
(NativeTrainer.step = teacher forward, student forward, unprojection + L1 over the four branches, backward, RCCL
gradient all-reduce, fused Adam).  New flags: --synthetic N (device-resident synthetic frames instead of the LMDB
dataset), --iters_per_epoch, and one-process-per-GPU launch through torch.distributed.run.

CoordConverter / LocationLoss keep the reference's call signatures (train_image_phase1.py:35-70) for callers that
want the map-space waypoints; they are thin torch expressions over (N,4,5,2) tensors, the timed path uses the fused
HIP loss kernel instead."""
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
SAVE_EPOCHS = [1, 2, 4, 8, 16, 32, 64, 128, 192, 256]


class CoordConverter():
    def __init__(self, w=384, h=160, fov=90, world_y=1.4, fixed_offset=2.0, device="cuda"):
        self._img_size = torch.tensor([float(w), float(h)], device=device)
        self._w, self._h, self._fov, self._world_y, self._fixed_offset = w, h, fov, world_y, fixed_offset

    def __call__(self, camera_locations):
        loc = (camera_locations + 1) * self._img_size / 2
        f = self._w / (2 * np.tan(self._fov * np.pi / 360))
        xt = (loc[..., 0] - self._w / 2) / f
        yt = (loc[..., 1] - self._h / 2) / f
        world_z = self._world_y / yt
        world_x = world_z * xt
        return torch.stack([world_x * PIXELS_PER_METER + CROP_SIZE / 2,
                            CROP_SIZE - world_z * PIXELS_PER_METER + self._fixed_offset * PIXELS_PER_METER], dim=-1)


class LocationLoss(torch.nn.Module):
    def forward(self, pred_locations, teac_locations):
        pred_locations = pred_locations / (0.5 * CROP_SIZE) - 1
        return torch.mean(torch.abs(pred_locations - teac_locations), dim=(1, 2, 3))


def train_or_eval(trainer, data, is_train, config, is_first_epoch):
    """reference train_image_phase1.py:157-229; epoch 0 is the reference's 11-iteration dry run without updates"""
    tick = time.time()
    device = config["device"]
    for i, (rgb_image, birdview, location, command, speed) in enumerate(data):
        command = one_hot(command).to(device)
        if is_train and config["speed_noise"] > 0:
            speed = torch.clamp(speed + torch.randn_like(speed) * config["speed_noise"], 0, 10)
        loss = trainer.step(rgb_image, speed, command, birdview=birdview, update=is_train and not is_first_epoch, train_mode=is_train)
        should_log = (i % int(config["log_iterations"]) == 0) or (not is_train) or is_first_epoch
        if should_log:
            lm = loss.mean().item()          # device->host sync only when logging, as the reference (:207-221)
            if not np.isfinite(lm):
                raise FloatingPointError("phase-1 loss is %s: a predicted waypoint reached the horizon (1/y pole of the "
                                         "unprojection); start from a phase-0 checkpoint" % lm)
            bzu.log.scalar(is_train=is_train, loss_mean=lm)
        now = time.time()
        bzu.log.scalar(is_train=is_train, fps=1.0 / max(now - tick, 1e-9), images_per_sec=rgb_image.shape[0] * config["world_size"] / max(now - tick, 1e-9))
        tick = now
        if is_first_epoch and i == 10:
            break


def train(config):
    rank, world = config["rank"], config["world_size"]
    device = config["device"]
    bzu.log.init(config["log_dir"], rank)
    bzu.log.save_config({k: v for k, v in config.items() if k not in ("rank", "world_size")})
    teacher_config = bzu.log.load_config(config["teacher_args"]["model_path"]) if config["teacher_args"]["model_path"] else {"model_args": {"backbone": "resnet18"}}

    net = ImagePolicyModelSS(config["model_args"]["backbone"], pretrained=config["model_args"]["imagenet_pretrained"], all_branch=True).to(device)
    if config["phase0_ckpt"]:
        net.load_state_dict(torch.load(config["phase0_ckpt"], map_location=device))
    teacher_net = BirdViewPolicyModelSS(teacher_config["model_args"]["backbone"], pretrained=True, all_branch=True).to(device)
    net.precision = teacher_net.precision = config.get("precision", "fp32")
    if config["teacher_args"]["model_path"]:
        teacher_net.load_state_dict(torch.load(config["teacher_args"]["model_path"], map_location=device))
    teacher_net.eval()
    broadcast_module(net)
    broadcast_module(teacher_net)

    bs = config["data_args"]["batch_size"] * int(config["data_args"].get("batch_aug", 1) or 1)
    data_train, data_val = make_loaders(config, device, rank, world)
    cam = camera_struct(**{k: float(v) for k, v in config["agent_args"]["camera_args"].items()})
    trainer = NativeTrainer(net, teacher_net, bs, (3, 160, 384), device, phase=1, lr=config["optimizer_args"]["lr"],
                            world_size=world, camera=cam)
    for epoch in range(int(config["max_epoch"]) + 1):
        net.train()
        train_or_eval(trainer, data_train, True, config, epoch == 0)
        net.eval()                              # reference train_image_phase1.py:255-256: a validation pass after every epoch
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
    parser.add_argument("--max_epoch", default=256)
    parser.add_argument("--pretrained", action="store_true")
    parser.add_argument("--ckpt", default=None, help="phase-0 checkpoint (required by the reference; optional with --synthetic)")
    parser.add_argument("--teacher_path", default=None)
    parser.add_argument("--fixed_offset", type=float, default=4.)
    parser.add_argument("--batch_aug", type=int, default=1)
    parser.add_argument("--dataset_dir", default=None)
    parser.add_argument("--batch_size", type=int, default=24)
    parser.add_argument("--speed_noise", type=float, default=0.0)
    parser.add_argument("--augment", choices=["medium", "medium_harder", "super_hard", "None", "custom"], default="super_hard")
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--synthetic", type=int, default=2048, help="number of device-resident synthetic frames (used when no --dataset_dir is given)")
    parser.add_argument("--iters_per_epoch", type=int, default=1000)
    parser.add_argument("--precision", choices=["fp32", "bf16", "bf16_mfma"], default="fp32",
                        help="fp32 = the reference arithmetic; bf16 = bf16 MFMA operands + bf16 activation storage, f32 master weights")
    parsed = parser.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a ROCm GPU")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")
    config = {
        "log_dir": parsed.log_dir, "log_iterations": parsed.log_iterations, "max_epoch": parsed.max_epoch,
        "device": torch.device("cuda", local), "precision": parsed.precision, "phase0_ckpt": parsed.ckpt, "optimizer_args": {"lr": parsed.lr},
        "speed_noise": parsed.speed_noise,
        "data_args": {"dataset_dir": parsed.dataset_dir, "batch_size": parsed.batch_size, "n_step": N_STEP, "gap": GAP,
                      "augment": parsed.augment, "batch_aug": parsed.batch_aug, "num_workers": 8},
        "model_args": {"model": "image_ss", "imagenet_pretrained": parsed.pretrained, "backbone": BACKBONE},
        "teacher_args": {"model_path": parsed.teacher_path},
        "agent_args": {"camera_args": {"w": 384, "h": 160, "fov": 90, "world_y": 1.4, "fixed_offset": 4.0}},
        "synthetic": parsed.synthetic, "iters_per_epoch": parsed.iters_per_epoch, "rank": rank, "world_size": world,
    }
    train(config)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
