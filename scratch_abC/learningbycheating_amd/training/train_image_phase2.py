"""Phase-2 (on-policy distillation / DAgger) of the sensorimotor agent -- the offline half
(reference training/train_image_phase2.py:152-258 `_train`), MI355X-native.

`rollout` (train_image_phase2.py:61-149) drives a live CARLA server and is outside the hot path; here the replay buffer is
filled from device-resident synthetic frames (BASELINE.json config 5).  `_train` is the phase-1 step plus, per sample,
the resampling weight get_weight(...) of the selected branch (HIP kernel lbc_phase2_weight) written back to the buffer;
the optimizer is re-created every epoch exactly as the reference does (train_image_phase2.py:164, moments reset)."""
import argparse
import ctypes
import os
import time
from pathlib import Path

import torch
import torch.distributed as dist

from .. import _lib
from ..bird_view.models.birdview import BirdViewPolicyModelSS
from ..bird_view.models.image import ImagePolicyModelSS
from ..bird_view.utils import bz_utils as bzu
from ..bird_view.utils.train_utils import one_hot
from ..optim import FusedAdam
from ..parallel import broadcast_module
from .native import NativeTrainer, camera_struct
from .phase2_utils import ReplayBuffer

BACKBONE = "resnet34"
SAVE_EPISODES = list(range(20))


def phase2_weights(trainer, p_sel, t_sel):
    n = p_sel.shape[0]
    w = torch.empty(n, dtype=torch.float32, device=p_sel.device)
    _lib.check(_lib.get().lbc_phase2_weight(ctypes.byref(trainer.cam), _lib.ptr(p_sel), _lib.ptr(t_sel), n, _lib.ptr(w),
                                            _lib.stream_for(p_sel)), "phase2_weight")
    return w


def _train(replay_buffer, trainer, config, episode):
    device = config["device"]
    bs = config["batch_size"]
    net, teacher_net = trainer.student, trainer.teacher
    for epoch in range(config["epoch_per_episode"]):
        trainer.opt = FusedAdam(list(net.named_parameters()), trainer.eng.grad_views, lr=1e-4)   # fresh moments each epoch
        net.train()
        replay_buffer.init_new_weights()
        for i in range(len(replay_buffer) // bs):                                              # drop_last=True
            idx = replay_buffer.sample_indices(bs)
            rgb, bv, cmd, speed = replay_buffer.batch(idx)
            command = one_hot(cmd).to(device)
            if config["speed_noise"] > 0:
                speed = torch.clamp(speed + torch.randn_like(speed) * config["speed_noise"], 0, 10)
            loss = trainer.step(rgb, speed, command, birdview=bv)
            replay_buffer.update_weights(idx, phase2_weights(trainer, trainer.last_pred[0], trainer.last_teacher[0]))
            if i % int(config["log_iterations"]) == 0:
                bzu.log.scalar(loss_mean=loss.mean().item())
        replay_buffer.normalize_weights()
        # the reference evaluates (eval mode) and visualises the 32 highest-weight samples here (:229-250); visualisation
        # is outside the hot path, the forward is kept so that the same kernels run
        top, rgb, bv, cmd, speed = replay_buffer.get_highest_k(min(32, len(replay_buffer)))
        net.eval()
        with torch.no_grad():
            net(rgb, speed, one_hot(cmd).to(device))
        net.train()
        bzu.log.end_epoch()
    if episode in SAVE_EPISODES and config["rank"] == 0:
        torch.save(net.state_dict(), str(Path(config["log_dir"]) / ("model-%d.th" % episode)))


def synthetic_buffer(n_frames, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    buf = ReplayBuffer(device, buffer_limit=n_frames, seed=seed)
    step = 1024
    for s in range(0, n_frames, step):
        n = min(step, n_frames - s)
        buf.add_batch(torch.randint(0, 256, (n, 160, 384, 3), generator=g, dtype=torch.uint8),
                      (torch.rand((n, 192, 192, 7), generator=g) < 0.1).to(torch.uint8),
                      torch.randint(1, 5, (n,), generator=g), torch.rand(n, generator=g) * 10, [1.0] * n)
    return buf


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--log_dir", required=True)
    parser.add_argument("--log_iterations", default=100)
    parser.add_argument("--max_episode", default=20)
    parser.add_argument("--epoch_per_episode", default=5)
    parser.add_argument("--ckpt", default=None)
    parser.add_argument("--teacher_path", default=None)
    parser.add_argument("--batch_size", type=int, default=128)
    parser.add_argument("--speed_noise", type=float, default=0.0)
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--synthetic", type=int, default=20000, help="frames in the synthetic replay buffer")
    parser.add_argument("--precision", choices=["fp32", "bf16", "bf16_mfma"], default="fp32",
                        help="fp32 = the reference arithmetic; bf16 = bf16 MFMA operands + bf16 activation storage, f32 master weights")
    parsed = parser.parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs a ROCm GPU")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl")
    config = {"log_dir": parsed.log_dir, "log_iterations": parsed.log_iterations, "epoch_per_episode": int(parsed.epoch_per_episode),
              "batch_size": parsed.batch_size, "speed_noise": parsed.speed_noise, "device": device, "rank": rank,
              "precision": parsed.precision,
              "model_args": {"model": "image_ss", "backbone": BACKBONE},
              "agent_args": {"camera_args": {"w": 384, "h": 160, "fov": 90, "world_y": 1.4, "fixed_offset": 4.0}}}
    bzu.log.init(parsed.log_dir, rank)
    bzu.log.save_config({k: v for k, v in config.items() if k != "rank"})
    net = ImagePolicyModelSS(BACKBONE, all_branch=True).to(device)
    if parsed.ckpt:
        net.load_state_dict(torch.load(parsed.ckpt, map_location=device))
    teacher = BirdViewPolicyModelSS("resnet18", all_branch=True).to(device)
    net.precision = teacher.precision = parsed.precision
    if parsed.teacher_path:
        teacher.load_state_dict(torch.load(parsed.teacher_path, map_location=device))
    broadcast_module(net)
    broadcast_module(teacher)
    trainer = NativeTrainer(net, teacher, parsed.batch_size, (3, 160, 384), device, phase=1, lr=parsed.lr, world_size=world,
                            camera=camera_struct())
    buf = synthetic_buffer(parsed.synthetic // world, device, seed=rank)
    for episode in range(int(parsed.max_episode)):
        t0 = time.time()
        _train(buf, trainer, config, episode)
        if rank == 0:
            print("episode %d: %.1f s" % (episode, time.time() - t0))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
