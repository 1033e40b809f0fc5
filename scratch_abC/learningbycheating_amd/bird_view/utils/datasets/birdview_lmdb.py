"""The privileged agent's dataset (reference bird_view/utils/datasets/birdview_lmdb.py:33-199, get_birdview :247-285): the same LMDB
episodes as ImageDataset, sampled with a random rotation of the bird-view about the ego pixel (160, 260) (angle_jitter degrees), a jittered
192 x 192 window (crop_x_jitter, crop_y_jitter) with the waypoints shifted and rotated to match, optionally command-biased sampling and a
frame cap.  Bytes stay uint8 until the GPU: DeviceLoader uploads the stored 320 x 320 x 7 map and lbc_birdview_warp_crop_u8 does the
cv2.warpAffine + crop of the reference's CPU workers there."""
import glob
import os
from pathlib import Path

import numpy as np

from .image_lmdb import PIXEL_OFFSET, DeviceLoader, ImageDataset
from .lmdb_format import LmdbReader


class BirdViewDataset(ImageDataset):
    """reference birdview_lmdb.py:33-166 (same constructor arguments; episodes in reverse-sorted order, frame cap as there)"""
    needs_rgb = False        # reference birdview_lmdb.py:107: rgb_image = None -- raw() does not read it, DeviceLoader does not upload it

    def __init__(self, dataset_path, img_size=320, crop_size=192, gap=5, n_step=5, crop_x_jitter=5, crop_y_jitter=5, angle_jitter=5,
                 down_ratio=4, gaussian_radius=1.0, max_frames=None):
        self.rgb_shape, self.img_size, self.crop_size = (160, 384, 3), img_size, crop_size
        self.gap, self.n_step, self.down_ratio, self.batch_aug = gap, n_step, down_ratio, 1
        self.augment_strategy, self.batch_read_number = None, 819200
        self.crop_x_jitter, self.crop_y_jitter, self.angle_jitter, self.max_frames = int(crop_x_jitter), int(crop_y_jitter), int(angle_jitter), max_frames
        self.envs, self.file_map, self.idx_map = [], [], []
        n_episodes = 0
        for full_path in sorted(glob.glob("%s/**" % dataset_path), reverse=True):
            if not (os.path.isdir(full_path) and os.path.exists(os.path.join(full_path, "data.mdb"))):
                continue
            env = LmdbReader(full_path)
            n = int(bytes(env.get("len"))) - self.gap * self.n_step
            e = len(self.envs)
            self.envs.append(env)
            for i in range(max(n, 0)):
                if max_frames and len(self) >= max_frames:
                    break
                self.file_map.append(e)
                self.idx_map.append(i)
            n_episodes += 1
            if max_frames and len(self) >= max_frames:
                break
        if not self.envs:
            raise RuntimeError("no LMDB episodes under %s" % dataset_path)
        print("%s: %d frames, %d episodes." % (dataset_path, len(self), n_episodes))

    def __getitem__(self, idx):
        # the reference's per-sample path (birdview_lmdb.py:95-151) rotates and crops the map with cv2 on a CPU worker; here that is
        # one GPU kernel over the batch (lbc_birdview_warp_crop_u8), so a single CPU sample is not what this class hands out
        raise NotImplementedError("BirdViewDataset yields batches through DeviceLoader (get_birdview_device): the rotation + crop of the "
                                  "map runs on the GPU; use raw(idx, delta_angle, dx, dy) for the stored bytes and the jittered waypoints")

    def draw_jitter(self, rng):
        """(delta_angle, dx, dy) as birdview_lmdb.py:103-105 draws them (integers; dy carries the fixed -PIXEL_OFFSET)"""
        delta_angle = int(rng.randint(-self.angle_jitter, self.angle_jitter + 1))
        dx = int(rng.randint(-self.crop_x_jitter, self.crop_x_jitter + 1))
        dy = int(rng.randint(0, self.crop_y_jitter + 1)) - PIXEL_OFFSET
        return delta_angle, dx, dy


class BiasedBirdViewDataset(BirdViewDataset):
    """reference birdview_lmdb.py:169-199: a sample = a command drawn with the given ratios, then a frame of that command (frames that
    stand still or follow the lane count as command 4)"""

    def __init__(self, dataset_path, left_ratio=0.25, right_ratio=0.25, straight_ratio=0.25, **kwargs):
        super().__init__(dataset_path, **kwargs)
        self._choices = [1, 2, 3, 4]
        self._weights = [left_ratio, right_ratio, straight_ratio, 1 - left_ratio - right_ratio - straight_ratio]
        self.cmd_map = {i: [] for i in range(1, 5)}
        for idx in range(len(self.file_map)):
            m = np.frombuffer(self.envs[self.file_map[idx]].get("measurements_%04d" % self.idx_map[idx]), np.float32)
            cmd, speed = m[11], np.linalg.norm(m[5:8])
            self.cmd_map[int(cmd) if (cmd != 4 and speed > 1.0) else 4].append(idx)
        for cmd, nums in self.cmd_map.items():
            print(cmd, len(nums))

    def sample_index(self, rng):
        cmd = int(rng.choice(self._choices, p=self._weights))
        frames = self.cmd_map[cmd]
        if not frames:
            raise RuntimeError("command-biased sampling: the dataset holds no frame of command %d" % cmd)
        return frames[int(rng.randint(len(frames)))]


def get_birdview_device(dataset_dir, batch_size, device, crop_x_jitter=0, crop_y_jitter=0, angle_jitter=0, n_step=5, gap=5, max_frames=None,
                        cmd_biased=False, samples=(1000, 10), seed=0, rank=0):
    """(train, val) DeviceLoaders as reference get_birdview (birdview_lmdb.py:247-285): jitter, frame cap and biased sampling on the
    training split only; 1000 / 10 random batches per epoch"""
    cls = BiasedBirdViewDataset if cmd_biased else BirdViewDataset
    train = cls(str(Path(dataset_dir) / "train"), gap=gap, n_step=n_step, crop_x_jitter=crop_x_jitter, crop_y_jitter=crop_y_jitter,
                angle_jitter=angle_jitter, max_frames=max_frames)
    val = BirdViewDataset(str(Path(dataset_dir) / "val"), gap=gap, n_step=n_step, crop_x_jitter=0, crop_y_jitter=0, angle_jitter=0)
    return (DeviceLoader(train, batch_size, samples[0], device, seed=seed, rank=rank),
            DeviceLoader(val, batch_size, samples[1], device, seed=seed + 1, rank=rank))
