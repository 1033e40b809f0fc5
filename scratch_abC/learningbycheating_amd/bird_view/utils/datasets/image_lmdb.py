"""The offline CARLA frame dataset (reference bird_view/utils/datasets/image_lmdb.py:59-293), MI355X-native.

On-disk format = the reference's: one LMDB environment per episode with keys `len`, `rgb_%04d` (160x384x3 uint8),
`birdview_%04d` (320x320x7 uint8), `measurements_%04d` (17 float32: position 3, orientation 2, velocity 3, acceleration 3,
command, steer, throttle, brake, manual, gear) written by data_collector.py:234-252; read here through the format restatement
in lmdb_format.py (the `lmdb` package is not needed).

Two ways out of the dataset:
  * `ImageDataset.__getitem__` / `get_image()`: the reference's per-sample contract (float CHW tensors, torch DataLoader
    over `Wrap`), for code that wants exactly that.  Host-side numpy only, no augmentation (imgaug is not available).
  * `DeviceLoader`: what the training scripts use.  A batch leaves the host as the stored uint8 bytes (rgb 184 KB +
    bird-view 717 KB per frame, straight from the memory-mapped file into pinned staging, asynchronous H2D); the fixed
    bird-view crop (rows 58:250, cols 64:256 -- image_lmdb.py:150-163 with dx = 0, dy = -PIXEL_OFFSET), the colour
    augmentation (bird_view/augmenter.py) and the `--batch_aug` replication run on the GPU; /255, ImageNet normalisation
    and the NHWC repack are fused into the networks' first kernel (lbc_net_forward_u8).  The zero-angle cv2.warpAffine
    of image_lmdb.py:155-156 is the identity and is skipped.
"""
import glob
import os
from pathlib import Path

import numpy as np
import torch

from ... import augmenter as augmenter_mod
from .... import _lib
from .lmdb_format import LmdbReader, write_lmdb

PIXEL_OFFSET = 10
PIXELS_PER_METER = 5
CROP_Y0, CROP_X0 = 58, 64        # -PIXEL_OFFSET + (260 - 96) - 96, 160 - 96


def world_to_pixel(x, y, ox, oy, ori_ox, ori_oy, offset=(-80, 160), size=320, angle_jitter=15):
    """reference image_lmdb.py:22-30"""
    pixel_dx, pixel_dy = (x - ox) * PIXELS_PER_METER, (y - oy) * PIXELS_PER_METER
    pixel_x = pixel_dx * ori_ox + pixel_dy * ori_oy
    pixel_y = -pixel_dx * ori_oy + pixel_dy * ori_ox
    pixel_x = 320 - pixel_x
    return np.array([pixel_x, pixel_y]) + offset


def warp_params(delta_angle, dx, dy, crop_size=192, center=(160, 260)):
    """lbc_warp_params of one sample as 7 float64 (6 matrix entries + the two int32 origins packed into the 7th): the INVERSE of
    cv2.getRotationMatrix2D(center, delta_angle, 1.0) exactly as cv2.warpAffine derives it (imgwarp.cpp: D = M0 M4 - M1 M3, ...), and the
    window origin of birdview_lmdb.py:116-121 (rows dy + 164 - 96 .., columns dx + 160 - 96 ..)"""
    a = np.deg2rad(np.float64(delta_angle))
    alpha, beta = np.cos(a), np.sin(a)
    cx, cy = np.float64(center[0]), np.float64(center[1])
    m = np.array([alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy], np.float64)
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m0, m1, m3, m4 = A11, m[1] * -D, m[3] * -D, A22
    b1 = -m0 * m[2] - m1 * m[5]
    b2 = -m3 * m[2] - m4 * m[5]
    out = np.empty(7, np.float64)
    out[:6] = (m0, m1, b1, m3, m4, b2)
    center_y = center[1] - crop_size // 2
    out[6:7].view(np.int32)[:] = (int(dy) + center_y - crop_size // 2, int(dx) + center[0] - crop_size // 2)
    return out


class ImageDataset(torch.utils.data.Dataset):
    """reference image_lmdb.py:59-222 (same constructor arguments)"""

    def __init__(self, dataset_path, rgb_shape=(160, 384, 3), img_size=320, crop_size=192, gap=5, n_step=5, gaussian_radius=1.,
                 down_ratio=4, augment_strategy=None, batch_read_number=819200, batch_aug=1):
        self.rgb_shape, self.img_size, self.crop_size = rgb_shape, img_size, crop_size
        self.gap, self.n_step, self.down_ratio, self.batch_aug = gap, n_step, down_ratio, batch_aug
        self.augment_strategy = augment_strategy if augment_strategy not in (None, "None") else None
        self.batch_read_number = batch_read_number
        self.envs, self.file_map, self.idx_map = [], [], []
        for full_path in sorted(glob.glob("%s/**" % dataset_path)):
            if not (os.path.isdir(full_path) and os.path.exists(os.path.join(full_path, "data.mdb"))):
                continue
            env = LmdbReader(full_path)
            n = int(bytes(env.get("len"))) - self.gap * self.n_step
            e = len(self.envs)
            self.envs.append(env)
            for i in range(max(n, 0)):
                self.file_map.append(e)
                self.idx_map.append(i)
        if not self.envs:
            raise RuntimeError("no LMDB episodes under %s" % dataset_path)
        print("Finished loading %s. Length: %d" % (dataset_path, len(self.file_map)))

    def __len__(self):
        return len(self.file_map)

    def raw(self, idx, delta_angle=0, dx=0, dy=-PIXEL_OFFSET):
        """the stored bytes of sample idx as zero-copy uint8 views + the derived targets:
        (rgb (160,384,3) u8, birdview (320,320,7) u8, locations (n_step,2) f64 pixels of the 192-crop, cmd f32, speed f32).
        delta_angle (degrees) / dx / dy: the rotation and window jitter of the privileged agent's loader (reference
        birdview_lmdb.py:103-151; image_lmdb.py fixes them at 0 / 0 / -PIXEL_OFFSET) -- the waypoints follow, the caller warps the map"""
        env = self.envs[self.file_map[idx]]
        index = self.idx_map[idx]
        bird_view = np.frombuffer(env.get("birdview_%04d" % index), np.uint8).reshape(320, 320, 7)
        measurement = np.frombuffer(env.get("measurements_%04d" % index), np.float32)
        # (the privileged agent never looks at the camera frame: reference birdview_lmdb.py:107 sets rgb_image = None)
        rgb_image = np.frombuffer(env.get("rgb_%04d" % index), np.uint8).reshape(160, 384, 3) if getattr(self, "needs_rgb", True) else None
        ox, oy, oz, ori_ox, ori_oy, vx, vy, vz, ax, ay, az, cmd, steer, throttle, brake, manual, gear = measurement
        speed = np.linalg.norm([vx, vy, vz])
        angle = np.arctan2(ori_oy, ori_ox) + np.deg2rad(delta_angle)      # (image_lmdb.py:146,165-166 with delta_angle = 0)
        ori_ox, ori_oy = np.cos(angle), np.sin(angle)
        locations = []
        for dt in range(self.gap, self.gap * (self.n_step + 1), self.gap):
            f = np.frombuffer(env.get("measurements_%04d" % (index + dt)), np.float32)
            x, y = f[0], f[1]
            pixel_y, pixel_x = world_to_pixel(x, y, ox, oy, ori_ox, ori_oy, size=self.img_size)
            pixel_x = pixel_x - (self.img_size - self.crop_size) // 2
            pixel_y = self.crop_size - (self.img_size - pixel_y) + 70
            locations.append([pixel_x - dx, pixel_y - dy])
        return rgb_image, bird_view, np.array(locations), np.float32(cmd), np.float32(speed)

    def __getitem__(self, idx):
        rgb_image, bird_view, locations, cmd, speed = self.raw(idx)
        bird_view = bird_view[CROP_Y0:CROP_Y0 + self.crop_size, CROP_X0:CROP_X0 + self.crop_size]
        to_tensor = lambda a: torch.from_numpy(np.array(a)).permute(2, 0, 1).float().div(255.0)   # transforms.ToTensor on uint8 HWC
        rgb = to_tensor(rgb_image)
        if self.batch_aug > 1:
            rgb = torch.stack([rgb for _ in range(self.batch_aug)])
        self.batch_read_number += 1
        return rgb, to_tensor(bird_view), locations, cmd, speed


class Wrap(torch.utils.data.Dataset):
    """reference image_lmdb.py:250-260: an "epoch" is batch_size * samples random draws with replacement"""

    def __init__(self, data, batch_size, samples):
        self.data, self.batch_size, self.samples = data, batch_size, samples

    def __len__(self):
        return self.batch_size * self.samples

    def __getitem__(self, i):
        return self.data[np.random.randint(len(self.data))]


def get_image(dataset_dir, batch_size=32, num_workers=0, shuffle=True, augment=None, n_step=5, gap=5, batch_aug=1):
    """reference image_lmdb.py:270-293: (train, val) torch DataLoaders with the reference's per-sample tensors"""
    def make_dataset(dir_name, is_train):
        data = ImageDataset(str(Path(dataset_dir) / dir_name), gap=gap, n_step=n_step, augment_strategy=augment if is_train else None,
                            batch_aug=batch_aug if is_train else 1)
        data = Wrap(data, batch_size, 1000 if is_train else 10)
        return torch.utils.data.DataLoader(data, batch_size=batch_size, num_workers=num_workers if is_train else 0, shuffle=True, drop_last=True)
    return make_dataset("train", True), make_dataset("val", False)


class DeviceLoader:
    """Batches of the stored uint8 frames on the GPU.  Iterating yields `samples` batches of
    (rgb u8 (B*batch_aug,160,384,3), birdview u8 (B*batch_aug,192,192,7), location f32 (B*batch_aug,5,2), command f32 CPU
    (B*batch_aug,), speed f32 (B*batch_aug,)), sampled with replacement by a per-rank stream (Wrap semantics), the next
    batch's H2D copy overlapping the consumer's compute."""

    def __init__(self, dataset, batch_size, samples, device, augment=None, batch_aug=1, seed=0, rank=0, crop=True):
        self.data, self.batch, self.samples, self.device = dataset, batch_size, samples, torch.device(device)
        self.batch_aug = batch_aug
        self.rng = np.random.RandomState(seed * 9973 + rank)
        self.strategy = augmenter_mod.get(augment)
        self.aug = augmenter_mod.BatchAugmenter(None, seed=seed * 31 + rank) if self.strategy else None
        self.images_seen = dataset.batch_read_number
        cuda = self.device.type == "cuda"
        # Staging in pageable memory unless LBC_PIN_STAGING=1: on the MI355X boxes a pinned buffer that the GPU has read since
        # the CPU last wrote it is slow to rewrite (184 KB: 3.2 ms, then 5.7 ms for the next H2D -- scripts/diag_latency.py),
        # and these buffers are rewritten for every batch; from pageable memory the runtime stages the copy itself
        import os
        pin = (lambda t: t.pin_memory()) if (cuda and os.environ.get("LBC_PIN_STAGING") == "1") else (lambda t: t)
        B = batch_size
        # per-sample rotation / window jitter (BirdViewDataset): parameters of lbc_birdview_warp_crop_u8, 56 bytes per image
        self.jitter = bool(getattr(dataset, "angle_jitter", 0) or getattr(dataset, "crop_x_jitter", 0) or getattr(dataset, "crop_y_jitter", 0))
        self.with_rgb = bool(getattr(dataset, "needs_rgb", True))     # bird-view datasets: no camera frame is read, staged or uploaded
        rgb_b = B if self.with_rgb else 0
        self.stage = [{"rgb": pin(torch.empty((rgb_b, 160, 384, 3), dtype=torch.uint8)), "bv": pin(torch.empty((B, 320, 320, 7), dtype=torch.uint8)),
                       "loc": pin(torch.empty((B, dataset.n_step, 2))), "speed": pin(torch.empty(B)), "cmd": torch.empty(B),
                       "warp": pin(torch.zeros((B, 7), dtype=torch.float64))} for _ in range(2)]
        self.dev = [{"rgb": torch.empty((rgb_b, 160, 384, 3), dtype=torch.uint8, device=self.device),
                     "bv": torch.empty((B, 320, 320, 7), dtype=torch.uint8, device=self.device),
                     "loc": torch.empty((B, dataset.n_step, 2), device=self.device), "speed": torch.empty(B, device=self.device),
                     "warp": torch.zeros((B, 7), dtype=torch.float64, device=self.device)} for _ in range(2)]
        self.copy = torch.cuda.Stream(device=self.device) if cuda else None
        self.ready = [torch.cuda.Event(), torch.cuda.Event()] if cuda else None
        self.free = [torch.cuda.Event(), torch.cuda.Event()] if cuda else None

    def __len__(self):
        return self.samples

    def _fill(self, k):
        st = self.stage[k]
        if self.ready is not None:
            self.ready[k].synchronize()          # the previous H2D out of this staging slot is done
        rgb_np, bv_np = st["rgb"].numpy(), st["bv"].numpy()
        draw = getattr(self.data, "sample_index", None)          # command-biased sampling (BiasedBirdViewDataset), else uniform with replacement
        indices = [draw(self.rng) for _ in range(self.batch)] if draw else self.rng.randint(len(self.data), size=self.batch)
        for i, idx in enumerate(indices):
            if self.jitter:
                delta_angle, dx, dy = self.data.draw_jitter(self.rng)
                rgb, bv, loc, cmd, speed = self.data.raw(int(idx), delta_angle, dx, dy)
                st["warp"][i] = torch.from_numpy(warp_params(delta_angle, dx, dy, self.data.crop_size))
            else:
                rgb, bv, loc, cmd, speed = self.data.raw(int(idx))
            if self.with_rgb:
                np.copyto(rgb_np[i], rgb)
            np.copyto(bv_np[i], bv)
            st["loc"][i] = torch.from_numpy(loc.astype(np.float32))
            st["cmd"][i] = float(cmd)
            st["speed"][i] = float(speed)
        d = self.dev[k]
        keys = (("rgb",) if self.with_rgb else ()) + (("bv", "loc", "speed", "warp") if self.jitter else ("bv", "loc", "speed"))
        if self.copy is None:
            for key in keys:
                d[key].copy_(st[key])
            return
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(self.free[k])
            for key in keys:
                d[key].copy_(st[key], non_blocking=True)
            self.ready[k].record(self.copy)

    def __iter__(self):
        if self.free is not None:
            for e in self.free:
                e.record()
        self._fill(0)
        for i in range(self.samples):
            k = i & 1
            if i + 1 < self.samples:
                self._fill(k ^ 1)                # stage + enqueue the next batch while this one is consumed
            d, st = self.dev[k], self.stage[k]
            if self.ready is not None:
                torch.cuda.current_stream(self.device).wait_event(self.ready[k])
            n = self.batch
            bv = torch.empty((n, self.data.crop_size, self.data.crop_size, 7), dtype=torch.uint8, device=self.device)
            if self.jitter:
                _lib.check(_lib.get().lbc_birdview_warp_crop_u8(_lib.ptr(d["bv"]), _lib.ptr(bv), _lib.ptr(d["warp"]), n, 320, 320, 7, self.data.crop_size,
                                                                self.data.crop_size, _lib.stream_for(bv)), "birdview_warp_crop_u8")
            else:
                _lib.check(_lib.get().lbc_birdview_crop_u8(_lib.ptr(d["bv"]), _lib.ptr(bv), n, 320, 320, 7, CROP_Y0, CROP_X0, self.data.crop_size,
                                                           self.data.crop_size, _lib.stream_for(bv)), "birdview_crop_u8")
            rgb, loc, speed, cmd = (d["rgb"] if self.with_rgb else None), d["loc"], d["speed"], st["cmd"].clone()
            if self.batch_aug > 1:               # reference train_image_phase1.py:131-154,183-189: every frame batch_aug times, back to back
                r = self.batch_aug
                bv, loc, speed, cmd = (t.repeat_interleave(r, dim=0) for t in (bv, loc, speed, cmd))
                rgb = rgb.repeat_interleave(r, dim=0) if rgb is not None else None
            else:
                rgb = rgb.clone() if rgb is not None else None     # the augmentation works in place; the slot is refilled by the copy stream
                loc, speed = loc.clone(), speed.clone()
            if self.strategy is not None and rgb is not None:
                self.aug.recipe = self.strategy(self.images_seen)       # strength schedule by images read (image_lmdb.py:138,220)
                self.aug.augment_batch(rgb)
            self.images_seen += n * self.batch_aug
            if self.free is not None:
                self.free[k].record(torch.cuda.current_stream(self.device))
            yield rgb, bv, loc, cmd, speed


def get_image_device(dataset_dir, batch_size, device, augment=None, n_step=5, gap=5, batch_aug=1, samples=(1000, 10), seed=0, rank=0):
    """(train, val) DeviceLoaders with the reference's epoch definition (1000 / 10 random batches; validation without
    augmentation or batch_aug: image_lmdb.py:278-281)"""
    train = DeviceLoader(ImageDataset(str(Path(dataset_dir) / "train"), gap=gap, n_step=n_step, augment_strategy=augment, batch_aug=batch_aug),
                         batch_size, samples[0], device, augment=augment, batch_aug=batch_aug, seed=seed, rank=rank)
    val = DeviceLoader(ImageDataset(str(Path(dataset_dir) / "val"), gap=gap, n_step=n_step), batch_size, samples[1], device, seed=seed + 1, rank=rank)
    return train, val


# ---- synthetic episodes in the reference's on-disk format ------------------------------------------------------------
def write_synthetic_episode(path, n_frames, seed=0):
    """one LMDB episode with random frames and a smooth random trajectory (for tests and benchmarks without CARLA data)"""
    rng = np.random.RandomState(seed)
    items = {"len": str(n_frames).encode()}
    heading, pos = rng.uniform(-np.pi, np.pi), rng.uniform(-50, 50, size=2)
    speed = rng.uniform(2, 8)
    for i in range(n_frames):
        heading += rng.normal(0, 0.01) + 0.004
        speed = float(np.clip(speed + rng.normal(0, 0.05), 0.5, 10))
        pos = pos + 0.1 * speed * np.array([np.cos(heading), np.sin(heading)])
        m = np.zeros(17, np.float32)
        m[0:3] = [pos[0], pos[1], 0.0]
        m[3:5] = [np.cos(heading), np.sin(heading)]
        m[5:8] = [speed * np.cos(heading), speed * np.sin(heading), 0.0]
        m[11] = float(rng.randint(1, 5))
        items["rgb_%04d" % i] = rng.randint(0, 256, (160, 384, 3), dtype=np.uint8).tobytes()
        items["birdview_%04d" % i] = ((rng.random_sample((320, 320, 7)) < 0.1) * 255).astype(np.uint8).tobytes()
        items["measurements_%04d" % i] = m.tobytes()
        items["control_%04d" % i] = np.zeros(3, np.float32).tobytes()
    return write_lmdb(path, items)


def write_synthetic_dataset(root, episodes=2, frames=60, seed=0):
    for split, off in (("train", 0), ("val", 1000)):
        for e in range(episodes):
            write_synthetic_episode(os.path.join(root, split, "%03d" % e), frames, seed + off + e)
    return root
