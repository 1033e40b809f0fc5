"""Phase-2 (DAgger) helpers, offline part (reference training/phase2_utils.py).

ReplayBuffer keeps the reference's interface (add_data / init_new_weights / update_weights / normalize_weights /
get_highest_k, loss-prioritised eviction and weighted resampling, phase2_utils.py:190-289) but holds the frames as
uint8 tensors resident on the device, so a batch is a gather + /255 on the GPU instead of 4 DataLoader workers.
`repeat` is the np.repeat-style interleave used with --batch_aug (phase2_utils.py:24-47)."""
import numpy as np
import torch

CROP_SIZE = 192
PIXELS_PER_METER = 5


def repeat(a, repeats, dim=0):
    return torch.repeat_interleave(a, repeats, dim=dim)


class ReplayBuffer:
    def __init__(self, device, buffer_limit=100000, sampling=True, seed=0):
        self.device = device
        self.buffer_limit = buffer_limit
        self._sampling = sampling
        self.normalized = False
        self.rgb = None          # (n,160,384,3) u8
        self.birdview = None     # (n,192,192,7) u8
        self.cmd = None
        self.speed = None
        self._weights = np.zeros(0, dtype=np.float64)
        self._rng = np.random.RandomState(seed)
        self._perm, self._perm_pos = None, 0     # epoch shuffle while the weights are not normalised yet

    def __len__(self):
        return 0 if self.rgb is None else self.rgb.shape[0]

    def add_batch(self, rgb_u8, birdview_u8, cmd, speed, weight):
        """rgb_u8 (n,160,384,3), birdview_u8 (n,192,192,7), cmd (n,), speed (n,), weight (n,) -- bulk form of add_data"""
        self.normalized = False
        cat = (lambda a, b: b if a is None else torch.cat([a, b]))
        self.rgb = cat(self.rgb, rgb_u8.to(self.device))
        self.birdview = cat(self.birdview, birdview_u8.to(self.device))
        self.cmd = cat(self.cmd, cmd.float().cpu())
        self.speed = cat(self.speed, speed.float().to(self.device))
        self._weights = np.concatenate([self._weights, np.asarray(weight, dtype=np.float64)])
        if len(self) > self.buffer_limit:                     # pop the samples with the lowest loss (phase2_utils.py:257-261)
            keep = np.sort(np.argsort(self._weights)[len(self) - self.buffer_limit:])
            kt = torch.from_numpy(keep)
            self.rgb, self.birdview = self.rgb[kt.to(self.device)], self.birdview[kt.to(self.device)]
            self.cmd, self.speed = self.cmd[kt], self.speed[kt.to(self.device)]
            self._weights = self._weights[keep]

    def add_data(self, rgb_img, cmd, speed, target, birdview_img, weight):
        self.add_batch(torch.as_tensor(rgb_img)[None], torch.as_tensor(birdview_img)[None], torch.tensor([cmd]), torch.tensor([speed]), [weight])

    def init_new_weights(self):
        """start of an epoch (reference train_image_phase2.py:168): fresh write-back array, fresh shuffle"""
        self._new_weights = self._weights.copy()
        self._perm = None

    def update_weights(self, idxes, losses):
        idx = np.asarray(idxes)
        ok = idx < len(self)
        self._new_weights[idx[ok]] = losses.detach().float().cpu().numpy()[ok]

    def normalize_weights(self):
        self._weights = self._new_weights
        self.normalized = True

    def sample_indices(self, batch_size, epoch_pos=None):
        """Indices of the next batch.  Once the weights are normalised: loss-weighted resampling, one draw per sample
        (reference phase2_utils.py:219-227, weighted_random_choice).  Before that: the reference's
        DataLoader(shuffle=True, drop_last=True) (train_image_phase2.py:170), i.e. consecutive slices of ONE permutation
        per epoch, so every replay sample is visited (and gets its weight written back) exactly once."""
        if self._sampling and self.normalized:
            p = self._weights / self._weights.sum()
            return self._rng.choice(len(self), size=batch_size, p=p)
        if self._perm is None or self._perm_pos + batch_size > len(self._perm) or len(self._perm) != len(self):
            self._perm, self._perm_pos = self._rng.permutation(len(self)), 0      # a new epoch (or the buffer changed)
        out = self._perm[self._perm_pos:self._perm_pos + batch_size]
        self._perm_pos += batch_size
        return out

    def batch(self, idx):
        di = torch.as_tensor(idx, device=self.device)
        rgb = self.rgb[di].permute(0, 3, 1, 2).float().div_(255.0).contiguous()
        bv = self.birdview[di].permute(0, 3, 1, 2).float().contiguous()
        return rgb, bv, self.cmd[torch.as_tensor(idx)], self.speed[di]

    def get_highest_k(self, k):
        top = np.argsort(self._weights)[-k:]
        return (top,) + self.batch(top)
