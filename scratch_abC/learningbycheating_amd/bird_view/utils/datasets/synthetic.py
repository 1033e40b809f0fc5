"""Synthetic stand-in for the offline CARLA LMDB dataset when no --dataset_dir is given (reference
bird_view/utils/datasets/image_lmdb.py:59-293): frames are generated once as the uint8 tensors the dataset stores and kept
resident on the device (184 KB + 258 KB per frame), so a training run measures the hot path, not a host loader.
A batch = (rgb u8 (B,160,384,3), birdview u8 (B,192,192,7) 0/255 masks, location (B,5,2) pixels, command (B,) in {1..4} on
the host, speed (B,) m/s): the same contract as datasets.image_lmdb.DeviceLoader -- /255, normalisation and the NHWC repack
happen in the networks' first kernel (lbc_net_forward_u8), nothing is decoded with torch ops."""
import torch


class SyntheticFrames:
    def __init__(self, n_frames, device, seed=0, rank=0, world=1):
        g = torch.Generator().manual_seed(seed)
        self.rgb = torch.randint(0, 256, (n_frames, 160, 384, 3), generator=g, dtype=torch.uint8).to(device)
        self.birdview = ((torch.rand((n_frames, 192, 192, 7), generator=g) < 0.1).to(torch.uint8) * 255).to(device)
        self.speed = (torch.rand(n_frames, generator=g) * 10).to(device)
        self.command = torch.randint(1, 5, (n_frames,), generator=g).float()
        self.location = (torch.rand((n_frames, 5, 2), generator=g) * 192).to(device)
        self.n = n_frames
        self.gen = torch.Generator().manual_seed(seed * 7919 + rank)   # per-rank sampling stream (Wrap samples with replacement)
        self.device = device

    def batch(self, batch_size):
        idx = torch.randint(0, self.n, (batch_size,), generator=self.gen)
        di = idx.to(self.device)
        return self.rgb[di], self.birdview[di], self.location[di], self.command[idx], self.speed[di]


def loader(frames, batch_size, n_batches):
    for _ in range(n_batches):
        yield frames.batch(batch_size)
