"""Synthetic stand-in for the offline CARLA LMDB dataset (reference bird_view/utils/datasets/image_lmdb.py:59-293),
same tensor contract per batch: (rgb (B,3,160,384) f32 in [0,1], birdview (B,7,192,192) f32 {0,1}, location (B,5,2),
command (B,) in {1..4}, speed (B,) m/s).  Frames are generated once as uint8 and kept resident on the device
(113 KB + 252 KB per frame) so the step measures the hot path, not a host loader; the real LMDB reader needs the
`lmdb`/`cv2` packages that this image does not ship."""
import torch


class SyntheticFrames:
    def __init__(self, n_frames, device, seed=0, rank=0, world=1):
        g = torch.Generator().manual_seed(seed)
        self.rgb = torch.randint(0, 256, (n_frames, 160, 384, 3), generator=g, dtype=torch.uint8).to(device)
        self.birdview = (torch.rand((n_frames, 7, 192, 192), generator=g) < 0.1).to(torch.uint8).to(device)
        self.speed = (torch.rand(n_frames, generator=g) * 10).to(device)
        self.command = torch.randint(1, 5, (n_frames,), generator=g).float()
        self.location = (torch.rand((n_frames, 5, 2), generator=g) * 192).to(device)
        self.n = n_frames
        self.gen = torch.Generator().manual_seed(seed * 7919 + rank)   # per-rank sampling stream (Wrap samples with replacement)
        self.device = device

    def batch(self, batch_size):
        idx = torch.randint(0, self.n, (batch_size,), generator=self.gen)
        di = idx.to(self.device)
        rgb = self.rgb[di].permute(0, 3, 1, 2).float().div_(255.0).contiguous()
        bv = self.birdview[di].float()
        return rgb, bv, self.location[di], self.command[idx], self.speed[di]


def loader(frames, batch_size, n_batches):
    for _ in range(n_batches):
        yield frames.batch(batch_size)
