"""Train / validation batch sources of the training scripts: the LMDB dataset (--dataset_dir, reference
bird_view/utils/datasets/image_lmdb.py:270-293 get_image) or device-resident synthetic frames (--synthetic N).  Both hand out
the dataset's uint8 frames on the device: (rgb u8 (B,160,384,3), birdview u8 (B,192,192,7), location (B,5,2), command (B,) on
the host, speed (B,)); the colour augmentation (--augment) and --batch_aug run on the GPU for either source."""
import torch

from ..bird_view import augmenter as augmenter_mod
from ..bird_view.utils.datasets.synthetic import SyntheticFrames


class _SyntheticLoader:
    def __init__(self, frames, batch_size, n_batches, augment=None, batch_aug=1, seed=0):
        self.frames, self.batch, self.n_batches, self.batch_aug = frames, batch_size, n_batches, batch_aug
        self.strategy = augmenter_mod.get(augment)
        self.aug = augmenter_mod.BatchAugmenter(None, seed=seed) if self.strategy else None
        self.images_seen = 819200            # the reference's ImageDataset starts its strength counter here (image_lmdb.py:72)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for _ in range(self.n_batches):
            rgb, bv, loc, cmd, speed = self.frames.batch(self.batch)
            if self.batch_aug > 1:           # reference train_image_phase1.py:131-154,183-189
                rgb, bv, loc, cmd, speed = (t.repeat_interleave(self.batch_aug, dim=0) for t in (rgb, bv, loc, cmd, speed))
            if self.strategy is not None:
                self.aug.recipe = self.strategy(self.images_seen)
                self.aug.augment_batch(rgb)  # (rgb is a gathered copy of the resident frames)
            self.images_seen += rgb.shape[0]
            yield rgb, bv, loc, cmd, speed


def make_loaders(config, device, rank=0, world=1):
    """-> (train, val): iterables of batches; one pass = one epoch (reference: 1000 train + 10 validation batches)"""
    da = config["data_args"]
    bs, iters = da["batch_size"], int(config["iters_per_epoch"])
    val_iters = max(1, iters // 100)
    augment, batch_aug = da.get("augment"), int(da.get("batch_aug", 1) or 1)
    if da.get("dataset_dir"):
        if any(k in da for k in ("crop_x_jitter", "crop_y_jitter", "angle_jitter", "cmd_biased", "max_frames")):
            # train_birdview's loader (reference bird_view/utils/datasets/birdview_lmdb.py:247-285): rotation / window jitter with the
            # waypoints following, command-biased sampling, a frame cap -- the rotation runs on the GPU (lbc_birdview_warp_crop_u8)
            from ..bird_view.utils.datasets.birdview_lmdb import get_birdview_device
            return get_birdview_device(da["dataset_dir"], bs, device, crop_x_jitter=da.get("crop_x_jitter", 0) or 0,
                                       crop_y_jitter=da.get("crop_y_jitter", 0) or 0, angle_jitter=da.get("angle_jitter", 0) or 0,
                                       n_step=da.get("n_step", 5), gap=da.get("gap", 5), max_frames=da.get("max_frames"),
                                       cmd_biased=bool(da.get("cmd_biased")), samples=(iters, val_iters), seed=0, rank=rank)
        from ..bird_view.utils.datasets.image_lmdb import get_image_device
        return get_image_device(da["dataset_dir"], bs, device, augment=augment, n_step=da.get("n_step", 5), gap=da.get("gap", 5),
                                batch_aug=batch_aug, samples=(iters, val_iters), seed=0, rank=rank)
    train = SyntheticFrames(config["synthetic"], device, seed=0, rank=rank, world=world)
    val = SyntheticFrames(max(bs, config["synthetic"] // 8), device, seed=1, rank=rank, world=world)
    return (_SyntheticLoader(train, bs, iters, augment, batch_aug, seed=rank), _SyntheticLoader(val, bs, val_iters))
