"""BirdViewPolicyModelSS -- the privileged teacher (reference bird_view/models/birdview.py:47-79),
on the same gfx950 executor as the student (7-channel 192x192 input, 48x48 soft-argmax)."""
import torch.nn as nn

from . import common

STEPS = 5
SPEED_STEPS = 3
COMMANDS = 4
DT = 0.1
CROP_SIZE = 192
PIXELS_PER_METER = 5


class BirdViewPolicyModelSS(common.PolicyBase):
    def __init__(self, backbone="resnet18", input_channel=7, n_step=5, all_branch=False, **kwargs):
        super().__init__(backbone=backbone, input_channel=input_channel, bias_first=False)
        if input_channel != 7:
            raise NotImplementedError("the HIP stem is built for the 7-channel bird-view (and 3-channel RGB) inputs")
        self.deconv = common.spatial_softmax_decoder()
        # input_hw: not a reference argument (swallowed by **kwargs there, birdview.py:48) -- reduced map sizes for the CPU-emulated tests
        self.input_hw = tuple(kwargs.get("input_hw", (192, 192)))
        self.location_pred = nn.ModuleList([
            nn.Sequential(nn.BatchNorm2d(64), nn.Conv2d(64, STEPS, 1, 1, 0), common.SpatialSoftmax(self.input_hw[1] // 4, self.input_hw[0] // 4, STEPS))
            for _ in range(COMMANDS)])
        self.all_branch = all_branch
        self._finish_init()

    def forward(self, bird_view, velocity, command):
        if tuple(bird_view.shape[2:]) != self.input_hw:
            raise ValueError("BirdViewPolicyModelSS expects %dx%d maps (SpatialSoftmax(48,48) at reference birdview.py:58 for 192x192)" % self.input_hw)
        location_pred, location_preds = self._run(bird_view, velocity, command)
        if self.all_branch:
            return location_pred, location_preds
        return location_pred
