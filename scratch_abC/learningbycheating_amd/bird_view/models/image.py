"""ImagePolicyModelSS -- the sensorimotor student (reference bird_view/models/image.py:22-89).

Same constructor, forward(image, velocity, command) signature, attribute names and state_dict
layout as the reference (so training/benchmark code and .th checkpoints interchange), but the
forward/backward arithmetic is the gfx950 executor (csrc/engine.cpp): ImageNet normalisation ->
ResNet trunk -> velocity late fusion -> BN/ConvTranspose/ReLU decoder -> 4 x (BN, 1x1 conv,
spatial softmax) -> branch select.
"""
import torch.nn as nn

from . import common

CROP_SIZE = 192
STEPS = 5
COMMANDS = 4
DT = 0.1
PIXELS_PER_METER = 5


class ImagePolicyModelSS(common.PolicyBase):
    _normalize = True    # common.NormalizeV2(mean=[0.485,0.456,0.406], std=[0.229,0.224,0.225]) at image.py:32-35

    def __init__(self, backbone, warp=False, pretrained=False, all_branch=False, **kwargs):
        super().__init__(backbone, pretrained=pretrained, input_channel=3, bias_first=False)
        if warp:
            raise NotImplementedError("warp=True is dead code in the reference (image.py:65-68 uses undefined names)")
        self.c = {"resnet18": 512, "resnet34": 512}[backbone]
        self.warp = warp
        self.deconv = common.spatial_softmax_decoder()
        # input_hw: not a reference argument (the reference swallows unknown keywords, image.py:23) -- reduced frame sizes for the
        # CPU-emulated tests; the SpatialSoftmax grid is a quarter of the frame (96 x 40 for the reference's 160 x 384)
        self.input_hw = tuple(kwargs.get("input_hw", (160, 384)))
        ow, oh = self.input_hw[1] // 4, self.input_hw[0] // 4
        self.location_pred = nn.ModuleList([
            nn.Sequential(nn.BatchNorm2d(64), nn.Conv2d(64, STEPS, 1, 1, 0), common.SpatialSoftmax(ow, oh, STEPS))
            for _ in range(COMMANDS)])
        self.all_branch = all_branch
        self._finish_init()

    def forward(self, image, velocity, command):
        if tuple(image.shape[2:]) != self.input_hw:
            raise ValueError("ImagePolicyModelSS expects %dx%d frames (SpatialSoftmax(96,40) at reference image.py:52,58 for 160x384); "
                             "got %s" % (self.input_hw + (tuple(image.shape[2:]),)))
        location_pred, location_preds = self._run(image, velocity, command)
        if self.all_branch:
            return location_pred, location_preds
        return location_pred
