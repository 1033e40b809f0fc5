"""Parameter containers of the ResNet trunk (reference bird_view/models/resnet.py:95-180).

The reference's ResNet is an executable torch graph; here the modules below are *only* the
owners of the parameters/buffers (so that state_dict keys, shapes, registration order and
initial distributions are exactly the reference's: resnet.py:102-119 for the trunk).  The
arithmetic of ResNet.forward (resnet.py:148-159) runs in the HIP executor (csrc/engine.cpp).
Only the BasicBlock variants exist (resnet18/34 -- the only ones any reference script uses:
training/train_image_phase1.py:27, train_birdview.py:28).
"""
import torch.nn as nn

_BLOCKS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}
_C_OUT = {"resnet18": -1, "resnet34": 512}   # second value of reference model_funcs (resnet.py:162-168)


class BasicBlockParams(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class ResNetParams(nn.Module):
    def __init__(self, layers, input_channel=7, num_classes=1000, bias_first=True):
        super().__init__()
        if bias_first:
            raise NotImplementedError("conv1 with bias is never used by the LbC policies (bias_first=False at "
                                      "reference image.py:24 / birdview.py:49)")
        self.conv1 = nn.Conv2d(input_channel, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, nb in enumerate(layers):
            planes = 64 << li
            blocks = []
            for bi in range(nb):
                blocks.append(BasicBlockParams(inplanes, planes, 2 if (li > 0 and bi == 0) else 1))
                inplanes = planes
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))          # present in the reference (resnet.py:111-112), never called
        self.fc = nn.Linear(512, num_classes)                # -> conv.fc.* keys exist in every checkpoint
        for m in self.modules():                             # resnet.py:114-119
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        raise RuntimeError("the trunk has no stand-alone torch forward; it runs inside the HIP executor of the policy model")


def get_resnet(model_name="resnet18", pretrained=False, **kwargs):
    """reference resnet.py:171-180 -> (trunk, c_out)"""
    if model_name not in _BLOCKS:
        raise ValueError("backbone %r unsupported: the MI355X path implements the BasicBlock ResNets (resnet18/resnet34)" % model_name)
    if pretrained and kwargs.get("input_channel", 3) == 3:
        raise RuntimeError("pretrained=True downloads ImageNet weights (reference resnet.py:175-178); load a state_dict instead")
    return ResNetParams(_BLOCKS[model_name], **kwargs), _C_OUT[model_name]
