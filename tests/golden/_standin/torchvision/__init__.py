"""DEV-CONTAINER-ONLY stand-in for the `torchvision` namespace.

torchvision==0.7.0 (reference requirements.txt:4) is not installed in this image and not
vendored under /root/reference, yet the reference's models.py:4 / datasets.py:3 import it.
This namespace exists only so that tests/golden/generate_fixtures.py can import the
reference's own `models.py` / `datasets.py` / `train.py` UNMODIFIED and record golden
vectors from them.  It provides exactly the four names the reference touches:

  torchvision.models.resnet18      (models.py:49)  -> standard BasicBlock ResNet-18 wiring
                                                      over torch.nn layers (He et al. 2015)
  torchvision.ops.RoIPool          (models.py:58)  -> oracle/roipool_ref.c
  torchvision.datasets.VisionDataset (datasets.py:9)
  torchvision.transforms.{Compose,ToTensor} (datasets.py:41-45)

It never travels to the product path and is never imported by tests at run time; the
fixtures it helped to produce are data (tests/golden/*.npz).  Because the ResNet wiring
and RoIPool here are build-owned, parity for those two pieces is UNPINNED (see
oracle/cova_oracle.py header); everything in the reference's own files is pinned.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
from oracle import cova_oracle as _oracle  # noqa: E402


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64), _BasicBlock(64, 64))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256))
        self.layer4 = nn.Sequential(_BasicBlock(256, 512, 2), _BasicBlock(512, 512))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def _resnet18(pretrained=False, **kw):
    # ImageNet weights cannot be downloaded here; callers load an explicit state_dict.
    return _ResNet18()


class _RoIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale

    def forward(self, x, rois):
        return _oracle.roi_pool(x, rois, self.output_size, self.spatial_scale)


class _VisionDataset(torch.utils.data.Dataset):
    def __init__(self, root, *a, **kw):
        self.root = root


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _ToTensor:
    def __call__(self, pic):
        a = np.asarray(pic, dtype=np.uint8)
        return torch.from_numpy(a).permute(2, 0, 1).float().div(255)


models = types.ModuleType("torchvision.models")
models.resnet18 = _resnet18
ops = types.ModuleType("torchvision.ops")
ops.RoIPool = _RoIPool
datasets = types.ModuleType("torchvision.datasets")
datasets.VisionDataset = _VisionDataset
transforms = types.ModuleType("torchvision.transforms")
transforms.Compose = _Compose
transforms.ToTensor = _ToTensor
for _m in (models, ops, datasets, transforms):
    sys.modules[_m.__name__] = _m
