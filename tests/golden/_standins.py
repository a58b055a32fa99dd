"""Stand-ins that let the reference's Python hot path be imported in the build container, where torchvision
and shapely are not installed.  Used ONLY by make_golden.py (never by tests or the product).

`torchvision.models.resnet`: BasicBlock / Bottleneck / ResNet written from torchvision's public definition with
torchvision's attribute names (conv1, bn1, layer1..4, downsample, fc), so state_dict keys match Appendix D of
SURVEY.md.  Because this is a stand-in, arithmetic parity with the real torchvision is NOT pinned by the
fixtures ("parity unpinned" for third-party code) — only the wiring around it is.
"""
import sys
import types

import torch
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        width = planes
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


def resnet18(pretrained=False, **kw):
    return ResNet(BasicBlock, [2, 2, 2, 2])


def resnet34(pretrained=False, **kw):
    return ResNet(BasicBlock, [3, 4, 6, 3])


def _unavailable(*a, **k):
    raise RuntimeError("stand-in: only resnet18/34 are provided")


def install():
    """Register the stand-in modules in sys.modules (idempotent)."""
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_cobevt_standin", False):
        return  # a real torchvision is present; use it
    tv = types.ModuleType("torchvision")
    tv._cobevt_standin = True
    models = types.ModuleType("torchvision.models")
    resnet = types.ModuleType("torchvision.models.resnet")
    for name, obj in (("BasicBlock", BasicBlock), ("Bottleneck", Bottleneck), ("ResNet", ResNet),
                      ("resnet18", resnet18), ("resnet34", resnet34)):
        setattr(resnet, name, obj)
    for name, obj in (("resnet18", resnet18), ("resnet34", resnet34), ("resnet50", _unavailable),
                      ("resnet101", _unavailable), ("resnet152", _unavailable), ("resnet", resnet)):
        setattr(models, name, obj)
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.models.resnet"] = resnet
    if "shapely" not in sys.modules:
        sh = types.ModuleType("shapely")
        geo = types.ModuleType("shapely.geometry")
        geo.Polygon = type("Polygon", (), {})
        sh.geometry = geo
        sys.modules["shapely"] = sh
        sys.modules["shapely.geometry"] = geo


def install_torchmetrics():
    """`torchmetrics.Metric` as far as nuscenes/cross_view_transformer/metrics.py uses it: constructor keywords ignored,
    add_state(name, default, ...) sets an attribute.  (torchmetrics is not installed; the metric's arithmetic is the
    reference's own update / compute code, which this base class does not touch.)"""
    if "torchmetrics" in sys.modules:
        return
    tm = types.ModuleType("torchmetrics")

    class Metric(object):
        def __init__(self, **kwargs):
            pass

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone())
    tm.Metric = Metric
    sys.modules["torchmetrics"] = tm


def install_fvcore():
    """`fvcore.nn.sigmoid_focal_loss` (the only fvcore symbol nuscenes/cross_view_transformer/losses.py uses), written from its
    published definition: third-party arithmetic, parity unpinned.  The label grouping, visibility masking and reductions
    around it are the reference's own code."""
    if "fvcore" in sys.modules:
        return
    import torch.nn.functional as F

    def sigmoid_focal_loss(inputs, targets, alpha=-1, gamma=2, reduction="none"):
        p = torch.sigmoid(inputs)
        ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
        p_t = p * targets + (1 - p) * (1 - targets)
        loss = ce_loss * ((1 - p_t) ** gamma)
        if alpha >= 0:
            loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
        if reduction == "mean":
            loss = loss.mean()
        elif reduction == "sum":
            loss = loss.sum()
        return loss
    fv, fvnn = types.ModuleType("fvcore"), types.ModuleType("fvcore.nn")
    fvnn.sigmoid_focal_loss = sigmoid_focal_loss
    fv.nn = fvnn
    sys.modules["fvcore"] = fv
    sys.modules["fvcore.nn"] = fvnn
