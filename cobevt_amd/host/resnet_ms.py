"""ResnetEncoder — mirror of opv2v/opencood/models/backbones/resnet_ms.py.

torchvision is not a dependency: the ResNet-18/34 parameter container below is written from torchvision's
public definition with torchvision's attribute names (conv1, bn1, layer1..4.{j}.{conv1,bn1,conv2,bn2,
downsample.{0,1}}, fc) so reference checkpoints load key-for-key.  `pretrained` is
accepted and ignored (no network; the reference would download ImageNet weights, resnet_ms.py:38).
"""
import torch
import torch.nn as nn

from .. import lib as _L
from .. import ops
from ..lib import CobevtHipError
from . import runtime as rt
from . import training
from .runtime import HipModule

_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}


class BasicBlock(HipModule):
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

    def forward_nhwc(self, x):
        identity = x
        p1 = rt.conv_plan(self, "c1", self.conv1, self.bn1, act=1)
        p2 = rt.conv_plan(self, "c2", self.conv2, self.bn2, act=1)
        if self.downsample is not None:
            pd = rt.conv_plan(self, "ds", self.downsample[0], self.downsample[1])
            if ops.dsblock_fusable(x, p1, p2, pd):
                return ops.dsblock(x, p1, p2, pd)
            variant = ops.conv3_ds_fusable(x, p1, p2, pd)
            if variant:                                  # layer3.0 / layer4.0: the shortcut rides in conv2's launch
                return ops.conv3_ds(ops.conv2d(x, p1), x, p2, pd, variant)
            identity = ops.conv2d(x, pd)
        if self.downsample is None and self.stride == 1 and ops.basicblock_fusable(x, p1, p2):
            return ops.basicblock(x, p1, p2)
        y = ops.conv2d(x, p1)
        return ops.conv2d(y, p2, residual=identity)


class ResNet(HipModule):
    """Parameter container with torchvision.models.ResNet's layout (BasicBlock variants)."""

    def __init__(self, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)   # unused by the encoder, kept for checkpoint compatibility

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def stem_nhwc(self, images, lut=None, bgr=False):
        """images: (N, H, W, 3) channels-last -> (N, H/4, W/4, 64).  fp32 images are the reference's normalised input
        (resnet_ms.py:62-69); uint8 images are camera frames, normalised through `lut` (3, 256) inside the stem's gather
        (ResnetEncoder.set_rgb_normalisation).  bgr: the frames are BGR as cv2 decodes them - the channel swap of
        rgb_preprocessor.py:33-42 is folded into the stem weights (input channels reversed) and the table rows."""
        if images.dtype != torch.uint8:
            return ops.stem_pool(images, rt.conv_plan(self, "stem", self.conv1, self.bn1, act=1, smallc=True))
        if lut is None:
            raise CobevtHipError("uint8 camera frames need the normalisation table: call ResnetEncoder.set_rgb_normalisation(mean, "
                                 "std[, bgr2rgb]) (host/rgb_preprocessor.py) before feeding them")
        if not bgr:
            plan = rt.conv_plan(self, "stem", self.conv1, self.bn1, act=1, smallc=True)
        else:
            plan = self._plan("stem.bgr", rt.module_tensors(self.conv1, self.bn1),
                              lambda dt, dev: ops.ConvPlan(self.conv1.weight.detach().flip(1), None, bn=self.bn1, stride=2, pad=3, act=1,
                                                           dtype=dt, device=dev, smallc=True))
        return ops.stem_pool_u8(images, lut, plan)


class ResnetEncoder(HipModule):
    """resnet_ms.py:9-89."""

    def __init__(self, params):
        super().__init__()
        self.num_layers = params["num_layers"]
        self.pretrained = params["pretrained"]
        image_height = params["image_height"]
        image_width = params["image_width"]
        self.idx_pick = params["id_pick"]
        if self.num_layers not in (18, 34, 50, 101, 152):
            raise ValueError("{} is not a valid number of resnet layers".format(self.num_layers))
        if self.num_layers not in _BLOCKS:
            raise CobevtHipError("resnet%d (Bottleneck variants) is not lowered to HIP; the FAX configs use 18/34"
                                 % self.num_layers)
        self.encoder = ResNet(_BLOCKS[self.num_layers])
        self.register_buffer("ingest_lut", None, persistent=False)      # set_rgb_normalisation(): uint8 camera frames as `inputs`
        # shapes the reference obtains from a dummy forward (resnet_ms.py:41-44), computed analytically here
        def half(v, k, p):
            return (v + 2 * p - k) // 2 + 1
        h, w = half(half(image_height, 7, 3), 3, 1), half(half(image_width, 7, 3), 3, 1)
        shapes = []
        for i, c in enumerate((64, 128, 256, 512)):
            if i > 0:
                h, w = half(h, 3, 1), half(w, 3, 1)
            shapes.append(torch.Size((1, 1, 1, c, h, w)))
        self.output_shapes = [shapes[i] for i in self.idx_pick] if isinstance(self.idx_pick, list) else [shapes[self.idx_pick]]

    ingest_bgr = False

    def set_rgb_normalisation(self, mean, std, bgr2rgb=False):
        """Accept uint8 camera frames (B, L, M, H, W, 3) as `inputs`: the /255, (x - mean) / std of RgbPreProcessor
        (rgb_preprocessor.py:14-31) becomes a table lookup inside the stem kernel, so a frame crosses PCIe as 1 byte per value
        instead of the 4 of the normalised fp32 image (inference_camera.py:56-61 uploads 63 MB per 5-agent frame; this is 15.7).
        bgr2rgb: the frames are BGR (cv2.imread) and the config asks for the swap - folded into the stem weights.  The table is a
        non-persistent buffer: state_dict keys are unchanged.  fp32 `inputs` keep working as before."""
        from .rgb_preprocessor import normalisation_table
        t = torch.from_numpy(normalisation_table(mean, std))
        if bgr2rgb:
            t = t.flip(0)                                  # row j = channel of byte position j of a BGR pixel
        t = t.contiguous().to(next(self.parameters()).device)
        if self.ingest_lut is not None and self.ingest_lut.device == t.device:
            self.ingest_lut.copy_(t)          # in place: captured graphs hold the table's address, and the version bump re-captures them
        else:
            self.ingest_lut = t
            rt.bump_structure_epoch()         # a new tensor for the captured-plan fingerprints to watch
        self.ingest_bgr = bool(bgr2rgb)
        return self

    def _stem(self, x):
        if x.dtype == torch.uint8:
            return self.encoder.stem_nhwc(x if x.is_contiguous() else x.contiguous(), self.ingest_lut, self.ingest_bgr)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float32).contiguous()
        return self.encoder.stem_nhwc(x)

    def stages_nhwc(self, input_images):
        """Generator over (stage index 0..3, channels-last feature map) so a caller can overlap work that depends on
        an early stage with the later stages (CorpBEVT.encode_agents)."""
        b, l, m, h, w, c = input_images.shape
        # encoder_scope: under set_compute_dtype("fp32_fast") these launches go to the one-fp16-MFMA library (runtime.py); the
        # scope is closed around every yield - the caller's work between two stages must not inherit it
        with _L.encoder_scope():
            x = self._stem(input_images.reshape(b * l * m, h, w, c))
        for i, layer in enumerate((self.encoder.layer1, self.encoder.layer2, self.encoder.layer3, self.encoder.layer4)):
            with _L.encoder_scope():
                for blk in layer:
                    x = blk.forward_nhwc(x)
            yield i, x

    def forward(self, input_images):
        """(B, L, M, H, W, 3) channels-last fp32 -> list of (B, L, M, C, h, w) (channels-last views)."""
        if self.training:
            return training.resnet_encoder(self, input_images)
        self._require_inference(input_images)
        b, l, m, h, w, c = input_images.shape
        results = []
        with _L.encoder_scope():
            x = self._stem(input_images.reshape(b * l * m, h, w, c))
            for layer in (self.encoder.layer1, self.encoder.layer2, self.encoder.layer3, self.encoder.layer4):
                for blk in layer:
                    x = blk.forward_nhwc(x)
                v = rt.nchw_view(x)
                results.append(v.reshape(b, l, m, *v.shape[1:]))
        if isinstance(self.idx_pick, list):
            return [results[i] for i in self.idx_pick]
        return results[self.idx_pick]
