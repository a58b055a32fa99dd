"""nuScenes SinBEVT flavour of the FAX hot path — mirror of nuscenes/cross_view_transformer/model/
{encoder_pyramid_axial.py, cvt.py, decoder.py} (hydra `_target_` classes of config/model/cvt_pyramid_axial.yaml)."""
from .encoder_pyramid_axial import Normalize, PyramidAxialEncoder  # noqa: F401
from .decoder import Decoder, DecoderBlock  # noqa: F401
from .cvt import CrossViewTransformer  # noqa: F401
from .efficientnet import EfficientNetExtractor  # noqa: F401
from .metrics import BaseIoUMetric, IoUMetric  # noqa: F401
from .losses import BinarySegmentationLoss, CenterLoss, MultipleLoss, SigmoidFocalLoss  # noqa: F401
