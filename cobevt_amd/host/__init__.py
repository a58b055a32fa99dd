"""Host-side mirror of the reference's nn.Module interface for the FAX hot path."""
from .runtime import compute_dtype, get_compute_dtype, get_compute_mode, get_matrix_path, set_compute_dtype  # noqa: F401
from .fax_modules import (Attention as FaxAttention, BEVEmbedding, Bottleneck, CrossViewSwapAttention,  # noqa: F401
                          CrossWinAttention, FAXModule, generate_grid, get_view_matrix)
from .swap_fusion_modules import (Attention as SwapAttention, SwapFusionBlock, SwapFusionBlockMask,  # noqa: F401
                                  SwapFusionEncoder)
from .base_transformer import BaseEncoder, BaseTransformer, CavAttention, FeedForward, PreNorm, PreNormResidual  # noqa: F401
from .resnet_ms import ResnetEncoder  # noqa: F401
from .naive_decoder import NaiveDecoder  # noqa: F401
from .naive_compress import NaiveCompressor  # noqa: F401
from .bev_seg_head import BevSegHead  # noqa: F401
from .fuse_utils import regroup  # noqa: F401
from .corpbevt import STTF, CorpBEVT  # noqa: F401
from .fax_fused_transformer import FaxFusedTransformer  # noqa: F401
from .cvt_modules import CrossAttention, CrossViewAttention, CrossViewModule  # noqa: F401
from .cross_view_transformer import CrossViewTransformer  # noqa: F401
from .cross_view_transformer_swap_fuse import CrossViewTransformerSwapFuse  # noqa: F401
from .cross_view_transformer_fcooper import CrossViewTransformerFcooper  # noqa: F401
from .cross_view_transformer_att_fuse import CrossViewTransformerAttFuse  # noqa: F401
from .v2v_fuse import ConvGRU, DiscoNetFusion, PixelWeightedFusionSoftmax, V2VNetFusion  # noqa: F401
from .cross_view_transformer_v2vnet import CrossViewTransformerV2VNet  # noqa: F401
from .cross_view_transformer_disconet import CrossViewTransformerDiscoNet  # noqa: F401
from .pipeline import CapturedCall, CapturedCorpBEVT, HostFrameFeeder, PipelinedCorpBEVT  # noqa: F401
# the data formats either side of the path
from .camera_bev_postprocessor import CameraBevPostprocessor  # noqa: F401
from .rgb_preprocessor import RgbPreProcessor  # noqa: F401
from .intermediate_fusion_dataset import collate_batch  # noqa: F401
from .train_utils import load_saved_model  # noqa: F401
from .seg_utils import cal_iou_training, mean_IU, mean_precision  # noqa: F401
from .vanilla_seg_loss import VanillaSegLoss  # noqa: F401
from .train_graph import CapturedTrainStep  # noqa: F401
