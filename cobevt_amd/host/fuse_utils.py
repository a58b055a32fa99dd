"""regroup — mirror of opv2v/opencood/models/sub_modules/fuse_utils.py:8-61, without its device->host
synchronisation: record_len is read by the kernel on the device."""
import torch

from .. import ops
from . import runtime as rt


def regroup(dense_feature, record_len, max_len):
    """dense_feature (N, C, H, W)-shaped device tensor, record_len (B,) -> ((B, max_len, C, H, W), mask (B, max_len))"""
    x = rt.to_nhwc(dense_feature)                                          # (N, H, W, C)
    rl = torch.as_tensor(record_len).to(device=x.device, dtype=torch.int32)
    out, mask = ops.regroup(x, rl, max_len)                                # (B, L, H, W, C)
    return out.permute(0, 1, 4, 2, 3), mask
