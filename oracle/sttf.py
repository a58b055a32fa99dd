"""Oracle for regroup / STTF / ROI mask.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows opv2v/opencood/models/sub_modules/fuse_utils.py:8-61, corpbevt.py:22-64 and
opv2v/opencood/models/sub_modules/torch_transformation_utils.py:11-355.
"""
import torch
import torch.nn.functional as F


def regroup(dense_feature, record_len, max_len):
    """fuse_utils.py:8-61 — (N, C, H, W), record_len (B,) -> (B, max_len, C, H, W), mask (B, max_len) int64."""
    lens = [int(v) for v in record_len]
    feats, mask, off = [], [], 0
    for n in lens:
        f = dense_feature[off:off + n]
        off += n
        pad = torch.zeros(max_len - n, *f.shape[1:], dtype=f.dtype)
        feats.append(torch.cat([f, pad], 0)[None])
        mask.append([1] * n + [0] * (max_len - n))
    return torch.cat(feats, 0), torch.tensor(mask, dtype=torch.int64)


def discretized_matrix(matrix, discrete_ratio, downsample_rate):
    """torch_transformation_utils.py:108-134 — (B, L, 4, 4) -> (B, L, 2, 3), translation in feature cells."""
    m = matrix[:, :, [0, 1], :][:, :, :, [0, 1, 3]].clone()
    m[:, :, :, -1] = m[:, :, :, -1] / (discrete_ratio * downsample_rate)
    return m.float()


def transformation_matrix(M, dsize):
    """:254-297 — rotation about (W/2, H/2) plus translation. M (N, 2, 3)."""
    H, W = dsize
    N = M.shape[0]
    eye = torch.eye(3, dtype=M.dtype)[None].repeat(N, 1, 1)
    shift, shift_inv, rot = eye.clone(), eye.clone(), eye.clone()
    center = torch.tensor([W / 2, H / 2], dtype=M.dtype)
    shift[:, :2, 2] = center
    shift_inv[:, :2, 2] = -center
    rot[:, :2, :2] = M[:, :2, :2]
    T = (shift @ rot @ shift_inv)[:, :2, :].clone()
    T[..., 2] += M[..., 2]
    return T


def _normal_transform_pixel(height, width, dtype, eps=1e-14):
    """:160-191 helper"""
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
    wd = eps if width == 1 else width - 1.0
    hd = eps if height == 1 else height - 1.0
    t[0, 0] = t[0, 0] * 2.0 / wd
    t[1, 1] = t[1, 1] * 2.0 / hd
    return t[None]


def warp_affine(src, M, dsize, mode="bilinear"):
    """:317-355 — src (N, C, H, W), M (N, 2, 3) destination-from-source in pixels."""
    N, C, H, W = src.shape
    M3 = F.pad(M, [0, 0, 0, 1], "constant", value=0.0)
    M3[..., -1, -1] += 1.0
    src_norm = _normal_transform_pixel(H, W, M.dtype)
    dst_norm = _normal_transform_pixel(dsize[0], dsize[1], M.dtype)
    dst_norm_trans_src_norm = dst_norm @ (M3 @ torch.inverse(src_norm))
    theta = torch.inverse(dst_norm_trans_src_norm)[:, :2, :]
    grid = F.affine_grid(theta, [N, C, dsize[0], dsize[1]], align_corners=True)
    return F.grid_sample(src, grid, align_corners=True, mode=mode, padding_mode="zeros")


def sttf(x, spatial_correction_matrix, discrete_ratio, downsample_rate):
    """STTF.forward, corpbevt.py:28-64 — x (B L C H W) -> (B L H W C) warped into the ego frame."""
    dist = discretized_matrix(spatial_correction_matrix, discrete_ratio, downsample_rate)
    x = x.permute(0, 1, 2, 4, 3).flip(4)
    B, L, C, H, W = x.shape
    T = transformation_matrix(dist.reshape(-1, 2, 3), (H, W))
    y = warp_affine(x.reshape(-1, C, H, W), T, (H, W)).reshape(B, L, C, H, W)
    return y.flip(4).permute(0, 1, 4, 3, 2)          # 'b l c w h -> b l h w c'


def roi_and_cav_mask(shape, cav_mask, spatial_correction_matrix, discrete_ratio, downsample_rate):
    """get_roi_and_cav_mask, torch_transformation_utils.py:11-49 — shape (B, L, H, W, C) -> (B, H, W, 1, L)."""
    B, L, H, W, _ = shape
    dist = discretized_matrix(spatial_correction_matrix, discrete_ratio, downsample_rate)
    T = transformation_matrix(dist.reshape(-1, 2, 3), (H, W))
    ones = torch.ones((B * L, 1, H, W), dtype=T.dtype)
    roi = warp_affine(ones, T, (H, W), mode="nearest").reshape(B, L, 1, H, W)
    com = roi * cav_mask[:, :, None, None, None].to(roi.dtype)
    return com.permute(0, 3, 4, 2, 1)
