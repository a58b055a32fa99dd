"""Batch collation of the camera intermediate-fusion dataset — mirror of `collate_batch`,
opv2v/opencood/data_utils/datasets/camera_only/intermediate_fusion_dataset.py:231-317: the list of per-scenario `ego`
dicts the dataset yields -> the `batch_dict['ego']` CorpBEVT.forward consumes (agents of all scenarios concatenated along
the batch axis, `record_len` agents per scenario, images channels-last (sum L, 1, M, H, W, C) float32)."""
import numpy as np
import torch


def collate_batch(batch, train=True):
    if not train:
        assert len(batch) == 1
    rgb, extrinsic, intrinsic, gt_static, gt_dynamic, t_matrix, pairwise, record_len = [], [], [], [], [], [], [], []
    for sample in batch:
        ego = sample["ego"]
        cams = ego["camera_data"]
        assert cams.shape[0] == ego["camera_intrinsic"].shape[0] == ego["camera_extrinsic"].shape[0]
        record_len.append(cams.shape[0])
        rgb.append(cams)
        intrinsic.append(ego["camera_intrinsic"])
        extrinsic.append(ego["camera_extrinsic"])
        gt_dynamic.append(ego["gt_dynamic"])
        gt_static.append(ego["gt_static"])
        t_matrix.append(ego["transformation_matrix"])
        pairwise.append(ego["pairwise_t_matrix"])

    def agents(parts):                      # (sum L, 1, ...) float32
        return torch.from_numpy(np.concatenate(parts, axis=0)).unsqueeze(1).float()

    return {"ego": {
        "inputs": agents(rgb),
        "extrinsic": agents(extrinsic),
        "intrinsic": agents(intrinsic),
        "gt_static": torch.from_numpy(np.stack(gt_static)).long(),
        "gt_dynamic": torch.from_numpy(np.stack(gt_dynamic)).long(),
        "transformation_matrix": torch.from_numpy(np.stack(t_matrix)).float(),
        "pairwise_t_matrix": torch.from_numpy(np.stack(pairwise)).float(),
        "record_len": torch.from_numpy(np.array(record_len, dtype=int)),
    }}
