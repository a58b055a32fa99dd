"""Checkpoint loading — mirror of opv2v/opencood/tools/train_utils.py:24-65 (`load_saved_model`): the newest
`net_epoch<N>.pth` of a run directory is a plain state_dict with the reference's key names, which the HIP modules accept
unchanged (tests/golden/gv0_state_dict_schema.npz pins the key / shape schema)."""
import glob
import os
import re

import torch


def find_last_checkpoint(save_dir):
    """highest N among <save_dir>/*epoch<N>.pth, 0 when there is none (:42-52)"""
    best = 0
    for path in glob.glob(os.path.join(save_dir, "*epoch*.pth")):
        found = re.findall(".*epoch(.*).pth.*", path)
        best = max(best, int(found[0]))
    return best


def load_saved_model(saved_path, model):
    """-> (epoch, model); non-strict load as in the reference (:54-63), on the CPU first; cached kernel plans of the
    model are rebuilt on the next forward because the parameters' versions change."""
    assert os.path.exists(saved_path), "{} not found".format(saved_path)
    initial_epoch = find_last_checkpoint(saved_path)
    if initial_epoch > 0:
        print("resuming by loading epoch %d" % initial_epoch)
        checkpoint = torch.load(os.path.join(saved_path, "net_epoch%d.pth" % initial_epoch), map_location="cpu")
        model.load_state_dict(checkpoint, strict=False)
        del checkpoint
    return initial_epoch, model
