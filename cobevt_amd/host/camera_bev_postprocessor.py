"""CameraBevPostprocessor — mirror of opv2v/opencood/data_utils/post_processor/camera_bev_postprocessor.py:13-96 (what
the reference's datasets call on the ground-truth BEV images and on the model's logits before scoring).

Label side (`generate_label`, `merge_label`) is host numpy in the reference's data loader and stays host numpy here.
Logit side (`softmax_argmax`, `post_process_train`, `post_process`) runs on the tensors the model just produced, i.e. on
the GPU: one HIP launch per head (cobevt_softmax_argmax) instead of nn.Softmax + torch.argmax."""
import numpy as np

from .. import ops


def bgr_to_gray_u8(bgr):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for uint8 (camera_bev_postprocessor.py:33) from OpenCV's published fixed-point
    formula; opencv-python is not in this image, so this is pinned only against the oracle (parity unpinned for cv2)."""
    img = np.asarray(bgr)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) uint8 BGR image, got %s %s" % (img.shape, img.dtype))
    acc = img[..., 0].astype(np.int32) * 1868 + img[..., 1].astype(np.int32) * 9617 + img[..., 2].astype(np.int32) * 4899
    return ((acc + 8192) >> 14).astype(np.uint8)


class CameraBevPostprocessor(object):
    def __init__(self, anchor_params, train):
        self.params = anchor_params
        self.train = train

    # ---- ground-truth side (host) ----
    def generate_label(self, bev_map):
        """(H, W, 3) uint8 BGR rendering -> (H, W) float64 binary map (:24-37)"""
        gray = bgr_to_gray_u8(bev_map)
        return (gray > 0).astype(np.float64)

    def merge_label(self, road_map, lane_map):
        """road -> 1, lane -> 2 (lane wins), everything else 0 (:39-53)"""
        merged = np.zeros((road_map.shape[0], road_map.shape[1]))
        merged[road_map == 1] = 1
        merged[lane_map == 1] = 2
        return merged

    # ---- logit side (GPU) ----
    def softmax_argmax(self, seg_logits):
        """(B, C, H, W) -> (softmax over C fp32, argmax of the probabilities int64 (B, H, W))  (:55-59)"""
        return ops.softmax_argmax(seg_logits)

    def post_process_train(self, output_dict):
        """adds static_prob / static_map / dynamic_prob / dynamic_map for agent slot 0 of each sample (:61-89)"""
        static_prob, static_map = self.softmax_argmax(output_dict["static_seg"][:, 0])
        dynamic_prob, dynamic_map = self.softmax_argmax(output_dict["dynamic_seg"][:, 0])
        output_dict.update({"static_prob": static_prob, "static_map": static_map,
                            "dynamic_map": dynamic_map, "dynamic_prob": dynamic_prob})
        return output_dict

    def post_process(self, batch_dict, output_dict):
        return self.post_process_train(output_dict)
