"""RgbPreProcessor — mirror of opv2v/opencood/data_utils/pre_processor/rgb_preprocessor.py:12-55: what turns a camera
frame into the float image `batch_dict['inputs']` carries (channel swap -> resize -> /255 -> (x - mean) / std, float64
like the reference; the collate function casts to float32).  Data-loader code, host numpy as in the reference."""
import numpy as np



class RgbPreProcessor(object):
    def __init__(self, preprocess_params, train):
        self.params = preprocess_params
        self.train = train

    def preprocess(self, rgb_image):
        return self.standalize(self.normalize(self.resize_image(self.channel_swap(rgb_image))))

    def standalize(self, rgb_image):
        args = self.params["args"]
        return (rgb_image - np.array(args["mean"])) / np.array(args["std"])

    def normalize(self, rgb_image):
        return np.array(rgb_image, dtype=float) / 255.

    def normalisation_table(self):
        """(3, 256) float32: what `standalize(normalize(.))` followed by the collate function's cast to float32
        (intermediate_fusion_dataset.py:231-317) gives byte value u of channel c - the table the on-GPU ingest reads
        (ResnetEncoder.set_rgb_normalisation, csrc/stem7x7.hip): uint8 frames go over PCIe, the arithmetic of :14-31 is a lookup."""
        return normalisation_table(self.params["args"]["mean"], self.params["args"]["std"])

    def channel_swap(self, rgb_image):
        """BGR -> RGB when the config asks for it (cv2.COLOR_BGR2RGB is a pure channel reversal)"""
        return np.ascontiguousarray(rgb_image[..., ::-1]) if self.params["args"]["bgr2rgb"] else rgb_image

    def resize_image(self, rgb_image):
        """cv2.resize(image, (resize_x, resize_y)) with its default INTER_LINEAR (:46-55).  Where OpenCV is installed this IS
        that call (data-loader code, host side, exactly what the reference does).  OpenCV is third-party and not in the build
        image, so without it the resize is restated from OpenCV's published algorithm (`resize_linear`, below) - PARITY
        UNPINNED: there is no cv2 here to replay it against; frames already at the target resolution pass
        through."""
        args = self.params["args"]
        if rgb_image.shape[1] == args["resize_x"] and rgb_image.shape[0] == args["resize_y"]:
            return rgb_image
        try:
            import cv2
        except ImportError:
            return resize_linear(rgb_image, args["resize_x"], args["resize_y"])
        return cv2.resize(rgb_image, (args["resize_x"], args["resize_y"]))


def normalisation_table(mean, std):
    """float32[(3, 256)]: fl32((float64(u) / 255. - mean[c]) / std[c]), the exact operation order of rgb_preprocessor.py:14-31"""
    u = np.arange(256, dtype=float)[None, :] / 255.
    return ((u - np.array(mean, dtype=float)[:, None]) / np.array(std, dtype=float)[:, None]).astype(np.float32)


def _linear_taps(n_src, n_dst):
    """source index pairs and weights of OpenCV's INTER_LINEAR along one axis: pixel centres aligned
    (fx = (d + 0.5) * scale - 0.5), left tap clamped at the borders"""
    scale = n_src / float(n_dst)
    f = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(f).astype(np.int64)
    w = (f - i0).astype(np.float32)
    lo = i0 < 0
    i0[lo], w[lo] = 0, 0.0
    hi = i0 >= n_src - 1
    i0[hi], w[hi] = n_src - 1, 0.0
    i1 = np.minimum(i0 + 1, n_src - 1)
    return i0, i1, w


def resize_linear(image, width, height):
    """(H, W[, C]) -> (height, width[, C]).  uint8 images follow OpenCV's fixed-point path: 11-bit tap weights
    (INTER_RESIZE_COEF_SCALE = 2048, round-to-nearest-even as cvRound), an integer horizontal pass, and the vertical pass
    ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2; an exact 2x down-scale takes OpenCV's area shortcut
    (2 x 2 mean, (sum + 2) >> 2).  Other dtypes are interpolated in float32."""
    img = np.asarray(image)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w = img.shape[:2]
    if img.dtype == np.uint8 and h == 2 * height and w == 2 * width:
        s = img.astype(np.int32)
        out = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        return out[:, :, 0] if squeeze else out
    x0, x1, wx = _linear_taps(w, width)
    y0, y1, wy = _linear_taps(h, height)
    if img.dtype == np.uint8:
        ax1 = np.rint(wx * 2048.0).astype(np.int64)
        ax0 = 2048 - ax1
        ay1 = np.rint(wy * 2048.0).astype(np.int64)
        ay0 = 2048 - ay1
        src = img.astype(np.int64)
        rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]          # (H, width, C), scaled by 2^11
        s0, s1 = rows[y0], rows[y1]
        out = (((ay0[:, None, None] * (s0 >> 4)) >> 16) + ((ay1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
        out = np.clip(out, 0, 255).astype(np.uint8)
    else:
        src = img.astype(np.float32)
        rows = src[:, x0] * (1.0 - wx)[None, :, None] + src[:, x1] * wx[None, :, None]
        out = (rows[y0] * (1.0 - wy)[:, None, None] + rows[y1] * wy[:, None, None]).astype(img.dtype if img.dtype.kind == "f" else np.float32)
    return out[:, :, 0] if squeeze else out
