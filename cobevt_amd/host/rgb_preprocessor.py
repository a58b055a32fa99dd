"""RgbPreProcessor — mirror of opv2v/opencood/data_utils/pre_processor/rgb_preprocessor.py:12-55: what turns a camera
frame into the float image `batch_dict['inputs']` carries (channel swap -> resize -> /255 -> (x - mean) / std, float64
like the reference; the collate function casts to float32).  Data-loader code, host numpy as in the reference."""
import numpy as np

from ..lib import CobevtHipError


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

    def channel_swap(self, rgb_image):
        """BGR -> RGB when the config asks for it (cv2.COLOR_BGR2RGB is a pure channel reversal)"""
        return np.ascontiguousarray(rgb_image[..., ::-1]) if self.params["args"]["bgr2rgb"] else rgb_image

    def resize_image(self, rgb_image):
        """cv2.resize to (resize_x, resize_y) (:46-55).  OpenCV is third-party and not in this image: frames already at
        the target resolution pass through, anything else needs cv2 and says so."""
        args = self.params["args"]
        if rgb_image.shape[1] == args["resize_x"] and rgb_image.shape[0] == args["resize_y"]:
            return rgb_image
        try:
            import cv2
        except ImportError:
            raise CobevtHipError("RgbPreProcessor.resize_image: %dx%d -> %dx%d needs opencv-python (cv2.resize), which is "
                                 "not installed" % (rgb_image.shape[1], rgb_image.shape[0], args["resize_x"], args["resize_y"]))
        return cv2.resize(rgb_image, (args["resize_x"], args["resize_y"]))
