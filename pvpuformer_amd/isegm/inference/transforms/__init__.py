"""Predictor-side transforms of the NoBRS loop (module layout of isegm/inference/transforms/): every transform offers
``transform / inv_transform / reset / get_state / set_state`` and the ``image_changed`` flag."""
from . import _device
from .base import BaseTransform, SigmoidForPred
from .flip import AddHorizontalFlip
from .limit_longest_side import LimitLongestSide
from .roi import Roi
from .zoom_in import ZoomIn, crop_resize, get_roi_image_nd


def resize_align_corners(x, size):
    return _device.resize_align_corners(x, size)
