"""ZoomIn (isegm/inference/transforms/zoom_in.py:9-200): after the first clicks the network sees the crop around the
current prediction (grown by ``expansion_ratio``, at least ``min_crop_size``), resized to ``target_size``; the prediction
is resized back and pasted into a full-size map; the crop is re-taken when a positive click falls outside it or the box
of the new prediction overlaps it by less than ``recompute_thresh_iou``.

GPU-resident: the previous prediction stays a device tensor; its thresholded box (joined with the positive clicks) is one
reduction kernel, so each step moves five integers to the host instead of the 448 x 448 map, and both bilinear resizes
(``align_corners=True``) run in the HIP up-sampling kernel.  The integer / float bookkeeping lives in ``Roi``."""
import numpy as np
import torch

from . import _device
from .base import BaseTransform
from .roi import Roi


def crop_resize(image_nd, roi, target_size):
    """The crop ``roi`` of a [N,C,H,W] tensor, resized to ``target_size`` (a (height, width) pair, or the length of the
    longer side)."""
    r0, r1, c0, c1 = roi
    with torch.no_grad():
        return _device.resize_align_corners(image_nd[:, :, r0:r1 + 1, c0:c1 + 1], Roi(*roi).output_size(target_size))


get_roi_image_nd = crop_resize      # the reference's name for it (zoom_in.py:168)


class ZoomIn(BaseTransform):
    def __init__(self, target_size=400, skip_clicks=1, expansion_ratio=1.4, min_crop_size=200, recompute_thresh_iou=0.5,
                 prob_thresh=0.50):
        self.target_size, self.skip_clicks = target_size, skip_clicks
        self.expansion_ratio, self.min_crop_size = expansion_ratio, min_crop_size
        self.recompute_thresh_iou, self.prob_thresh = recompute_thresh_iou, prob_thresh
        self.reset()

    # the previous prediction: kept on the device; a numpy array handed in (restored state, tests) is uploaded lazily
    @property
    def _prev_probs(self):
        return self._probs

    @_prev_probs.setter
    def _prev_probs(self, value):
        self._probs = value

    def _probs_on(self, device):
        if self._probs is not None and not torch.is_tensor(self._probs):
            self._probs = torch.from_numpy(np.ascontiguousarray(self._probs, dtype=np.float32)).to(device)
        return self._probs

    def _predicted_box(self, device, clicks=()):
        """Grown, clipped box of the thresholded previous prediction (joined with the positive clicks); None while there
        is no prediction or it is empty."""
        probs = self._probs_on(device)
        if probs is None:
            return None
        positives = [(c.coords[0], c.coords[1]) for c in clicks if c.is_positive]
        count, r0, r1, c0, c1 = _device.mask_box(probs, self.prob_thresh, positives)
        if count == 0:
            return None
        return Roi(r0, r1, c0, c1).grown(self.expansion_ratio, self.min_crop_size).clipped(*probs.shape[2:])

    def transform(self, image_nd, clicks_lists):
        assert image_nd.shape[0] == 1 and len(clicks_lists) == 1
        clicks = clicks_lists[0]
        self.image_changed = False
        if len(clicks) <= self.skip_clicks:
            return image_nd, clicks_lists
        self._input_image_shape = image_nd.shape
        box = self._predicted_box(image_nd.device, clicks)
        if box is None:
            if self.skip_clicks >= 0:
                return image_nd, clicks_lists
            box = Roi.whole(*image_nd.shape[2:])         # skip_clicks < 0: zoom from the very first click on
        held = self._object_roi
        if held is None or not Roi(*held).encloses(clicks) or box.overlap(held) < self.recompute_thresh_iou:
            self._object_roi, self.image_changed = box, True
        self._roi_image = crop_resize(image_nd, self._object_roi, self.target_size)
        return self._roi_image, [self._to_crop(clicks)]

    def inv_transform(self, prob_map):
        if self._object_roi is None:
            self._probs = prob_map
            return prob_map
        assert prob_map.shape[0] == 1
        roi = Roi(*self._object_roi)
        patch = _device.resize_align_corners(prob_map, (roi.height, roi.width))
        probs = self._probs_on(prob_map.device)
        if probs is None:
            full = patch                                   # (no earlier prediction to take the canvas size from)
        else:
            full = torch.zeros(tuple(probs.shape), device=patch.device, dtype=patch.dtype)
            full[:, :, roi[0]:roi[1] + 1, roi[2]:roi[3] + 1] = patch
        self._probs = full
        return full

    def check_possible_recalculation(self):
        """With skip_clicks <= 0 and no crop chosen yet: would the first prediction's box cover less than half of the image
        (then the predictor runs the click again, zoomed in)."""
        if self._probs is None or self._object_roi is not None or self.skip_clicks > 0:
            return False
        device = self._probs.device if torch.is_tensor(self._probs) else "cuda"
        box = self._predicted_box(device)
        return box is not None and box.overlap(Roi.whole(*self._input_image_shape[2:])) < 0.50

    def get_state(self):
        probs = self._probs.clone() if torch.is_tensor(self._probs) else self._probs
        return self._input_image_shape, self._object_roi, probs, self._roi_image, self.image_changed

    def set_state(self, state):
        self._input_image_shape, self._object_roi, self._probs, self._roi_image, self.image_changed = state

    def reset(self):
        self._input_image_shape = self._object_roi = self._probs = self._roi_image = None
        self.image_changed = False

    def _to_crop(self, clicks):
        if self._object_roi is None:
            return clicks
        roi, hw = Roi(*self._object_roi), self._roi_image.shape[2:]
        return [c.copy(coords=roi.to_crop(c.coords, hw)) for c in clicks]
