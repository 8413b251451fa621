"""Predictor-side transforms of the NoBRS loop, API-compatible with isegm/inference/transforms/
(base.py:4-38, flip.py:8-37, zoom_in.py:9-200, limit_longest_side.py:4-22): ``transform / inv_transform / reset /
get_state / set_state`` and the ``image_changed`` flag.

ZoomIn keeps the reference's host-side bookkeeping (ROI from the previous prediction and the positive clicks, expansion,
clamping, IoU-triggered re-crop, click re-mapping) in integer / Python-float arithmetic so that it is bit-identical, and
does the two ``align_corners=True`` bilinear resizes (crop -> network size, prediction -> crop size) with the HIP kernel
``vpu_upsample_ac_fwd`` on the GPU.  There is no CPU resize path: the transforms need CUDA tensors.
"""
import numpy as np
import torch

from ... import ops


# ---- isegm/utils/misc.py:36-79
def get_bbox_from_mask(mask):
    rows, cols = np.any(mask, axis=1), np.any(mask, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    return rmin, rmax, cmin, cmax


def expand_bbox(bbox, expand_ratio, min_crop_size=None):
    rmin, rmax, cmin, cmax = bbox
    rcenter, ccenter = 0.5 * (rmin + rmax), 0.5 * (cmin + cmax)
    height, width = expand_ratio * (rmax - rmin + 1), expand_ratio * (cmax - cmin + 1)
    if min_crop_size is not None:
        height, width = max(height, min_crop_size), max(width, min_crop_size)
    return (int(round(rcenter - 0.5 * height)), int(round(rcenter + 0.5 * height)),
            int(round(ccenter - 0.5 * width)), int(round(ccenter + 0.5 * width)))


def clamp_bbox(bbox, rmin, rmax, cmin, cmax):
    return max(rmin, bbox[0]), min(rmax, bbox[1]), max(cmin, bbox[2]), min(cmax, bbox[3])


def get_segments_iou(s1, s2):
    (a, b), (c, d) = s1, s2
    return max(0, min(b, d) - max(a, c) + 1) / max(1e-6, max(b, d) - min(a, c) + 1)


def get_bbox_iou(b1, b2):
    return get_segments_iou(b1[:2], b2[:2]) * get_segments_iou(b1[2:4], b2[2:4])


# ---- resize on the GPU
def resize_align_corners(x, size):
    """``F.interpolate(x, size, mode='bilinear', align_corners=True)`` for a CUDA fp32 NCHW tensor (HIP kernel)."""
    if not x.is_cuda:
        raise RuntimeError("the predictor transforms run on the GPU only (no CPU resize path exists)")
    n, c, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    src = x.contiguous().float()
    if (h, w) == (H, W):
        return src.clone()
    dst = torch.empty(n, c, H, W, device=x.device, dtype=torch.float32)
    ops.upsample_ac_fwd(src, dst, n * c, h, w, H, W)
    return dst


def get_roi_image_nd(image_nd, object_roi, target_size):
    """zoom_in.py:168-185."""
    rmin, rmax, cmin, cmax = object_roi
    height, width = rmax - rmin + 1, cmax - cmin + 1
    if isinstance(target_size, tuple):
        new_height, new_width = target_size
    else:
        scale = target_size / max(height, width)
        new_height, new_width = int(round(height * scale)), int(round(width * scale))
    with torch.no_grad():
        return resize_align_corners(image_nd[:, :, rmin:rmax + 1, cmin:cmax + 1], (new_height, new_width))


def get_object_roi(pred_mask, clicks_list, expansion_ratio, min_crop_size):
    """zoom_in.py:153-165."""
    pred_mask = pred_mask.copy()
    for click in clicks_list:
        if click.is_positive:
            pred_mask[int(click.coords[0]), int(click.coords[1])] = 1
    bbox = expand_bbox(get_bbox_from_mask(pred_mask), expansion_ratio, min_crop_size)
    h, w = pred_mask.shape[0], pred_mask.shape[1]
    return clamp_bbox(bbox, 0, h - 1, 0, w - 1)


def check_object_roi(object_roi, clicks_list):
    """zoom_in.py:188-196 (the upper bounds are exclusive there; kept)."""
    for click in clicks_list:
        if click.is_positive:
            if click.coords[0] < object_roi[0] or click.coords[0] >= object_roi[1]:
                return False
            if click.coords[1] < object_roi[2] or click.coords[1] >= object_roi[3]:
                return False
    return True


class BaseTransform:
    def __init__(self):
        self.image_changed = False

    def transform(self, image_nd, clicks_lists):
        raise NotImplementedError

    def inv_transform(self, prob_map):
        raise NotImplementedError

    def reset(self):
        pass

    def get_state(self):
        return None

    def set_state(self, state):
        pass


class SigmoidForPred(BaseTransform):
    def transform(self, image_nd, clicks_lists):
        return image_nd, clicks_lists

    def inv_transform(self, prob_map):
        return torch.sigmoid(prob_map)


class AddHorizontalFlip(BaseTransform):
    def transform(self, image_nd, clicks_lists):
        assert image_nd.dim() == 4
        image_nd = torch.cat([image_nd, torch.flip(image_nd, dims=[3])], dim=0)
        w = image_nd.shape[3]
        flipped = [[c.copy(coords=(c.coords[0], w - c.coords[1] - 1)) for c in cl] for cl in clicks_lists]
        return image_nd, clicks_lists + flipped

    def inv_transform(self, prob_map):
        assert prob_map.dim() == 4 and prob_map.shape[0] % 2 == 0
        half = prob_map.shape[0] // 2
        return 0.5 * (prob_map[:half] + torch.flip(prob_map[half:], dims=[3]))


class ZoomIn(BaseTransform):
    def __init__(self, target_size=400, skip_clicks=1, expansion_ratio=1.4, min_crop_size=200,
                 recompute_thresh_iou=0.5, prob_thresh=0.50):
        super().__init__()
        self.target_size, self.min_crop_size, self.skip_clicks = target_size, min_crop_size, skip_clicks
        self.expansion_ratio, self.recompute_thresh_iou, self.prob_thresh = expansion_ratio, recompute_thresh_iou, prob_thresh
        self.reset()

    def transform(self, image_nd, clicks_lists):
        assert image_nd.shape[0] == 1 and len(clicks_lists) == 1
        self.image_changed = False
        clicks_list = clicks_lists[0]
        if len(clicks_list) <= self.skip_clicks:
            return image_nd, clicks_lists
        self._input_image_shape = image_nd.shape
        current_object_roi = None
        if self._prev_probs is not None:
            current_pred_mask = (self._prev_probs > self.prob_thresh)[0, 0]
            if current_pred_mask.sum() > 0:
                current_object_roi = get_object_roi(current_pred_mask, clicks_list, self.expansion_ratio,
                                                    self.min_crop_size)
        if current_object_roi is None:
            if self.skip_clicks >= 0:
                return image_nd, clicks_lists
            current_object_roi = 0, image_nd.shape[2] - 1, 0, image_nd.shape[3] - 1
        update = (self._object_roi is None or not check_object_roi(self._object_roi, clicks_list) or
                  get_bbox_iou(current_object_roi, self._object_roi) < self.recompute_thresh_iou)
        if update:
            self._object_roi = current_object_roi
            self.image_changed = True
        self._roi_image = get_roi_image_nd(image_nd, self._object_roi, self.target_size)
        return self._roi_image.to(image_nd.device), [self._transform_clicks(clicks_list)]

    def inv_transform(self, prob_map):
        if self._object_roi is None:
            self._prev_probs = prob_map.cpu().numpy()
            return prob_map
        assert prob_map.shape[0] == 1
        rmin, rmax, cmin, cmax = self._object_roi
        prob_map = resize_align_corners(prob_map, (rmax - rmin + 1, cmax - cmin + 1))
        if self._prev_probs is not None:
            new_prob_map = torch.zeros(*self._prev_probs.shape, device=prob_map.device, dtype=prob_map.dtype)
            new_prob_map[:, :, rmin:rmax + 1, cmin:cmax + 1] = prob_map
        else:
            new_prob_map = prob_map
        self._prev_probs = new_prob_map.cpu().numpy()
        return new_prob_map

    def check_possible_recalculation(self):
        if self._prev_probs is None or self._object_roi is not None or self.skip_clicks > 0:
            return False
        pred_mask = (self._prev_probs > self.prob_thresh)[0, 0]
        if pred_mask.sum() > 0:
            possible = get_object_roi(pred_mask, [], self.expansion_ratio, self.min_crop_size)
            image_roi = (0, self._input_image_shape[2] - 1, 0, self._input_image_shape[3] - 1)
            if get_bbox_iou(possible, image_roi) < 0.50:
                return True
        return False

    def get_state(self):
        roi_image = self._roi_image.cpu() if self._roi_image is not None else None
        return self._input_image_shape, self._object_roi, self._prev_probs, roi_image, self.image_changed

    def set_state(self, state):
        self._input_image_shape, self._object_roi, self._prev_probs, self._roi_image, self.image_changed = state

    def reset(self):
        self._input_image_shape = None
        self._object_roi = None
        self._prev_probs = None
        self._roi_image = None
        self.image_changed = False

    def _transform_clicks(self, clicks_list):
        if self._object_roi is None:
            return clicks_list
        rmin, rmax, cmin, cmax = self._object_roi
        crop_height, crop_width = self._roi_image.shape[2:]
        out = []
        for click in clicks_list:
            new_r = crop_height * (click.coords[0] - rmin) / (rmax - rmin + 1)
            new_c = crop_width * (click.coords[1] - cmin) / (cmax - cmin + 1)
            out.append(click.copy(coords=(new_r, new_c)))
        return out


class LimitLongestSide(ZoomIn):
    """limit_longest_side.py:4-22."""

    def __init__(self, max_size=800):
        super().__init__(target_size=max_size, skip_clicks=0)

    def transform(self, image_nd, clicks_lists):
        assert image_nd.shape[0] == 1 and len(clicks_lists) == 1
        self.image_changed = False
        if max(image_nd.shape[2:4]) <= self.target_size:
            return image_nd, clicks_lists
        self._input_image = image_nd
        self._object_roi = (0, image_nd.shape[2] - 1, 0, image_nd.shape[3] - 1)
        self._roi_image = get_roi_image_nd(image_nd, self._object_roi, self.target_size)
        self.image_changed = True
        return self._roi_image, [self._transform_clicks(clicks_lists[0])]
