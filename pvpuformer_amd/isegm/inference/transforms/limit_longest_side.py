"""LimitLongestSide (isegm/inference/transforms/limit_longest_side.py:4-22): images whose longer side exceeds ``max_size``
are shrunk to it as a whole (a ZoomIn whose crop is always the full image)."""
from .roi import Roi
from .zoom_in import ZoomIn, crop_resize


class LimitLongestSide(ZoomIn):
    def __init__(self, max_size=800):
        super().__init__(target_size=max_size, skip_clicks=0)

    def transform(self, image_nd, clicks_lists):
        (clicks,) = clicks_lists
        batch, _, height, width = image_nd.shape
        assert batch == 1
        self.image_changed = max(height, width) > self.target_size
        if not self.image_changed:
            return image_nd, clicks_lists
        self._object_roi = Roi.whole(height, width)
        self._roi_image = crop_resize(image_nd, self._object_roi, self.target_size)
        return self._roi_image, [self._to_crop(clicks)]
