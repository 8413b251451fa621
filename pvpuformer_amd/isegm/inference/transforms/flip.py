"""Horizontal-flip test-time augmentation (isegm/inference/transforms/flip.py:8-37): the batch becomes
[image, mirrored image] with mirrored clicks, the two predictions are averaged after un-mirroring the second."""
import torch

from .base import BaseTransform


class AddHorizontalFlip(BaseTransform):
    def transform(self, image_nd, clicks_lists):
        assert image_nd.dim() == 4
        width = image_nd.shape[3]
        mirrored = [[c.copy(coords=(c.coords[0], width - c.coords[1] - 1)) for c in clicks] for clicks in clicks_lists]
        return torch.cat([image_nd, image_nd.flip(3)], 0), clicks_lists + mirrored

    def inv_transform(self, prob_map):
        assert prob_map.dim() == 4 and prob_map.shape[0] % 2 == 0
        straight, mirrored = prob_map.chunk(2, 0)
        return 0.5 * (straight + mirrored.flip(3))
