"""Transform protocol of the predictors (isegm/inference/transforms/base.py:4-38): ``transform`` maps the network input and
the click lists, ``inv_transform`` maps the prediction back; ``image_changed`` tells the predictor that cached features
(there are none on this path) would be stale."""
import torch


class BaseTransform:
    image_changed = False

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
    """Logits -> probabilities on the way back (applied after the flip average: the pipeline is inverted in reverse)."""

    def transform(self, image_nd, clicks_lists):
        return image_nd, clicks_lists

    def inv_transform(self, prob_map):
        return torch.sigmoid(prob_map)
