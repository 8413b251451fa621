from .base import BasePredictor
from ..transforms import ZoomIn


def get_predictor(net, brs_mode, device, prob_thresh=0.49, with_flip=True, zoom_in_params=dict(), predictor_params=None,
                  **kwargs):
    """isegm/inference/predictors/__init__.py:9-99 restricted to the north star's NoBRS mode."""
    if brs_mode != 'NoBRS':
        raise NotImplementedError("only the NoBRS predictor is on the VPUFormer hot path (BRS variants are RITM legacy)")
    zoom_in = ZoomIn(**zoom_in_params) if zoom_in_params is not None else None
    params = dict(predictor_params or {})
    params.pop('optimize_after_n_clicks', None)
    return BasePredictor(net, device, zoom_in=zoom_in, with_flip=with_flip, **params)
