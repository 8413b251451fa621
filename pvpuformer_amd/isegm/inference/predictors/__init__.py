from .base import BasePredictor


def get_predictor(net, brs_mode, device, prob_thresh=0.49, with_flip=True, zoom_in_params=None, predictor_params=None,
                  **kwargs):
    """isegm/inference/predictors/__init__.py:9-99 restricted to the north star's NoBRS mode."""
    if brs_mode != 'NoBRS':
        raise NotImplementedError("only the NoBRS predictor is on the VPUFormer hot path (BRS variants are RITM legacy)")
    if zoom_in_params is not None:
        raise NotImplementedError("ZoomIn is not built yet (SURVEY.md section 8f); pass zoom_in_params=None")
    params = dict(predictor_params or {})
    return BasePredictor(net, device, with_flip=with_flip, **params)
