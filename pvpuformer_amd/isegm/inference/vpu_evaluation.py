"""NoBRS evaluation protocol, call-compatible with isegm/inference/vpu_evaluation.py:18-98: ``evaluate_dataset(dataset,
predictor, **kwargs)`` -> (list of per-object IoU series, elapsed seconds), ``evaluate_sample(image, gt_mask, predictor,
max_iou_thr, pred_thr=0.49, min_clicks=1, max_clicks=20, sample_id=None, callback=None)`` -> (clicks, IoU series,
last probability map).  Per click: the Clicker puts the next oracle click at the deepest point of the larger error
region, the predictor runs the network on (image, previous prediction, clicks [+ box prompt]) through its transform
pipeline, the IoU at ``pred_thr`` is recorded, and the loop stops once ``max_iou_thr`` is reached after ``min_clicks``.
As shipped the reference always evaluates with click prompts (``as_prompt_type = 0``, vpu_evaluation.py:51)."""
import time

import numpy as np
import torch

from . import utils
from .clicker import Clicker


def click_series(image, gt_mask, predictor, pred_thr=0.49, max_clicks=20, as_prompt_type=0):
    """Generator over the clicks of one object: yields (click index, probability map, IoU, clicker, prompts)."""
    clicker = Clicker(gt_mask=gt_mask)
    mask = np.zeros(gt_mask.shape, dtype=bool)
    predictor.set_input_image(image)
    for k in range(max_clicks):
        clicker.make_next_click(mask)
        probs, prompts = predictor.get_vqu_prediction(clicker, gt_mask=gt_mask, as_prompt_type=as_prompt_type, click_indx=k,
                                                      as_multi_prompts=True)
        mask = probs > pred_thr
        yield k, probs, utils.get_iou(gt_mask, mask), clicker, prompts


def evaluate_sample(image, gt_mask, predictor, max_iou_thr, pred_thr=0.49, min_clicks=1, max_clicks=20, sample_id=None,
                    callback=None, as_prompt_type=0):
    ious, probs, clicker = [], None, None
    # the callback (the reference's visualisation hook, vpu_evaluation.py:84-97) is the one consumer of ``prompts`` in this
    # protocol: with one attached the predictor derives the box prompt on every click as the reference does (base.py:176)
    keep = getattr(predictor, "always_simulate_prompts", None)
    if callback is not None and keep is not None:
        predictor.always_simulate_prompts = True
    try:
        return _evaluate_sample(image, gt_mask, predictor, max_iou_thr, pred_thr, min_clicks, max_clicks, sample_id, callback,
                                as_prompt_type)
    finally:
        if keep is not None:
            predictor.always_simulate_prompts = keep


def _evaluate_sample(image, gt_mask, predictor, max_iou_thr, pred_thr, min_clicks, max_clicks, sample_id, callback, as_prompt_type):
    ious, probs, clicker = [], None, None
    with torch.no_grad():
        for k, probs, iou, clicker, prompts in click_series(image, gt_mask, predictor, pred_thr, max_clicks, as_prompt_type):
            ious.append(iou)
            reached = iou >= max_iou_thr and k + 1 >= min_clicks
            if callback is not None:
                callback(image, gt_mask, probs, iou, sample_id, k, clicker.clicks_list, reached, predictor.zoom_in, prompts,
                         as_prompt_type)
            if reached:
                break
    return (clicker.clicks_list if clicker is not None else []), np.array(ious, dtype=np.float32), probs


def evaluate_dataset(dataset, predictor, **kwargs):
    """``dataset``: anything with ``len()`` and ``get_sample(i)`` returning an object with ``image``, ``objects_ids``
    and ``gt_mask(object_id)`` (the contract of the reference's isegm.data datasets)."""
    started = time.time()
    series = []
    for index in range(len(dataset)):
        sample = dataset.get_sample(index)
        series += [evaluate_sample(sample.image, sample.gt_mask(oid), predictor, sample_id=index, **kwargs)[1]
                   for oid in sample.objects_ids]
    return series, time.time() - started
