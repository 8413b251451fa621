"""NoBRS evaluation protocol, call-compatible with isegm/inference/vpu_evaluation.py:18-98: ``evaluate_dataset(dataset,
predictor, **kwargs)`` -> (list of per-object IoU series, elapsed seconds), ``evaluate_sample(image, gt_mask, predictor,
max_iou_thr, pred_thr=0.49, min_clicks=1, max_clicks=20, sample_id=None, callback=None)`` -> (clicks, IoU series,
last probability map).  Per click: the Clicker puts the next oracle click at the deepest point of the larger error
region, the predictor runs the network on (image, previous prediction, clicks [+ box prompt]) through its transform
pipeline, the IoU at ``pred_thr`` is recorded, and the loop stops once ``max_iou_thr`` is reached after ``min_clicks``.
As shipped the reference always evaluates with click prompts (``as_prompt_type = 0``, vpu_evaluation.py:51)."""
from time import time

import numpy as np
import torch

from . import utils
from .clicker import Clicker


def evaluate_sample(image, gt_mask, predictor, max_iou_thr, pred_thr=0.49, min_clicks=1, max_clicks=20, sample_id=None,
                    callback=None, as_prompt_type=0):
    clicker = Clicker(gt_mask=gt_mask)
    pred_mask = np.zeros_like(gt_mask)
    ious, pred_probs = [], None
    with torch.no_grad():
        predictor.set_input_image(image)
        for click_indx in range(max_clicks):
            clicker.make_next_click(pred_mask)
            pred_probs, prompts = predictor.get_vqu_prediction(clicker, gt_mask=gt_mask, as_prompt_type=as_prompt_type,
                                                               click_indx=click_indx, as_multi_prompts=True)
            pred_mask = pred_probs > pred_thr
            iou = utils.get_iou(gt_mask, pred_mask)
            ious.append(iou)
            done = iou >= max_iou_thr and click_indx + 1 >= min_clicks
            if callback is not None:
                callback(image, gt_mask, pred_probs, iou, sample_id, click_indx, clicker.clicks_list, done,
                         predictor.zoom_in, prompts, as_prompt_type)
            if done:
                break
    return clicker.clicks_list, np.array(ious, dtype=np.float32), pred_probs


def evaluate_dataset(dataset, predictor, **kwargs):
    """``dataset``: anything with ``len()`` and ``get_sample(i)`` returning an object with ``image``, ``objects_ids``
    and ``gt_mask(object_id)`` (the contract of the reference's isegm.data datasets)."""
    all_ious = []
    t0 = time()
    for index in range(len(dataset)):
        sample = dataset.get_sample(index)
        for object_id in sample.objects_ids:
            _, sample_ious, _ = evaluate_sample(sample.image, sample.gt_mask(object_id), predictor, sample_id=index, **kwargs)
            all_ious.append(sample_ious)
    return all_ious, time() - t0
