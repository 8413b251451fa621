"""IoU / NoC metrics, API-compatible with isegm/inference/utils.py:80-110."""
import numpy as np


def get_iou(gt_mask, pred_mask, ignore_label=-1):
    keep = gt_mask != ignore_label
    obj = gt_mask == 1
    inter = np.logical_and(np.logical_and(pred_mask, obj), keep).sum()
    union = np.logical_and(np.logical_or(pred_mask, obj), keep).sum()
    return inter / union


def compute_noc_metric(all_ious, iou_thrs, max_clicks=20):
    def noc(iou_arr, thr):
        vals = iou_arr >= thr
        return np.argmax(vals) + 1 if np.any(vals) else max_clicks
    noc_list, noc_std, over_max = [], [], []
    for thr in iou_thrs:
        scores = np.array([noc(a, thr) for a in all_ious], dtype=int)
        noc_list.append(scores.mean()); noc_std.append(scores.std()); over_max.append((scores == max_clicks).sum())
    return noc_list, noc_std, over_max
