"""The whole public surface of the reference's isegm/inference/utils.py (this module takes its name under the overlay;
``scripts/evaluate_vpumodel.py`` calls ``utils.get_dataset`` :114, ``utils.load_is_model`` :118, ``utils.find_checkpoint``
:234, ``utils.get_time_metrics`` :253, ``utils.compute_noc_metric`` :256 and ``utils.get_results_table`` :262).

Unlike the reference file (inference/utils.py:6-7) nothing is imported from ``isegm.data`` at module load: that package is
absent from the reference snapshot, so ``get_dataset`` looks the dataset classes up when it is called.
"""
from datetime import timedelta
from importlib import import_module
from pathlib import Path

import numpy as np
import torch

from ..utils.serialization import load_model


def get_time_metrics(all_ious, elapsed_time):
    """(seconds per click, seconds per image) of an evaluation run (inference/utils.py:11-18)."""
    clicks = sum(len(ious) for ious in all_ious)
    return elapsed_time / clicks, elapsed_time / len(all_ious)


def load_single_is_model(state_dict, device, eval_ritm=False, **kwargs):
    """Rebuilds the network from a ``{'state_dict', 'config'}`` checkpoint, strict key match, frozen, eval mode
    (inference/utils.py:38-46)."""
    model = load_model(state_dict['config'], eval_ritm, **kwargs)
    model.load_state_dict(state_dict['state_dict'], strict=True)
    model.requires_grad_(False)
    model.to(device)
    return model.eval()


def load_is_model(checkpoint, device, eval_ritm=False, **kwargs):
    """``checkpoint`` is a path or an already loaded checkpoint; a list of checkpoints gives ``(first model, all models)``
    (inference/utils.py:21-35).  The file is a pickle of plain containers plus the ``config`` objects, hence
    ``weights_only=False`` (torch >= 2.6 defaults to True and would refuse the pickled loss module of released files)."""
    if isinstance(checkpoint, (str, Path)):
        checkpoint = torch.load(str(checkpoint), map_location='cpu', weights_only=False)
    if isinstance(checkpoint, list):
        models = [load_single_is_model(c, device, eval_ritm, **kwargs) for c in checkpoint]
        return models[0], models
    return load_single_is_model(checkpoint, device, eval_ritm, **kwargs)


# dataset name -> (class in isegm.data.datasets, attribute of cfg holding its root, constructor keywords)
_DATASETS = {
    'GrabCut': ('GrabCutDataset', 'GRABCUT_PATH', {}),
    'Berkeley': ('BerkeleyDataset', 'BERKELEY_PATH', {}),
    'DAVIS': ('DavisDataset', 'DAVIS_PATH', {}),
    'SBD': ('SBDEvaluationDataset', 'SBD_PATH', {}),
    'SBD_Train': ('SBDEvaluationDataset', 'SBD_PATH', {'split': 'train'}),
    'PascalVOC': ('PascalVocDataset', 'PASCALVOC_PATH', {'split': 'val'}),
    'COCO_MVal': ('DavisDataset', 'COCO_MVAL_PATH', {}),
    'BraTS': ('BraTSDataset', 'BraTS_PATH', {}),
    'ssTEM': ('ssTEMDataset', 'ssTEM_PATH', {}),
    'OAIZIB': ('OAIZIBDataset', 'OAIZIB_PATH', {}),
    'HARD': ('HARDDataset', 'HARD_PATH', {}),
    'ADE20K': ('ADE20kDataset', 'ADE20K_PATH', {'split': 'val'}),
}


def get_dataset(dataset_name, cfg):
    """The evaluation dataset of that name built from the paths in ``cfg``; ``None`` for an unknown name
    (inference/utils.py:49-77).  Needs an ``isegm.data.datasets`` from the tree the overlay sits on."""
    entry = _DATASETS.get(dataset_name)
    if entry is None:
        return None
    cls_name, path_key, kw = entry
    datasets = import_module('isegm.data.datasets')
    root = cfg[path_key] if isinstance(cfg, dict) else getattr(cfg, path_key)
    return getattr(datasets, cls_name)(root, **kw)


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


def find_checkpoint(weights_folder, checkpoint_name):
    """Resolves ``[model_folder_prefix:]name`` below ``weights_folder`` to one ``.pth`` path (inference/utils.py:113-132):
    an explicit ``*.pth`` is taken as given if it exists, else relative to the weights folder; a bare name must match
    exactly one file under the (unique) model folder."""
    weights_folder = Path(weights_folder)
    folder = weights_folder
    if ':' in checkpoint_name:
        model_name, checkpoint_name = checkpoint_name.split(':')
        candidates = [d for d in weights_folder.glob(f'{model_name}*') if d.is_dir()]
        assert len(candidates) == 1, f'{len(candidates)} model folders match {model_name!r}'
        folder = candidates[0]
    if checkpoint_name.endswith('.pth'):
        found = checkpoint_name if Path(checkpoint_name).exists() else weights_folder / checkpoint_name
    else:
        matches = list(folder.rglob(f'{checkpoint_name}*.pth'))
        assert len(matches) == 1, f'{len(matches)} checkpoints match {checkpoint_name!r} under {folder}'
        found = matches[0]
    return str(found)


def get_results_table(noc_list, over_max_list, brs_type, dataset_name, mean_spc, elapsed_time, n_clicks=20, model_name=None):
    """(header, row) of the NoC table printed by evaluate_vpumodel.py:262-264; column layout of inference/utils.py:135-159:
    NoC@80/85/90/95 %, failures (>= n_clicks) @85/90/95 %, seconds per click, wall time; '?' where fewer thresholds ran."""
    def cell(values, i, fmt):
        return f'{values[i]:{fmt}}' if len(noc_list) > i else f'{"?":^9}'
    over = f'>={n_clicks}@'
    columns = [f'{"BRS Type":^13}', f'{"Dataset":^11}'] + [f'{"NoC@" + p + "%":^9}' for p in ('80', '85', '90', '95')] \
        + [f'{over + p + "%":^9}' for p in ('85', '90', '95')] + [f'{"SPC,s":^7}', f'{"Time":^9}']
    head = '|' + '|'.join(columns) + '|'
    rule = '-' * len(head)
    header = (f'Eval results for model: {model_name}\n' if model_name is not None else '') + rule + '\n' + head + '\n' + rule
    cells = [f'{brs_type:^13}', f'{dataset_name:^11}', f'{noc_list[0]:^9.2f}'] \
        + [cell(noc_list, i, '^9.2f') for i in (1, 2, 3)] + [cell(over_max_list, i, '^9') for i in (1, 2, 3)] \
        + [f'{mean_spc:^7.3f}', f'{str(timedelta(seconds=int(elapsed_time))):^9}']
    return header, '|' + '|'.join(cells) + '|'
