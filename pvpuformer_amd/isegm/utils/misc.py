"""The whole public surface of the reference's isegm/utils/misc.py (this module takes its name under the overlay, so the
reference's own losses / metrics / datasets find every helper they call here).

* ``save_checkpoint`` -- file-format compatible with misc.py:15-33: ``{'state_dict', 'config'}`` where ``config`` is the
  ``@serialize`` capture of the constructor (dotted class path + keyword arguments), so
  ``isegm.inference.utils.load_is_model`` / ``load_model`` of either code base can rebuild the network from the file.
* the reduction-axis and bounding-box helpers of misc.py:7-13,36-86 (callers: losses.py:80-83,128-131,176,
  metrics.py:90, the data pipeline and the ZoomIn transform), written fresh on numpy.
"""
import os

import numpy as np
import torch


def get_dims_with_exclusion(dim, exclude=None):
    """Axes ``0..dim-1`` without ``exclude`` (misc.py:7-12; like ``list.remove`` an absent axis is an error)."""
    if exclude is not None and not 0 <= exclude < dim:
        raise ValueError(f"axis {exclude} is not one of the {dim} axes")
    return [d for d in range(dim) if d != exclude]


def save_checkpoint(net, checkpoints_path, epoch=None, prefix='', verbose=True, multi_gpu=False):
    name = 'last_checkpoint.pth' if epoch is None else f'{epoch:03d}.pth'
    if prefix:
        name = f'{prefix}_{name}'
    os.makedirs(str(checkpoints_path), exist_ok=True)
    path = os.path.join(str(checkpoints_path), name)
    if verbose:
        try:
            from .log import logger            # the other tree's logger when the overlay has one
            logger.info(f'Save checkpoint to {path}')
        except ImportError:
            print(f'Save checkpoint to {path}')
    net = net.module if multi_gpu and hasattr(net, 'module') else net
    torch.save({'state_dict': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'config': net._config}, path)
    return path


def get_bbox_from_mask(mask):
    """(rmin, rmax, cmin, cmax), inclusive, of the non-zero pixels (misc.py:36-42)."""
    mask = np.asarray(mask)
    r = np.flatnonzero(mask.any(axis=1))
    c = np.flatnonzero(mask.any(axis=0))
    return r[0], r[-1], c[0], c[-1]


def expand_bbox(bbox, expand_ratio, min_crop_size=None):
    """Scales an inclusive box about its centre; python ``round`` (half to even) as in misc.py:45-60."""
    rmin, rmax, cmin, cmax = bbox
    extent = [expand_ratio * (rmax - rmin + 1), expand_ratio * (cmax - cmin + 1)]
    if min_crop_size is not None:
        extent = [max(e, min_crop_size) for e in extent]
    out = []
    for centre, e in zip((0.5 * (rmin + rmax), 0.5 * (cmin + cmax)), extent):
        out += [int(round(centre - 0.5 * e)), int(round(centre + 0.5 * e))]
    return tuple(out)


def clamp_bbox(bbox, rmin, rmax, cmin, cmax):
    return max(rmin, bbox[0]), min(rmax, bbox[1]), max(cmin, bbox[2]), min(cmax, bbox[3])


def get_segments_iou(s1, s2):
    """IoU of two inclusive integer intervals (misc.py:74-79)."""
    lo = max(s1[0], s2[0]); hi = min(s1[1], s2[1])
    inter = max(0, hi - lo + 1)
    union = max(1e-6, max(s1[1], s2[1]) - min(s1[0], s2[0]) + 1)
    return inter / union


def get_bbox_iou(b1, b2):
    return get_segments_iou(b1[:2], b2[:2]) * get_segments_iou(b1[2:4], b2[2:4])


def get_labels_with_sizes(x):
    """Non-zero labels present in an integer label image and their pixel counts (misc.py:82-86)."""
    sizes = np.bincount(np.asarray(x).reshape(-1))
    labels = [int(l) for l in np.flatnonzero(sizes) if l != 0]
    return labels, sizes[labels].tolist()
