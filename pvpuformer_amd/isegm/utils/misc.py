"""Checkpoint writer, file-format compatible with isegm/utils/misc.py:15-33: ``{'state_dict', 'config'}`` where
``config`` is the ``@serialize`` capture of the constructor (dotted class path + keyword arguments), so
``isegm.inference.utils.load_is_model`` / ``load_model`` of either code base can rebuild the network from the file."""
import os

import torch


def save_checkpoint(net, checkpoints_path, epoch=None, prefix='', verbose=False, multi_gpu=False):
    name = 'last_checkpoint.pth' if epoch is None else f'{epoch:03d}.pth'
    if prefix:
        name = f'{prefix}_{name}'
    os.makedirs(str(checkpoints_path), exist_ok=True)
    path = os.path.join(str(checkpoints_path), name)
    net = net.module if multi_gpu and hasattr(net, 'module') else net
    torch.save({'state_dict': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'config': net._config}, path)
    return path
