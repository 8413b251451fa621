"""``CrossEntropyLoss`` under the dotted path released VPUFormer checkpoints pickle
(reference: isegm/model/modeling/transformer_helper/cross_entropy_loss.py:140-203, constructed at
models/iSegNet/vpu_base448_cocolvis.py:39 and stored in the model's ``_config``).

The trainer never calls it (the head's ``loss_decode`` is not on the training path: trainer.py:399-419 applies NFL, Dice
and the P2CL BCE), so this is a small torch-only restatement with the same constructor keywords, attribute names (the
pickle restores ``__dict__``) and the three criterion functions the pickle references by name -- no mmcv registry."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reduce(loss, weight, reduction, avg_factor):
    """Element weights, then 'none' | 'mean' | 'sum'; ``avg_factor`` replaces the element count of 'mean'."""
    if weight is not None:
        loss = loss * weight.float()
    if reduction == 'none':
        return loss
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else loss.sum()
    if reduction != 'mean':
        raise ValueError('avg_factor can only be used with reduction="mean"')
    return loss.sum() / avg_factor


def cross_entropy(pred, label, weight=None, class_weight=None, reduction='mean', avg_factor=None, ignore_index=-100):
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none', ignore_index=ignore_index)
    return _reduce(loss, weight, reduction, avg_factor)


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None, ignore_index=255):
    if pred.dim() != label.dim():                       # class indices -> one-hot planes, ignored pixels weighted 0
        valid = (label >= 0) & (label != ignore_index)
        onehot = torch.zeros_like(pred)
        onehot.scatter_(1, label.clamp(min=0, max=pred.shape[1] - 1).unsqueeze(1), 1.0)
        onehot = onehot * valid.unsqueeze(1)
        valid = valid.unsqueeze(1).expand_as(pred).float()
        weight = valid if weight is None else weight.unsqueeze(1).expand_as(pred) * valid
        label = onehot
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction='none')
    return _reduce(loss, weight, reduction, avg_factor)


def mask_cross_entropy(pred, target, label, reduction='mean', avg_factor=None, class_weight=None, ignore_index=None):
    assert ignore_index is None and reduction == 'mean' and avg_factor is None
    rows = torch.arange(pred.shape[0], device=pred.device)
    return F.binary_cross_entropy_with_logits(pred[rows, label].squeeze(1), target, weight=class_weight, reduction='mean')[None]


class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert not (use_sigmoid and use_mask)
        self.use_sigmoid = use_sigmoid
        self.use_mask = use_mask
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.class_weight = np.load(class_weight) if isinstance(class_weight, str) else class_weight
        self.cls_criterion = binary_cross_entropy if use_sigmoid else mask_cross_entropy if use_mask else cross_entropy

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        class_weight = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=class_weight,
                                                     reduction=reduction_override or self.reduction,
                                                     avg_factor=avg_factor, **kwargs)
