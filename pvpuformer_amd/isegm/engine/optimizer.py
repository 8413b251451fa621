"""Optimizer factories, API-compatible with isegm/engine/optimizer.py:6-41, returning the fused single-launch optimizer
instead of a torch optimizer with one param group per tensor."""
import math

from ...optim import FusedAdam
from ..utils import lr_decay as lrd


def _check(opt_name):
    if opt_name.lower() not in ("adam", "adamw"):
        raise NotImplementedError("the fused optimizer implements Adam and AdamW (the reference's configs use 'adam')")
    return opt_name.lower() == "adamw"


def get_optimizer(model, opt_name, opt_kwargs):
    """optimizer.py:6-27: base lr for every tensor, ``param.lr_mult`` honoured where set."""
    decoupled = _check(opt_name)
    kw = dict(opt_kwargs)
    wd = kw.pop("weight_decay", 0.0)
    per_param = {}
    for name, param in model.named_parameters():
        mult = getattr(param, "lr_mult", 1.0)
        if param.requires_grad and not math.isclose(mult, 1.0):
            per_param[name] = (mult, wd)
    return FusedAdam(model, weight_decay=wd, decoupled_weight_decay=decoupled, per_param=per_param or None, **kw)


def get_optimizer_with_layerwise_decay(model, opt_name, opt_kwargs):
    """optimizer.py:29-41: layer decay 0.75, weight decay 0.02; tensors in no group (patch_embed_coords, pe_layer, ...)
    are not optimised by the reference -- they get learning-rate scale 0 here."""
    decoupled = _check(opt_name)
    kw = dict(opt_kwargs)
    lr = kw["lr"]
    kw.pop("weight_decay", None)
    groups = lrd.param_groups_lrd(model, lr, weight_decay=0.02, no_weight_decay_list=model.backbone.no_weight_decay(),
                                  layer_decay=0.75)
    table = lrd.per_param_table(groups, lr)
    per_param = {n: table.get(n, (0.0, 0.0)) for n, _ in model.named_parameters()}
    return FusedAdam(model, weight_decay=0.0, decoupled_weight_decay=decoupled, per_param=per_param, **kw)
