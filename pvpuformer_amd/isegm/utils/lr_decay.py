"""Layer-wise learning-rate decay groups, API-compatible with isegm/utils/lr_decay.py:15-84 (BEiT scheme): returns the
same torch-style list of param groups; ``per_param_table`` flattens it to {name: (lr_scale, weight_decay)} for the fused
optimizer."""


def get_layer_id_for_vit(name, num_layers):
    """lr_decay.py:72-84."""
    if name in ['cls_token', 'pos_embed']:
        return 0
    if name.startswith('patch_embed'):
        return 0
    if name.startswith('blocks'):
        return int(name.split('.')[1]) + 1
    return num_layers


def param_groups_lrd(model, lr, weight_decay=0.05, no_weight_decay_list=(), layer_decay=.75):
    """lr_decay.py:15-69: backbone tensors grouped by (layer id, decay / no decay) with lr * layer_decay**(L - id);
    1-D tensors and the no-decay list get weight_decay 0; neck / head tensors one group each, base lr, full decay."""
    param_groups = {}
    num_layers = len(model.backbone.blocks) + 1
    layer_scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    for n, p in model.backbone.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim == 1 or n in no_weight_decay_list:
            g_decay, this_decay = "no_decay", 0.
        else:
            g_decay, this_decay = "decay", weight_decay
        layer_id = get_layer_id_for_vit(n, num_layers)
        group_name = "layer_%d_%s" % (layer_id, g_decay)
        if group_name not in param_groups:
            this_scale = layer_scales[layer_id]
            param_groups[group_name] = {"lr_scale": this_scale, "lr": lr * this_scale, "weight_decay": this_decay,
                                        "params": [], "names": []}
        param_groups[group_name]["params"].append(p)
        param_groups[group_name]["names"].append("backbone." + n)
    params = list(param_groups.values())
    for prefix, sub in (("neck.", model.neck), ("head.", model.head)):
        for n, p in sub.named_parameters():
            if p.requires_grad:
                params.append({"params": p, "weight_decay": weight_decay, "names": [prefix + n]})
    return params


def per_param_table(groups, base_lr):
    """{full parameter name: (lr / base_lr, weight_decay)} for every tensor that appears in ``groups``."""
    table = {}
    for g in groups:
        scale = g.get("lr", base_lr) / base_lr
        for n in g["names"]:
            table[n] = (scale, g.get("weight_decay", 0.0))
    return table
