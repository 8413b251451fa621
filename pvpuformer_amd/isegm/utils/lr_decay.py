"""Layer-wise learning-rate decay (BEiT scheme) behind the reference's API, isegm/utils/lr_decay.py:15-84:
``param_groups_lrd`` returns a torch-style list of param groups whose ORDER, lr / weight-decay values and backbone group
keys ("layer_<id>_<decay|no_decay>") are the reference's -- they are behaviour: the optimizer state of a checkpoint is
indexed by group position.  Built here as a two-step table: every backbone tensor is first mapped to a (depth, decays)
bucket, the buckets are then emitted in first-seen order; neck and head tensors follow, one group per tensor.
``per_param_table`` flattens the groups to {name: (lr_scale, weight_decay)} for the fused optimizer's segment table."""
import re

_BLOCK = re.compile(r"blocks\.(\d+)\.")


def get_layer_id_for_vit(name, num_layers):
    """Depth of a backbone tensor (lr_decay.py:72-84): the embedding stage (cls_token, pos_embed, patch_embed*) is depth 0,
    transformer block i is depth i + 1, everything after the blocks (norms, classifier) is depth ``num_layers``."""
    m = _BLOCK.match(name)
    if m:
        return 1 + int(m.group(1))
    embedding_stage = name in ("cls_token", "pos_embed") or name.startswith("patch_embed")
    return 0 if embedding_stage else num_layers


def _bucket_of(name, tensor, skip_decay, depth_count):
    """(depth, regularised?) of one backbone tensor: vectors (biases, norm scales) and the names the model exempts are not
    weight-decayed (lr_decay.py:32-37)."""
    regularised = tensor.ndim != 1 and name not in skip_decay
    return get_layer_id_for_vit(name, depth_count), regularised


def param_groups_lrd(model, lr, weight_decay=0.05, no_weight_decay_list=(), layer_decay=.75):
    """lr_decay.py:15-69.  Backbone: one group per (depth, regularised?) bucket at lr * layer_decay ** (L - depth), L =
    number of blocks + 1, weight decay 0 for the unregularised buckets.  Neck and head: one group per tensor at the
    optimizer's base lr with the full weight decay.  Each group also carries ``names`` (full parameter names: the fused
    optimizer addresses the flat buffer by name)."""
    depth_count = len(model.backbone.blocks) + 1
    rate_at = {d: layer_decay ** (depth_count - d) for d in range(depth_count + 1)}
    buckets = {}          # insertion-ordered: a bucket's position is where its first tensor appears
    for name, tensor in model.backbone.named_parameters():
        if not tensor.requires_grad:
            continue
        depth, regularised = _bucket_of(name, tensor, no_weight_decay_list, depth_count)
        key = f"layer_{depth}_{'decay' if regularised else 'no_decay'}"
        entry = buckets.get(key)
        if entry is None:
            entry = buckets[key] = dict(lr_scale=rate_at[depth], lr=lr * rate_at[depth],
                                        weight_decay=weight_decay if regularised else 0., params=[], names=[])
        entry["params"].append(tensor)
        entry["names"].append(f"backbone.{name}")
    groups = list(buckets.values())
    for owner in ("neck", "head"):
        for name, tensor in getattr(model, owner).named_parameters():
            if tensor.requires_grad:
                groups.append(dict(params=tensor, weight_decay=weight_decay, names=[f"{owner}.{name}"]))
    return groups


def per_param_table(groups, base_lr):
    """{full parameter name: (lr / base_lr, weight_decay)} for every tensor that appears in ``groups``."""
    return {name: (g.get("lr", base_lr) / base_lr, g.get("weight_decay", 0.0)) for g in groups for name in g["names"]}
