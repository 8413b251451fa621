"""Position-embedding re-gridding, API-compatible with isegm/model/modeling/pos_embed.py:75-128 (host-side,
load-time only: bicubic interpolation through torch is plumbing, not the hot path)."""
import torch
import torch.nn.functional as F


def _regrid(pos_tokens, old, new, dim):
    pos_tokens = pos_tokens.reshape(-1, old[0], old[1], dim).permute(0, 3, 1, 2)
    pos_tokens = F.interpolate(pos_tokens, size=new, mode="bicubic", align_corners=False)
    return pos_tokens.permute(0, 2, 3, 1).flatten(1, 2)


def interpolate_pos_embed(model, checkpoint_model):
    """MAE 14x14 -> model grid at load (pos_embed.py:75-96); mutates ``checkpoint_model`` in place."""
    if "pos_embed" not in checkpoint_model:
        return
    pe = checkpoint_model["pos_embed"]
    dim = pe.shape[-1]
    num_patches = model.patch_embed.num_patches
    extra = model.pos_embed.shape[-2] - num_patches
    orig = int((pe.shape[-2] - extra) ** 0.5)
    new = model.patch_embed.grid_size
    if (orig, orig) != tuple(new):
        tokens = _regrid(pe[:, extra:], (orig, orig), tuple(new), dim)
        checkpoint_model["pos_embed"] = torch.cat((pe[:, :extra], tokens), dim=1)


def interpolate_pos_embed_inference(model, infer_img_size, device):
    """Eval-time re-gridding (pos_embed.py:99-128, called by scripts/evaluate_vpumodel.py:125 before an evaluation at
    another input size).  The reference REPLACES ``model.pos_embed`` by its bicubic re-gridding; here the trained embedding
    stays where it is -- inside the engine's flat parameter buffer -- and the engine derives the embedding of whatever
    token grid an input has from it on demand (``Engine._pos_for``: the same bicubic interpolation, cached per grid), so
    this call only records the evaluation grid on ``patch_embed`` as the reference does."""
    ps = model.patch_embed.patch_size
    new = (infer_img_size[0] // ps[0], infer_img_size[1] // ps[1])
    model.patch_embed.grid_size = new
    model.patch_embed.num_patches = new[0] * new[1]
    model.infer_grid_size = new


def regridded_pos_embed(model, infer_img_size):
    """The tensor the reference would install as ``pos_embed`` for ``infer_img_size`` (class token kept, grid tokens
    bicubically re-gridded) -- for checkpoints meant to be read back by the reference at that size."""
    pe = model.pos_embed.detach()
    dim = pe.shape[-1]
    ps = model.patch_embed.patch_size
    new = (infer_img_size[0] // ps[0], infer_img_size[1] // ps[1])
    old = int(round((pe.shape[-2] - 1) ** 0.5))
    if (old, old) == tuple(new):
        return pe.clone()
    return torch.cat((pe[:, :1], _regrid(pe[:, 1:], (old, old), new, dim)), dim=1)
