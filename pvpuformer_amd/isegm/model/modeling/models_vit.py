"""Parameter container with the public surface of the reference's VisionTransformer
(isegm/model/modeling/models_vit.py:107-319): ``pos_embed``, ``patch_embed.{num_patches,grid_size,patch_size}``,
``blocks``, ``no_weight_decay()``, ``init_weights_from_pretrained()``, factories B/L/H.  The arithmetic of
``forward_backbone`` lives in pvpuformer_amd/engine.py (HIP kernels); this class never computes."""
import torch
import torch.nn as nn

from ..param_table import Container, register_tree, vit_shapes
from .pos_embed import interpolate_pos_embed


class PatchEmbed(Container):
    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, embed_dim=768, norm_layer=None,
                 flatten=True):
        super().__init__()
        self.in_chans, self.img_size, self.patch_size = in_chans, tuple(img_size), tuple(patch_size)
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        register_tree(self, {"proj.weight": (embed_dim, in_chans, patch_size[0], patch_size[1]),
                             "proj.bias": (embed_dim,)})


class VisionTransformer(Container):
    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, num_classes=1000, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True, pos_drop_rate=0., attn_drop_rate=0.,
                 proj_drop_rate=0., norm_layer=None, act_layer=None, cls_feature_dim=None, global_pool=False):
        super().__init__()
        assert qkv_bias and in_chans == 3 and cls_feature_dim is None
        assert pos_drop_rate == 0. and attn_drop_rate == 0. and proj_drop_rate == 0., "VPU configs use no dropout here"
        self.num_classes, self.embed_dim, self.num_features = num_classes, embed_dim, embed_dim
        self.depth, self.num_heads, self.mlp_ratio, self.global_pool = depth, num_heads, mlp_ratio, global_pool
        shapes = vit_shapes(embed_dim, depth, tuple(img_size), tuple(patch_size), mlp_ratio, num_classes)
        self.cls_token = nn.Parameter(torch.zeros(shapes.pop("cls_token")))
        self.pos_embed = nn.Parameter(torch.zeros(shapes.pop("pos_embed")))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        shapes.pop("patch_embed.proj.weight"); shapes.pop("patch_embed.proj.bias")
        register_tree(self, shapes)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def init_weights_from_pretrained(self, pretrained_path):
        """MAE checkpoint import (models_vit.py:150-166): {'model'| 'state_dict'} -> interpolate pos_embed -> non-strict load."""
        if not pretrained_path:
            return None
        ckpt = torch.load(pretrained_path, map_location="cpu")
        sd = ckpt["model"] if "model" in ckpt else ckpt["state_dict"]
        interpolate_pos_embed(self, sd)
        with torch.no_grad():
            return self.load_state_dict(sd, strict=False)


def vit_base_patch16(**kw):
    return VisionTransformer(patch_size=(16, 16), embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, **kw)


def vit_large_patch16(**kw):
    return VisionTransformer(patch_size=(16, 16), embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True, **kw)


def vit_huge_patch14(**kw):
    return VisionTransformer(patch_size=(14, 14), embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4, qkv_bias=True, **kw)
