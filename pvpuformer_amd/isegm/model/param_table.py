"""Parameter table of VitMultiGaussianVector_ed_Model: every state-dict entry the reference registers
(isegm/model/is_vpu_model.py:140-186, models_vit.py:109-148, transformer.py:222-521, swin_transformer.py:666-721,
transformer_helper/decode_head.py:82), including the ones the VPU path never uses -- ``load_state_dict(strict=True)``
of released checkpoints needs all 349 keys for ViT-B -- plus the reference's initial distributions."""
import math

import torch
import torch.nn as nn


class Container(nn.Module):
    """Name-only module: holds parameters / children; integer-named children are indexable like nn.Sequential."""

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())


def vit_shapes(D, depth, img, patch, mlp_ratio, num_classes=1000):
    n_tok = (img[0] // patch[0]) * (img[1] // patch[1])
    hid = int(D * mlp_ratio)
    s = {"cls_token": (1, 1, D), "pos_embed": (1, n_tok + 1, D),
         "patch_embed.proj.weight": (D, 3, patch[0], patch[1]), "patch_embed.proj.bias": (D,)}
    for i in range(depth):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (D,); s[p + "norm1.bias"] = (D,)
        s[p + "norm2.weight"] = (D,); s[p + "norm2.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D); s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D); s[p + "attn.proj.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (hid, D); s[p + "mlp.fc1.bias"] = (hid,)
        s[p + "mlp.fc2.weight"] = (D, hid); s[p + "mlp.fc2.bias"] = (D,)
    s["fc_norm.weight"] = (D,); s["fc_norm.bias"] = (D,)
    s["head.weight"] = (num_classes, D); s["head.bias"] = (num_classes,)
    return s


def neck_shapes(D, out_dims, img):
    o = out_dims
    s = {"ffn_layer.lin1.weight": (2048, 2 * img[0] + 3), "ffn_layer.lin1.bias": (2048,),
         "ffn_layer.lin2.weight": (D, 2048), "ffn_layer.lin2.bias": (D,)}

    def attn(prefix, internal):
        for nm in ("q_proj", "k_proj", "v_proj"):
            s[f"{prefix}.{nm}.weight"] = (internal, D); s[f"{prefix}.{nm}.bias"] = (internal,)
        s[f"{prefix}.out_proj.weight"] = (D, internal); s[f"{prefix}.out_proj.bias"] = (D,)
    for l in range(3):
        p = f"att.layers.{l}"
        attn(p + ".self_attn", D)
        s[p + ".norm1.weight"] = (D,); s[p + ".norm1.bias"] = (D,)
        attn(p + ".cross_attn_token_to_image", D // 2)
        s[p + ".norm2.weight"] = (D,); s[p + ".norm2.bias"] = (D,)
        s[p + ".mlp.lin1.weight"] = (1024, D); s[p + ".mlp.lin1.bias"] = (1024,)
        s[p + ".mlp.lin2.weight"] = (D, 1024); s[p + ".mlp.lin2.bias"] = (D,)
        s[p + ".norm3.weight"] = (D,); s[p + ".norm3.bias"] = (D,)
        s[p + ".norm4.weight"] = (D,); s[p + ".norm4.bias"] = (D,)
        attn(p + ".cross_attn_image_to_token", D // 2)
    attn("att.final_attn_token_to_image", D // 2)
    s["att.norm_final_attn.weight"] = (D,); s["att.norm_final_attn.bias"] = (D,)
    c4 = max(o[0] * 2, D // 2)
    s["down_4.0.weight"] = (D, c4, 2, 2); s["down_4.0.bias"] = (c4,)
    s["down_4.1.weight"] = (c4,); s["down_4.1.bias"] = (c4,)
    s["down_4.3.weight"] = (c4, c4 // 2, 2, 2); s["down_4.3.bias"] = (c4 // 2,)
    s["down_4.4.weight"] = (c4 // 2,); s["down_4.4.bias"] = (c4 // 2,)
    s["down_4.5.weight"] = (o[0], c4 // 2, 1, 1); s["down_4.5.bias"] = (o[0],)
    s["down_4.6.weight"] = (o[0],); s["down_4.6.bias"] = (o[0],)
    c8 = max(o[1], D // 2)
    s["down_8.0.weight"] = (D, c8, 2, 2); s["down_8.0.bias"] = (c8,)
    s["down_8.1.weight"] = (c8,); s["down_8.1.bias"] = (c8,)
    s["down_8.2.weight"] = (o[1], c8, 1, 1); s["down_8.2.bias"] = (o[1],)
    s["down_8.3.weight"] = (o[1],); s["down_8.3.bias"] = (o[1],)
    s["down_16.0.weight"] = (o[2], D, 1, 1); s["down_16.0.bias"] = (o[2],)
    s["down_16.1.weight"] = (o[2],); s["down_16.1.bias"] = (o[2],)
    c32 = max(o[3], D * 2)
    s["down_32.0.weight"] = (c32, D, 2, 2); s["down_32.0.bias"] = (c32,)
    s["down_32.1.weight"] = (c32,); s["down_32.1.bias"] = (c32,)
    s["down_32.2.weight"] = (o[3], c32, 1, 1); s["down_32.2.bias"] = (o[3],)
    s["down_32.3.weight"] = (o[3],); s["down_32.3.bias"] = (o[3],)
    return s


def head_shapes(in_channels, channels, num_classes, d_model):
    C = channels
    s = {"logit_scale": (), "conv_seg.weight": (num_classes, C, 1, 1), "conv_seg.bias": (num_classes,)}
    for i, ci in enumerate(in_channels):
        s[f"convs.{i}.conv.weight"] = (C, ci, 1, 1); s[f"convs.{i}.conv.bias"] = (C,)
    s["fusion_conv.conv.weight"] = (C, C * len(in_channels), 1, 1); s["fusion_conv.conv.bias"] = (C,)
    for name, cin in (("up_conv1", C), ("up_conv2", C // 2)):
        s[f"{name}.0.weight"] = (cin, cin // 2, 2, 2); s[f"{name}.0.bias"] = (cin // 2,)
        s[f"{name}.1.weight"] = (cin // 2,); s[f"{name}.1.bias"] = (cin // 2,)
        s[f"{name}.2.weight"] = (cin // 2, cin // 2, 1, 1); s[f"{name}.2.bias"] = (cin // 2,)
        s[f"{name}.3.weight"] = (cin // 2,); s[f"{name}.3.bias"] = (cin // 2,)
    s["ffn_layer.lin1.weight"] = (2 * d_model, d_model); s["ffn_layer.lin1.bias"] = (2 * d_model,)
    s["ffn_layer.lin2.weight"] = (C, 2 * d_model); s["ffn_layer.lin2.bias"] = (C,)
    return s


def register_tree(root, shapes, buffers=()):
    """Creates nested Containers along each dotted name and registers a zero Parameter (or buffer) at the leaf, in table
    order -- which is the reference's registration order, so state_dict() key order matches too."""
    for name, shape in shapes.items():
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, Container())
            mod = mod._modules[p]
        t = torch.zeros(shape)
        if name in buffers:
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(t))


def _kaiming_uniform_default(w, b):
    """torch's default reset_parameters of nn.Linear / nn.Conv*: kaiming_uniform(a=sqrt(5)), bias U(+-1/sqrt(fan_in))."""
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    if b is not None:
        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(w)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(b, -bound, bound)


@torch.no_grad()
def init_like_reference(model):
    """Initial distributions of the reference at construction: the ViT follows models_vit.py:168-188 (xavier-uniform
    Linears, zero biases, unit LayerNorms, N(0, .02) cls/pos tokens, patch-embed xavier on the flattened kernel);
    everything else keeps torch's module defaults (the reference's SimpleFPN.init_weights is a no-op and mmcv's
    init_cfg is never applied, is_vpu_model.py:90-91)."""
    sd = dict(model.named_parameters())
    for name, p in sd.items():
        leaf = name.split(".")[-1]
        is_norm = p.dim() == 1 and leaf == "weight"
        if name.startswith("backbone."):
            if name in ("backbone.cls_token", "backbone.pos_embed"):
                nn.init.normal_(p, std=.02)
            elif name == "backbone.patch_embed.proj.weight":
                nn.init.xavier_uniform_(p.view(p.shape[0], -1))
            elif name == "backbone.patch_embed.proj.bias":
                fan_in = sd["backbone.patch_embed.proj.weight"][0].numel()
                nn.init.uniform_(p, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            elif is_norm:
                nn.init.ones_(p)
            elif leaf == "bias":
                nn.init.zeros_(p)
            else:
                nn.init.xavier_uniform_(p)
        elif name == "head.logit_scale":
            p.fill_(math.log(1 / 0.07))
        elif name.startswith("point_embeddings") or name.startswith("not_a_point_embed"):
            nn.init.normal_(p)
        elif is_norm:
            nn.init.ones_(p)
        elif leaf == "bias" and name[:-4] + "weight" in sd and sd[name[:-4] + "weight"].dim() == 1:
            nn.init.zeros_(p)
        elif leaf == "weight":
            _kaiming_uniform_default(p, sd.get(name[:-6] + "bias"))
    model.pe_layer.positional_encoding_gaussian_matrix.copy_(torch.randn(2, model.embed_dim // 2))
