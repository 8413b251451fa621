"""CPU oracle for the VPUFormer forward/backward hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker / the CPU
baseline.  The product path (``pvpuformer_amd``) never imports it and has no CPU fallback.

It is a from-scratch restatement (numpy for the integer / byte bookkeeping, functional
torch-CPU fp32 for the floating-point path) of the reference algorithm; every function cites the
reference lines it follows (paths relative to /root/reference).  The reference publishes no tests
or golden vectors, so the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF,
generated in the build container by ``oracle/make_golden.py`` (which imports the reference) and
committed under ``tests/golden/``;  ``tests/test_oracle_golden.py`` checks it against them.

Parity status: PINNED for a1,a2,a4-a8,a10-a15.  UNPINNED for the OpenCV
rasterisers (a3: cv2.rectangle / cv2.polylines, thickness 3) -- cv2 is not installable here.

State-dict keys are the reference's (``backbone.blocks.0.attn.qkv.weight`` ...).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# deterministic, platform-independent pseudo-random numbers (shared by fixtures, tests, bench)
# ----------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_uniform(n, seed):
    """n floats in [-1, 1): splitmix64 of (index, seed); exact integer arithmetic, so the same on
    every machine.  float32."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)
             + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 bits -> exact in fp32
    return (u * 2.0 - 1.0).astype(np.float32)


def _name_seed(name):
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def synth_state_dict(shapes, seed=0):
    """Deterministic weights for a {name: shape} table (same values on every machine).
    Matrices: uniform(+-sqrt(6/(fan_in+fan_out))) as the reference's xavier init
    (isegm/model/modeling/models_vit.py:168-188); norm weights 1+-0.1; biases / vectors +-0.05;
    pos_embed +-0.04."""
    sd = {}
    for name, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        u = hash_uniform(n, _name_seed(name) + 7919 * seed)
        leaf = name.split(".")[-1]
        if len(shape) >= 2 and "pos_embed" not in name and "cls_token" not in name:
            fan_out = shape[0]
            fan_in = int(np.prod(shape[1:]))
            if "down_" in name and len(shape) == 4 and _is_convT(name):
                fan_in, fan_out = shape[0] * shape[2] * shape[3], shape[1]
            a = math.sqrt(6.0 / (fan_in + fan_out))
            v = u * np.float32(a)
        elif leaf == "weight" and len(shape) == 1:
            v = np.float32(1.0) + u * np.float32(0.1)
        elif "pos_embed" in name:
            v = u * np.float32(0.04)
        elif len(shape) == 0:
            v = np.float32(2.6593) + u * np.float32(0.0)
        else:
            v = u * np.float32(0.05)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v.reshape(shape)))
    return sd


def _is_convT(name):
    return name.startswith("neck.down_4.0.") or name.startswith("neck.down_4.3.") \
        or name.startswith("neck.down_8.0.") or ".up_conv" in name and name.split(".")[-2] == "0"


# ----------------------------------------------------------------------------------------------
# model configuration + parameter table
# ----------------------------------------------------------------------------------------------
def make_cfg(embed_dim=768, depth=12, num_heads=12, img=448, patch=16, mlp_ratio=4,
             out_dims=(128, 256, 512, 1024), head_channels=256, num_max_points=24,
             head_d_model=None):
    """ViT-B/448 defaults = models/iSegNet/vpu_base448_cocolvis.py:13-56.  ``head_d_model`` is the
    reference's hard-coded 768 (swin_transformer.py:668); for embed_dim != 768 the golden script
    patches the reference head to ``embed_dim``."""
    return dict(embed_dim=embed_dim, depth=depth, num_heads=num_heads, img=img, patch=patch,
                mlp_ratio=mlp_ratio, out_dims=tuple(out_dims), head_channels=head_channels,
                num_max_points=num_max_points,
                head_d_model=head_d_model if head_d_model is not None else embed_dim)


def param_shapes(cfg):
    """All state-dict entries of VitMultiGaussianVector_ed_Model (is_vpu_model.py:140-186)
    including the never-used ones, in the reference's registration order."""
    D, P, img = cfg["embed_dim"], cfg["patch"], cfg["img"]
    n_tok = (img // P) ** 2
    hid = D * cfg["mlp_ratio"]
    o = cfg["out_dims"]
    C = cfg["head_channels"]
    s = {}
    s["patch_embed_coords.proj.weight"] = (D, 3, P, P)
    s["patch_embed_coords.proj.bias"] = (D,)
    s["backbone.cls_token"] = (1, 1, D)
    s["backbone.pos_embed"] = (1, n_tok + 1, D)
    s["backbone.patch_embed.proj.weight"] = (D, 3, P, P)
    s["backbone.patch_embed.proj.bias"] = (D,)
    for i in range(cfg["depth"]):
        p = f"backbone.blocks.{i}."
        s[p + "norm1.weight"] = (D,); s[p + "norm1.bias"] = (D,)
        s[p + "norm2.weight"] = (D,); s[p + "norm2.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D); s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D); s[p + "attn.proj.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (hid, D); s[p + "mlp.fc1.bias"] = (hid,)
        s[p + "mlp.fc2.weight"] = (D, hid); s[p + "mlp.fc2.bias"] = (D,)
    s["backbone.fc_norm.weight"] = (D,); s["backbone.fc_norm.bias"] = (D,)
    s["backbone.head.weight"] = (1000, D); s["backbone.head.bias"] = (1000,)
    # neck (is_vpu_model.py:18-91): hide_dim = 1024 hard-coded
    s["neck.ffn_layer.lin1.weight"] = (2048, 2 * img + 3); s["neck.ffn_layer.lin1.bias"] = (2048,)
    s["neck.ffn_layer.lin2.weight"] = (D, 2048); s["neck.ffn_layer.lin2.bias"] = (D,)

    def attn(prefix, internal):
        for nm in ("q_proj", "k_proj", "v_proj"):
            s[f"{prefix}.{nm}.weight"] = (internal, D); s[f"{prefix}.{nm}.bias"] = (internal,)
        s[f"{prefix}.out_proj.weight"] = (D, internal); s[f"{prefix}.out_proj.bias"] = (D,)
    for l in range(3):
        p = f"neck.att.layers.{l}"
        attn(p + ".self_attn", D)
        s[p + ".norm1.weight"] = (D,); s[p + ".norm1.bias"] = (D,)
        attn(p + ".cross_attn_token_to_image", D // 2)
        s[p + ".norm2.weight"] = (D,); s[p + ".norm2.bias"] = (D,)
        s[p + ".mlp.lin1.weight"] = (1024, D); s[p + ".mlp.lin1.bias"] = (1024,)
        s[p + ".mlp.lin2.weight"] = (D, 1024); s[p + ".mlp.lin2.bias"] = (D,)
        s[p + ".norm3.weight"] = (D,); s[p + ".norm3.bias"] = (D,)
        s[p + ".norm4.weight"] = (D,); s[p + ".norm4.bias"] = (D,)
        attn(p + ".cross_attn_image_to_token", D // 2)
    attn("neck.att.final_attn_token_to_image", D // 2)
    s["neck.att.norm_final_attn.weight"] = (D,); s["neck.att.norm_final_attn.bias"] = (D,)
    c4 = max(o[0] * 2, D // 2)
    s["neck.down_4.0.weight"] = (D, c4, 2, 2); s["neck.down_4.0.bias"] = (c4,)
    s["neck.down_4.1.weight"] = (c4,); s["neck.down_4.1.bias"] = (c4,)
    s["neck.down_4.3.weight"] = (c4, c4 // 2, 2, 2); s["neck.down_4.3.bias"] = (c4 // 2,)
    s["neck.down_4.4.weight"] = (c4 // 2,); s["neck.down_4.4.bias"] = (c4 // 2,)
    s["neck.down_4.5.weight"] = (o[0], c4 // 2, 1, 1); s["neck.down_4.5.bias"] = (o[0],)
    s["neck.down_4.6.weight"] = (o[0],); s["neck.down_4.6.bias"] = (o[0],)
    c8 = max(o[1], D // 2)
    s["neck.down_8.0.weight"] = (D, c8, 2, 2); s["neck.down_8.0.bias"] = (c8,)
    s["neck.down_8.1.weight"] = (c8,); s["neck.down_8.1.bias"] = (c8,)
    s["neck.down_8.2.weight"] = (o[1], c8, 1, 1); s["neck.down_8.2.bias"] = (o[1],)
    s["neck.down_8.3.weight"] = (o[1],); s["neck.down_8.3.bias"] = (o[1],)
    s["neck.down_16.0.weight"] = (o[2], D, 1, 1); s["neck.down_16.0.bias"] = (o[2],)
    s["neck.down_16.1.weight"] = (o[2],); s["neck.down_16.1.bias"] = (o[2],)
    c32 = max(o[3], D * 2)
    s["neck.down_32.0.weight"] = (c32, D, 2, 2); s["neck.down_32.0.bias"] = (c32,)
    s["neck.down_32.1.weight"] = (c32,); s["neck.down_32.1.bias"] = (c32,)
    s["neck.down_32.2.weight"] = (o[3], c32, 1, 1); s["neck.down_32.2.bias"] = (o[3],)
    s["neck.down_32.3.weight"] = (o[3],); s["neck.down_32.3.bias"] = (o[3],)
    # head (swin_transformer.py:666-721, decode_head.py:82)
    s["head.logit_scale"] = ()
    s["head.conv_seg.weight"] = (1, C, 1, 1); s["head.conv_seg.bias"] = (1,)
    for i in range(4):
        s[f"head.convs.{i}.conv.weight"] = (C, o[i], 1, 1); s[f"head.convs.{i}.conv.bias"] = (C,)
    s["head.fusion_conv.conv.weight"] = (C, 4 * C, 1, 1); s["head.fusion_conv.conv.bias"] = (C,)
    s["head.up_conv1.0.weight"] = (C, C // 2, 2, 2); s["head.up_conv1.0.bias"] = (C // 2,)
    s["head.up_conv1.1.weight"] = (C // 2,); s["head.up_conv1.1.bias"] = (C // 2,)
    s["head.up_conv1.2.weight"] = (C // 2, C // 2, 1, 1); s["head.up_conv1.2.bias"] = (C // 2,)
    s["head.up_conv1.3.weight"] = (C // 2,); s["head.up_conv1.3.bias"] = (C // 2,)
    s["head.up_conv2.0.weight"] = (C // 2, C // 4, 2, 2); s["head.up_conv2.0.bias"] = (C // 4,)
    s["head.up_conv2.1.weight"] = (C // 4,); s["head.up_conv2.1.bias"] = (C // 4,)
    s["head.up_conv2.2.weight"] = (C // 4, C // 4, 1, 1); s["head.up_conv2.2.bias"] = (C // 4,)
    s["head.up_conv2.3.weight"] = (C // 4,); s["head.up_conv2.3.bias"] = (C // 4,)
    dm = cfg["head_d_model"]
    s["head.ffn_layer.lin1.weight"] = (2 * dm, dm); s["head.ffn_layer.lin1.bias"] = (2 * dm,)
    s["head.ffn_layer.lin2.weight"] = (C, 2 * dm); s["head.ffn_layer.lin2.bias"] = (C,)
    s["pe_layer.positional_encoding_gaussian_matrix"] = (2, D // 2)
    for i in range(4):
        s[f"point_embeddings.{i}.weight"] = (1, D)
    s["not_a_point_embed.weight"] = (1, D)
    s["head_aux.weight"] = (1, 128, 1, 1); s["head_aux.bias"] = (1,)
    return s


# Tensors that never receive a gradient on the VPU path.
def unused_param_names(cfg):
    names = ["backbone.cls_token", "backbone.fc_norm.weight", "backbone.fc_norm.bias",
             "backbone.head.weight", "backbone.head.bias", "head.logit_scale",
             "not_a_point_embed.weight", "head_aux.weight", "head_aux.bias"]
    for u in ("up_conv1", "up_conv2"):
        for i in range(4):
            names += [f"head.{u}.{i}.weight", f"head.{u}.{i}.bias"]
    names += [f"point_embeddings.{i}.weight" for i in range(4)]
    return names


# ----------------------------------------------------------------------------------------------
# a7/a8: Prompt-unified Encoder (PuE) Gaussian vectors -- integer bookkeeping, numpy
# ----------------------------------------------------------------------------------------------
def click_lut(sigma=3):
    """19-tap clip exp(-(d^2)/(2 sigma^2)) in float32 with the peak raised by 1
    (isegm/model/ops.py:51-61)."""
    r = int(sigma * 3)
    t = np.arange(0, 2 * r + 1, 1, np.float32)
    lut = np.exp(-((t - (2 * r + 1) // 2) ** 2) / (2 * sigma ** 2))
    lut[r] += 1
    return lut  # float32


def _in_img(x, y, w, h):
    return not (x < 0 or x > w or y < 0 or y > h)  # ops.py:63-67 (inclusive upper bound)


def _gauss_pair(x, y, rx, ry, lut_x, lut_y, size):
    """Two 1-D vectors with the reference's clipping rule (ops.py:80-104 / 170-201)."""
    vx = np.zeros(size, np.float64)
    vy = np.zeros(size, np.float64)
    ulx, uly, brx, bry = x - rx, y - ry, x + rx + 1, y + ry + 1
    if (not _in_img(ulx, uly, size, size)) and (not _in_img(brx, bry, size, size)):
        return vx, vy  # "corner-drop" quirk: both corners outside in EITHER coordinate
    for j in range(max(0, ulx), min(size, brx)):
        vx[j] = lut_x[j - ulx]
    for j in range(max(0, uly), min(size, bry)):
        vy[j] = lut_y[j - uly]
    return vx, vy


def pue_click(points, num_max_points=24, img=448):
    """_guassinvector_click (is_vpu_model.py:189-230).  points [B,2n,3] (row, col, order).
    Returns float64 [B, 2*num_max_points, 2*img+3].  Note the reference feeds (row, col) as
    (x, y): the first img entries encode points[...,0]."""
    pts = np.asarray(points, dtype=np.float32)
    B, N, _ = pts.shape
    n = N // 2
    E = 2 * img + 3
    lut = click_lut()
    rows = np.zeros((B, N, E), np.float64)
    for b in range(B):
        for i in range(N):
            if pts[b, i, 2] == -1:
                rows[b, i, E - 1] = 1.0
                continue
            xy = (pts[b, i, :2] * 4 / 4).astype("int32")  # truncation toward zero (ops.py:81)
            vx, vy = _gauss_pair(int(xy[0]), int(xy[1]), 9, 9, lut, lut, img)
            rows[b, i, :img] = vx
            rows[b, i, img:2 * img] = vy
            rows[b, i, 2 * img + (0 if i < n else 1)] = 1.0
    return _pad_slots(rows, n, num_max_points, E)


def _pad_slots(rows, n, num_max_points, E):
    if n == num_max_points:
        return rows
    B = rows.shape[0]
    nap = np.zeros((B, num_max_points - n, E), np.float64)
    nap[:, :, E - 1] = 1.0
    return np.concatenate([rows[:, :n], nap, rows[:, n:], nap], axis=1)  # is_vpu_model.py:218-228


def box_vectors(cx, cy, w, h, img=448):
    """GaussianVector_box.gen_guassian_vector (ops.py:138-202): kernel w//2*2-1, sigma=radius//3
    (integer), float32 exp, no peak raise; all-zero if a sigma is 0 or the box is all-zero."""
    zx, zy = np.zeros(img, np.float64), np.zeros(img, np.float64)
    if cx + cy + w + h == 0:
        return zx, zy
    luts, rads = [], []
    for L in (w, h):
        k = L // 2 * 2 - 1
        r = (k - 1) // 2
        s = r // 3
        if s == 0:
            return zx, zy
        t = np.arange(0, k, 1, np.float32)
        luts.append(np.exp(-((t - k // 2) ** 2) / (2 * s ** 2)))
        rads.append(r)
    return _gauss_pair(int(cx), int(cy), rads[0], rads[1], luts[0], luts[1], img)


def pue_box(points, boxes, num_max_points=24, img=448):
    """_guassinvector_box (is_vpu_model.py:233-291).  boxes [B,5] int32 (xc, yc, w, h, slot)."""
    pts = np.asarray(points, dtype=np.float32)
    bx = np.asarray(boxes).astype(np.int64)
    B, N, _ = pts.shape
    n = N // 2
    E = 2 * img + 3
    rows = pue_click(pts, n, img)  # un-padded [B, N, E]
    for b in range(B):
        vx, vy = box_vectors(int(bx[b, 0]), int(bx[b, 1]), int(bx[b, 2]), int(bx[b, 3]), img)
        slot = int(bx[b, 4])
        rows[b, slot, :] = 0.0
        rows[b, slot, :img] = vx
        rows[b, slot, img:2 * img] = vy
        rows[b, slot, 2 * img + (0 if slot < n else 1)] = 1.0
    return _pad_slots(rows, n, num_max_points, E)


# ----------------------------------------------------------------------------------------------
# a2/a3: click disk maps + box outline
# ----------------------------------------------------------------------------------------------
def scribble_vectors(scribble, rect, rng, img=448):
    """GaussianVector_scribble.gen_guassian_vector (ops.py:244-296), sigma 3, scale 1.  ``scribble`` int [P,2] (x, y),
    ``rect`` (x0, y0, w0, h0); ``rng`` is a ``random.Random`` (the reference draws from the global ``random`` module).
    Quirks kept: the drawn index addresses the scribble array itself, not the list of matching points (ops.py:272-275,
    :288-290); a chosen point is removed (all its duplicates) in the x pass only."""
    vx, vy = np.zeros(img, np.float64), np.zeros(img, np.float64)
    scribble = np.asarray(scribble).astype(np.int32)
    rect = np.asarray(rect)
    if np.sum(scribble) + np.sum(rect) == 0:
        return vx, vy
    scribble = (scribble * 4 / 4).astype("int32")
    sigma = 3
    x0, y0, w0, h0 = (int(min(int(v), img)) for v in rect)
    w_box, h_box = x0 - w0 // 2, y0 - h0 // 2
    for xi in range(w0):
        idx = np.argwhere(scribble[:, 0] == xi)
        if len(idx) != 0:
            point = scribble[rng.randint(0, len(idx) - 1)]
            xs, hs = int(point[0]), int(point[1])
            vx[xi] = np.exp(-((hs - h_box) ** 2) / (2 * sigma ** 2))
            scribble = np.delete(scribble, np.argwhere((scribble[:, 0] == xs) & (scribble[:, 1] == hs)), axis=0)
    for yj in range(h0):
        idx = np.argwhere(scribble[:, 1] == yj)
        if len(idx) != 0:
            point = scribble[rng.randint(0, len(idx) - 1)]
            vy[yj] = np.exp(-((int(point[0]) - w_box) ** 2) / (2 * sigma ** 2))
    return vx, vy


def pue_scribble(points, scribbles, rects, rng, num_max_points=24, img=448):
    """_guassinvector_scribble (is_vpu_model.py:294-352): click rows, then the LAST valid positive row of each sample is
    overwritten by the scribble vector of that sample (label one-hot 0).  scribbles int [B,1,P,2], rects int [B,1,4].
    a9 is never reached by the shipped trainer / evaluator; CPU restatement + seeded golden only."""
    pts = np.asarray(points, dtype=np.float32)
    B, N, _ = pts.shape
    n = N // 2
    E = 2 * img + 3
    rows = pue_click(pts, n, img)  # un-padded [B, N, E]
    vecs = [scribble_vectors(scribbles[b][0], rects[b][0], rng, img) for b in range(B)]   # all samples first, as the reference
    for b in range(B):
        valid = np.nonzero(pts[b, :n, 2] != -1)[0]
        if len(valid):
            i = int(valid[-1])
            rows[b, i, :] = 0.0
            rows[b, i, :img], rows[b, i, img:2 * img] = vecs[b]
            rows[b, i, 2 * img] = 1.0
    return _pad_slots(rows, n, num_max_points, E)


def disk_maps(points, H, W, radius=5):
    """DistMaps.get_coord_features torch path, use_disks=True, spatial_scale=1 (ops.py:347-379).
    fp32 arithmetic with separately rounded sub/mul/add.  Returns float32 [B,2,H,W] in {0,1}."""
    pts = np.asarray(points, dtype=np.float32)
    B, N, _ = pts.shape
    n = N // 2
    rr = np.arange(H, dtype=np.float32)[:, None]
    cc = np.arange(W, dtype=np.float32)[None, :]
    out = np.zeros((B, 2, H, W), np.float32)
    thr = np.float32(radius * radius)
    for b in range(B):
        for g in range(2):
            best = np.full((H, W), np.float32(1e6), np.float32)
            for i in range(g * n, (g + 1) * n):
                pr, pc = pts[b, i, 0], pts[b, i, 1]
                if max(pr, pc) < 0:
                    continue
                dr = rr - pr
                dc = cc - pc
                d = (dr * dr).astype(np.float32) + (dc * dc).astype(np.float32)
                best = np.minimum(best, d.astype(np.float32))
            out[b, g] = (best <= thr).astype(np.float32)
    return out


# ---- OpenCV's thick-line drawing, restated (a3) ------------------------------------------------------------------------------
# ISModel.draw_box / draw_scribble (is_model.py:97-146) call cv2.rectangle(..., 3) and cv2.polylines(..., False, ..., 3) on a
# uint8 single-channel canvas with the default LINE_8 / shift 0.  OpenCV 4.7.0 (requirements.txt:88) is not installable here,
# so the functions below RESTATE its algorithm from modules/imgproc/src/drawing.cpp, function for function: rectangle ->
# PolyLine(closed) ; polylines -> PolyLine ; PolyLine -> ThickLine per segment (end-cap flags 3 for the first segment of an
# open poly-line, 2 for every other one) ; ThickLine (thickness > 1) = FillConvexPoly of the quadrilateral p +- dp, dp = the
# perpendicular of length (thickness * 2^15 + odd * 2^15) / |p1 - p0| in 16.16 fixed point (cvRound of doubles), plus a filled
# Circle of radius (thickness * 2^15 + 2^15) >> 16 at the flagged ends ; FillConvexPoly = the outline through Line2 (16.16
# DDA, end point first) plus the two-edge scan conversion with its +2^15 rounding ; clipLine as Line2 uses it.
# PARITY UNPINNED against real cv2 (no fixture of the reference's holds a drawn outline): restated from the algorithm as
# published, integer for integer; what IS pinned is HIP == these functions bit for bit, and every downstream tensor of the
# box / scribble modes through them (tests/golden/tiny*.npz, vitl.npz).
_XY_SHIFT = 16
_XY_ONE = 1 << _XY_SHIFT


def _cdiv(a, b):
    """C++ int64 division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _cv_round(x):
    """cvRound(double): round half to even (lrint under the default rounding mode)."""
    return int(np.rint(np.float64(x)))


def _cv_clip_line(width, height, p1, p2):
    """clipLine(Size2l, Point2l&, Point2l&) (drawing.cpp): Cohen-Sutherland with the intersections through doubles truncated
    to int64.  Returns (visible, p1, p2)."""
    x1, y1 = p1
    x2, y2 = p2
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, p1, p2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(np.float64(a - y1) * np.float64(x2 - x1) / np.float64(y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(np.float64(a - y2) * np.float64(x2 - x1) / np.float64(y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(np.float64(a - x1) * np.float64(y2 - y1) / np.float64(x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(np.float64(a - x2) * np.float64(y2 - y1) / np.float64(x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def _cv_line2(img, p1, p2):
    """Line2 (drawing.cpp): a line between two 16.16 points as a fixed-point DDA along the major axis, the rounded END point
    first.  img: bool [H, W]."""
    H, W = img.shape
    ok, (x1, y1), (x2, y2) = _cv_clip_line(W << _XY_SHIFT, H << _XY_SHIFT, p1, p2)
    if not ok:
        return

    def put(x, y):
        if 0 <= x < W and 0 <= y < H:
            img[y, x] = True
    dx, dy = x2 - x1, y2 - y1
    ax, ay = abs(dx), abs(dy)
    if ax > ay:
        if dx < 0:
            x1, x2, y1, y2, dy = x2, x1, y2, y1, -dy
        y_step = _cdiv(dy << _XY_SHIFT, ax | 1)
        ecount = (x2 - x1) >> _XY_SHIFT
    else:
        if dy < 0:
            x1, x2, y1, y2, dx = x2, x1, y2, y1, -dx
        x_step = _cdiv(dx << _XY_SHIFT, ay | 1)
        ecount = (y2 - y1) >> _XY_SHIFT
    x1 += _XY_ONE >> 1
    y1 += _XY_ONE >> 1
    put((x2 + (_XY_ONE >> 1)) >> _XY_SHIFT, (y2 + (_XY_ONE >> 1)) >> _XY_SHIFT)
    if ax > ay:
        x = x1 >> _XY_SHIFT
        for k in range(ecount + 1):
            put(x + k, (y1 + k * y_step) >> _XY_SHIFT)
    else:
        y = y1 >> _XY_SHIFT
        for k in range(ecount + 1):
            put((x1 + k * x_step) >> _XY_SHIFT, y + k)


def _cv_fill_convex_poly(img, v):
    """FillConvexPoly(img, v, npts, color, LINE_8, shift = XY_SHIFT) (drawing.cpp): the outline edge by edge through Line2, then
    the scan conversion between a left-going and a right-going edge walker started at the top-most vertex.  v: 16.16 points."""
    H, W = img.shape
    npts, shift = len(v), _XY_SHIFT
    delta = (1 << shift) >> 1
    delta1 = delta2 = _XY_ONE >> 1
    xmin = xmax = v[0][0]
    ymin = ymax = v[0][1]
    imin = 0
    p0 = v[-1]
    for i, p in enumerate(v):
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax, xmax, xmin = max(ymax, p[1]), max(xmax, p[0]), min(xmin, p[0])
        _cv_line2(img, p0, p)
        p0 = p
    xmin, xmax = (xmin + delta) >> shift, (xmax + delta) >> shift
    ymin, ymax = (ymin + delta) >> shift, (ymax + delta) >> shift
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= W or ymin >= H:
        return
    ymax = min(ymax, H - 1)
    e_idx, e_di = [imin, imin], [1, npts - 1]
    e_ye, e_x, e_dx = [ymin, ymin], [-_XY_ONE, -_XY_ONE], [0, 0]
    edges, y = npts, ymin
    while True:
        for i in range(2):
            if y >= e_ye[i]:
                idx0, di = e_idx[i], e_di[i]
                idx = idx0 + di
                if idx >= npts:
                    idx -= npts
                while True:
                    edges -= 1
                    if not edges + 1 > 0:         # for (; edges-- > 0; )
                        break
                    ty = (v[idx][1] + delta) >> shift
                    if ty > y:
                        xs, xe = v[idx0][0], v[idx][0]
                        e_ye[i] = ty
                        e_dx[i] = _cdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e_x[i] = xs
                        e_idx[i] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= npts:
                        idx -= npts
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if e_x[0] > e_x[1] else (0, 1)
            xx1 = (e_x[left] + delta1) >> _XY_SHIFT
            xx2 = (e_x[right] + delta2) >> _XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                img[y, max(xx1, 0):min(xx2, W - 1) + 1] = True
        e_x[0] += e_dx[0]
        e_x[1] += e_dx[1]
        y += 1
        if not y <= ymax:
            break


def _cv_circle_fill(img, cx, cy, radius):
    """Circle(img, center, radius, color, fill = 1) (drawing.cpp): the midpoint circle's horizontal spans, clipped."""
    H, W = img.shape
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1

    def hline(y, xa, xb):
        if 0 <= y < H and xb >= 0 and xa < W:
            img[y, max(xa, 0):min(xb, W - 1) + 1] = True
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        if err > 0:             # mask = (err <= 0) - 1
            err -= minus
            dx -= 1
            minus -= 2


def _cv_thick_line(img, p0, p1, thickness, flags):
    """ThickLine(img, p0, p1, color, thickness > 1, LINE_8, flags, shift = 0) (drawing.cpp).  flags bit 0 / 1: end cap at p0 / p1."""
    assert thickness > 1
    p0 = (int(p0[0]) << _XY_SHIFT, int(p0[1]) << _XY_SHIFT)
    p1 = (int(p1[0]) << _XY_SHIFT, int(p1[1]) << _XY_SHIFT)
    inv = np.float64(1.0 / _XY_ONE)
    dx, dy = np.float64(p0[0] - p1[0]) * inv, np.float64(p1[1] - p0[1]) * inv
    r = dx * dx + dy * dy
    odd = thickness & 1
    th = thickness << (_XY_SHIFT - 1)
    if abs(r) > np.finfo(np.float64).eps:
        r = np.float64(th + odd * _XY_ONE * 0.5) / np.sqrt(r)
        dpx, dpy = _cv_round(dy * r), _cv_round(dx * r)
        _cv_fill_convex_poly(img, [(p0[0] + dpx, p0[1] + dpy), (p0[0] - dpx, p0[1] - dpy),
                                   (p1[0] - dpx, p1[1] - dpy), (p1[0] + dpx, p1[1] + dpy)])
    for i, p in enumerate((p0, p1)):
        if flags & (i + 1):
            _cv_circle_fill(img, (p[0] + (_XY_ONE >> 1)) >> _XY_SHIFT, (p[1] + (_XY_ONE >> 1)) >> _XY_SHIFT,
                            (th + (_XY_ONE >> 1)) >> _XY_SHIFT)


def cv_polyline_mask(H, W, pts, closed, thickness=3):
    """PolyLine(img, v, count, is_closed, color, thickness, LINE_8, 0) (drawing.cpp) on an empty [H, W] canvas -> bool mask of
    the pixels it sets.  pts: integer (x, y) pairs."""
    img = np.zeros((H, W), bool)
    pts = [(int(x), int(y)) for x, y in pts]
    if not pts:
        return img
    i = len(pts) - 1 if closed else 0
    flags = 2 + (0 if closed else 1)
    p0 = pts[i]
    for p in pts[(0 if closed else 1):]:
        _cv_thick_line(img, p0, p, thickness, flags)
        p0, flags = p, 2
    return img


def box_outline(canvas, box, n_points, thickness=3):
    """ISModel.draw_box (is_model.py:97-121): cv2.rectangle((x0, y0), (x1, y1), 255, 3) = the closed 4-segment poly-line
    (x0,y0) (x1,y0) (x1,y1) (x0,y1) OR-ed into channel 0 (slot < n) or 1, through the restated ThickLine above."""
    xc, yc, w, h, slot = [int(v) for v in box]
    ch = 0 if slot < n_points else 1
    x0, x1, y0, y1 = xc - w // 2, xc + w // 2, yc - h // 2, yc + h // 2
    H, W = canvas.shape[-2:]
    band = cv_polyline_mask(H, W, [(x0, y0), (x1, y0), (x1, y1), (x0, y1)], True, thickness)
    # the reference round-trips the channel through uint8 (x.astype(int)*255 // 255): identity on {0,1}
    canvas[ch] = np.where(band, np.float32(1.0), canvas[ch])
    return canvas


def polyline_raster(canvas, curve, thickness=3):
    """ISModel.draw_scribble (is_model.py:123-146): cv2.polylines(image, [curve], False, 255, 3) -- the open poly-line
    through the int32-truncated scribble points, OR-ed into channel 0 (always the positive channel) -- through the restated
    ThickLine above (a one-point curve draws nothing: PolyLine's loop starts at the second point)."""
    pts = np.asarray(curve)
    pts = np.column_stack((pts[:, 0].astype(np.int32), pts[:, 1].astype(np.int32))).astype(np.int64)
    H, W = canvas.shape[-2:]
    hit = cv_polyline_mask(H, W, pts, False, thickness)
    canvas[0] = np.where(hit, np.float32(1.0), canvas[0])
    return canvas


# weights of OpenCV's 5x5 chamfer mask for DIST_L2 (getDistanceTransformMask: a = 1, b = 1.4, c = 2.1969, float32) in the
# 16-bit fixed point of distanceTransform_5x5: CV_FLT_TO_FIX(x, 16) = cvRound(x * 65536)
_CH_A = int(np.rint(np.float32(1.0) * np.float32(65536.0)))        # 65536
_CH_B = int(np.rint(np.float32(1.4) * np.float32(65536.0)))        # 91750
_CH_C = int(np.rint(np.float32(2.1969) * np.float32(65536.0)))     # 143976


def chamfer_l2_5x5(mask):
    """cv2.distanceTransform(mask, cv2.DIST_L2, 5) of the training simulators (isegm/engine/trainer.py:628-629,673-674,
    736-737): the two-pass 5x5 chamfer approximation of the distance to the nearest zero pixel.  OpenCV 4.7.0 (pinned at
    requirements.txt:88) is not installable here, so this RESTATES its published algorithm (modules/imgproc/src/
    distransform.cpp, distanceTransform_5x5, the C++ path without IPP): integer distances in 16-bit fixed point, a forward
    raster pass over the upper half of the mask ((-2,+-1), (-1,+-2) at c; (-1,+-1) at b; (-1,0), (0,-1) at a), a backward
    pass over the mirrored half, everything outside the image at "infinity", result = float32(t) * 2^-16.  PARITY UNPINNED
    against real OpenCV (no cv2, no fixture of the reference's): what is pinned is HIP == this function, bit for bit.
    mask: [H, W] bool / uint8; returns float32 [H, W].  Row recurrences are prefix minima: t[j] = min(cand[j], t[j-1] + a)
    = a j + min_{k <= j} (cand[k] - a k)."""
    m = np.asarray(mask) != 0
    H, W = m.shape
    INF = np.int64(1) << 40
    a, b, c = np.int64(_CH_A), np.int64(_CH_B), np.int64(_CH_C)
    T = np.full((H + 4, W + 4), INF, np.int64)      # two border rows / columns of "infinity" on every side
    cols = np.arange(W, dtype=np.int64) * a

    def sh(row, d):                                  # row shifted by d columns (image columns live at [2, W + 2))
        return row[2 + d: 2 + d + W]
    for i in range(H):                               # forward pass
        r1, r2 = T[i + 1], T[i]                      # image rows i - 1 and i - 2
        base = np.minimum.reduce([sh(r2, -1) + c, sh(r2, 1) + c, sh(r1, -2) + c, sh(r1, 2) + c,
                                  sh(r1, -1) + b, sh(r1, 1) + b, sh(r1, 0) + a])
        cand = np.where(m[i], base, 0)
        T[i + 2, 2:W + 2] = np.minimum.accumulate(cand - cols) + cols
    for i in range(H - 1, -1, -1):                   # backward pass
        r1, r2 = T[i + 3], T[i + 4]                  # image rows i + 1 and i + 2
        base = np.minimum.reduce([sh(r2, -1) + c, sh(r2, 1) + c, sh(r1, -2) + c, sh(r1, 2) + c,
                                  sh(r1, -1) + b, sh(r1, 1) + b, sh(r1, 0) + a])
        cand = np.minimum(T[i + 2, 2:W + 2], base)
        rev = (cand + cols)[::-1]                    # t[j] = min(cand[j], t[j+1] + a) = -a j + min_{k >= j} (cand[k] + a k)
        T[i + 2, 2:W + 2] = np.minimum.accumulate(rev)[::-1] - cols
    t = np.minimum(T[2:H + 2, 2:W + 2], np.int64(0xFFFFFFFF) - c)       # DIST_MAX saturation (an all-foreground image)
    return t.astype(np.float32) * np.float32(1.0 / 65536.0)


def coord_features(prev_mask, points, boxes=None, prompt_type=0, radius=5, scribbles=None):
    """ISModel.get_coord_features_with_prompt (is_model.py:78-95): cat(prev_mask, disks).  ``scribbles`` (prompt type 2):
    array [B,1,P,2] of (x, y)."""
    B, _, H, W = prev_mask.shape
    d = disk_maps(points.detach().cpu().numpy(), H, W, radius)
    if prompt_type == 2:
        for b in range(B):
            d[b] = polyline_raster(d[b], np.asarray(scribbles)[b][0])
    if prompt_type == 1:
        n = points.shape[1] // 2
        bx = boxes.detach().cpu().numpy() if torch.is_tensor(boxes) else np.asarray(boxes)
        for b in range(B):
            d[b] = box_outline(d[b], bx[b], n)
    return torch.cat([prev_mask, torch.from_numpy(d).to(prev_mask.dtype)], dim=1)


# ----------------------------------------------------------------------------------------------
# a1, a4-a6: normalisation, patch embedding, windowed MAE-ViT
# ----------------------------------------------------------------------------------------------
_MEAN = (.485, .456, .406)
_STD = (.229, .224, .225)


def normalize_image(rgb):
    """BatchImageNormalize (ops.py:398-407)."""
    m = torch.tensor(_MEAN, dtype=rgb.dtype).view(1, 3, 1, 1)
    s = torch.tensor(_STD, dtype=rgb.dtype).view(1, 3, 1, 1)
    return (rgb - m) / s


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _patch_tokens(x, sd, p, P):
    y = F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], stride=P)  # models_vit.py:91,100
    return y.flatten(2).transpose(1, 2)


def _to_windows(x, g, wg):
    """patchify (models_vit.py:225-239): [B, g*g, C] -> [B*nw*nw, wg*wg, C]."""
    B, N, C = x.shape
    nw = g // wg
    return x.view(B, nw, wg, nw, wg, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nw * nw, wg * wg, C)


def _from_windows(x, g, wg):
    """unpatchify (models_vit.py:242-255)."""
    nw = g // wg
    B = x.shape[0] // (nw * nw)
    C = x.shape[-1]
    return x.view(B, nw, nw, wg, wg, C).permute(0, 1, 3, 2, 4, 5).reshape(B, g * g, C)


def _vit_block(x, sd, p, heads):
    """Block / Attention / Mlp (models_vit.py:9-75), LayerNorm eps 1e-6 (:126)."""
    B, N, C = x.shape
    hd = C // heads
    h = _ln(x, sd, p + "norm1", 1e-6)
    qkv = _lin(h, sd, p + "attn.qkv").view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax((qkv[0] @ qkv[1].transpose(-2, -1)) * (hd ** -0.5), dim=-1)
    o = (att @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + _lin(o, sd, p + "attn.proj")
    h = _ln(x, sd, p + "norm2", 1e-6)
    h = _lin(F.gelu(_lin(h, sd, p + "mlp.fc1")), sd, p + "mlp.fc2")
    return x + h


def vit_backbone(sd, cfg, rgb_norm, coord, taps=None):
    """VisionTransformer.forward_backbone (models_vit.py:257-287), shuffle=False."""
    P = cfg["patch"]
    g = cfg["img"] // P
    wg = 224 // P
    x = _patch_tokens(rgb_norm, sd, "backbone.patch_embed", P) \
        + _patch_tokens(coord, sd, "patch_embed_coords", P)          # is_vpu_model.py:385, :260
    x = x + sd["backbone.pos_embed"][:, 1:]
    if taps is not None:
        taps["tokens0"] = x
    depth = cfg["depth"]
    group = 6 if depth == 12 else depth // 4
    windowed = False
    for i in range(1, depth + 1):
        if i % group:
            if not windowed:
                x = _to_windows(x, g, wg)
                windowed = True
        else:
            x = _from_windows(x, g, wg)
            windowed = False
        x = _vit_block(x, sd, f"backbone.blocks.{i - 1}.", cfg["num_heads"])
        if taps is not None:
            taps[f"block{i}"] = _from_windows(x, g, wg) if windowed else x
    return x


# ----------------------------------------------------------------------------------------------
# a10/a11: DMA neck
# ----------------------------------------------------------------------------------------------
def pos2d(d_model, height, width):
    """TwoWayTransformer.pos2d (transformer.py:290-318) -> [1, H*W, d_model]."""
    pe = torch.zeros(d_model, height, width)
    half = d_model // 2
    div = torch.exp(torch.arange(0., half, 2) * -(math.log(10000.0) / half))
    pw = torch.arange(0., width).unsqueeze(1) * div     # [W, half/2]
    ph = torch.arange(0., height).unsqueeze(1) * div
    pe[0:half:2] = torch.sin(pw).t().unsqueeze(1).expand(-1, height, -1)
    pe[1:half:2] = torch.cos(pw).t().unsqueeze(1).expand(-1, height, -1)
    pe[half::2] = torch.sin(ph).t().unsqueeze(2).expand(-1, -1, width)
    pe[half + 1::2] = torch.cos(ph).t().unsqueeze(2).expand(-1, -1, width)
    return pe.reshape(d_model, height * width).t().unsqueeze(0)


def _mha(sd, p, q, k, v, heads=8):
    """transformer.py Attention.forward (:499-521)."""
    q, k, v = _lin(q, sd, p + ".q_proj"), _lin(k, sd, p + ".k_proj"), _lin(v, sd, p + ".v_proj")
    B, nq, ci = q.shape
    hd = ci // heads

    def split(t):
        return t.view(B, t.shape[1], heads, hd).transpose(1, 2)
    a = torch.softmax(split(q) @ split(k).transpose(-2, -1) / math.sqrt(hd), dim=-1)
    o = (a @ split(v)).transpose(1, 2).reshape(B, nq, ci)
    return _lin(o, sd, p + ".out_proj")


def dma_transformer(sd, queries0, keys0):
    """TwoWayTransformer.forward with return_intermediate (transformer.py:323-384) and
    TwoWayAttentionBlock.forward (:432-463).  LayerNorm eps 1e-5 (nn.LayerNorm default)."""
    B, n_img, C = keys0.shape
    side = int(math.sqrt(n_img))
    kpe = pos2d(C, side, side).to(keys0.dtype)
    qpe = queries0
    q, k = queries0, keys0
    outs = []
    for l in range(3):
        p = f"neck.att.layers.{l}"
        if l == 0:
            q = _mha(sd, p + ".self_attn", q, q, q)
        else:
            qq = q + qpe
            q = q + _mha(sd, p + ".self_attn", qq, qq, q)
        q = _ln(q, sd, p + ".norm1", 1e-5)
        q = _ln(q + _mha(sd, p + ".cross_attn_token_to_image", q + qpe, k + kpe, k), sd, p + ".norm2", 1e-5)
        m = _lin(F.relu(_lin(q, sd, p + ".mlp.lin1")), sd, p + ".mlp.lin2")
        q = _ln(q + m, sd, p + ".norm3", 1e-5)
        k = _ln(k + _mha(sd, p + ".cross_attn_image_to_token", k + kpe, q + qpe, q), sd, p + ".norm4", 1e-5)
        if l != 2:
            outs.append((q, k))
    q = _ln(q + _mha(sd, "neck.att.final_attn_token_to_image", q + qpe, k + kpe, k),
            sd, "neck.att.norm_final_attn", 1e-5)
    outs.append((q, k))
    return outs


def _gn(x, sd, p):
    return F.group_norm(x, 1, sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def neck_forward(sd, cfg, x, pue, taps=None):
    """SimpleFPN.forward (is_vpu_model.py:93-136)."""
    q = _lin(F.relu(_lin(pue.to(x.dtype), sd, "neck.ffn_layer.lin1")), sd, "neck.ffn_layer.lin2")
    hs = dma_transformer(sd, q, x)
    q_out = q + hs[0][0] + hs[1][0] + hs[2][0]
    B, N, C = x.shape
    g = int(math.sqrt(N))
    feats = [x]
    for qi, ki in hs:
        cg = qi.max(dim=1).values.sigmoid().unsqueeze(1)      # channel gate, max over queries
        sg = ki.max(dim=2).values.sigmoid().unsqueeze(2)      # spatial gate, max over channels
        feats.append(x + x * cg + x * sg)
    maps = [f.transpose(1, 2).reshape(B, C, g, g) for f in feats]
    if taps is not None:
        taps["q_ffn"] = q
        taps["q_out"] = q_out
        for i, (qi, ki) in enumerate(hs):
            taps[f"dma_q{i}"] = qi
            taps[f"dma_k{i}"] = ki
    d4 = F.conv_transpose2d(maps[0], sd["neck.down_4.0.weight"], sd["neck.down_4.0.bias"], stride=2)
    d4 = F.gelu(_gn(d4, sd, "neck.down_4.1"))
    d4 = F.conv_transpose2d(d4, sd["neck.down_4.3.weight"], sd["neck.down_4.3.bias"], stride=2)
    d4 = _gn(d4, sd, "neck.down_4.4")
    d4 = F.gelu(_gn(F.conv2d(d4, sd["neck.down_4.5.weight"], sd["neck.down_4.5.bias"]), sd, "neck.down_4.6"))
    d8 = F.conv_transpose2d(maps[1], sd["neck.down_8.0.weight"], sd["neck.down_8.0.bias"], stride=2)
    d8 = _gn(d8, sd, "neck.down_8.1")
    d8 = F.gelu(_gn(F.conv2d(d8, sd["neck.down_8.2.weight"], sd["neck.down_8.2.bias"]), sd, "neck.down_8.3"))
    d16 = F.gelu(_gn(F.conv2d(maps[2], sd["neck.down_16.0.weight"], sd["neck.down_16.0.bias"]), sd, "neck.down_16.1"))
    d32 = _gn(F.conv2d(maps[3], sd["neck.down_32.0.weight"], sd["neck.down_32.0.bias"], stride=2), sd, "neck.down_32.1")
    d32 = F.gelu(_gn(F.conv2d(d32, sd["neck.down_32.2.weight"], sd["neck.down_32.2.bias"]), sd, "neck.down_32.3"))
    return [d4, d8, d16, d32], q_out


# ----------------------------------------------------------------------------------------------
# a12/a13: segmentation head, P2CL logits, final upsample
# ----------------------------------------------------------------------------------------------
def head_forward(sd, cfg, feats, q_out, drop_mask=None, taps=None):
    """SwinTransfomerSegHead.forward_feat (swin_transformer.py:723-767), upsample='x1'.
    ``drop_mask`` [B,C,1,1] is the Dropout2d(0.1) keep-mask/(1-p) in train mode (decode_head.py:210-215);
    None = eval."""
    size = feats[0].shape[2:]
    outs = []
    for i, f in enumerate(feats):
        y = F.relu(F.conv2d(f, sd[f"head.convs.{i}.conv.weight"], sd[f"head.convs.{i}.conv.bias"]))
        outs.append(F.interpolate(y, size=size, mode="bilinear", align_corners=False))
    fused = F.relu(F.conv2d(torch.cat(outs, 1), sd["head.fusion_conv.conv.weight"],
                            sd["head.fusion_conv.conv.bias"]))
    query = _lin(F.relu(_lin(q_out, sd, "head.ffn_layer.lin1")), sd, "head.ffn_layer.lin2")
    emb = fused.flatten(2)
    seg_in = fused if drop_mask is None else fused * drop_mask
    seg = F.conv2d(seg_in, sd["head.conv_seg.weight"], sd["head.conv_seg.bias"])
    sim = (F.normalize(query, p=2, dim=2) @ F.normalize(emb, p=2, dim=1) + 1) / 2
    B, Nq, HW = sim.shape
    if taps is not None:
        taps["fused"] = fused
        taps["query"] = query
    return seg, sim.view(B, Nq, size[0], size[1])


def vpu_forward(sd, cfg, image4, points, boxes=None, prompt_type=0, drop_mask=None, taps=None,
                pue_override=None, scribbles=None, rng=None):
    """VitMultiGaussianVector_ed_Model.forward (is_vpu_model.py:422-438) with edloss=True.  Prompt type 2:
    ``scribbles`` = (array [B,1,P,2] of (x, y), rects [B,1,4] of (x_center, y_center, width, height)) and ``rng`` a
    ``random.Random`` standing for the reference's global ``random`` state at the call."""
    rgb = normalize_image(image4[:, :3])
    prev = image4[:, 3:]
    coord = coord_features(prev, points, boxes, prompt_type, scribbles=None if scribbles is None else scribbles[0])
    x = vit_backbone(sd, cfg, rgb, coord, taps)
    if taps is not None:
        taps["backbone"] = x
        taps["coord"] = coord
    if pue_override is not None:
        pue = pue_override
    elif prompt_type == 0:
        pue = pue_click(points.detach().cpu().numpy(), cfg["num_max_points"], cfg["img"])
    elif prompt_type == 2:
        pue = pue_scribble(points.detach().cpu().numpy(), np.asarray(scribbles[0]).astype(np.int32), np.asarray(scribbles[1]),
                           rng, cfg["num_max_points"], cfg["img"])
    else:
        pue = pue_box(points.detach().cpu().numpy(),
                      boxes.detach().cpu().numpy() if torch.is_tensor(boxes) else boxes,
                      cfg["num_max_points"], cfg["img"])
    pue = torch.as_tensor(pue)
    feats, q_out = neck_forward(sd, cfg, x, pue, taps)
    seg, sim = head_forward(sd, cfg, feats, q_out, drop_mask, taps)
    if taps is not None:
        taps["seg_lowres"] = seg
        taps["sim_lowres"] = sim
    H = image4.shape[2:]
    return {"instances": F.interpolate(seg, size=H, mode="bilinear", align_corners=True),
            "instances_aux": F.interpolate(sim, size=H, mode="bilinear", align_corners=True)}


# ----------------------------------------------------------------------------------------------
# a14/a15: losses
# ----------------------------------------------------------------------------------------------
def nfl_loss(logits, label, alpha=0.5, gamma=2, eps=1e-12):
    """NormalizedFocalLossSigmoid.forward (losses.py:39-87), max_mult=-1, detach_delimeter,
    size_average, ignore_label=-1, weight 1.  Returns [B]."""
    pos = label > 0.5
    w = (label != -1).to(logits.dtype)
    p = torch.sigmoid(logits)
    a = torch.where(pos, alpha * w, (1 - alpha) * w)
    pt = torch.where(w > 0, 1.0 - torch.abs(label - p), torch.ones_like(p))
    beta = (1 - pt) ** gamma
    mult = (w.sum(dim=(-2, -1), keepdim=True) / (beta.sum(dim=(-2, -1), keepdim=True) + eps)).detach()
    beta = beta * mult
    l = -a * beta * torch.log(torch.clamp_max(pt + eps, 1.0)) * w
    dims = tuple(range(1, l.dim()))
    return l.sum(dims) / (w.sum(dims) + eps)


def dice_loss_naive(logits, target, eps=1e-3):
    """DiceLoss(use_sigmoid, activate, naive_dice, reduction='mean') (losses.py:227-363) -> scalar."""
    p = torch.sigmoid(logits).flatten(1)
    t = target.flatten(1).float()
    d = (2 * (p * t).sum(1) + eps) / (p.sum(1) + t.sum(1) + eps)
    return (1 - d).mean()


def bce_from_sigmoid(prob, label):
    """SigmoidBinaryCrossEntropyLoss(from_sigmoid=True).forward (losses.py:163-176) -> [B]."""
    w = (label != -1).to(prob.dtype)
    y = torch.where(w > 0, label, torch.zeros_like(label))
    l = -(torch.log(prob + 1e-12) * y + torch.log(1. - prob + 1e-12) * (1. - y)) * w
    return l.mean(dim=tuple(range(1, l.dim())))


def ed_mask_label(gt, num_max_points=24):
    """trainer.py:329-331."""
    return torch.cat([gt.repeat(1, num_max_points, 1, 1),
                      torch.logical_not(gt).to(gt.dtype).repeat(1, num_max_points, 1, 1)], dim=1)


def step_loss(out, gt, ed_label, iter_weight=1.0):
    """ISTrainer.add_loss x4 for one click iteration (trainer.py:399-419, :533-554):
    NFL*1 + Dice*1 + P2CL(BCE)*2, each times the iteration weight."""
    l_nfl = nfl_loss(out["instances"], gt).mean()
    l_dice = dice_loss_naive(out["instances"], gt).mean()
    l_pcl = bce_from_sigmoid(out["instances_aux"], ed_label).mean()
    total = (1.0 * l_nfl + 1.0 * l_dice + 2.0 * l_pcl) * iter_weight
    return total, {"nfl": l_nfl, "dice": l_dice, "p2cl": l_pcl}


# ----------------------------------------------------------------------------------------------
# synthetic batches -- shared by tests and bench
# ----------------------------------------------------------------------------------------------
def synth_batch(B, img=448, seed=0, num_max_points=24, integer_clicks=True):
    """images U[0,1); gt = ellipse/rectangle union; points [B,48,3] with 1-3 positive clicks inside gt
    and 0-2 negative outside (row, col, order; -1 pad); boxes [B,5] from the gt bbox (cal_box rule,
    trainer.py:1113-1125, no jitter, slot = first free positive slot)."""
    rs = np.random.RandomState(seed)
    images = hash_uniform(B * 3 * img * img, 1000 + seed).reshape(B, 3, img, img) * 0.5 + 0.5
    yy, xx = np.mgrid[0:img, 0:img]
    gt = np.zeros((B, 1, img, img), np.float32)
    pts = -np.ones((B, 2 * num_max_points, 3), np.float32)
    boxes = np.zeros((B, 5), np.int32)
    for b in range(B):
        cy, cx = rs.randint(img // 4, 3 * img // 4, size=2)
        ry, rx = rs.randint(img // 10, img // 4, size=2)
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        y0, x0 = rs.randint(img // 8, img // 2, size=2)
        hh, ww = rs.randint(img // 10, img // 3, size=2)
        m |= (yy >= y0) & (yy < y0 + hh) & (xx >= x0) & (xx < x0 + ww)
        gt[b, 0] = m
        inside = np.argwhere(m)
        outside = np.argwhere(~m)
        kp, kn = rs.randint(1, 4), rs.randint(0, 3)
        order = 0
        for i in range(kp):
            r, c = inside[rs.randint(len(inside))]
            off = 0.0 if integer_clicks else rs.rand() * 0.9
            pts[b, i] = (r + off, c + off, order); order += 1
        for i in range(kn):
            r, c = outside[rs.randint(len(outside))]
            pts[b, num_max_points + i] = (r, c, order); order += 1
        ys, xs = inside[:, 0], inside[:, 1]
        bx0, bx1, by0, by1 = xs.min(), xs.max(), ys.min(), ys.max()
        boxes[b] = (int(0.5 * (bx0 + bx1)), int(0.5 * (by0 + by1)), int(bx1 - bx0), int(by1 - by0), kp)
    return {"images": torch.from_numpy(images.astype(np.float32)), "instances": torch.from_numpy(gt),
            "points": torch.from_numpy(pts), "boxes": torch.from_numpy(boxes)}
