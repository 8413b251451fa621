"""Synthetic training batches of the shape the reference's data pipeline produces (the
``isegm/data`` package is absent from the reference snapshot): images U[0,1), an ellipse+rectangle ground-truth mask,
1-3 positive clicks inside it and 0-2 negative clicks outside (row, col, order; -1 padded to 24+24), and the
ground-truth bounding box in ``cal_box`` form (isegm/engine/trainer.py:1113-1125, no jitter)."""
import numpy as np
import torch


def synth_batch(B, img=448, seed=0, num_max_points=24, device="cpu"):
    rs = np.random.RandomState(seed)
    images = rs.rand(B, 3, img, img).astype(np.float32)
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
        inside, outside = np.argwhere(m), np.argwhere(~m)
        kp, kn = rs.randint(1, 4), rs.randint(0, 3)
        order = 0
        for i in range(kp):
            r, c = inside[rs.randint(len(inside))]
            pts[b, i] = (r, c, order); order += 1
        for i in range(kn):
            r, c = outside[rs.randint(len(outside))]
            pts[b, num_max_points + i] = (r, c, order); order += 1
        ys, xs = inside[:, 0], inside[:, 1]
        boxes[b] = (int(0.5 * (xs.min() + xs.max())), int(0.5 * (ys.min() + ys.max())), int(xs.max() - xs.min()),
                    int(ys.max() - ys.min()), kp)
    out = {"images": torch.from_numpy(images), "instances": torch.from_numpy(gt), "points": torch.from_numpy(pts),
           "boxes": torch.from_numpy(boxes)}
    return {k: v.to(device) for k, v in out.items()}


def vitb_model_kwargs(embed_dim=768, depth=12, num_heads=12, img=448, patch=16, out_dims=(128, 256, 512, 1024),
                      channels=256):
    """Constructor arguments of models/iSegNet/vpu_base448_cocolvis.py:13-56."""
    bp = dict(img_size=(img, img), patch_size=(patch, patch), in_chans=3, embed_dim=embed_dim, depth=depth,
              num_heads=num_heads, mlp_ratio=4, qkv_bias=True)
    npar = dict(in_dim=embed_dim, out_dims=list(out_dims), img_size=(img, img))
    hp = dict(in_channels=list(out_dims), in_index=[0, 1, 2, 3], dropout_ratio=0.1, num_classes=1, loss_decode=None,
              align_corners=False, upsample='x1', ed_loss=True, channels=channels)
    return dict(use_disks=True, norm_radius=5, with_prev_mask=True, backbone_params=bp, neck_params=npar,
                head_params=hp, random_split=False, residual=True, with_aux_output=True)
