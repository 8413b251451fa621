"""Generates tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference with the
run-time stand-ins of oracle/ref_import.py) and, in the same run, checks the oracle restatement
(oracle/vpu_oracle.py) against it.  Build-container only; the fixtures it writes are data (inputs +
expected outputs), never reference source.

    python oracle/make_golden.py            # writes tests/golden/{pue,disk,tiny,vitb,...}.npz
    python oracle/make_golden.py vitl vith  # the full-size ViT-L / ViT-H fixtures (minutes of CPU, ~20 GB of memory; not
                                            # in the default list)

Weights are NOT stored: they are regenerated bit-identically from ``vpu_oracle.synth_state_dict``
(integer hash), so fixtures stay small.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import vpu_oracle as vo  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def build_reference(cfg, ref_vpu):
    from isegm.model.modeling.common import FFNBlock
    D = cfg["embed_dim"]
    bp = dict(img_size=(cfg["img"],) * 2, patch_size=(cfg["patch"],) * 2, in_chans=3, embed_dim=D,
              depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"], qkv_bias=True)
    npar = dict(in_dim=D, out_dims=list(cfg["out_dims"]), img_size=(cfg["img"],) * 2)
    hp = dict(in_channels=list(cfg["out_dims"]), in_index=[0, 1, 2, 3], dropout_ratio=0.1, num_classes=1,
              loss_decode=None, align_corners=False, upsample='x1', ed_loss=True,
              channels=cfg["head_channels"])
    m = ref_vpu.VitMultiGaussianVector_ed_Model(
        use_disks=True, norm_radius=5, with_prev_mask=True, backbone_params=bp, neck_params=npar,
        head_params=hp, random_split=False, residual=True, with_aux_output=True)
    if D != 768:  # reference hard-codes d_model=768 in the head (swin_transformer.py:668)
        m.head.d_model = D
        m.head.ffn_layer = FFNBlock(embedding_dim=D, mlp_dim=2 * D, out_dim=cfg["head_channels"])
    shapes = vo.param_shapes(cfg)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state-dict key order differs from reference"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    sd = vo.synth_state_dict(shapes, seed=0)
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m, sd


def patch_box_rasteriser(model):
    """cv2 is absent: route the reference's draw_box through the oracle's rasteriser so that every
    DOWNSTREAM tensor of the box path is still produced by the reference."""
    def draw_box(image_, box_, points, gt_mask=None):
        n = points.shape[1] // 2
        arr = vo.box_outline(image_.cpu().numpy().copy(), box_.cpu().numpy(), n)
        image_[:] = torch.from_numpy(arr)
        return image_
    model.draw_box = draw_box


def sub(t, step=7):
    return t[..., ::step, ::step].contiguous().numpy()


def synth_scribbles(gt, P=200, seed=17):
    """Seeded strokes across each sample's ground-truth box (float (x, y) samples + the box as (xc, yc, w, h)), the layout
    ``forward(as_prompt_type=2)`` takes (is_vpu_model.py:294-352)."""
    B = gt.shape[0]
    rs = np.random.RandomState(seed)
    t = np.linspace(0, 1, P)
    scr = np.zeros((B, 1, P, 2), np.float64)
    rects = np.zeros((B, 1, 4), np.int64)
    for b in range(B):
        ys, xs = np.nonzero(gt[b, 0].numpy() > 0.5)
        x0, x1, y0, y1 = xs.min(), xs.max(), ys.min(), ys.max()
        scr[b, 0, :, 0] = x0 + (x1 - x0) * t + rs.uniform(-1.5, 1.5, P)
        scr[b, 0, :, 1] = (y0 + y1) / 2 + 0.35 * (y1 - y0) * np.sin(5 * t + b) + rs.uniform(-1.5, 1.5, P)
        rects[b, 0] = ((x0 + x1) // 2, (y0 + y1) // 2, x1 - x0, y1 - y0)
    return scr, rects


def patch_scribble_rasteriser(model):
    """cv2 is absent: the reference's draw_scribble goes through the oracle's poly-line rasteriser (same arrangement as the
    box outline); its debug ``ops.draw_scribble`` (cv2.imwrite to a hard-coded path, ops.py:409-419) becomes a no-op."""
    import isegm.model.ops as ref_ops
    ref_ops.draw_scribble = lambda *a, **k: None

    def draw_scribble(image_, scribble_, bounding_rectangle_, gt_mask=None):
        arr = vo.polyline_raster(image_.cpu().numpy().copy(), np.asarray(scribble_[0]))
        image_[:] = torch.from_numpy(arr)
        return image_
    model.draw_scribble = draw_scribble


SCRIBBLE_SEED = 321


def run_model_fixture(name, cfg, B, ref_vpu, ref_losses, with_grads=True, store_full_small=True,
                      modes=(("click", 0), ("box", 1))):
    import random
    t0 = time.time()
    model, sd = build_reference(cfg, ref_vpu)
    patch_box_rasteriser(model)
    patch_scribble_rasteriser(model)
    batch = vo.synth_batch(B, cfg["img"], seed=3)
    img4 = torch.cat([batch["images"], torch.zeros(B, 1, cfg["img"], cfg["img"])], 1)
    # a non-trivial previous mask for sample 0 (as in click iteration > 0)
    img4[0, 3] = torch.sigmoid(4 * (batch["instances"][0, 0] - 0.5))
    pts, boxes, gt = batch["points"], batch["boxes"], batch["instances"]
    fx = {"cfg_" + k: np.asarray(v) for k, v in cfg.items()}
    fx["B"] = np.asarray(B)
    scr, rects = synth_scribbles(gt)
    if any(pt == 2 for _, pt in modes):
        fx.update(scribbles=scr, rects=rects, scribble_seed=np.asarray(SCRIBBLE_SEED))

    for mode, ptype in modes:
        taps_ref = {}
        hooks = []
        hooks.append(model.backbone.register_forward_hook(lambda m_, i, o: None))
        feats = {}
        def bb_hook(m_, i, o):
            feats["bb"] = o
        orig_fb = model.backbone.forward_backbone
        def fb(*a, **k):
            o = orig_fb(*a, **k)
            feats["bb"] = o
            return o
        model.backbone.forward_backbone = fb
        orig_neck = model.neck.forward
        def nk(*a, **k):
            o = orig_neck(*a, **k)
            feats["neck"] = o
            return o
        model.neck.forward = nk
        orig_head = model.head.forward_feat
        def hd(*a, **k):
            o = orig_head(*a, **k)
            feats["head"] = o
            return o
        model.head.forward_feat = hd

        params = [p for p in model.parameters()]
        for p in params:
            p.grad = None
        prompts = [pts, boxes, [scr, rects] if ptype == 2 else [None, None]] if ptype else None
        random.seed(SCRIBBLE_SEED)          # the scribble vectors draw from the global ``random`` state (ops.py:274,290)
        out = model(img4.clone(), pts.clone(), prompts, ptype, True, False)
        model.backbone.forward_backbone, model.neck.forward, model.head.forward_feat = orig_fb, orig_neck, orig_head
        for h in hooks:
            h.remove()

        # ---- oracle vs reference (this is the pin) ----
        taps = {}
        # box mode: numpy >= 2 makes the reference evaluate the box Gaussians in float64 (np.int32 scalar promotion),
        # numpy 1.23 (the reference's requirements.txt) and the oracle in float32; the <= 1e-7 difference is checked in
        # prompt_fixtures().  Feed the reference's own rows here so that everything downstream is compared exactly.
        pue_ref = None
        if ptype == 1:
            with torch.no_grad():
                pue_ref = model._guassinvector_box(pts, boxes)
            assert np.abs(vo.pue_box(pts.numpy(), boxes.numpy()) - pue_ref.numpy()).max() <= 2e-7
        okw = dict(scribbles=(scr, rects)) if ptype == 2 else {}
        with torch.no_grad():
            o_or = vo.vpu_forward(sd, cfg, img4, pts, boxes, ptype, taps=taps, pue_override=pue_ref,
                                  rng=random.Random(SCRIBBLE_SEED), **okw)
        for k in ("instances", "instances_aux"):
            err = (o_or[k] - out[k]).abs().max().item()
            ref_mag = out[k].abs().max().item()
            print(f"[{name}/{mode}] oracle vs reference {k}: max abs err {err:.3e} (max |ref| {ref_mag:.3f})")
            assert err <= 2e-5 * max(1.0, ref_mag), "oracle restatement diverges from the reference"
        assert (taps["backbone"] - feats["bb"]).abs().max().item() < 2e-4
        assert (taps["q_out"] - feats["neck"][1]).abs().max().item() < 2e-4

        fx[f"{mode}_backbone_sub"] = feats["bb"].detach()[:, ::37, ::5].contiguous().numpy()
        fx[f"{mode}_backbone_abs_mean"] = np.asarray(feats["bb"].detach().abs().mean().item())
        fx[f"{mode}_q_out"] = feats["neck"][1].detach().numpy()
        for i, f in enumerate(feats["neck"][0]):
            fx[f"{mode}_fpn{i}_abs_mean"] = np.asarray(f.detach().abs().mean().item())
            fx[f"{mode}_fpn{i}_sub"] = f.detach()[:, ::9, ::3, ::3].contiguous().numpy()
        fx[f"{mode}_seg_lowres"] = feats["head"][0].detach().numpy()
        fx[f"{mode}_sim_lowres_sub"] = feats["head"][1].detach()[:, ::6, ::2, ::2].contiguous().numpy()
        fx[f"{mode}_sim_lowres_slot_mean"] = feats["head"][1].detach().mean(dim=(2, 3)).numpy()
        fx[f"{mode}_instances_sub"] = sub(out["instances"].detach())
        fx[f"{mode}_instances_aux_sub"] = sub(out["instances_aux"].detach()[:, ::6])
        fx[f"{mode}_instances_mean"] = np.asarray(out["instances"].detach().mean().item())
        fx[f"{mode}_instances_aux_mean"] = np.asarray(out["instances_aux"].detach().mean().item())

        if with_grads:
            ed = vo.ed_mask_label(gt, cfg["num_max_points"])
            nfl = ref_losses.NormalizedFocalLossSigmoid(alpha=0.5, gamma=2, penalty_loss=False)
            dice = ref_losses.DiceLoss(use_sigmoid=True, activate=True, naive_dice=True, loss_weight=1.0)
            bce = ref_losses.SigmoidBinaryCrossEntropyLoss(from_sigmoid=True)
            l_nfl = torch.mean(nfl(out["instances"], gt))
            l_dice = torch.mean(dice(out["instances"], gt))
            l_pcl = torch.mean(bce(out["instances_aux"], ed))
            loss = 1.0 * l_nfl + 1.0 * l_dice + 2.0 * l_pcl
            loss.backward()
            tot_or, parts = vo.step_loss({k: v for k, v in out.items()}, gt, ed)
            assert abs(tot_or.item() - loss.item()) < 1e-5 * max(1, abs(loss.item())), (tot_or.item(), loss.item())
            fx[f"{mode}_loss"] = np.asarray([loss.item(), l_nfl.item(), l_dice.item(), l_pcl.item()])
            names = [n for n, _ in model.named_parameters()]
            gnorm = {}
            for n, p in model.named_parameters():
                gnorm[n] = float(p.grad.norm().item()) if p.grad is not None else -1.0
            fx[f"{mode}_grad_names"] = np.asarray(names)
            fx[f"{mode}_grad_norms"] = np.asarray([gnorm[n] for n in names])
            no_grad = sorted(n for n in names if gnorm[n] < 0)
            assert no_grad == sorted(vo.unused_param_names(cfg)), no_grad
            # oracle backward vs reference backward
            sd_g = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
            o2 = vo.vpu_forward(sd_g, cfg, img4, pts, boxes, ptype, pue_override=pue_ref, rng=random.Random(SCRIBBLE_SEED), **okw)
            t2, _ = vo.step_loss(o2, gt, ed)
            t2.backward()
            worst = 0.0
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                g2 = sd_g[n].grad
                # k_proj.bias gradients are analytically zero (softmax is invariant to a key bias):
                # only rounding noise ~1e-10 lives there, hence the absolute floor.
                rel = max(0.0, (g2 - p.grad).norm().item() - 5e-9) / (p.grad.norm().item() + 1e-12)
                worst = max(worst, rel)
            print(f"[{name}/{mode}] oracle vs reference grads: worst rel-L2 err {worst:.3e}")
            assert worst < 1e-3
            # a few full gradients (small tensors) and slices of big ones
            keep = ["backbone.blocks.0.norm1.weight", "backbone.blocks.0.attn.qkv.bias",
                    f"backbone.blocks.{cfg['depth'] - 1}.mlp.fc2.bias", "backbone.patch_embed.proj.bias",
                    "patch_embed_coords.proj.bias", "neck.att.layers.0.norm1.weight",
                    "neck.att.layers.2.cross_attn_image_to_token.out_proj.bias",
                    "neck.att.norm_final_attn.bias", "neck.down_4.6.weight", "neck.down_32.3.bias",
                    "head.conv_seg.weight", "head.fusion_conv.conv.bias", "head.ffn_layer.lin2.bias",
                    "neck.ffn_layer.lin2.bias"]
            for n, p in model.named_parameters():
                if n in keep:
                    fx[f"{mode}_grad::{n}"] = p.grad.detach().numpy().copy()
            fx[f"{mode}_grad_slice::backbone.blocks.0.attn.qkv.weight"] = \
                dict(model.named_parameters())["backbone.blocks.0.attn.qkv.weight"].grad[::17, ::13].numpy().copy()
            fx[f"{mode}_grad_slice::neck.ffn_layer.lin1.weight"] = \
                dict(model.named_parameters())["neck.ffn_layer.lin1.weight"].grad[::64, ::29].numpy().copy()

    fx["images_seed"] = np.asarray(3)
    fx["points"] = pts.numpy()
    fx["boxes"] = boxes.numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
    print(f"[{name}] written in {time.time() - t0:.1f}s")


def prompt_fixtures(ref_vpu):
    """Known-answer vectors for the integer bookkeeping: PuE click/box rows and disk maps on
    crafted edge cases (corner-drop quirk, truncation, invalid rows, tiny boxes, fewer than 24 slots)."""
    from isegm.model.ops import DistMaps
    cfg = vo.make_cfg(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)
    model, _ = build_reference(cfg, ref_vpu)
    P = -np.ones((6, 48, 3), np.float32)
    # sample 0: interior, fractional, borders
    P[0, 0] = (200, 150, 0); P[0, 1] = (123.7, 200.2, 1); P[0, 2] = (0, 0, 2); P[0, 3] = (447, 447, 3)
    P[0, 24] = (5, 5, 4); P[0, 25] = (438, 5, 5)
    # sample 1: corner-drop quirk cases
    P[1, 0] = (445, 5, 0); P[1, 1] = (5, 445, 1); P[1, 2] = (439, 5, 2); P[1, 3] = (445, 445, 3)
    P[1, 24] = (9, 438, 4); P[1, 25] = (10, 439, 5); P[1, 26] = (8.99, 438.5, 6)
    # sample 2: valid order but coordinates -1 / order -1 with valid coordinates / half-integer
    P[2, 0] = (100.5, 200.5, 0); P[2, 1] = (-1, -1, 1); P[2, 2] = (50, 60, -1); P[2, 24] = (300.25, 10.75, 2)
    # sample 3: all 24 positive slots used
    for i in range(24):
        P[3, i] = (10 + 17 * i, 430 - 16 * i, i)
    P[3, 47] = (222, 111, 24)
    # sample 4: nothing valid;  sample 5: near-border sweep
    for i, v in enumerate((0, 1, 8, 9, 10, 437, 438, 439, 440, 446, 447)):
        P[5, i] = (v, 224, i); P[5, 24 + i] = (224, v, 11 + i)
    pts = torch.from_numpy(P)
    fx = {"points": P}
    with torch.no_grad():
        ref_click = model._guassinvector_click(pts).numpy()
    assert ref_click.dtype == np.float64
    mine = vo.pue_click(P)
    assert np.array_equal(mine, ref_click), "PuE click restatement is not bit-exact"
    fx["pue_click"] = ref_click
    # fewer than 24 slots per polarity (predictor path, base.py:195-213)
    P6 = -np.ones((2, 6, 3), np.float32)
    P6[0, 0] = (200, 150, 0); P6[0, 3] = (20, 30, 1); P6[1, 1] = (444, 3, 0); P6[1, 5] = (100.9, 100.1, 1)
    with torch.no_grad():
        ref6 = model._guassinvector_click(torch.from_numpy(P6)).numpy()
    assert np.array_equal(vo.pue_click(P6), ref6)
    fx["points_n3"] = P6
    fx["pue_click_n3"] = ref6
    # boxes: (xc, yc, w, h, slot)
    BX = np.array([[224, 224, 100, 60, 1], [224, 224, 7, 60, 0], [30, 420, 100, 100, 2], [100, 100, 8, 8, 30],
                   [0, 0, 0, 0, 0], [440, 10, 40, 30, 47]], np.int32)
    with torch.no_grad():
        ref_box = model._guassinvector_box(pts, torch.from_numpy(BX)).numpy()
    mine_b = vo.pue_box(P, BX)
    err = np.abs(mine_b - ref_box).max()
    print("PuE box: max abs err", err, " nonzero pattern equal:", np.array_equal(mine_b != 0, ref_box != 0))
    assert err <= 1e-7 and np.array_equal(mine_b != 0, ref_box != 0)
    fx["boxes"] = BX
    fx["pue_box"] = ref_box
    fx["click_lut"] = vo.click_lut()
    np.savez_compressed(os.path.join(OUT, "pue.npz"), **fx)

    dm = DistMaps(norm_radius=5, spatial_scale=1.0, cpu_mode=False, use_disks=True)
    with torch.no_grad():
        ref_d = dm(torch.zeros(6, 3, 448, 448), pts).numpy()
    mine_d = vo.disk_maps(P, 448, 448)
    assert np.array_equal(mine_d, ref_d), "disk-map restatement is not bit-exact"
    # non-square / small canvas, random fractional clicks
    rs = np.random.RandomState(5)
    Pr = -np.ones((3, 10, 3), np.float32)
    for b in range(3):
        for i in (0, 1, 2, 5, 6):
            Pr[b, i] = (rs.rand() * 95, rs.rand() * 130, i)
    with torch.no_grad():
        ref_r = dm(torch.zeros(3, 3, 96, 131), torch.from_numpy(Pr)).numpy()
    assert np.array_equal(vo.disk_maps(Pr, 96, 131), ref_r)
    np.savez_compressed(os.path.join(OUT, "disk.npz"), points=P, disks_packed=np.packbits(ref_d.astype(np.uint8)),
                        disks_shape=np.asarray(ref_d.shape), disk_counts=ref_d.sum(axis=(2, 3)),
                        points_small=Pr, disks_small=ref_r.astype(np.uint8))
    print("disk counts sample0:", ref_d[0].sum(axis=(1, 2)), " half-integer click:", ref_d[2, 0].sum())


def simulator_fixtures():
    """a16 / a18 bookkeeping: the reference's get_next_promts (click slot / order / label-mask / box logic) and
    BasePredictor.get_points_nd (click packing), run on seeded inputs.  OpenCV / skimage calls inside are replaced by the
    stand-ins of ref_import.install_simulator_standins(); cal_scribble (bezier, unused by click / box prompts) is disabled."""
    import random
    ref_import.install_simulator_standins()
    import isegm.engine.trainer as rt
    from isegm.inference.predictors.base import BasePredictor
    from isegm.inference.clicker import Click
    rt.cal_scribble = lambda *a, **k: None
    B, H = 4, 448
    batch = vo.synth_batch(B, H, seed=11)
    gt = batch["instances"]
    yy, xx = np.mgrid[0:H, 0:H]
    fx = {}
    rs = np.random.RandomState(5)
    for rnd_i, jitter in enumerate((True, False, True)):
        # a plausible prediction: gt shifted / eroded per sample plus a spurious blob
        pred = np.zeros((B, 1, H, H), np.float32)
        for b in range(B):
            sh = rs.randint(-30, 30, size=2)
            pred[b, 0] = np.roll(gt[b, 0].numpy(), tuple(sh), axis=(0, 1)) * 0.9
            cy, cx, r = rs.randint(50, 400), rs.randint(50, 400), rs.randint(10, 40)
            pred[b, 0] = np.maximum(pred[b, 0], (((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r) * 0.8)
        if rnd_i == 2:
            pred[0] = 0.0      # first-iteration situation: nothing predicted yet
            pred[1, 0] = gt[1, 0].numpy()   # perfect prediction: no click is added
        pts = batch["points"].clone()
        if rnd_i == 1:
            pts[0, :24, 2] = torch.arange(24).float()      # all positive slots taken -> falls back to slot n-1
            pts[0, :24, :2] = 5.0
        ed = vo.ed_mask_label(gt).clone()
        np.random.seed(100 + rnd_i); random.seed(200 + rnd_i)
        new_pts, boxes, _, ed_out = rt.get_next_promts(torch.from_numpy(pred), gt, pts, ed, as_allmask=False,
                                                       jitter_box=jitter)
        changed = (ed_out != vo.ed_mask_label(gt)).flatten(2).any(2).numpy()          # [B, 48]
        fx[f"r{rnd_i}_pred"] = np.packbits(pred > 0.49)
        fx[f"r{rnd_i}_pred_vals"] = np.unique(pred)
        fx[f"r{rnd_i}_points_in"] = pts.numpy()
        fx[f"r{rnd_i}_points_out"] = new_pts.numpy()
        fx[f"r{rnd_i}_boxes"] = boxes.numpy()
        fx[f"r{rnd_i}_changed_slots"] = changed
        fx[f"r{rnd_i}_changed_sums"] = (ed_out.sum(dim=(2, 3)) * torch.from_numpy(changed)).numpy()
        fx[f"r{rnd_i}_jitter"] = np.asarray(jitter)
        # the click-only simulator (trainer.py:615-654) on the same inputs, its own seed
        np.random.seed(300 + rnd_i)
        fx[f"r{rnd_i}_next_points"] = rt.get_next_points(torch.from_numpy(pred), gt, pts).numpy()
    fx["gt_seed"] = np.asarray(11)
    # click packing of the predictor (base.py:195-213)
    bp = BasePredictor.__new__(BasePredictor)
    bp.net_clicks_limit, bp.device = None, "cpu"
    lists = [[Click(True, (10, 20), 0), Click(False, (30, 40), 1), Click(True, (50, 60), 2)],
             [Click(False, (1, 2), 0)]]
    fx["points_nd_a"] = bp.get_points_nd(lists).numpy()
    fx["points_nd_b"] = bp.get_points_nd([[Click(True, (7, 8), 0)]]).numpy()
    bp.net_clicks_limit = 2
    fx["points_nd_limit2"] = bp.get_points_nd(lists).numpy()
    np.savez_compressed(os.path.join(OUT, "sim.npz"), **fx)
    print("[sim] written; boxes round0:", fx["r0_boxes"].tolist())


def zoom_fixtures():
    """ZoomIn / LimitLongestSide bookkeeping and resizes of the reference (isegm/inference/transforms/zoom_in.py,
    limit_longest_side.py) on a seeded click sequence: regions of interest, re-mapped click coordinates, the cropped +
    resized network input and the un-zoomed probability map (sub-sampled)."""
    from isegm.inference.transforms import ZoomIn, LimitLongestSide
    from isegm.inference.clicker import Click
    H, W = 300, 420
    g = torch.Generator().manual_seed(11)
    image_nd = torch.rand(1, 4, H, W, generator=g)
    yy, xx = np.mgrid[0:H, 0:W]
    blob1 = (((yy - 140) / 60.0) ** 2 + ((xx - 200) / 90.0) ** 2 < 1).astype(np.float32)
    blob2 = (((yy - 90) / 30.0) ** 2 + ((xx - 330) / 40.0) ** 2 < 1).astype(np.float32)
    clicks = [Click(True, (140, 200), 0), Click(False, (30, 40), 1), Click(True, (150.5, 260.25), 2),
              Click(True, (95, 335), 3)]
    fx = {"image_seed": np.asarray(11), "H": np.asarray(H), "W": np.asarray(W),
          "clicks": np.asarray([[c.is_positive, c.coords[0], c.coords[1], c.indx] for c in clicks], np.float64)}
    for name, kw in (("vpu", dict(target_size=(448, 448), skip_clicks=-1)), ("ritm", dict(target_size=400, skip_clicks=1))):
        z = ZoomIn(**kw)
        probs = [blob1 * 0.9, np.maximum(blob1, blob2) * 0.8, np.maximum(blob1, blob2) * 0.8]
        for step in range(3):
            cl = clicks[:step + 2]
            img_t, tcl = z.transform(image_nd, [cl])
            fx[f"{name}_{step}_roi"] = np.asarray(z._object_roi if z._object_roi is not None else (-1, -1, -1, -1))
            fx[f"{name}_{step}_changed"] = np.asarray(z.image_changed)
            fx[f"{name}_{step}_img_shape"] = np.asarray(img_t.shape)
            fx[f"{name}_{step}_img_sub"] = img_t[:, :, ::13, ::11].numpy()
            fx[f"{name}_{step}_tclicks"] = np.asarray([[c.coords[0], c.coords[1]] for c in tcl[0]], np.float64)
            # the "network output": a smooth function at the transformed size
            hh, ww = img_t.shape[2:]
            ty, tx = torch.meshgrid(torch.linspace(0, 1, hh), torch.linspace(0, 1, ww), indexing="ij")
            net_out = (torch.sin(3 * ty + step) * torch.cos(5 * tx) * 0.5 + 0.5)[None, None]
            # what the model "predicted" in image space decides the next region: paste the blob through the inverse
            back = z.inv_transform(net_out)
            fx[f"{name}_{step}_back_shape"] = np.asarray(back.shape)
            fx[f"{name}_{step}_back_sub"] = back[:, :, ::7, ::9].numpy()
            z._prev_probs = probs[step][None, None]          # drive the region logic with the seeded blobs
            fx[f"{name}_{step}_recalc"] = np.asarray(z.check_possible_recalculation())
    lim = LimitLongestSide(max_size=256)
    img_t, tcl = lim.transform(image_nd, [clicks[:2]])
    fx["lim_roi"] = np.asarray(lim._object_roi)
    fx["lim_img_shape"] = np.asarray(img_t.shape)
    fx["lim_img_sub"] = img_t[:, :, ::13, ::11].numpy()
    fx["lim_tclicks"] = np.asarray([[c.coords[0], c.coords[1]] for c in tcl[0]], np.float64)
    np.savez_compressed(os.path.join(OUT, "zoom.npz"), **fx)
    print("[zoom] written", {k: v.tolist() for k, v in fx.items() if k.endswith("_roi")})


def lrd_fixture(ref_vpu):
    """Layer-wise lr decay groups of the reference (isegm/utils/lr_decay.py:15-69 as called by
    isegm/engine/optimizer.py:29-35) on the tiny model: per tensor name its learning rate and weight decay."""
    import isegm.utils.lr_decay as lrd
    cfg = vo.make_cfg(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)
    m, _ = build_reference(cfg, ref_vpu)
    groups = lrd.param_groups_lrd(m, 5e-5, weight_decay=0.02, no_weight_decay_list=m.backbone.no_weight_decay(),
                                  layer_decay=0.75)
    by_id = {id(p): n for n, p in m.named_parameters()}
    names, lrs, wds = [], [], []
    for g in groups:
        ps = g["params"] if isinstance(g["params"], (list, tuple)) else [g["params"]]
        for p_ in ps:
            names.append(by_id[id(p_)]); lrs.append(g.get("lr", 5e-5)); wds.append(g["weight_decay"])
    np.savez_compressed(os.path.join(OUT, "lrd.npz"), names=np.asarray(names), lr=np.asarray(lrs, np.float64),
                        wd=np.asarray(wds, np.float64), all_names=np.asarray([n for n, _ in m.named_parameters()]))
    print("[lrd] written", len(names), "of", len(by_id), "tensors in groups")


def scribble_fixture(ref_vpu):
    """a9: the reference's _guassinvector_scribble (is_vpu_model.py:294-352) on seeded scribbles; its debug
    ``draw_scribble`` (cv2.imwrite to a hard-coded path, ops.py:409-419) is replaced by a no-op."""
    import random
    import isegm.model.ops as ref_ops
    ref_ops.draw_scribble = lambda *a, **k: None
    cfg = vo.make_cfg(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)
    m, _ = build_reference(cfg, ref_vpu)
    rs = np.random.RandomState(7)
    B, n, P = 3, 5, 60
    pts = -np.ones((B, 2 * n, 3), np.float32)
    pts[0, 0] = (100, 200, 0); pts[0, 1] = (50.7, 60.2, 1); pts[0, n] = (300, 10, 2)
    pts[1, 0] = (5, 440, 0)
    pts[2, n] = (10, 10, 0)                                   # no valid positive row: nothing is overwritten
    t = np.linspace(0, 1, P)
    scr = np.zeros((B, 1, P, 2), np.float64)
    scr[0, 0] = np.stack([100 + 80 * t + rs.randint(-2, 3, P), 150 + 40 * np.sin(6 * t) + rs.randint(-2, 3, P)], 1)
    scr[1, 0] = np.stack([300 + 20 * np.cos(5 * t), 200 + 100 * t], 1)
    scr[2, 0] = np.stack([40 + 10 * t, 40 + 10 * t], 1)
    rects = np.zeros((B, 1, 4), np.int64)
    rects[0, 0] = (140, 150, 90, 90); rects[1, 0] = (300, 250, 50, 110); rects[2, 0] = (45, 45, 12, 12)
    random.seed(123)
    with torch.no_grad():
        ref = m._guassinvector_scribble(torch.from_numpy(pts), [scr, rects]).numpy()
    got = vo.pue_scribble(pts, scr.astype(np.int32), rects, random.Random(123), 24, cfg["img"])
    err = np.abs(got - ref).max()
    print(f"[scribble] oracle vs reference: max abs err {err:.3e}; non-zero entries {int((ref != 0).sum())}")
    assert err == 0.0
    np.savez_compressed(os.path.join(OUT, "scribble.npz"), points=pts, scribbles=scr.astype(np.int32), rects=rects,
                        seed=np.asarray(123), pue=ref)


def scribble_model_fixture(ref_vpu):
    """a3 poly-line + a9 through the whole model (prompt type 2), tiny configuration: the reference's forward with its
    ``draw_scribble`` routed through the oracle's rasteriser (cv2 is absent; same arrangement as the box outline) and the
    debug ``ops.draw_scribble`` (cv2.imwrite to a hard-coded path) replaced by a no-op.  The vectors draw from the global
    ``random`` state: seeded right before the call, recorded in the fixture."""
    import random
    import isegm.model.ops as ref_ops
    ref_ops.draw_scribble = lambda *a, **k: None
    cfg = vo.make_cfg(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)
    model, sd = build_reference(cfg, ref_vpu)

    def draw_scribble(image_, scribble_, bounding_rectangle_, gt_mask=None):
        arr = vo.polyline_raster(image_.cpu().numpy().copy(), np.asarray(scribble_[0]))
        image_[:] = torch.from_numpy(arr)
        return image_
    model.draw_scribble = draw_scribble
    B, P, H = 2, 200, cfg["img"]
    batch = vo.synth_batch(B, H, seed=3)
    img4 = torch.cat([batch["images"], torch.zeros(B, 1, H, H)], 1)
    img4[0, 3] = torch.sigmoid(4 * (batch["instances"][0, 0] - 0.5))
    pts, boxes, gt = batch["points"], batch["boxes"], batch["instances"]
    rs = np.random.RandomState(17)
    t = np.linspace(0, 1, P)
    scr = np.zeros((B, 1, P, 2), np.float64)
    rects = np.zeros((B, 1, 4), np.int64)
    for b in range(B):
        ys, xs = np.nonzero(gt[b, 0].numpy() > 0.5)
        x0, x1, y0, y1 = xs.min(), xs.max(), ys.min(), ys.max()
        scr[b, 0, :, 0] = x0 + (x1 - x0) * t + rs.uniform(-1.5, 1.5, P)
        scr[b, 0, :, 1] = (y0 + y1) / 2 + 0.35 * (y1 - y0) * np.sin(5 * t + b) + rs.uniform(-1.5, 1.5, P)
        rects[b, 0] = ((x0 + x1) // 2, (y0 + y1) // 2, x1 - x0, y1 - y0)
    seed = 321
    random.seed(seed)
    with torch.no_grad():
        out = model(img4.clone(), pts.clone(), [pts, boxes, [scr, rects]], 2, True, False)
    random.seed(seed)
    with torch.no_grad():
        pue_ref = model._guassinvector_scribble(pts, [scr, rects]).numpy()
    taps = {}
    with torch.no_grad():
        o_or = vo.vpu_forward(sd, cfg, img4, pts, boxes, 2, taps=taps, scribbles=(scr, rects), rng=random.Random(seed))
    for k in ("instances", "instances_aux"):
        err = (o_or[k] - out[k]).abs().max().item()
        print(f"[tiny_scribble] oracle vs reference {k}: max abs err {err:.3e}")
        assert err <= 2e-5 * max(1.0, out[k].abs().max().item())
    got = vo.pue_scribble(pts.numpy(), scr.astype(np.int32), rects, random.Random(seed), 24, H)
    assert np.abs(got - pue_ref).max() == 0.0
    fx = {"cfg_" + k: np.asarray(v) for k, v in cfg.items()}
    fx.update(B=np.asarray(B), images_seed=np.asarray(3), seed=np.asarray(seed), points=pts.numpy(), boxes=boxes.numpy(),
              scribbles=scr, rects=rects, pue=pue_ref, coord_sum=taps["coord"].sum(dim=(2, 3)).numpy(),
              coord_bits=np.packbits(taps["coord"][:, 1].numpy() > 0.5),
              q_out=taps["q_out"].numpy(), seg_lowres=taps["seg_lowres"].numpy(),
              sim_lowres_sub=taps["sim_lowres"][:, ::6, ::2, ::2].contiguous().numpy(),
              instances_sub=sub(out["instances"].detach()), instances_aux_sub=sub(out["instances_aux"].detach()[:, ::6]))
    np.savez_compressed(os.path.join(OUT, "tiny_scribble.npz"), **fx)
    print("[tiny_scribble] written; pixels the poly-line adds per sample:",
          (taps["coord"][:, 1].sum(dim=(1, 2)) - torch.from_numpy(vo.disk_maps(pts.numpy(), H, H))[:, 0].sum(dim=(1, 2))).tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_vpu, ref_losses = ref_import.import_reference()
    which = sys.argv[1:] or ["pue", "tiny", "tinyh", "vitl8", "vitb", "sim", "zoom", "lrd", "scribble", "tiny_scribble"]
    if "zoom" in which:
        zoom_fixtures()
    if "lrd" in which:
        lrd_fixture(ref_vpu)
    if "scribble" in which:
        scribble_fixture(ref_vpu)
    if "tiny_scribble" in which:
        scribble_model_fixture(ref_vpu)
    if "sim" in which:
        simulator_fixtures()
    if "pue" in which:
        prompt_fixtures(ref_vpu)
    if "tiny" in which:
        cfg = vo.make_cfg(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)
        run_model_fixture("tiny", cfg, 2, ref_vpu, ref_losses)
    if "tinyh" in which:   # ViT-H geometry in small: patch 14 (32 x 32 tokens, 16 x 16 windows), head dim 80
        cfg = vo.make_cfg(embed_dim=640, depth=8, num_heads=8, patch=14, out_dims=(16, 32, 64, 128), head_channels=32)
        run_model_fixture("tinyh", cfg, 2, ref_vpu, ref_losses)
    if "vitl8" in which:   # ViT-L width (D = 1024, 16 heads of 64) at reduced depth: config 4's GEMM / attention shapes
        run_model_fixture("vitl8", vo.make_cfg(embed_dim=1024, depth=8, num_heads=16), 2, ref_vpu, ref_losses)
    if "vitb" in which:
        run_model_fixture("vitb", vo.make_cfg(), 2, ref_vpu, ref_losses)
    if "vitl" in which:    # config 4's own model: ViT-L at its full depth 24 (models_vit.py:310-313), click / box / scribble
        run_model_fixture("vitl", vo.make_cfg(embed_dim=1024, depth=24, num_heads=16), 2, ref_vpu, ref_losses,
                          modes=(("click", 0), ("box", 1), ("scribble", 2)))
    if "vith" in which:    # config 5's own model: ViT-H, D = 1280, depth 32, 16 heads of 80, patch 14 (models_vit.py:315-319)
        run_model_fixture("vith", vo.make_cfg(embed_dim=1280, depth=32, num_heads=16, patch=14), 2, ref_vpu, ref_losses)


if __name__ == "__main__":
    main()
