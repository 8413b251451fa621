"""Host-side prompt simulation of the training / evaluation loops: where the next click goes, which slot it takes,
which error mask the P2CL label of that slot becomes, and the box prompt.  Mirrors
``get_next_points`` / ``get_next_promts`` / ``cal_box`` / ``max_connected_regions`` of the reference
(isegm/engine/trainer.py:615-768, 1061-1131, 1175-1190).  SURVEY.md section 8f ranks a GPU-resident version as the next
row; this one keeps the reference's CPU structure (numpy + scipy) so that the integer bookkeeping is identical.

PARITY NOTE: the reference calls ``cv2.distanceTransform(mask, DIST_L2, 5)`` (5x5 chamfer approximation) and
``skimage.measure.label(connectivity=2)``; neither library is in this image.  Round 4: the chamfer transform is RESTATED from
OpenCV's published two-pass fixed-point algorithm (``chamfer_l2_5x5`` below, the same arithmetic as
oracle/vpu_oracle.py::chamfer_l2_5x5 and the HIP kernel ``vpu_chamfer5``) and ``scipy.ndimage.label`` (8-connected) stands in
for skimage, so slot / order / label-mask / box bookkeeping AND the click coordinates are pinned by ``tests/golden/sim.npz``
(generated from the reference with the same two stand-ins) -- against real OpenCV the coordinates stay unpinned.
"""
import random

import numpy as np
import torch
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=bool)
# The connected components of cal_box on the GPU too (ops.cc_roots + ops.cc_table: union-find labels, then one reduction
# pass to a per-component size / bounding-box table), so that no mask crosses to the host.  Identical results (tests).
# GPU_CC = False: one uint8 mask per sample goes to the host and is labelled there (scipy) inside its bounding box.
import os as _os
GPU_CC = True      # connected components on the device (module attribute: tests compare against the host path)


def distance_transform(mask_u8):
    """Distance of every non-zero pixel to the nearest zero pixel (cv2.distanceTransform DIST_L2, mask size 0 = precise)."""
    return ndimage.distance_transform_edt(mask_u8).astype(np.float32)


# OpenCV's 5x5 chamfer weights for DIST_L2 (1, 1.4, 2.1969 as float32) in its 16-bit fixed point (cvRound(x * 65536))
_CH_A, _CH_B, _CH_C = 65536, 91750, 143976


def chamfer_l2_5x5(mask):
    """cv2.distanceTransform(mask, cv2.DIST_L2, 5) (trainer.py:628-629, 673-674, 736-737), restated from OpenCV's two-pass
    algorithm (imgproc/distransform.cpp, distanceTransform_5x5, the C++ path): fixed-point distances, a forward raster pass
    over the upper half of the 5x5 mask, a backward pass over the mirrored half, outside the image = infinity, result
    float32(t) * 2^-16.  A row of a pass is the prefix minimum t[j] = a j + min_{k <= j} (cand[k] - a k)."""
    m = np.asarray(mask) != 0
    H, W = m.shape
    INF = np.int64(1) << 40
    a, b, c = np.int64(_CH_A), np.int64(_CH_B), np.int64(_CH_C)
    T = np.full((H + 4, W + 4), INF, np.int64)
    cols = np.arange(W, dtype=np.int64) * a
    sh = lambda row, d: row[2 + d: 2 + d + W]
    nb = lambda r1, r2: np.minimum.reduce([sh(r2, -1) + c, sh(r2, 1) + c, sh(r1, -2) + c, sh(r1, 2) + c,
                                           sh(r1, -1) + b, sh(r1, 1) + b, sh(r1, 0) + a])
    for i in range(H):
        cand = np.where(m[i], nb(T[i + 1], T[i]), 0)
        T[i + 2, 2:W + 2] = np.minimum.accumulate(cand - cols) + cols
    for i in range(H - 1, -1, -1):
        cand = np.minimum(T[i + 2, 2:W + 2], nb(T[i + 3], T[i + 4]))
        T[i + 2, 2:W + 2] = np.minimum.accumulate((cand + cols)[::-1])[::-1] - cols
    t = np.minimum(T[2:H + 2, 2:W + 2], np.int64(0xFFFFFFFF) - c)
    return t.astype(np.float32) * np.float32(1.0 / 65536.0)


def distance_transform_batch(masks, device=None, chamfer=False):
    """Distance transforms of N masks [N,H,W] (bool / uint8), each treated as surrounded by zero pixels (the reference pads
    by one pixel first, trainer.py:626-629) -> float32 [N,H,W].  ``chamfer``: the 5x5 chamfer approximation of the TRAINING
    simulators (cv2 DIST_L2, 5) instead of the exact Euclidean transform (the Clicker's DIST_L2, 0).  On a CUDA ``device`` the
    HIP kernels (ops.edt / ops.chamfer5, bit-identical to the host forms: tests/test_ops_gpu.py) do all N at once."""
    masks = np.ascontiguousarray(masks).astype(np.uint8)
    if device is not None and torch.device(device).type == "cuda":
        from pvpuformer_amd import ops
        f = ops.chamfer5 if chamfer else ops.edt
        return f(torch.from_numpy(masks).to(device), zero_border=True).cpu().numpy()
    out = np.empty(masks.shape, np.float32)
    for i, m in enumerate(masks):
        if chamfer:
            out[i] = chamfer_l2_5x5(np.pad(m, 1, "constant"))[1:-1, 1:-1]
        else:
            out[i] = ndimage.distance_transform_edt(np.pad(m, 1, "constant")).astype(np.float32)[1:-1, 1:-1]
    return out


def max_connected_regions(mask):
    """trainer.py:1175-1190 incl. its quirk: every component larger than 10 % of the foreground is merged into the
    largest-so-far label while scanning labels in ascending order."""
    mask = np.asarray(mask)
    rows, cols = np.flatnonzero(mask.any(1)), np.flatnonzero(mask.any(0))
    if len(rows) == 0:
        return np.zeros(mask.shape, np.int64)
    if (rows[-1] - rows[0] + 1) * (cols[-1] - cols[0] + 1) * 2 < mask.size:
        # label only the bounding box of the foreground (same components, same raster order of their first pixels)
        out = np.zeros(mask.shape, np.int8)
        sl = (slice(rows[0], rows[-1] + 1), slice(cols[0], cols[-1] + 1))
        out[sl] = max_connected_regions(np.ascontiguousarray(mask[sl]))
        return out
    labels, n = ndimage.label(mask, structure=_EIGHT)
    if n == 0:
        return labels.astype(np.int64)
    return _kept_labels(labels, n)[labels].astype(np.int8)


def _kept_labels(labels, n):
    """bool [n + 1]: which of the labels 1..n ``max_connected_regions`` keeps.  The reference relabels the image once per
    component (O(n * H * W)); the same scan on the component sizes: a label is evaluated before anything can be merged
    into it, so its size at that moment is its original size."""
    counts = np.bincount(labels.ravel(), minlength=n + 1)
    total = int(counts[1:].sum())
    target = np.arange(n + 1)
    max_num, max_pixel = 0, 0
    for j in range(1, n + 1):
        cnt = int(counts[j])
        if cnt > max_num:
            max_num, max_pixel = cnt, j
        if cnt > 0.1 * total:
            target[j] = max_pixel
    keep = target == max_pixel
    keep[0] = False
    return keep


def _kept_region_crop(mask):
    """``max_connected_regions(mask) == 1`` restricted to the bounding box of the foreground: (bool crop, first row, first
    column), or None for an empty mask -- what the stroke simulator needs, without the full-size passes."""
    mask = np.asarray(mask)
    rows = np.flatnonzero(mask.any(1))
    if len(rows) == 0:
        return None
    cols = np.flatnonzero(mask.any(0))
    crop = mask[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1]
    labels, n = ndimage.label(crop, structure=_EIGHT)
    region = (labels > 0) if n == 1 else _kept_labels(labels, n)[labels]
    return region, int(rows[0]), int(cols[0])


def _first_free(points_b, lo, hi, default):
    """index of the first row in [lo, hi) whose order (column 2) is < 0, else ``default``."""
    free = np.nonzero(points_b[lo:hi, 2] < 0)[0]
    return lo + int(free[0]) if len(free) else default


def cal_box(gt_mask, fn_mask, fp_mask, points, as_allmask=True, jitter_box=True, set_offset=10, rng=random):
    """trainer.py:1061-1131.  gt/fn/fp: bool [B,H,W]; points: float array [B,2n,3].  Returns int32 [B,5]
    (x_center, y_center, width, height, slot)."""
    B, H, W = gt_mask.shape
    n = points.shape[1] // 2
    out = np.zeros((B, 5), np.int32)
    for b in range(B):
        if as_allmask:
            idx = np.argwhere(gt_mask[b])
            loc = _first_free(points[b], 0, n, n - 1)
        else:
            if np.sum(fn_mask[b]) > np.sum(fp_mask[b]):
                idx = np.argwhere(max_connected_regions(fn_mask[b]) == 1)
                loc = n - 1                                           # trainer.py:1088 (always the last positive slot)
            else:
                idx = np.argwhere(max_connected_regions(fp_mask[b]) == 1)
                loc = _first_free(points[b], n, 2 * n, 2 * n - 1)
        if len(idx) == 0:
            continue
        y0, y1, x0, x1 = idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max()
        if jitter_box:
            off = rng.randint(-set_offset, 0)
            bx = min(max(x0 + off, 0), W - set_offset)
            off = rng.randint(0, set_offset)
            ex = max(min(x1 + off, W), bx + set_offset)
            off = rng.randint(-set_offset, 0)
            by = min(max(y0 + off, 0), H - set_offset)
            off = rng.randint(0, set_offset)
            ey = max(min(y1 + off, H), by + set_offset)
            y0, y1, x0, x1 = by, ey, bx, ex
        xc, yc, bw, bh = int(0.5 * (x0 + x1)), int(0.5 * (y0 + y1)), int(x1 - x0), int(y1 - y0)
        if xc >= 1 and yc >= 1 and bw >= 1 and bh >= 1:
            out[b] = (xc, yc, bw, bh, loc)
    return out


def _bezier(ctrl, ts):
    """Bezier curve of degree len(ctrl) - 1 through the Bernstein basis (what bezier.Curve.evaluate_multi returns; that
    package is not in this image)."""
    from math import comb
    k = len(ctrl) - 1
    basis = np.stack([comb(k, i) * ts ** i * (1 - ts) ** (k - i) for i in range(k + 1)], 1)   # [T, k+1]
    return basis @ ctrl


def cal_scribble(gt_mask, min_p=3, max_p=10, num_samples=1000, rng=random, np_rng=np.random):
    """Scribble simulator (trainer.py:1133-1243): per sample a smooth stroke through 3-10 random pixels of the largest
    region of ``gt_mask`` [B,H,W] -- one per stratum of the region's row range -- either as the Bezier curve of those
    control points or as the interpolating spline over the rows (coin flip per sample), clipped to the region's bounding
    box and truncated to integers.  Returns [scribbles float [B,1,num_samples,2] of (x, y), rectangles int [B,1,4] of
    (x_center, y_center, width, height)], the layout the model's scribble branch takes.  PARITY UNPINNED: the reference
    draws its curves with the ``bezier`` package (absent here) and is never asked for scribbles by the shipped trainer
    (trainer.py:367); the generator order of the draws (num_p, then per stratum row and pixel, then the coin) follows it."""
    from scipy.interpolate import make_interp_spline
    B = len(gt_mask)
    scr = np.zeros((B, 1, num_samples, 2), np.float64)
    rects = np.zeros((B, 1, 4), np.int64)
    for b in range(B):
        kept = _kept_region_crop(gt_mask[b])                            # the kept region inside the foreground's box (its
        if kept is None:                                                # pixels in raster order are the reference's point
            continue                                                    # list: a row of it = that row's columns)
        region, ro, co = kept
        num_p = rng.randint(min_p, max_p)
        rows_any, cols_any = np.flatnonzero(region.any(1)), np.flatnonzero(region.any(0))
        r0, r1, c0, c1 = ro + rows_any[0], ro + rows_any[-1], co + cols_any[0], co + cols_any[-1]
        gap = int(r1 - r0) // num_p
        ctrl, lo = [], int(r0)
        for _ in range(num_p):
            row = rng.randint(lo, lo + gap - 1) if gap > 0 else rng.randint(lo, lo + gap)
            cand = np.flatnonzero(region[row - ro]) if 0 <= row - ro < region.shape[0] else ()
            if len(cand):
                ctrl.append((row, co + cand[rng.randint(0, len(cand) - 1)]))
            lo += gap
        if not ctrl:
            continue
        ctrl = np.asarray(ctrl, np.float64)
        inline = np_rng.rand() > 0.5
        curve = None
        if not inline:
            try:
                spline = make_interp_spline(ctrl[:, 0], ctrl[:, 1])
                rows = np.linspace(ctrl[:, 0].min(), ctrl[:, 0].max(), num_samples)
                curve = np.stack([rows, spline(rows)], 1)
            except Exception:
                curve = None
        if curve is None:
            curve = _bezier(ctrl, np.linspace(0.0, 1.0, num_samples))
        rows_i = np.clip(curve[:, 0], r0, r1).astype(int)
        cols_i = np.clip(curve[:, 1], c0, c1).astype(int)
        scr[b, 0] = np.stack([cols_i, rows_i], 1)
        rects[b, 0] = (int(0.5 * (c0 + c1)), int(0.5 * (r0 + r1)), int(c1 - c0), int(r1 - r0))
    return [scr, rects]


def next_click(pred, gt, points, pred_thresh=0.49, np_rng=np.random, device=None):
    """The shared body of get_next_points / get_next_promts (trainer.py:615-654, 733-764): for each sample the new
    click (or None), its slot, and whether it is positive.  pred: float [B,H,W]; gt: bool [B,H,W]; points float
    [B,2n,3] (modified copy is returned).  Also returns the false-negative / false-positive masks."""
    fn = np.logical_and(gt, pred < pred_thresh)
    fp = np.logical_and(np.logical_not(gt), pred > pred_thresh)
    dts = distance_transform_batch(np.concatenate([fn, fp], 0), device, chamfer=True)   # zero border = the reference's np.pad(.., 1)
    n = points.shape[1] // 2
    points = points.copy()
    picks = []
    B = gt.shape[0]
    for b in range(B):
        fn_dt, fp_dt = dts[b], dts[B + b]
        fn_max, fp_max = float(np.max(fn_dt)), float(np.max(fp_dt))
        is_pos = fn_max > fp_max
        dt = fn_dt if is_pos else fp_dt
        inner = np.argwhere(dt > max(fn_max, fp_max) / 2.0)
        if len(inner) == 0:
            picks.append(None)
            continue
        coords = inner[np_rng.randint(0, len(inner))]
        order = max(float(points[b, :, 2].max()), 0.0) + 1
        loc = _first_free(points[b], 0, n, n - 1) if is_pos else _first_free(points[b], n, 2 * n, 2 * n - 1)
        points[b, loc] = (float(coords[0]), float(coords[1]), float(order))
        picks.append((loc, bool(is_pos)))
    return points, picks, fn, fp


class PromptState:
    """The P2CL label (``ed_mask_label``, trainer.py:329-331) kept in factored form for the fused loss kernel:
    label[b][s] = gt[b] (s < S/2) or 1 - gt[b], unless slot_idx[b][s] >= 0, then override[slot_idx[b][s]]."""

    def __init__(self, B, S, H, W, device, max_rounds=3, buffers=None):
        """``buffers`` = (slot_idx, override): tensors at fixed addresses (a captured training pass reads them); the caller
        has reset ``slot_idx`` to -1."""
        if buffers is not None:
            self.slot_idx, self.override = buffers
        else:
            self.slot_idx = -torch.ones(B, S, dtype=torch.int32, device=device)
            self.override = torch.empty(max_rounds * B, H, W, dtype=torch.float32, device=device)   # (only assigned planes are read)
        self.used = 0

    def assign(self, b, slot, mask_np):
        self.override[self.used].copy_(torch.from_numpy(np.ascontiguousarray(mask_np, dtype=np.float32)))
        self.slot_idx[b, slot] = self.used
        self.used += 1

    def assign_device(self, b, slot, mask_t):
        """assign() for a mask that already lives on the device."""
        self.override[self.used].copy_(mask_t.to(torch.float32))
        self.slot_idx[b, slot] = self.used
        self.used += 1

    def dense(self, gt):
        """Materialises ed_mask_label [B,S,H,W] (tests only)."""
        B, S = self.slot_idx.shape
        lab = torch.cat([gt.repeat(1, S // 2, 1, 1), (1 - gt).repeat(1, S // 2, 1, 1)], 1).clone()
        idx = self.slot_idx.cpu().numpy()
        for b in range(B):
            for s in range(S):
                if idx[b, s] >= 0:
                    lab[b, s] = self.override[idx[b, s]]
        return lab


def get_next_promts(pred, gt, points, state=None, pred_thresh=0.49, as_allmask=False, jitter_box=True,
                    np_rng=np.random, rng=random):
    """trainer.py:703-768.  pred [B,1,H,W] probabilities, gt [B,1,H,W], points [B,2n,3] (tensors).  Returns
    (points, boxes int32 [B,5]) as tensors on points' device and updates ``state`` (the slot that received the click now
    predicts the false-negative / false-positive mask of this round)."""
    dev = points.device
    if pred.is_cuda:
        return _get_next_promts_gpu(pred, gt, points, state, pred_thresh, as_allmask, jitter_box, np_rng, rng)
    pred_np = pred.detach().float().cpu().numpy()[:, 0]
    gt_np = gt.detach().cpu().numpy()[:, 0] > 0.5
    pts_np = points.detach().float().cpu().numpy()
    fn0 = np.logical_and(gt_np, pred_np < pred_thresh)
    fp0 = np.logical_and(np.logical_not(gt_np), pred_np > pred_thresh)
    boxes = cal_box(gt_np, fn0, fp0, pts_np, as_allmask=as_allmask, jitter_box=jitter_box, rng=rng)
    new_pts, picks, fn, fp = next_click(pred_np, gt_np, pts_np, pred_thresh, np_rng,
                                        device=pred.device if pred.is_cuda else None)
    if state is not None:
        for b, pk in enumerate(picks):
            if pk is not None:
                state.assign(b, pk[0], fn[b] if pk[1] else fp[b])
    return torch.from_numpy(new_pts).to(dev), torch.from_numpy(boxes).to(dev)


def get_iou(pred, gt, pred_thresh=0.49):
    """trainer.py:1045-1051."""
    pm, gm = pred > pred_thresh, gt > 0.5
    return (pm & gm).sum() / (pm | gm).sum()


def _get_next_promts_gpu(pred, gt, points, state, pred_thresh, as_allmask, jitter_box, np_rng, rng):
    """get_next_promts with the masks, distance transforms, maxima and the k-th-candidate lookup on the GPU: what crosses
    to the host per call is one uint8 mask per sample (for the connected-component labelling of cal_box), a few scalars
    per sample and the chosen coordinates -- not 2B float distance maps.  Same random draws in the same order, same
    results as the host path (tests/test_ops_gpu.py::test_get_next_promts_gpu_equals_host)."""
    from pvpuformer_amd import ops
    dev = points.device
    B, _, H, W = pred.shape
    p = pred.detach().float()[:, 0]
    g = gt.detach()[:, 0] > 0.5
    fn = g & (p < pred_thresh)
    fp = (~g) & (p > pred_thresh)
    pts_np = points.detach().float().cpu().numpy()
    n = pts_np.shape[1] // 2
    # ---- cal_box (trainer.py:1061-1131): only bounding boxes are needed -- of gt, or of the region that
    # max_connected_regions keeps of the larger error mask; the components are found on the GPU (ops.cc_roots) and
    # their sizes / boxes come to the host as a few numbers per component
    if as_allmask:
        use_fn = None
        anyr, anyc = g.any(2).cpu().numpy(), g.any(1).cpu().numpy()
        bbox = []
        for b in range(B):
            rows, cols = np.flatnonzero(anyr[b]), np.flatnonzero(anyc[b])
            bbox.append(None if len(rows) == 0 else (rows[0], rows[-1], cols[0], cols[-1]))
    else:
        cnt = torch.stack([fn.flatten(1).sum(1), fp.flatten(1).sum(1)]).cpu().numpy()     # [2, B]
        use_fn = cnt[0] > cnt[1]
        sel = torch.from_numpy(use_fn).to(pred.device)
        chosen = torch.where(sel[:, None, None], fn, fp).to(torch.uint8)
        if GPU_CC:
            bbox = _kept_region_boxes(chosen)
        else:   # GPU_CC = False: one uint8 mask per sample to the host, labelled there inside its bounding box
            bbox = []
            for m in chosen.cpu().numpy():
                region = max_connected_regions(m > 0) == 1
                rows, cols = np.flatnonzero(region.any(1)), np.flatnonzero(region.any(0))
                bbox.append(None if len(rows) == 0 else (rows[0], rows[-1], cols[0], cols[-1]))
    boxes = np.zeros((B, 5), np.int32)
    set_offset = 10
    for b in range(B):
        if as_allmask:
            loc = _first_free(pts_np[b], 0, n, n - 1)
        else:
            loc = n - 1 if use_fn[b] else _first_free(pts_np[b], n, 2 * n, 2 * n - 1)
        if bbox[b] is None:
            continue
        y0, y1, x0, x1 = bbox[b]
        if jitter_box:
            off = rng.randint(-set_offset, 0)
            bx = min(max(x0 + off, 0), W - set_offset)
            off = rng.randint(0, set_offset)
            ex = max(min(x1 + off, W), bx + set_offset)
            off = rng.randint(-set_offset, 0)
            by = min(max(y0 + off, 0), H - set_offset)
            off = rng.randint(0, set_offset)
            ey = max(min(y1 + off, H), by + set_offset)
            y0, y1, x0, x1 = by, ey, bx, ex
        xc, yc, bw, bh = int(0.5 * (x0 + x1)), int(0.5 * (y0 + y1)), int(x1 - x0), int(y1 - y0)
        if xc >= 1 and yc >= 1 and bw >= 1 and bh >= 1:
            boxes[b] = (xc, yc, bw, bh, loc)
    # ---- next_click (trainer.py:615-654, 733-764)
    dts = ops.chamfer5(torch.cat([fn, fp], 0).to(torch.uint8), zero_border=True)           # [2B, H, W] (cv2 DIST_L2, 5)
    mx = dts.flatten(1).amax(1).cpu().numpy()                                              # fn maxima, then fp maxima
    is_pos = mx[:B] > mx[B:]
    pick_t = torch.from_numpy(np.where(is_pos, np.arange(B), np.arange(B) + B)).to(pred.device)
    dt = dts[pick_t]                                                                       # the map each sample draws from
    thr = torch.from_numpy((np.maximum(mx[:B], mx[B:]) / 2.0).astype(np.float32)).to(pred.device)
    inner = (dt > thr[:, None, None]).flatten(1)
    counts = inner.sum(1).cpu().numpy()
    ks = np.full(B, -1, np.int64)
    for b in range(B):                                   # the draws of the host path, in its order
        if counts[b] > 0:
            ks[b] = np_rng.randint(0, counts[b])
    # k-th candidate in raster order = first position where the running count reaches k + 1
    cs = inner.cumsum(1)
    kt = torch.from_numpy(ks).to(pred.device)
    pos = ((cs == (kt[:, None] + 1)) & inner).to(torch.uint8).argmax(1).cpu().numpy()
    new_pts = pts_np.copy()
    for b in range(B):
        if ks[b] < 0:
            continue
        order = max(float(new_pts[b, :, 2].max()), 0.0) + 1
        loc = _first_free(new_pts[b], 0, n, n - 1) if is_pos[b] else _first_free(new_pts[b], n, 2 * n, 2 * n - 1)
        new_pts[b, loc] = (float(pos[b] // W), float(pos[b] % W), float(order))
        if state is not None:
            state.assign_device(b, loc, (fn if is_pos[b] else fp)[b])
    return torch.from_numpy(new_pts).to(dev), torch.from_numpy(boxes).to(dev)


def _kept_region_boxes(masks, kmax=4096):
    """Bounding box (y0, y1, x0, x1) of ``max_connected_regions(mask) == 1`` for every mask of a uint8 CUDA batch
    [B,H,W] (None where the mask is empty).  Components by union-find on the GPU (``vpu_cc_roots``) and their sizes /
    boxes by one reduction pass (``vpu_cc_table``): what crosses to the host is six integers per component.  Sorted by
    root (= smallest pixel index) the components are in scipy's / skimage's label order, so the host scan over the
    component SIZES (largest-so-far, merge everything above 10 % of the foreground) is the one of the reference
    (trainer.py:1175-1190).  More than ``kmax`` components (salt-and-pepper masks): the host labelling takes over."""
    from pvpuformer_amd import ops
    B, H, W = masks.shape
    flat = ops.cc_table(ops.cc_roots(masks), kmax).cpu().numpy()
    K = int(flat[-1])
    if K > kmax:
        out = []
        for m in masks.cpu().numpy():
            region = max_connected_regions(m > 0) == 1
            rows, cols = np.flatnonzero(region.any(1)), np.flatnonzero(region.any(0))
            out.append(None if len(rows) == 0 else (int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1])))
        return out
    table = flat[:6 * K].reshape(K, 6)
    table = table[np.argsort(table[:, 0], kind="stable")]          # ascending roots = label order
    owner = table[:, 0] // (H * W)
    out = [None] * B
    for b in range(B):
        comp = table[owner == b]
        if len(comp) == 0:
            continue
        cnts = comp[:, 1]
        total = int(cnts.sum())
        target = np.arange(len(comp))
        max_num, max_pixel = 0, -1
        for j in range(len(comp)):
            c = int(cnts[j])
            if c > max_num:
                max_num, max_pixel = c, j
            if c > 0.1 * total:
                target[j] = max_pixel
        keep = comp[target == max_pixel]
        out[b] = (int(keep[:, 2].min()), int(keep[:, 3].max()), int(keep[:, 4].min()), int(keep[:, 5].max()))
    return out
