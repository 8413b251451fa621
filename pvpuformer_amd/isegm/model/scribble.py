"""Host side of the scribble prompt (prompt type 2): the two 1-D profiles of ``GaussianVector_scribble``
(isegm/model/ops.py:244-296) and the integer poly-line the rasteriser draws (isegm/model/is_model.py:123-130).

Why this stays on the host: the profile walk is sequential by construction -- for each column (then each row) of the
bounding rectangle it draws an index from Python's global ``random`` and, in the column pass, deletes the chosen point
from the list before the next draw -- so its numbers match the reference only when drawn, in the same order, from the
same generator.  It touches <= 1000 points per sample; the results (2 x img float64 per sample) go to the GPU once
(``ops.pue_scribble_rows``)."""
import random as _random
from bisect import bisect_left, bisect_right

import numpy as np


def scribble_profiles(scribbles, rects, img, rng=None):
    """scribbles: array-like [B,1,P,2] of (x, y); rects: [B,1,4] of (x_center, y_center, width, height).
    Returns float64 [B, 2*img]: per sample the x profile then the y profile, samples processed in order (the order of the
    reference's draws).  Faithful to the reference's indexing: the drawn number indexes the point list ITSELF, not the
    list of points in the current column (ops.py:272-275, 288-290); sigma = 3."""
    rng = rng or _random
    # randint(0, k - 1) IS _randbelow(k) (random.py: randint -> randrange -> istart + _randbelow(width)): the same draws
    # without two layers of argument checks, ~100 draws per sample
    below = getattr(getattr(rng, "_inst", rng), "_randbelow", None) or (lambda k: rng.randint(0, k - 1))
    scribbles, rects = np.asarray(scribbles), np.asarray(rects)
    B = scribbles.shape[0]
    out = np.zeros((B, 2 * img), np.float64)
    for b in range(B):
        pts = scribbles[b, 0].astype(np.int32)
        rect = rects[b, 0]
        if int(pts.sum()) + int(np.sum(rect)) == 0:
            continue
        xc, yc, bw, bh = (min(int(v), img) for v in rect)
        left, top = xc - bw // 2, yc - bh // 2
        # the list of points still alive (ascending original index), how many of them sit in each column / row of the
        # walk, and the copies of each distinct point (a stable sort of the packed coordinates: one bisect finds them, in
        # ascending index order): the same walk as a mask-and-count per column, without a pass over all points per step
        px64, py64 = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
        px, py = px64.tolist(), py64.tolist()
        alive = list(range(len(px)))
        cx = np.bincount(px64[(px64 >= 0) & (px64 < bw)], minlength=max(bw, 1)).tolist()
        cy = np.bincount(py64[(py64 >= 0) & (py64 < bh)], minlength=max(bh, 1)).tolist()
        key = (px64 << 32) + (py64 & 0xFFFFFFFF)
        order = np.argsort(key, kind="stable")
        skey, order = key[order].tolist(), order.tolist()
        key = key.tolist()
        for col in range(bw):
            k = cx[col]
            if k:
                i = alive[below(k)]                   # (the drawn number indexes the LIST, not the column's points)
                x, y = px[i], py[i]
                out[b, col] = np.exp(-((y - top) ** 2) / 18)
                gone = order[bisect_left(skey, key[i]):bisect_right(skey, key[i])]   # every copy of the chosen point
                for j in gone:                                                       # leaves the list
                    del alive[bisect_left(alive, j)]
                if 0 <= x < bw:
                    cx[x] -= len(gone)
                if 0 <= y < bh:
                    cy[y] -= len(gone)
        for row in range(bh):
            k = cy[row]
            if k:
                x = px[alive[below(k)]]
                out[b, img + row] = np.exp(-((x - left) ** 2) / 18)
    return out


def scribble_curves(scribbles):
    """int32 [B,P,2] poly-line vertices: the scribble points truncated as ``astype(np.int32)`` (is_model.py:128)."""
    s = np.asarray(scribbles)
    return np.ascontiguousarray(s[:, 0].astype(np.int32))
