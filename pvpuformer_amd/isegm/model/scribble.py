"""Host side of the scribble prompt (prompt type 2): the two 1-D profiles of ``GaussianVector_scribble``
(isegm/model/ops.py:244-296) and the integer poly-line the rasteriser draws (isegm/model/is_model.py:123-130).

Why this stays on the host: the profile walk is sequential by construction -- for each column (then each row) of the
bounding rectangle it draws an index from Python's global ``random`` and, in the column pass, deletes the chosen point
from the list before the next draw -- so its numbers match the reference only when drawn, in the same order, from the
same generator.  It touches <= 1000 points per sample; the results (2 x img float64 per sample) go to the GPU once
(``ops.pue_scribble_rows``)."""
import random as _random
from bisect import bisect_left

import numpy as np


def scribble_profiles(scribbles, rects, img, rng=None):
    """scribbles: array-like [B,1,P,2] of (x, y); rects: [B,1,4] of (x_center, y_center, width, height).
    Returns float64 [B, 2*img]: per sample the x profile then the y profile, samples processed in order (the order of the
    reference's draws).  Faithful to the reference's indexing: the drawn number indexes the point list ITSELF, not the
    list of points in the current column (ops.py:272-275, 288-290); sigma = 3."""
    rng = rng or _random
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
        # the list of points still alive (ascending original index), how many of them sit in each column / row, and the
        # copies of each distinct point: the same walk as a mask-and-count per column, without a pass over all points per step
        px, py = pts[:, 0].tolist(), pts[:, 1].tolist()
        alive = list(range(len(px)))
        cx, cy, copies = {}, {}, {}
        for i, (x, y) in enumerate(zip(px, py)):
            cx[x] = cx.get(x, 0) + 1
            cy[y] = cy.get(y, 0) + 1
            copies.setdefault((x, y), []).append(i)
        for col in range(bw):
            k = cx.get(col, 0)
            if k:
                i = alive[rng.randint(0, k - 1)]       # (the drawn number indexes the LIST, not the column's points)
                x, y = px[i], py[i]
                out[b, col] = np.exp(-((y - top) ** 2) / 18)
                gone = copies.pop((x, y))              # every copy of the chosen point leaves the list
                for j in gone:
                    del alive[bisect_left(alive, j)]
                cx[x] -= len(gone)
                cy[y] -= len(gone)
        for row in range(bh):
            k = cy.get(row, 0)
            if k:
                x = px[alive[rng.randint(0, k - 1)]]
                out[b, img + row] = np.exp(-((x - left) ** 2) / 18)
    return out


def scribble_curves(scribbles):
    """int32 [B,P,2] poly-line vertices: the scribble points truncated as ``astype(np.int32)`` (is_model.py:128)."""
    s = np.asarray(scribbles)
    return np.ascontiguousarray(s[:, 0].astype(np.int32))
