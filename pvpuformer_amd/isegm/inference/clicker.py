"""Oracle clicks of the NoBRS evaluation protocol (isegm/inference/clicker.py:7-118): the next click goes to the pixel of
the larger error region (false negatives vs false positives) that lies deepest inside it -- the arg-max, first in raster
order, of the region's distance transform, already-clicked pixels excluded.

GPU-resident when a device is available: ground truth, label-validity and not-yet-clicked maps live on the device, one
click is three kernels (error masks, exact distance transform of both masks at once, packed arg-max) and ONE 16-byte
read-back; the host path does the same with scipy.  ``cv2.distanceTransform(DIST_L2, 0)`` (precise) of the reference is
replaced by the exact Euclidean transform in both paths -- identical values between the two (tests), unpinned against
OpenCV itself (not installable here).
"""
import copy
import struct

import numpy as np


class Click:
    """One click: polarity, (row, col) and its ordinal.  (attribute names are read by the predictors)"""

    def __init__(self, is_positive, coords, indx=None):
        self.is_positive, self.coords, self.indx = is_positive, coords, indx

    @property
    def coords_and_indx(self):
        return (*self.coords, self.indx)

    def copy(self, **changes):
        twin = copy.deepcopy(self)
        vars(twin).update(changes)
        return twin


class Clicker:
    def __init__(self, gt_mask=None, init_clicks=None, ignore_label=-1, click_indx_offset=0, device="auto"):
        """``device`` (not in the reference): "auto" = the GPU when there is one, None = host only."""
        if device == "auto":
            import torch
            device = "cuda" if torch.cuda.is_available() else None
        self.device, self.click_indx_offset = device, click_indx_offset
        self.gt_mask = None if gt_mask is None else (gt_mask == 1)
        if gt_mask is not None:
            self.not_ignore_mask = gt_mask != ignore_label
        self._dev = None            # device mirrors: (gt, valid, not_clicked) uint8
        self._clicks = []
        self.reset_clicks()
        self.set_state(init_clicks or (), keep=True)

    # ------------------------------------------------------------------ click list
    @property
    def clicks_list(self):
        return self._clicks

    @property
    def num_pos_clicks(self):
        return sum(1 for c in self._clicks if c.is_positive)

    @property
    def num_neg_clicks(self):
        return len(self._clicks) - self.num_pos_clicks

    def __len__(self):
        return len(self._clicks)

    def get_clicks(self, clicks_limit=None):
        return self._clicks[:clicks_limit]

    def add_click(self, click):
        click.indx = self.click_indx_offset + len(self._clicks)
        self._clicks.append(click)
        self._mark(click.coords, clicked=True)

    def _remove_last_click(self):
        self._mark(self._clicks.pop().coords, clicked=False)

    def reset_clicks(self):
        self._clicks = []
        if self.gt_mask is None:
            return
        self.not_clicked_map = np.ones(self.gt_mask.shape, dtype=bool)
        if self._dev is not None:
            self._dev[2].fill_(1)

    def get_state(self):
        return copy.deepcopy(self._clicks)

    def set_state(self, state, keep=False):
        if not keep:
            self.reset_clicks()
        for click in state:
            self.add_click(click)

    def _mark(self, coords, clicked):
        if self.gt_mask is None:
            return
        r, c = int(coords[0]), int(coords[1])
        self.not_clicked_map[r, c] = not clicked
        if self._dev is not None:
            self._dev[2][r, c] = 0 if clicked else 1

    # ------------------------------------------------------------------ next click
    def make_next_click(self, pred_mask):
        assert self.gt_mask is not None
        self.add_click(self._get_next_click(pred_mask))

    def _get_next_click(self, pred_mask, padding=True):
        if self.device is not None:
            (fn_max, fn_at), (fp_max, fp_at) = self._deepest_errors_device(pred_mask, padding)
        else:
            (fn_max, fn_at), (fp_max, fp_at) = self._deepest_errors_host(np.asarray(pred_mask, dtype=bool), padding)
        positive = fn_max > fp_max
        return Click(is_positive=bool(positive), coords=fn_at if positive else fp_at)

    def _deepest_errors_host(self, pred, padding):
        from scipy import ndimage
        out = []
        for err in (self.gt_mask & ~pred & self.not_ignore_mask, ~self.gt_mask & pred & self.not_ignore_mask):
            dist = ndimage.distance_transform_edt(np.pad(err, 1) if padding else err).astype(np.float32)
            dist = (dist[1:-1, 1:-1] if padding else dist) * self.not_clicked_map
            at = int(np.argmax(dist))                    # first maximum in raster order
            out.append((float(dist.flat[at]), divmod(at, dist.shape[1])))
        return out

    def _deepest_errors_device(self, pred_mask, padding):
        import torch
        from pvpuformer_amd import ops
        if self._dev is None:
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(self.device)
            self._dev = (up(self.gt_mask), up(self.not_ignore_mask), up(self.not_clicked_map))
        gt, valid, keep = self._dev
        if torch.is_tensor(pred_mask):
            pred = pred_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        else:
            pred = torch.from_numpy(np.ascontiguousarray(pred_mask, dtype=np.uint8)).to(self.device, non_blocking=True)
        dist = ops.edt(ops.error_masks(pred, gt, valid), zero_border=padding)
        keys = ops.masked_argmax(dist, keep).tolist()                       # the click's one synchronisation
        width = gt.shape[1]
        out = []
        for key in keys:
            key &= 0xFFFFFFFFFFFFFFFF
            value = struct.unpack("<f", struct.pack("<I", key >> 32))[0]
            out.append((value, divmod(0xFFFFFFFF - (key & 0xFFFFFFFF), width)))
        return out
