"""The two device operations the predictor transforms are built on (module attributes, looked up at call time, so a CPU
test can substitute stand-ins): the ``align_corners=True`` bilinear resize and the bounding box of a thresholded
probability map joined with the positive clicks.  Both are HIP kernels (``vpu_upsample_ac_fwd``, ``vpu_mask_bbox``);
there is no CPU path in the product."""
import torch

from pvpuformer_amd import ops


def resize_align_corners(x, size):
    """``F.interpolate(x, size, mode='bilinear', align_corners=True)`` for a CUDA fp32 NCHW tensor."""
    if not x.is_cuda:
        raise RuntimeError("the predictor transforms run on the GPU only (no CPU resize path exists)")
    n, c, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    src = x.contiguous().float()
    if (h, w) == (H, W):
        return src.clone()
    dst = torch.empty(n, c, H, W, device=x.device, dtype=torch.float32)
    ops.upsample_ac_fwd(src, dst, n * c, h, w, H, W)
    return dst


def mask_box(prob, thr, positive_clicks=()):
    """(count, rmin, rmax, cmin, cmax) of ``prob[0, 0] > thr`` joined with the integer (row, col) positive clicks: the map
    stays on the device, five integers come back (the one host synchronisation of a ZoomIn step)."""
    if not prob.is_cuda:
        raise RuntimeError("the predictor transforms run on the GPU only (no CPU path exists)")
    clicks = None
    if len(positive_clicks):
        clicks = torch.tensor([[int(r), int(c)] for r, c in positive_clicks], dtype=torch.int32).to(prob.device, non_blocking=True)
    p = prob[0, 0].contiguous().float().unsqueeze(0)
    return tuple(ops.mask_bbox(p, thr, clicks)[0].tolist())
