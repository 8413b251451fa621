"""ZoomIn / LimitLongestSide mirror (pvpuformer_amd/isegm/inference/transforms.py) against a click sequence recorded from
the reference's own classes (tests/golden/zoom.npz, oracle/make_golden.py zoom_fixtures): regions of interest, re-mapped
click coordinates and the recalculation flag bit-exact; resized images within 1e-5 (HIP kernel, GPU test)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pvpuformer_amd.isegm.inference import transforms as T
from pvpuformer_amd.isegm.inference.clicker import Click


def _sequence(fx, device, check_images, atol):
    H, W = int(fx["H"]), int(fx["W"])
    g = torch.Generator().manual_seed(int(fx["image_seed"]))
    image_nd = torch.rand(1, 4, H, W, generator=g).to(device)
    yy, xx = np.mgrid[0:H, 0:W]
    blob1 = (((yy - 140) / 60.0) ** 2 + ((xx - 200) / 90.0) ** 2 < 1).astype(np.float32)
    blob2 = (((yy - 90) / 30.0) ** 2 + ((xx - 330) / 40.0) ** 2 < 1).astype(np.float32)
    clicks = [Click(bool(r[0]), (r[1], r[2]), int(r[3])) for r in fx["clicks"]]
    clicks[0].coords = (int(clicks[0].coords[0]), int(clicks[0].coords[1]))
    for name, kw in (("vpu", dict(target_size=(448, 448), skip_clicks=-1)), ("ritm", dict(target_size=400, skip_clicks=1))):
        z = T.ZoomIn(**kw)
        probs = [blob1 * 0.9, np.maximum(blob1, blob2) * 0.8, np.maximum(blob1, blob2) * 0.8]
        for step in range(3):
            cl = clicks[:step + 2]
            img_t, tcl = z.transform(image_nd, [cl])
            roi = z._object_roi if z._object_roi is not None else (-1, -1, -1, -1)
            assert tuple(int(v) for v in roi) == tuple(int(v) for v in fx[f"{name}_{step}_roi"]), (name, step)
            assert bool(z.image_changed) == bool(fx[f"{name}_{step}_changed"])
            assert tuple(img_t.shape) == tuple(fx[f"{name}_{step}_img_shape"])
            got = np.asarray([[c.coords[0], c.coords[1]] for c in tcl[0]], np.float64)
            assert np.array_equal(got, fx[f"{name}_{step}_tclicks"]), (name, step)       # click re-mapping: bit-exact
            if check_images:
                np.testing.assert_allclose(img_t[:, :, ::13, ::11].cpu().numpy(), fx[f"{name}_{step}_img_sub"], atol=atol)
            hh, ww = img_t.shape[2:]
            ty, tx = torch.meshgrid(torch.linspace(0, 1, hh), torch.linspace(0, 1, ww), indexing="ij")
            net_out = (torch.sin(3 * ty + step) * torch.cos(5 * tx) * 0.5 + 0.5)[None, None].to(device)
            back = z.inv_transform(net_out)
            assert tuple(back.shape) == tuple(fx[f"{name}_{step}_back_shape"])
            if check_images:
                np.testing.assert_allclose(back[:, :, ::7, ::9].cpu().numpy(), fx[f"{name}_{step}_back_sub"], atol=atol)
            z._prev_probs = probs[step][None, None]
            assert bool(z.check_possible_recalculation()) == bool(fx[f"{name}_{step}_recalc"])
    lim = T.LimitLongestSide(max_size=256)
    img_t, tcl = lim.transform(image_nd, [clicks[:2]])
    assert tuple(lim._object_roi) == tuple(int(v) for v in fx["lim_roi"])
    assert tuple(img_t.shape) == tuple(fx["lim_img_shape"])
    assert np.array_equal(np.asarray([[c.coords[0], c.coords[1]] for c in tcl[0]], np.float64), fx["lim_tclicks"])
    if check_images:
        np.testing.assert_allclose(img_t[:, :, ::13, ::11].cpu().numpy(), fx["lim_img_sub"], atol=atol)


def test_zoom_bookkeeping_matches_reference_cpu(golden_dir, monkeypatch):
    """Integer / float bookkeeping of the whole click sequence on the CPU.  The product resize is GPU-only, so a torch
    stand-in is injected here for the duration of the test (the images it produces are not what is being tested)."""
    fx = np.load(os.path.join(golden_dir, "zoom.npz"))
    with pytest.raises(RuntimeError):
        T.resize_align_corners(torch.zeros(1, 1, 4, 4), (8, 8))          # no CPU path in the product
    monkeypatch.setattr(T, "resize_align_corners",
                        lambda x, size: F.interpolate(x.float(), size=tuple(int(s) for s in size), mode="bilinear",
                                                      align_corners=True))
    _sequence(fx, "cpu", check_images=False, atol=0.0)


def test_bbox_helpers():
    m = np.zeros((50, 60), bool)
    m[10:21, 30:41] = True
    assert T.get_bbox_from_mask(m) == (10, 20, 30, 40)
    assert T.expand_bbox((10, 20, 30, 40), 1.4, 20) == (5, 25, 25, 45)
    assert T.clamp_bbox((-5, 70, 3, 90), 0, 49, 0, 59) == (0, 49, 3, 59)
    assert abs(T.get_bbox_iou((0, 9, 0, 9), (5, 14, 0, 9)) - 5 / 15) < 1e-12
    assert T.check_object_roi((10, 20, 30, 40), [Click(True, (10, 30))]) and not T.check_object_roi((10, 20, 30, 40), [Click(True, (20, 30))])


@pytest.mark.gpu
def test_zoom_sequence_matches_reference_gpu(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    fx = np.load(os.path.join(golden_dir, "zoom.npz"))
    _sequence(fx, "cuda", check_images=True, atol=1e-5)


@pytest.mark.gpu
def test_resize_align_corners_vs_torch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    x = torch.rand(2, 3, 37, 53, device="cuda")
    for size in ((448, 448), (20, 31), (37, 53), (111, 40)):
        ref = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
        torch.testing.assert_close(T.resize_align_corners(x, size), ref, atol=1e-5, rtol=1e-5)
