"""ZoomIn / LimitLongestSide (pvpuformer_amd/isegm/inference/transforms/) against a click sequence recorded from the
reference's own classes (tests/golden/zoom.npz, oracle/make_golden.py zoom_fixtures): regions of interest, re-mapped
click coordinates and the recalculation flag bit-exact; resized images within 1e-5 (HIP kernels, GPU test)."""
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


def _host_mask_box(prob, thr, positive_clicks=()):
    """numpy stand-in of the device reduction (vpu_mask_bbox), same contract"""
    m = prob[0, 0].cpu().numpy() > thr
    if not m.any():
        return (0, m.shape[0], -1, m.shape[1], -1)
    rows, cols = np.flatnonzero(m.any(1)), np.flatnonzero(m.any(0))
    pts = [(int(r), int(c)) for r, c in positive_clicks]
    rr, cc = [rows[0], rows[-1]] + [p[0] for p in pts], [cols[0], cols[-1]] + [p[1] for p in pts]
    return (int(m.sum()), int(min(rr)), int(max(rr)), int(min(cc)), int(max(cc)))


def test_zoom_bookkeeping_matches_reference_cpu(golden_dir, monkeypatch):
    """Integer / float bookkeeping of the whole click sequence on the CPU.  The product's two device operations are
    GPU-only, so stand-ins are injected for the duration of the test (the images they produce are not what is tested)."""
    fx = np.load(os.path.join(golden_dir, "zoom.npz"))
    with pytest.raises(RuntimeError):
        T.resize_align_corners(torch.zeros(1, 1, 4, 4), (8, 8))          # no CPU path in the product
    with pytest.raises(RuntimeError):
        T._device.mask_box(torch.zeros(1, 1, 4, 4), 0.5)
    monkeypatch.setattr(T._device, "resize_align_corners",
                        lambda x, size: F.interpolate(x.float(), size=tuple(int(s) for s in size), mode="bilinear",
                                                      align_corners=True))
    monkeypatch.setattr(T._device, "mask_box", _host_mask_box)
    _sequence(fx, "cpu", check_images=False, atol=0.0)


def test_roi_arithmetic():
    R = T.Roi
    assert R(10, 20, 30, 40).grown(1.4, 20) == (5, 25, 25, 45) and R(10, 20, 30, 40).grown(1.0) == (10, 20, 30, 40)   # (round half to even)
    assert R(-5, 70, 3, 90).clipped(50, 60) == (0, 49, 3, 59) and R.whole(50, 60) == (0, 49, 0, 59)
    assert abs(R(0, 9, 0, 9).overlap((5, 14, 0, 9)) - 5 / 15) < 1e-12 and R(0, 9, 0, 9).overlap((20, 29, 0, 9)) == 0
    assert R(10, 20, 30, 40).encloses([Click(True, (10, 30)), Click(False, (0, 0))])
    assert not R(10, 20, 30, 40).encloses([Click(True, (20, 30))])          # upper bounds are exclusive in this check
    assert R(10, 19, 30, 49).to_crop((15, 40), (100, 40)) == (50.0, 20.0)
    assert R(0, 99, 0, 49).output_size(400) == (400, 200) and R(0, 99, 0, 49).output_size((448, 448)) == (448, 448)
    assert (R(3, 9, 0, 5).height, R(3, 9, 0, 5).width) == (7, 6)


@pytest.mark.gpu
def test_mask_bbox_and_clicker_device_path_equal_host():
    """The two device reductions of the NoBRS loop: vpu_mask_bbox == the numpy box of (prob > thr) joined with the positive
    clicks (empty masks, a single pixel, clicks outside the mask); the Clicker's device path (error masks -> exact EDT ->
    packed first-arg-max) picks the same clicks as its scipy host path over a 6-click sequence, incl. ignore labels, the
    exclusion of already-clicked pixels and tie-breaking by raster order."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pvpuformer_amd.isegm.inference.clicker import Clicker
    g = torch.Generator().manual_seed(2)
    prob = torch.rand(1, 1, 97, 131, generator=g) * 0.4
    prob[0, 0, 20:41, 50:90] = 0.9
    for clicks in ([], [(5, 7)], [(95, 3), (30.9, 120.2)]):
        assert T._device.mask_box(prob.cuda(), 0.5, clicks) == _host_mask_box(prob, 0.5, clicks), clicks
    assert T._device.mask_box(torch.zeros(1, 1, 97, 131).cuda(), 0.5, [(3, 4)])[0] == 0
    one = torch.zeros(1, 1, 97, 131); one[0, 0, 96, 130] = 1.0
    assert T._device.mask_box(one.cuda(), 0.5) == (1, 96, 96, 130, 130)
    yy, xx = np.mgrid[0:120, 0:150]
    gt = ((((yy - 60) / 30.0) ** 2 + ((xx - 70) / 45.0) ** 2) < 1).astype(np.int32)
    gt[0:10, 0:10] = -1                                                   # ignore region
    preds = [np.zeros_like(gt, bool), (((yy - 50) / 20.0) ** 2 + ((xx - 60) / 30.0) ** 2) < 1, (xx > 40) & (yy > 30),
             gt == 1, np.ones_like(gt, bool), (yy < 60) & (gt == 1)]
    dev, host = Clicker(gt_mask=gt, device="cuda"), Clicker(gt_mask=gt, device=None)
    for p in preds:
        dev.make_next_click(p)
        host.make_next_click(p)
    a = [(c.is_positive, int(c.coords[0]), int(c.coords[1]), c.indx) for c in dev.get_clicks()]
    b = [(c.is_positive, int(c.coords[0]), int(c.coords[1]), c.indx) for c in host.get_clicks()]
    assert a == b and len(a) == 6 and dev.num_pos_clicks == host.num_pos_clicks
    st = dev.get_state(); dev._remove_last_click(); assert len(dev) == 5 and dev.not_clicked_map.sum() == gt.size - 5
    dev.set_state(st); assert len(dev) == 6 and float(dev._dev[2].sum()) == gt.size - 6
    dev.make_next_click(torch.from_numpy(preds[1]).cuda())                 # a device tensor is taken as it is
    host.make_next_click(preds[1])
    assert tuple(int(v) for v in dev.get_clicks()[-1].coords) == tuple(int(v) for v in host.get_clicks()[-1].coords)


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
