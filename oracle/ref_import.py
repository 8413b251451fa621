"""Import harness for the read-only reference at /root/reference (build container only).

TEST INFRASTRUCTURE.  Only ``oracle/make_golden.py`` (fixture generation) and
``tests/test_overlay_reference_cpu.py`` (the overlay against the real checkout; skipped where
``/root/reference`` is absent) use this file; nothing under ``pvpuformer_amd/``, ``bench.py`` or
the ``-m gpu`` tests may import it (``/root/reference`` does not exist on the GPU box).

The reference imports several third-party packages that are absent from this image
(cv2, mmcv, timm, easydict, torchvision, tensorboard).  Nothing of their arithmetic is
needed on the VPU hot path except ``mmcv.cnn.ConvModule`` (conv + ReLU, restated below from
mmcv-full 1.6.2 semantics: ``self.conv`` = nn.Conv2d(bias=True) when ``norm_cfg`` is None,
``self.activate`` = nn.ReLU unless ``act_cfg`` is None).  The stand-ins are injected into
``sys.modules`` at run time; no stub file is ever written to disk and no reference source is
copied.
"""
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package so "import a.b" works
    sys.modules[name] = m
    return m


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Registry:
    def __init__(self, name, parent=None, **kw):
        self.name = name
        self._d = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._d[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def build(self, cfg):
        raise RuntimeError("registry build is not available in the stub")


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule for norm_cfg=None: conv(bias=True) [+ ReLU(inplace)]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, **kw):
        super().__init__()
        assert norm_cfg is None and conv_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups,
                              bias=True if bias == 'auto' else bias)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


def _identity_decorator(*a, **kw):
    if len(a) == 1 and callable(a[0]) and not kw:
        return a[0]

    def deco(f):
        return f
    return deco


def install_stubs():
    sys.dont_write_bytecode = True
    for alias, t in (("bool", bool), ("int", int), ("float", float)):
        if alias not in np.__dict__:
            setattr(np, alias, t)

    ed = _mod("easydict")
    ed.EasyDict = _EasyDict

    cv2 = _mod("cv2")
    cv2.DIST_L1, cv2.DIST_L2 = 1, 2

    def _no_cv2(*a, **k):
        raise RuntimeError("cv2 is not available in this image (parity unpinned for OpenCV paths)")
    for fn in ("rectangle", "polylines", "distanceTransform", "imwrite", "resize", "circle"):
        setattr(cv2, fn, _no_cv2)

    timm = _mod("timm")
    tm = _mod("timm.models")
    tl = _mod("timm.models.layers")
    timm.models = tm
    tm.layers = tl

    class DropPath(nn.Module):
        def __init__(self, p=0.):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0. or not self.training
            return x
    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: x if isinstance(x, (tuple, list)) else (x, x)
    tl.trunc_normal_ = lambda t, std=.02, **k: nn.init.normal_(t, std=std)

    tv = _mod("torchvision")
    tvt = _mod("torchvision.transforms")
    tv.transforms = tvt

    class ToTensor:
        def __call__(self, img):
            t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float()
            return t / 255.0 if img.dtype == np.uint8 else t
    tvt.ToTensor = ToTensor

    tb = _mod("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass
    tb.SummaryWriter = SummaryWriter

    mmcv = _mod("mmcv")
    mmcv.jit = _identity_decorator
    u = _mod("mmcv.utils")
    u.Registry = _Registry
    u.build_from_cfg = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    import logging
    u.get_logger = lambda name, log_file=None, log_level=logging.INFO: logging.getLogger(name)
    u.mkdir_or_exist = lambda *a, **k: None
    cnn = _mod("mmcv.cnn")
    cnn.MODELS = _Registry("model")
    cnn.ConvModule = _ConvModule
    cnn.build_conv_layer = lambda cfg, *a, **k: nn.Conv2d(*a, **k)
    cnn.build_norm_layer = lambda cfg, n, **k: ("ln", nn.LayerNorm(n))
    br = _mod("mmcv.cnn.bricks")
    reg = _mod("mmcv.cnn.bricks.registry")
    reg.ATTENTION = _Registry("attention")
    cnn.bricks = br
    br.registry = reg
    run = _mod("mmcv.runner")
    run.BaseModule = _BaseModule
    run.auto_fp16 = _identity_decorator
    run.force_fp32 = _identity_decorator
    run.get_dist_info = lambda: (0, 1)
    bm = _mod("mmcv.runner.base_module")
    bm.BaseModule = _BaseModule
    fio = _mod("mmcv.fileio")
    fio.FileClient = object
    fio.load = lambda *a, **k: None
    par = _mod("mmcv.parallel")
    par.is_module_wrapper = lambda m: False
    mmcv.utils, mmcv.cnn, mmcv.runner, mmcv.fileio, mmcv.parallel = u, cnn, run, fio, par

    _mod("bezier")            # imported at the top of isegm/engine/trainer.py, used by the scribble simulator only
    _mod("tqdm").tqdm = lambda it, *a, **k: it
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def install_simulator_standins():
    """Stand-ins for the two third-party calls inside the click / box simulators (isegm/engine/trainer.py:628,736,1176),
    so that the reference's BOOKKEEPING around them can be run here: cv2.distanceTransform(DIST_L2, 5) -> the oracle's
    restatement of OpenCV's 5 x 5 chamfer transform (vpu_oracle.chamfer_l2_5x5; mask size 0 = precise -> the exact Euclidean
    transform), skimage.measure.label(connectivity=2) -> scipy 8-connected labelling.  Parity of the click COORDINATES
    against real OpenCV therefore stays unpinned."""
    from scipy import ndimage
    import vpu_oracle as vo
    cv2 = sys.modules["cv2"]

    def distance_transform(m, dist_type, mask_size):
        if dist_type == cv2.DIST_L2 and mask_size == 5:
            return vo.chamfer_l2_5x5(m)
        return ndimage.distance_transform_edt(m).astype(np.float32)
    cv2.distanceTransform = distance_transform
    sk = _mod("skimage")
    skm = _mod("skimage.measure")
    sk.measure = skm
    skm.label = lambda mask, connectivity=2: ndimage.label(mask, structure=np.ones((3, 3), bool))[0]


def import_reference():
    """Returns the reference's VPU model module and loss module."""
    install_stubs()
    import isegm.model.is_vpu_model as ref_vpu  # noqa
    import isegm.model.losses as ref_losses  # noqa
    return ref_vpu, ref_losses
