"""pvpuformer_amd -- MI355X-native implementation of the VPUFormer forward/backward hot path.

Hand-written HIP kernels for gfx950 behind a C ABI (``include/vpu_hip.h``, ``libvpu_hip.so``), driven from Python
through ctypes, mirroring the reference's ``isegm.model.is_vpu_model`` / ``isegm.inference`` API.
"""
__version__ = "0.2.0"


def _other_isegm_dir(mirror_dir, reference_root=None):
    """The ``isegm`` package directory that the interpreter would import if this package did not exist: the first
    ``<entry>/isegm`` along ``reference_root`` / ``sys.path`` that is not the mirror itself."""
    import os
    import sys
    entries = ([reference_root] if reference_root else []) + list(sys.path)
    for e in entries:
        d = os.path.join(e or os.getcwd(), "isegm")
        if os.path.isdir(d) and os.path.realpath(d) != os.path.realpath(mirror_dir):
            return d
    return None


def install(reference_root=None):
    """Makes ``import isegm...`` resolve to this implementation FOR THE HOT PATH AND TO THE REST OF THE REFERENCE FOR
    EVERYTHING ELSE (an overlay, not a replacement):

    * every module of the API mirror (``pvpuformer_amd/isegm/**``) is registered under its ``isegm.*`` name, so
      ``isegm.model.is_vpu_model``, ``isegm.engine.trainer``, ``isegm.inference.predictors`` ... -- and the dotted class
      paths stored in checkpoints -- are this package's;
    * the mirror packages' ``__path__`` is extended with the matching directories of the next ``isegm`` package on
      ``sys.path`` (or under ``reference_root``), so modules the mirror does not ship -- ``isegm.utils.exp``,
      ``isegm.utils.vis``, ``isegm.utils.log``, ``isegm.model.losses``, ``isegm.data`` ... -- are imported from there,
      as sub-modules of the same ``isegm`` package (their relative imports keep working);
    * a mirror module that hides a file of that tree (``isegm.utils.misc``, ``isegm.inference.utils``,
      ``isegm.utils.serialization`` ...) offers the file's public surface itself and, for any other name, falls through
      to the hidden file (``pvpuformer_amd._overlay``).

    Call it before the first ``import isegm``: if the reference's own package is already imported it is left alone and
    returned.  Idempotent.  In a job launched with ``WORLD_SIZE`` > 1 AND ``VPU_DIST_RESERVE_CUS`` set explicitly it also
    caps RCCL's channel count to those CUs (``parallel.configure_rccl_env``, logged on rank 0) -- ``install()`` runs before
    the reference's ``init_experiment`` creates the process group (exp.py:29-32), which is when that has to be in the
    environment.  Opt-in: the cap applies to EVERY collective of the process and is unmeasured on a multi-GPU node."""
    import importlib
    import os
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 and os.environ.get("VPU_DIST_RESERVE_CUS", "") != "":
        from .parallel import configure_rccl_env
        configure_rccl_env()
    import pkgutil
    import sys
    cur = sys.modules.get("isegm")
    if cur is not None and not cur.__name__.startswith("pvpuformer_amd"):
        return cur
    pkg = importlib.import_module("pvpuformer_amd.isegm")
    mirror_dir = os.path.dirname(os.path.abspath(pkg.__file__))
    prefix = "pvpuformer_amd.isegm."
    names = [m.name[len(prefix):] for m in pkgutil.walk_packages([mirror_dir], prefix)]   # before __path__ grows
    sys.modules["isegm"] = pkg
    for name in names:
        sys.modules["isegm." + name] = importlib.import_module(prefix + name)
    other = _other_isegm_dir(mirror_dir, reference_root)
    if other is not None:
        def extend(mod, rel):
            d = os.path.join(other, *rel.split(".")) if rel else other
            if os.path.isdir(d) and d not in list(mod.__path__):
                mod.__path__.append(d)
        from ._overlay import attach, shadowed_file
        extend(pkg, "")
        for name in names:
            mod = sys.modules["isegm." + name]
            if hasattr(mod, "__path__"):
                extend(mod, name)
            path, is_pkg = shadowed_file(other, name)
            if path is not None and "__getattr__" not in vars(mod):
                attach(mod, "isegm." + name, path, is_pkg)
    pkg.__vpu_overlay__ = other
    return pkg
