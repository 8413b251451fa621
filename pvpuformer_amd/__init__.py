"""pvpuformer_amd -- MI355X-native implementation of the VPUFormer forward/backward hot path.

Hand-written HIP kernels for gfx950 behind a C ABI (``include/vpu_hip.h``, ``libvpu_hip.so``), driven from Python
through ctypes, mirroring the reference's ``isegm.model.is_vpu_model`` / ``isegm.inference`` API.
"""
__version__ = "0.1.0"


def install():
    """Registers the API mirror as the top-level ``isegm`` package (if the reference's own ``isegm`` is not already
    imported), so reference drivers and checkpoint ``config['class']`` paths resolve to this implementation."""
    import importlib
    import sys
    if "isegm" in sys.modules and not sys.modules["isegm"].__name__.startswith("pvpuformer_amd"):
        return sys.modules["isegm"]
    pkg = importlib.import_module("pvpuformer_amd.isegm")
    sys.modules["isegm"] = pkg
    prefix = "pvpuformer_amd.isegm."
    for name in ("model", "model.is_model", "model.is_vpu_model", "model.modeling", "model.modeling.models_vit",
                 "model.modeling.pos_embed", "utils", "utils.serialization", "engine", "engine.trainer", "inference",
                 "inference.clicker", "inference.utils", "inference.predictors", "inference.predictors.base"):
        sys.modules["isegm." + name] = importlib.import_module(prefix + name)
    return pkg
