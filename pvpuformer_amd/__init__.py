"""pvpuformer_amd -- MI355X-native implementation of the VPUFormer forward/backward hot path.

Hand-written HIP kernels for gfx950 behind a C ABI (``include/vpu_hip.h``, ``libvpu_hip.so``), driven from Python
through ctypes, mirroring the reference's ``isegm.model.is_vpu_model`` / ``isegm.inference`` API.
"""
__version__ = "0.1.0"
