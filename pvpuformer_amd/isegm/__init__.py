"""Mirror of the reference's ``isegm`` package for the VPUFormer hot path (same module paths, class names, constructor
arguments, state-dict keys, call signatures and output dict).  ``pvpuformer_amd.install()`` registers it as the
top-level ``isegm`` so that the reference's drivers and released checkpoints (which name
``isegm.model.is_vpu_model.VitMultiGaussianVector_ed_Model``) resolve to this implementation."""
