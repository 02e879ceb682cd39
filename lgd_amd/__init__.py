"""lgd_amd -- MI355X-native LGD (label-guided self-distillation) training path.

Public surface mirrors the reference's (megvii-research/LGD): registries, `build_model`,
`DistillatorRetinaNet` / `DistillatorFCOS`, `DynamicTeacher`, `SequentialConvs`,
`build_distillator_configs`.  The LGD hot path runs as hand-written HIP kernels
(lgd_amd/csrc, C-ABI in include/lgd_hip.h); see DESIGN.md.
"""
__version__ = "0.1.0"
