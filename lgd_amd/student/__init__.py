"""Student detectors (callers of the LGD hot path): plain PyTorch-ROCm restatements of the
detectron2 v0.3 RetinaNet / cvpods-style FCOS the reference subclasses (neither package exists in
this environment; SURVEY.md appendix A/B, parity unpinned)."""
from .retinanet import RetinaNetCT  # noqa: F401
from .fcos import FCOSCT  # noqa: F401
