"""Host-side mirror of the reference's modeling package: importing it registers every plugin."""
from .backbone import HipResNet50, build_backbone, build_resnet_backbone  # noqa: F401
from .camera_head import PlaneCameraHead, build_camera_head  # noqa: F401
from .matching_head import MatchingHead, build_matching_head  # noqa: F401
from .meta_arch import PlaneTR_NopeSAC, decode_masks  # noqa: F401
from .plane_head import PlaneTRHead, build_planeTR_head, post_select  # noqa: F401

# detectron2 present -> the drop-in is also reachable through detectron2's own META_ARCH_REGISTRY / build_model
from ..registry import register_into_detectron2 as _register_into_detectron2  # noqa: E402

_register_into_detectron2()
