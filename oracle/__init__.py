"""oracle/ -- CPU restatement of the reference's detector forward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``two_stage_object_detection_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the reported CPU baseline.

Pinning status (see DESIGN.md "Oracle"):

* anchors, loc2bbox, bbox_iou, ResNet / HarDNet forward, RPN + RoI-head glue:
  pinned against the reference's own Python modules run in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the
  reference's known answers (utils/loc_bbox_iou.py:99-103, utils/basic_anchors.py:60-63).
* ``nms`` and ``roi_pool`` (torchvision, absent from /root/reference and from the image,
  version unpinned by the reference): PARITY UNPINNED -- restated from the published
  algorithm, pinned only by hand-derived known-answer cases.  Likewise ``roi_align``, the checker of the
  ADDED ``roi_op="align"`` option (the reference itself uses RoIPool).
* ``targets`` (AnchorTargetCreator / ProposalTargetCreator): pinned bit-exact against the reference's own
  classes (tests/golden/targets_*.npz).
* ``eval_transform`` (torchvision v2 Resize + ToTensor on a float tensor image): PARITY UNPINNED for
  the torchvision glue; the arithmetic is torch's own antialiased bilinear ``interpolate`` (oracle/transform.py).
"""
from .box import (  # noqa: F401
    generate_basic_anchor, enumerate_shifted_anchor, loc2bbox, bbox_iou,
    nms, roi_pool, roi_align, proposal_layer, rpn_forward, roi_head_forward,
)
from .backbones import resnet_trunk, hardnet_trunk, calibrate_bn  # noqa: F401
from .detector import detector_forward, detections_from_outputs, postprocess  # noqa: F401
from .transform import eval_transform  # noqa: F401
