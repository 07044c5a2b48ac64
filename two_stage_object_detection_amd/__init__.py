"""two_stage_object_detection_amd -- MI355X (gfx950) native forward path of the two-stage detector of
3SAILab/two_stage_object_detection, behind the reference's own module surface.

    from two_stage_object_detection_amd.nets.frcnn import FasterRCNN
    from two_stage_object_detection_amd.nets.rpn import RegionProposalNetwork, ProposalCreator
    from two_stage_object_detection_amd.nets.classify import HarNetRoIHead
    from two_stage_object_detection_amd.models.resnet import resnet50
    from two_stage_object_detection_amd.models.hardnet import HarDNetFeatureExtraction, HarNetClassifier
    from two_stage_object_detection_amd.utils.basic_anchors import generate_basic_anchor, enumerate_shifted_anchor
    from two_stage_object_detection_amd.utils.loc_bbox_iou import bbox_iou, loc2bbox

``install_dropin()`` aliases these sub-packages as top-level ``nets`` / ``models`` / ``utils`` so that
code written against the reference imports them unchanged.

Compute happens only in libtsod.so (hand-written HIP, include/tsod.h); there is no CPU fallback:
the modules raise on CPU tensors and on a missing extension.
"""
import importlib
import sys

__version__ = "0.2.0"


def install_dropin(force: bool = False) -> None:
    """Register ``nets``, ``models``, ``utils`` and ``dataset`` (and their sub-modules) as aliases of this package's
    mirrors of the reference modules."""
    for top, subs in (("utils", ("basic_anchors", "loc_bbox_iou")), ("models", ("resnet", "hardnet")),
                      ("nets", ("rpn", "classify", "frcnn", "frcnn_training")), ("dataset", ("transform",))):
        if top in sys.modules and not force and not getattr(sys.modules[top], "__tsod_dropin__", False):
            raise ImportError(f"a different top-level package named {top!r} is already imported")
        pkg = importlib.import_module(f"{__name__}.{top}")
        pkg.__tsod_dropin__ = True
        sys.modules[top] = pkg
        for s in subs:
            sys.modules[f"{top}.{s}"] = importlib.import_module(f"{__name__}.{top}.{s}")
