"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Imports the REAL reference (/root/reference, read-only) in the build container so that
oracle/make_golden.py can pin the oracle against it.  /root/reference does not exist on the GPU
box: nothing under tests/, bench.py or __graft_entry__.py imports this module.

Two modules the reference imports at module top but never calls on the hot path are stubbed
(SURVEY.md section 8c): ``cv2`` (mvn/utils/volumetric.py:2, mvn/utils/img.py:2) and ``easydict``
(mvn/utils/cfg.py:2).
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "mvn"))


def load():
    """Returns the reference's ``mvn`` package (models.triangulation, utils.op, ... imported)."""
    if not available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "easydict" not in sys.modules:
        from .synth import AttrDict
        ed = types.ModuleType("easydict")
        ed.EasyDict = AttrDict
        sys.modules["easydict"] = ed
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import mvn  # noqa: F401
    import mvn.models.triangulation  # noqa: F401
    import mvn.models.pose_resnet  # noqa: F401
    import mvn.models.v2v  # noqa: F401
    import mvn.utils.op  # noqa: F401
    import mvn.utils.multiview  # noqa: F401
    import mvn.utils.volumetric  # noqa: F401
    assert os.path.realpath(mvn.__file__).startswith(REFERENCE_ROOT), mvn.__file__
    return mvn
