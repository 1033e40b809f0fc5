"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference model classes from /root/reference.

Only usable in the build container (the GPU box has no /root/reference); used by
oracle/make_golden.py to (a) validate the torch-functional restatement in lbc_oracle.py
and (b) generate the committed fixtures under tests/golden/.

Recipe: the reference imports torchvision / carla / cv2 at module
import time (bird_view/models/common.py:9, agent.py:3-5, birdview.py:1) and calls .cuda()
inside NormalizeV2.__init__ (common.py:105-106); none of that touches the arithmetic of
the hot path, so stub modules are injected and .cuda() is made a no-op while importing /
constructing.
"""
import ast
import contextlib
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


def available():
    import os
    return os.path.isdir(REFERENCE_ROOT + "/bird_view/models")


@contextlib.contextmanager
def _cuda_noop():
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


def _install_stubs():
    for name in ("torchvision", "torchvision.transforms", "carla", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].ToTensor = lambda: (lambda x: x)
    bv = REFERENCE_ROOT + "/bird_view"
    if bv not in sys.path:
        sys.path.insert(0, bv)


def load_model_classes():
    """-> (ImagePolicyModelSS, BirdViewPolicyModelSS) reference classes."""
    _install_stubs()
    with _cuda_noop():
        from models.image import ImagePolicyModelSS
        from models.birdview import BirdViewPolicyModelSS
    return ImagePolicyModelSS, BirdViewPolicyModelSS


def build(kind, backbone, **kw):
    Image, Bird = load_model_classes()
    with _cuda_noop():
        if kind == "image":
            return Image(backbone, **kw)
        return Bird(backbone, **kw)


def load_one_hot():
    _install_stubs()
    from utils.train_utils import one_hot
    return one_hot


class _Cv2ProjectPointsOnly:
    """Stand-in for the ONE cv2 call on the path (train_image_phase0.py:61 cv2.projectPoints(xyz, tvec=0, rvec=0, A, None)).
    opencv-python==4.0.0.21 (environment.yml:186) is not installed and cannot be fetched.  Published algorithm
    (OpenCV calib3d docs, projectPoints): x' = X/Z, y' = Y/Z after [R|t] (identity here), no distortion terms when
    distCoeffs is None, then u = fx*x' + cx, v = fy*y' + cy, computed in float64, returned as (N, 1, 2) plus a jacobian.
    Everything else of the reference's CoordConverter (frame changes, metres, offset, clipping, reshapes) is the
    reference's own code, executed unchanged around this call."""

    @staticmethod
    def projectPoints(xyz, rvec, tvec, A, dist):
        assert dist is None and not np.any(rvec) and not np.any(tvec), "only the reference's call pattern is restated"
        xyz = np.asarray(xyz, dtype=np.float64)
        u = A[0, 0] * xyz[:, 0] / xyz[:, 2] + A[0, 2]
        v = A[1, 1] * xyz[:, 1] / xyz[:, 2] + A[1, 2]
        return np.stack([u, v], -1)[:, None, :], None


def extract_training_defs(script, names):
    """AST-extract classes/functions from a non-importable training script
    (training/train_image_phase{0,1}.py import modules that do not exist here)."""
    src = open(REFERENCE_ROOT + "/training/" + script).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "np": np, "PIXELS_PER_METER": 5, "CROP_SIZE": 192, "N_STEP": 5, "cv2": _Cv2ProjectPointsOnly}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            code = compile(ast.Module([node], []), script, "exec")
            exec(code, ns)
    return {n: ns[n] for n in names}
