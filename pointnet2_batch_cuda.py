"""Drop-in for the reference's compiled module of the same name.

The reference does `import pointnet2_batch_cuda as pointnet2_cuda`
(openpoints/cpp/pointnet2_batch/__init__.py:2) and calls nine `*_wrapper`
functions bound at openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24.
With this file (and the `adaptpoint_amd` package) on sys.path -- or installed
by `python setup.py install` -- `openpoints` runs unmodified on MI355X: the same
symbols launch hand-written gfx950 kernels through libadaptpoint_amd.so.
"""
from adaptpoint_amd.ops import (  # noqa: F401
    ball_query_wrapper,
    furthest_point_sampling_wrapper,
    gather_points_grad_wrapper,
    gather_points_wrapper,
    group_points_grad_wrapper,
    group_points_wrapper,
    three_interpolate_grad_wrapper,
    three_interpolate_wrapper,
    three_nn_wrapper,
)
from adaptpoint_amd import _lib as _apn_lib

# Fail at import time, like a missing compiled extension would.
_apn_lib.load()

# APN_PATCH_OPENPOINTS=1: besides the nine operators, route the reference's SetAbstraction / PointsetGrouper /
# Anchor_selfattention / ConvBNReLU1D forwards to the fused kernels -- without touching the reference tree
# (adaptpoint_amd/integrate.py; the classes are patched as their modules are imported).
from adaptpoint_amd import integrate as _apn_integrate
if _apn_integrate.requested_by_environment():
    _apn_integrate.patch_openpoints(lazy=True)
