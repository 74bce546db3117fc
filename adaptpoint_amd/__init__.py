"""adaptpoint_amd -- MI355X-native set-abstraction hot path of AdaptPoint / OpenPoints.

    csrc/      hand-written gfx950 HIP kernels + the C ABI (include/adaptpoint_amd.h)
    ops.py     the nine `*_wrapper` operators of the reference extension
    layers.py  host-side mirror of openpoints/models/layers/{subsample,group,upsampling}.py
    set_abstraction.py  PointNeXt SetAbstraction block over those layers

`pointnet2_batch_cuda.py` at the repository root is the drop-in module the
reference imports.
"""
__version__ = "0.1.0"
