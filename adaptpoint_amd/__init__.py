"""adaptpoint_amd -- MI355X-native set-abstraction hot path of AdaptPoint / OpenPoints.

    csrc/      hand-written gfx950 HIP kernels + the C ABI (include/adaptpoint_amd.h)
    ops.py     the nine `*_wrapper` operators of the reference extension (+ the resampler)
    layers.py  operator layer: the reference's callable names over those operators
    set_abstraction.py, pointnext.py   PointNeXt SetAbstraction block, PointNeXt-S classifier
    fused.py, fused_wide.py            the fused grouped MLP (32->32->64 / every width)
    pointset.py, attention.py, imitator.py, augmentor.py, discriminator.py, gan.py
                                       the AdaptPoint generator / discriminator / training steps
    dp.py      data-parallel plumbing (flat gradient all-reduce, all-reduce SyncBatchNorm)

`pointnet2_batch_cuda.py` at the repository root is the drop-in module the
reference imports.
"""
__version__ = "0.2.0"
