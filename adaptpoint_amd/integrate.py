"""Zero-edit opt-in for an UNMODIFIED OpenPoints / AdaptPoint tree: `patch_openpoints()` rebinds the `forward` of the four
reference classes that sit on the hot path to the fused operators of this package -- nothing in the reference tree is
edited, no parameter is added, renamed or copied (state_dict keys and checkpoints stay the reference's), and every
call that the fused kernels do not cover goes to the reference's own `forward`, counted by reason.

    import openpoints                         # the reference, unmodified, with `pointnet2_batch_cuda` = this repo's drop-in
    from adaptpoint_amd import integrate
    integrate.patch_openpoints()              # or: APN_PATCH_OPENPOINTS=1 in the environment (pointnet2_batch_cuda.py
                                              # installs the same patch lazily, as each module is imported)

    class (reference file:line)                                              -> runs on
    SetAbstraction.forward      (openpoints/models/backbone/pointnext.py:140-170)   adaptpoint_amd.set_abstraction: FPS + ball
                                query + grouped MLP + pool + skip in the fused launches (csrc/sa_fused.hip, sa_wide*.hip)
    PointsetGrouper.forward     (models_adaptpoint/generator_component4_15.py:394-431)   adaptpoint_amd.pointset (csrc/pointset_group.hip)
    Anchor_selfattention.forward (:448-480)                                   adaptpoint_amd.attention (csrc/attention.hip)
    ConvBNReLU1D.forward        (:93-105)                                     adaptpoint_amd.pointwise (csrc/pointwise.hip)

Without the patch the reference reaches the nine drop-in operators only (`pointnet2_batch_cuda`): exact, eager,
13.8 k clouds/s on the headline block; with it the same cfgs run the fused path (bench.py's `value`).
`COUNTS` tells which path every call took; `unpatch_openpoints()` restores the reference's methods.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

import torch.nn as nn

COUNTS = {}                     # "<Class>.fused" / "<Class>.reference: <reason>" -> calls
_ORIGINAL = {}                  # (class, attribute) -> the reference's own function
TARGET_MODULES = ("openpoints.models.backbone.pointnext", "openpoints.models_adaptpoint.generator_component4_15")


def _count(key):
    COUNTS[key] = COUNTS.get(key, 0) + 1


def _swap(cls, name, fn):
    if (cls, name) not in _ORIGINAL:
        _ORIGINAL[(cls, name)] = cls.__dict__[name]
    setattr(cls, name, fn)


# ------------------------------------------------------------------------------------------- SetAbstraction
def _adapter_for(ref):
    """A `adaptpoint_amd.set_abstraction.SetAbstraction` that SHARES the reference module's sub-modules (convs, skipconv,
    act: same objects, same parameters) -- or the reason (str) why this instance stays on the reference's forward."""
    from .layers import BallGrouper, make_grouper
    from .set_abstraction import SetAbstraction
    if getattr(ref, "feature_type", None) != "dp_fj":
        return f"feature_type {getattr(ref, 'feature_type', None)!r}"
    if not ref.is_head:
        if getattr(getattr(ref, "sample_fn", None), "__name__", "") != "furthest_point_sample":
            return "sampler is not FPS"
        g = ref.grouper
        if ref.all_aggr:
            if type(g).__name__ != "GroupAll":
                return f"grouper {type(g).__name__}"
        else:
            if type(g).__name__ != "QueryAndGroup":
                return f"grouper {type(g).__name__}"
            if (not g.relative_xyz or g.normalize_by_std or g.normalize_by_allstd or g.normalize_by_allstd2
                    or g.return_only_idx):
                return "QueryAndGroup options outside dp / radius"
    for blk in ref.convs:
        mods = list(blk)
        if not (isinstance(mods[0], (nn.Conv1d, nn.Conv2d)) and mods[0].kernel_size in ((1,), (1, 1))
                and all(isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.ReLU)) for m in mods[1:])):
            return "convolution block other than conv(1x1)-BatchNorm-ReLU"
    if ref.use_res and not (isinstance(ref.skipconv, (nn.Sequential, nn.Identity)) and isinstance(ref.act, nn.ReLU)):
        return "residual branch other than Conv1d + ReLU"
    ad = SetAbstraction.__new__(SetAbstraction)
    nn.Module.__init__(ad)
    ad.stride, ad.is_head, ad.all_aggr, ad.use_res = ref.stride, ref.is_head, ref.all_aggr, ref.use_res
    ad.feature_type, ad.fused, ad.sync_bn = ref.feature_type, True, False
    ad.convs = ref.convs
    if ref.use_res:
        ad.skipconv, ad.act = ref.skipconv, ref.act
    if not ref.is_head:
        ad.grouper = (make_grouper({'NAME': 'ballquery', 'radius': None, 'nsample': None}) if ref.all_aggr
                      else BallGrouper(ref.grouper.radius, ref.grouper.nsample, normalize_dp=ref.grouper.normalize_dp))
    return ad


def _sa_forward(self, pf):
    ad = self.__dict__.get("_apn_adapter")
    if ad is None:
        ad = _adapter_for(self)
        self.__dict__["_apn_adapter"] = ad          # (in __dict__, not a registered sub-module: state_dict keys unchanged)
    if isinstance(ad, str):
        _count("SetAbstraction.reference: " + ad)
        return _sa_original(self, pf)
    why = _call_outside_fused(ad, pf)
    if why is not None:                              # per CALL: the structure fits, this input does not
        _count("SetAbstraction.reference: " + why)
        return _sa_original(self, pf)
    ad.training = self.training
    _count("SetAbstraction.fused")
    return ad(pf)


# the fused index stage (csrc/sa_seq.hip: apn_sa_sample_seq -> the resident samplers of csrc/fps.hip) takes clouds of at
# most this many points; larger ones (S3DIS rooms at 24 k) go to the reference's own forward, whose furthest_point_sample
# reaches the streaming sampler through the drop-in operator
MAX_FUSED_POINTS = 16384


def _call_outside_fused(ad, pf):
    """Why THIS call cannot take the adapter (None: it can).  The adapter was chosen from the module's structure alone;
    the tensors are checked here, per call, so that an uncovered input never raises out of a patched tree."""
    import torch
    p, f = pf[0], pf[1]
    if not (torch.is_tensor(p) and torch.is_tensor(f)):
        return "inputs are not tensors"
    # (CPU tensors are not diverted: the reference's forward would hand them to the same drop-in operators, which refuse
    # them just as loudly -- adaptpoint_amd/ops.py has no CPU path -- and tests/test_integrate_reference_cpu.py runs the
    # adapters' composed mirrors on CPU tensors over the oracle to compare them with the REAL reference classes)
    if p.dtype != torch.float32 or f.dtype != torch.float32:
        return f"dtype {p.dtype} / {f.dtype}"
    if not ad.is_head and not ad.all_aggr and p.shape[1] > MAX_FUSED_POINTS:
        return f"N > {MAX_FUSED_POINTS}"
    return None


def _sa_original(self, pf):
    for cls in type(self).__mro__:
        if (cls, "forward") in _ORIGINAL:
            return _ORIGINAL[(cls, "forward")](self, pf)
    raise RuntimeError("adaptpoint_amd.integrate: the reference's SetAbstraction.forward was not recorded")


# ------------------------------------------------------------------------------------------- the imitator's three classes
def _conv_bn_relu_forward(self, x):
    from . import pointwise
    net = self.net
    if (len(net) == 3 and isinstance(net[0], nn.Conv1d) and isinstance(net[1], nn.BatchNorm1d)
            and isinstance(net[2], nn.ReLU) and pointwise.supported(x, net[0], net[1])):
        _count("ConvBNReLU1D.fused")
        return pointwise.conv_bn_act(x, net[0], net[1], relu=True)
    _count("ConvBNReLU1D.reference: shape / mode not served by csrc/pointwise.hip"
           if x.is_cuda else "ConvBNReLU1D.reference: CPU tensor")
    return _ORIGINAL[(type(self), "forward")](self, x)


def _patch_pointnext(mod):
    _swap(mod.SetAbstraction, "forward", _sa_forward)


def _patch_generator(mod):
    from .attention import AnchorSelfAttention
    from .pointset import PointsetGrouper
    _swap(mod.ConvBNReLU1D, "forward", _conv_bn_relu_forward)
    # the mirrors' forwards read exactly the attributes the reference classes define (reduce, kneighbors, radi,
    # normalize, affine_alpha / affine_beta; to_qkv, pos_embedding, res, head_num) plus the class-level switch `fused`
    mirror_group, mirror_attn = PointsetGrouper.forward, AnchorSelfAttention.forward

    def group_forward(self, xyz, points, index=None):
        _count("PointsetGrouper.fused" if (points.is_cuda and self.normalize == "anchor") else "PointsetGrouper.composed")
        return mirror_group(self, xyz, points, index)

    def attn_forward(self, x, xyz=None):
        _count("Anchor_selfattention.fused" if x.is_cuda else "Anchor_selfattention.composed")
        return mirror_attn(self, x, xyz)
    mod.PointsetGrouper.fused = True
    mod.Anchor_selfattention.fused = True
    _swap(mod.PointsetGrouper, "forward", group_forward)
    _swap(mod.Anchor_selfattention, "forward", attn_forward)


_PATCHERS = {TARGET_MODULES[0]: _patch_pointnext, TARGET_MODULES[1]: _patch_generator}


class _PostImport(importlib.abc.MetaPathFinder):
    """Patches a target module right after its import (for APN_PATCH_OPENPOINTS=1: the drop-in module is imported while
    `openpoints` is still half way through its own import)."""
    _busy = False

    def find_spec(self, name, path, target=None):
        if name not in _PATCHERS or _PostImport._busy:
            return None
        _PostImport._busy = True
        try:
            spec = importlib.util.find_spec(name)
        finally:
            _PostImport._busy = False
        if spec is None or spec.loader is None or not hasattr(spec.loader, "exec_module"):
            return None
        run = spec.loader.exec_module

        def exec_module(module, _run=run, _name=name):
            _run(module)
            _PATCHERS[_name](module)
        spec.loader.exec_module = exec_module
        return spec


def patch_openpoints(lazy=False):
    """Rebind the four forwards (see the module docstring).  Modules already imported are patched now; with lazy=True the
    others are patched when they are imported, else they are imported here.  Returns the patched module names."""
    done = []
    for name, patch in _PATCHERS.items():
        if name in sys.modules:
            patch(sys.modules[name])
            done.append(name)
        elif not lazy:
            patch(importlib.import_module(name))
            done.append(name)
    if lazy and not any(isinstance(f, _PostImport) for f in sys.meta_path):
        sys.meta_path.insert(0, _PostImport())
    return done


def unpatch_openpoints():
    for (cls, name), fn in list(_ORIGINAL.items()):
        setattr(cls, name, fn)
        if "fused" in cls.__dict__:
            delattr(cls, "fused")
    _ORIGINAL.clear()
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _PostImport)]


def requested_by_environment():
    return os.environ.get("APN_PATCH_OPENPOINTS") == "1"
