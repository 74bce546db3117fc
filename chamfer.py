"""Inert stand-in for the reference's compiled `chamfer` module.

`openpoints/models/__init__.py:9` imports it transitively
(reconstruction/maskedpointvit.py:10 -> cpp/chamfer_dist/__init__.py:10) even for
classification configs.  Chamfer distance is outside the set-abstraction hot
path, so both entry points refuse to run.
"""


def forward(*args, **kwargs):
    raise NotImplementedError("chamfer.forward is outside the MI355X hot-path build")


def backward(*args, **kwargs):
    raise NotImplementedError("chamfer.backward is outside the MI355X hot-path build")
